#!/bin/bash
# usage: tools/train_sweep.sh VAR v1 v2 ...   (training bench per value of an env switch, e.g. VY_TRAIN_SIDE_STREAM 0 1)
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --mode train --steps 8 --warmup 3 > /tmp/o.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/o.json')); print('train $var=$v', round(d['value'],1), 'fwd', round(d['roofline']['forward_ms'],2), 'bwd', round(d['roofline']['backward_ms'],2))"
done
