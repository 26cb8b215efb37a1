#!/bin/bash
for v in ${@:-0 3 4}; do
  VY_CONV_SCHED=$v python bench.py --steps 5 --warmup 2 --cpu-frames 0 > /tmp/o.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/o.json')); print('infer sched $v', round(d['value'],1), round(d['roofline']['achieved'],1), d['roofline']['by_kernel_ms'])"
  VY_CONV_SCHED=$v python bench.py --mode train --steps 8 --warmup 3 > /tmp/o.json 2>/dev/null
  python -c "import json; d=json.load(open('/tmp/o.json')); print('train sched $v', round(d['value'],1), round(d['roofline']['forward_ms'],2), round(d['roofline']['backward_ms'],2))"
done
