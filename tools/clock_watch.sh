#!/bin/bash
# sample the GPU clocks / power while the inference bench runs (is the MFMA peak clock sustained?)
python bench.py --steps 150 --warmup 5 --cpu-frames 0 --no-roofline > /tmp/b.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W): /W=/' | tr '\n' ' '
  echo
  sleep 0.7
done
cat /tmp/b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['value'])"
