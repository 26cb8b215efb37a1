#!/bin/bash
# the other configurations quoted in DESIGN.md section 5
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$1', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms', round(r.get('achieved',0),1), 'TF', round(r.get('whole_step_tflops',0),1))"; }
python bench.py --size 416 --cpu-frames 0 2>/dev/null | p infer416_b64
python bench.py --classes 30 --cpu-frames 0 2>/dev/null | p infer608_b64_c30
python bench.py --obj-bias -5 --cpu-frames 0 2>/dev/null | p infer608_b64_objbias-5
python bench.py --mode train --size 608 2>/dev/null | p train608_b16
python bench.py --mode train --batch 32 2>/dev/null | p train416_b32
python tools/latency.py 2>/dev/null | tail -4
