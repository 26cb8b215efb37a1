#!/bin/bash
# training step 416x416 batch 16: exact against conv mode split_bf16x3_train with its parts switched
# (VY_SPLIT_TRAIN: 0 none, 1 forward + dgrad, 2 forward only, 3 dgrad only;  VY_SPLIT_WGRAD: weight gradients)
for m in "exact 1 1" "split_bf16x3_train 1 1" "split_bf16x3_train 0 1" "split_bf16x3_train 1 0" "split_bf16x3_train 3 1" "split_bf16x3_train 2 1"; do
  set -- $m
  VY_SPLIT_TRAIN=$2 VY_SPLIT_WGRAD=$3 python3 $GRAFT_REPO_ROOT/bench.py --mode train --conv-mode $1 --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-20s fwd/dgrad switch $2  wgrad $3 : %7.1f fps  %6.2f ms   fwd %s  bwd %s' % ('$1', r['value'], r['ms_per_step'], r.get('forward_ms'), r.get('backward_ms')))
"
done
