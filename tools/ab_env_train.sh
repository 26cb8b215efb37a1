#!/bin/bash
# same-box A/B of an environment switch on the training step (416x416 batch 16): tools/ab_env_train.sh VAR A B
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for v in $2 $3; do
env $1=$v python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-pmc 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d.get('step_split', {})
print('$1=$v  %.1f frames/s  %.2f ms  fwd %.2f bwd %.2f  loss %.6f' % (d['value'], d['ms_per_step'], s.get('forward_ms', 0), s.get('backward_ms', 0), d['config']['loss_rank0']))"
done; done
