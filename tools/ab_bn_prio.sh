#!/bin/bash
# A/B of the issue priority of the BatchNorm passes (VY_BN_PRIO: s_setprio 3 in bn_apply / bn_bwd_reduce / bn_bwd_apply) and of
# the queue priority of the weight-gradient stream (VY_SIDE_LOW_PRIO), training step 416x416 batch 16, same box, alternating.
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  out=$(VY_BN_PRIO=$1 VY_SIDE_LOW_PRIO=$2 python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('step_split',{}); print('%.1f frames/s  %.2f ms  fwd %.2f bwd %.2f' % (d['value'], d['ms_per_step'], s.get('forward_ms',0), s.get('backward_ms',0)))")
  echo "VY_BN_PRIO=$1 VY_SIDE_LOW_PRIO=$2  $out"
done; done
