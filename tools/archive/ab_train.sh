#!/bin/bash
# same-box A/B of a training-step switch: tools/ab_train.sh VAR A B [reps]   (e.g. VY_WGRAD_XCD 0 1)
# prints frames/s of `bench.py --mode train` alternating VAR=A / VAR=B, then the wgrad / conv HBM bytes per step of each
var=$1; a=$2; b=$3; reps=${4:-2}
R=${GRAFT_REPO_ROOT:-$(dirname $0)/..}
cd /tmp; export TMPDIR=/tmp
for i in $(seq $reps); do
  for v in $a $b; do
    env $var=$v python3 $R/bench.py --mode train --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('$var=$v  %.1f frames/s  fwd %.2f  bwd %.2f ms' % (r['value'], r['roofline']['forward_ms'], r['roofline']['backward_ms']))"
  done
done
for v in $a $b; do
  env $var=$v python3 $R/bench.py --mode train --steps 5 --warmup 2 --no-roofline 2>/dev/null | tail -1 > /tmp/ab_$v.json
done
for v in $a $b; do
  env $var=$v python3 $R/bench.py --mode train --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r['roofline']
print('$var=$v  traffic %.1f GB/step' % (t['traffic'] / 1e9), t.get('traffic_detail', {}).get('top_kernels_MB_per_step'))"
done
