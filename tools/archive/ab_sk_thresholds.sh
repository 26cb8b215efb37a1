#!/bin/bash
# stream-K criterion: threshold (VY_CONV_SK_GAIN) x hand-off cost (VY_CONV_SK_COST), same box, two alternations
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for cfg in "0.03 7.5" "0.02 7.5" "0.01 7.5" "0.03 5" "0.02 5" "0.02 3"; do set -- $cfg
VY_CONV_SK_GAIN=$1 VY_CONV_SK_COST=$2 python3 $R/bench.py --no-pmc --cpu-frames 0 --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r['also_train416']
print('gain>=$1 cost $2: infer %.1f fps  416: %.1f   train %.1f fps fwd %.2f' % (r['value'], r['also_416']['frames_per_s'], t['frames_per_s'], t['forward_ms']))"
done; done
