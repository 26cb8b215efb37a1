#!/bin/bash
# same-box A/B of the default (stream-K launches chosen by the shared cost model) against VY_CONV_SK=0
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for v in 0 1; do
VY_CONV_SK=$v python3 $R/bench.py --no-pmc --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r['also_train416']
print('SK=$v infer %.1f fps  frac %.4f  b1 %.3f ms  416: %.1f   train %.1f fps (%.4f) fwd %.2f bwd %.2f' % (r['value'], r['roofline']['frac'], r['latency_batch1']['eager_ms'], r['also_416']['frames_per_s'], t['frames_per_s'], t['frac'], t['forward_ms'], t['backward_ms']))"
done; done
