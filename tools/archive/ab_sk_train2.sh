#!/bin/bash
# stream-K (hybrid schedule, write-through hand-off, predicted-saving criterion): same-box A/B, inference and training
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
run() { # sk gain dgrad
VY_CONV_SK=$1 VY_CONV_SK_GAIN=$2 VY_CONV_SK_DGRAD=$3 python3 $R/bench.py --mode train --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('SK=$1 min gain $2 dgrad $3: train %.1f fps  fwd %.2f  bwd %.2f ms' % (r['value'], r['roofline']['forward_ms'], r['roofline']['backward_ms']))"
}
inf() {
VY_CONV_SK=$1 VY_CONV_SK_GAIN=$2 python3 $R/bench.py --no-pmc --no-train-legs --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('SK=$1 min gain $2 infer %.1f fps  frac %.4f  b1 %.3f ms  416: %.1f' % (r['value'], r['roofline']['frac'], r['latency_batch1']['eager_ms'], r['also_416']['frames_per_s']))"
}
for i in 1 2; do
run 0 0 0; run 1 0.01 0; run 1 0.01 1; run 1 0.03 0
inf 0 0; inf 1 0.01; inf 1 0.03
done
