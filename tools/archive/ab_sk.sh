#!/bin/bash
# stream-K (conv_igemm.hip, SK instances): parity with the hand-off forced on small shapes, then a same-box A/B
for slots in 5; do echo "== SK slots $slots"; VY_CONV_SK_SLOTS=$slots timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py -m gpu -q -x -k "not small_tile and not recycled" 2>&1 | tail -2; done
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for v in 0 1; do
VY_CONV_SK=$v python3 $GRAFT_REPO_ROOT/bench.py --no-pmc --no-train-legs --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('SK=$v infer %.1f fps  frac %.4f  b1 %.3f ms  416: %.1f' % (r['value'], r['roofline']['frac'], r['latency_batch1']['eager_ms'], r['also_416']['frames_per_s']))"
VY_CONV_SK=$v python3 $GRAFT_REPO_ROOT/bench.py --mode train --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('SK=$v train %.1f fps  fwd %.2f  bwd %.2f ms' % (r['value'], r['roofline']['forward_ms'], r['roofline']['backward_ms']))"
done; done
