#!/bin/bash
# stream-K with the grid sized so that a share is at least one tile (1 .. resident blocks per CU): parity with the hand-off
# forced, per-layer training tables with the switch off / on, then a same-box A/B of the training step at two thresholds
R=$GRAFT_REPO_ROOT
cd $R
for n in 24 13; do VY_CONV_SK=1 VY_CONV_SK_SLOTS=$n timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_parity.py -m gpu -q -x -k "not small_tile and not recycled" 2>&1 | tail -2; done
VY_CONV_SK=1 timeout 900 python -m pytest tests/test_gpu_train_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
VY_CONV_SK=0 tools/train_layers.sh r03_sk0t > /dev/null 2>&1
VY_CONV_SK=1 VY_CONV_SK_GAIN=0.03 tools/train_layers.sh r03_sk1t > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for v in "0 0.03" "1 0.25" "1 0.08" "1 0.03"; do set -- $v
VY_CONV_SK=$1 VY_CONV_SK_GAIN=$2 python3 $R/bench.py --mode train --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('SK=$1 gain>=$2 train %.1f fps  fwd %.2f  bwd %.2f ms' % (r['value'], r['roofline']['forward_ms'], r['roofline']['backward_ms']))"
done; done
