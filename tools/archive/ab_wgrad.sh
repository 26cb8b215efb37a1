#!/bin/bash
# training bench + per-kernel durations of the weight-gradient kernels (compare with profiles/r03_train416_b16_kernel_stats.csv)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do python3 $R/bench.py --mode train --no-pmc --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('train %.1f fps  fwd %.2f  bwd %.2f ms  frac %.4f' % (r['value'], r['roofline']['forward_ms'], r['roofline']['backward_ms'], r['roofline']['frac']))"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_wgrad_trn -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-pmc > /dev/null 2>&1
grep -h "wgrad_kernel\|conv_igemm_kernel<64, 64, 2, 2, true, 2" $(ls $R/gpurun_out/ab_wgrad_trn/*/*kernel_stats.csv | head -1) | cut -d, -f1-4
