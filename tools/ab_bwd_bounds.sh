#!/bin/bash
# NEEDS a measurement build of the library (the shipped one never reads VY_TRAIN_ABL):
#   VY_BUILD_EXTRA_FLAGS=-DVY_TRAIN_ABL_BUILD python -m videoyolo_amd.build --force     (rebuild without it afterwards)
# Which stream of the backward pass holds the training step?  VY_TRAIN_ABL=8: no weight-gradient kernels (side stream),
# 16: no data-gradient kernels (main stream), 24: neither, 31: neither and no BatchNorm passes (forward + fixed parts only)
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
{
for mode in exact split_bf16x3_train; do
  for abl in 0 8 16 24 31 0; do
    fps=$(VY_TRAIN_ABL=$abl python3 $R/bench.py --mode train --conv-mode $mode --steps 20 --warmup 5 --no-pmc --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))")
    echo "$mode  VY_TRAIN_ABL=$abl  $fps"
  done
done
} 2>&1 | tee $R/gpurun_out/ab_bwd_bounds.txt
