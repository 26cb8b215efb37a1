#!/bin/bash
# NEEDS a measurement build of the library (the shipped one never reads VY_TRAIN_ABL):
#   VY_BUILD_EXTRA_FLAGS=-DVY_TRAIN_ABL_BUILD python -m videoyolo_amd.build --force     (rebuild without it afterwards)
# What could fusing the BatchNorm passes of the training step into the neighbouring conv launches return AT MOST?
# The step with those launches simply skipped (VY_TRAIN_ABL, train.hip; results are garbage), exact and split mode.
# bits: 1 bn_bwd_reduce (+ finalize)   2 forward bn_apply   4 bn_bwd_apply
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out
{
for mode in exact split_bf16x3_train; do
  for abl in 0 1 2 4 7 0; do
    fps=$(VY_TRAIN_ABL=$abl python3 $R/bench.py --mode train --conv-mode $mode --steps 20 --warmup 5 --no-pmc --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))")
    echo "$mode  VY_TRAIN_ABL=$abl  $fps"
  done
done
} 2>&1 | tee $R/gpurun_out/ab_bn_bounds.txt
