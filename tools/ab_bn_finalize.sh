#!/bin/bash
# What would statistics reductions that FINISH THEMSELVES return to the training step?  Each BatchNorm cell launches a small
# ordered-reduce + finalize kernel between its reduce pass and its apply pass, forward (bn_reduce_finalize, 5.9 us x 72) and
# backward (bn_bwd_reduce_finalize, 12.3 us x 72 in the two-stream trace): 1.3 ms of kernel time on the dependent chain.
# Bound: the step with those launches skipped after warm-up (VY_TRAIN_ABL=64 backward, 128 forward, 192 both; measurement build,
# the apply passes then use the previous step's coefficients) against the full step, same box, alternating.
# NEEDS videoyolo_amd/libvyolo_trainabl.so: VY_BUILD_EXTRA_FLAGS=-DVY_TRAIN_ABL_BUILD python -m videoyolo_amd.build --force
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
cp $R/videoyolo_amd/libvyolo.so /tmp/libvyolo_keep.so
cp $R/videoyolo_amd/libvyolo_trainabl.so $R/videoyolo_amd/libvyolo.so
for i in 1 2 3; do for abl in 0 64 128 192; do
  fps=$(VY_TRAIN_ABL=$abl python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-pmc --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d.get('step_split',{}); print('%.1f frames/s  %.2f ms  fwd %.2f bwd %.2f' % (d['value'], d['ms_per_step'], s.get('forward_ms',0), s.get('backward_ms',0)))")
  echo "VY_TRAIN_ABL=$abl  $fps"
done; done
cp /tmp/libvyolo_keep.so $R/videoyolo_amd/libvyolo.so
