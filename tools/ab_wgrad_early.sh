#!/bin/bash
# What would a PERFECT weight-gradient kernel for the early cells (stages.0.1 ... 0.5: N <= 128, K <= 576 — output tiles that
# are mostly padding, 82-104 TF; stages.0.2.body.0 25.7 TF) return to the training step?  The step with those launches simply
# skipped (VY_TRAIN_ABL=32 in a measurement build, results are garbage) against the full step, same box, alternating.
# NEEDS videoyolo_amd/libvyolo_trainabl.so: VY_BUILD_EXTRA_FLAGS=-DVY_TRAIN_ABL_BUILD python -m videoyolo_amd.build --force
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
cp $R/videoyolo_amd/libvyolo.so /tmp/libvyolo_keep.so
cp $R/videoyolo_amd/libvyolo_trainabl.so $R/videoyolo_amd/libvyolo.so
for i in 1 2 3; do for abl in 0 32 8; do
  fps=$(VY_TRAIN_ABL=$abl python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-pmc --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.2f ms' % (d['value'], d['ms_per_step']))")
  echo "VY_TRAIN_ABL=$abl  $fps"
done; done
cp /tmp/libvyolo_keep.so $R/videoyolo_amd/libvyolo.so
