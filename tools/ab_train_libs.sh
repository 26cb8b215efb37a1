#!/bin/bash
# same-box A/B of library builds on the training step (416x416 batch 16): tools/ab_train_libs.sh new wgskip ...
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
cp $R/videoyolo_amd/libvyolo.so /tmp/libvyolo_keep.so
for i in 1 2 3; do for v in "$@"; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/bench.py --mode train --steps 20 --warmup 5 --no-pmc 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d.get('step_split', {})
print('%-8s %.1f frames/s  %.2f ms  fwd %.2f bwd %.2f' % ('$v', d['value'], d['ms_per_step'], s.get('forward_ms', 0), s.get('backward_ms', 0)))"
done; done
cp /tmp/libvyolo_keep.so $R/videoyolo_amd/libvyolo.so
