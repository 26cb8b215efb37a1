#!/bin/bash
# same-box A/B of two builds of the library: videoyolo_amd/libvyolo_prev.so against libvyolo_new.so (both made beforehand;
# the .so files are git-ignored and travel with the gpurun snapshot).  Headline step, one-frame latency, 416 batch.
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for v in prev new; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/bench.py --no-pmc --cpu-frames 0 --steps 20 --warmup 5 --no-host-legs --no-split-leg ${AB_ARGS:---no-train-legs} 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r.get('also_train416')
print('$v: infer %.1f fps  dominant %.4f  all-conv %.4f  b1 %.3f ms  416: %.1f' % (r['value'], r['roofline']['frac'], r['roofline']['frac_all_conv'], r['latency_batch1']['eager_ms'], r['also_416']['frames_per_s']) + (('  train %.1f fps fwd %.2f bwd %.2f' % (t['frames_per_s'], t['forward_ms'], t['backward_ms'])) if t else ''))"
done; done
cp $R/videoyolo_amd/libvyolo_new.so $R/videoyolo_amd/libvyolo.so
