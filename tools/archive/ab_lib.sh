#!/bin/bash
# same-box A/B of two builds of the library: videoyolo_amd/libvyolo_prev.so against libvyolo_new.so (both made beforehand)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for v in prev new; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/bench.py --no-pmc --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r['also_train416']
print('$v: infer %.1f fps  b1 %.3f ms  416: %.1f   train %.1f fps (%.4f) fwd %.2f bwd %.2f' % (r['value'], r['latency_batch1']['eager_ms'], r['also_416']['frames_per_s'], t['frames_per_s'], t['frac'], t['forward_ms'], t['backward_ms']))"
done; done
for v in prev new; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/tools/layer_profile.py --size 608 --out $R/gpurun_out/r03_lay_608_$v.txt > /dev/null 2>&1
done
cp $R/videoyolo_amd/libvyolo_new.so $R/videoyolo_amd/libvyolo.so
