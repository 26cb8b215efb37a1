#!/bin/bash
# same-box A/B of two builds (libvyolo_prev.so / libvyolo_new.so): headline, 416, training, and the small batches
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do for v in prev new; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/bench.py --no-pmc --cpu-frames 0 --no-roofline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
t = r['also_train416']
print('$v: infer %.1f fps  416: %.1f   train %.1f fps fwd %.2f bwd %.2f' % (r['value'], r['also_416']['frames_per_s'], t['frames_per_s'], t['forward_ms'], t['backward_ms']))"
done; done
for v in prev new prev new; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
echo "== $v"; python3 $R/tools/small_batch_latency.py --batches 2,4,8,16,32 2>/dev/null
done
cp $R/videoyolo_amd/libvyolo_new.so $R/videoyolo_amd/libvyolo.so
