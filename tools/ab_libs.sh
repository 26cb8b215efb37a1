#!/bin/bash
# same-box A/B of several builds of the library: tools/ab_libs.sh prev new rule1 ... (videoyolo_amd/libvyolo_<tag>.so, built
# beforehand; *.so is git-ignored and travels with the gpurun snapshot).  Headline step + one-frame latencies.
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for v in "$@"; do
cp $R/videoyolo_amd/libvyolo_$v.so $R/videoyolo_amd/libvyolo.so
python3 $R/bench.py --no-pmc --cpu-frames 0 --steps 20 --warmup 5 --no-host-legs --no-split-leg --no-train-legs 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('%-6s infer %.1f fps  dominant %.4f  all-conv %.4f  416b64 %.1f  b1@608 %.3f ms  b1@416 %.3f ms' % ('$v', r['value'], r['roofline']['frac'], r['roofline']['frac_all_conv'], r['also_416']['frames_per_s'], r['latency_batch1']['eager_ms'], r['latency_batch1_416']['eager_ms']))"
done; done
cp $R/videoyolo_amd/libvyolo_new.so $R/videoyolo_amd/libvyolo.so
