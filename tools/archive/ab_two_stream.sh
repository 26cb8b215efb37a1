#!/bin/bash
# the frames of a batch as two half-batches on two streams (VY_TWO_STREAM_BATCH) against one launch sequence, same box
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for v in 0 64; do
VY_TWO_STREAM_BATCH=$v python3 $R/bench.py --no-pmc --no-train-legs --no-roofline --cpu-frames 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('two-stream from batch $v: infer %.1f fps   416: %.1f' % (r['value'], r['also_416']['frames_per_s']))"
done; done
