#!/bin/bash
# conv-only rate and tile / stream-K choices of the training step at every size of the reference's default multi-scale
# mode (train_yolov3.py:258-271: 320 ... 608): one rocprofv3 kernel trace per size (weight gradients on the main stream:
# clean durations), joined with the library's launch labels by tools/train_layers.py.
# usage: tools/multiscale_layers.sh [tag] [sizes...]   ->  gpurun_out/<tag>_multiscale_layers.txt (+ per-size tables)
tag=${1:-r06}; shift
sizes=${@:-320 352 384 416 448 480 512 544 576 608}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_multiscale_layers.txt
echo "# per size: kind, total ms, TFLOP/s of the matrix-core launches (fp32 roof 157.3); batch 16, exact mode" > $out
for s in $sizes; do
  bash $R/tools/train_layers.sh ${tag}_ms$s --size $s --batch 16 > /dev/null 2>&1
  f=$R/gpurun_out/${tag}_ms${s}_layers.txt
  echo "== size $s" >> $out
  grep -E "^(fwd|dgrad|wgrad) total" $f >> $out
  python3 - $f $s >> $out <<'PY'
import re, sys
t = fl = 0.0
tiles = {}
kind = None
for line in open(sys.argv[1]):
    if line.startswith("----"):
        kind = line.split()[1]
        continue
    m = re.match(r"(fwd|dgrad|wgrad) total ([\d.]+) ms +([\d.]+) TF", line)
    if m:
        t += float(m.group(2)); fl += float(m.group(2)) * float(m.group(3))
        continue
    m = re.search(r"x \d+\s+(\S+)\s+([\d.]+) us\s+([\d.]+) TF", line)
    if m and kind:
        d = tiles.setdefault((kind, m.group(1)), [0, 0.0])
        d[0] += 1; d[1] += float(m.group(2))
print("all matrix-core launches: %.2f ms, %.1f TF = %.3f of the fp32 roof" % (t, fl / t, fl / t / 157.3))
for (k, tile), (n, us) in sorted(tiles.items(), key=lambda kv: -kv[1][1]):
    print("   %-6s %-28s %3d launches %8.2f ms" % (k, tile, n, us / 1e3))
PY
done
tail -40 $out
