#!/bin/bash
# One frame, per layer: the default tile choice against the 16x16-wave-tile kernel (conv_small.hip) forced on every forward
# launch it has an instance for (VY_CONV_FORCE=32x32 / 32x64; launches summed in runs stay on their split-K form).
# Both are bit-exact; the question is which launches of ONE frame the small tiles would win (round 3 measured the 3x3 cells
# only).  -> gpurun_out/<tag>_small_tiles_<size>.txt : launch, default us, 32x32 us, 32x64 us
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
for size in 608 416; do
  for f in default 32x32 32x64; do
    if [ $f = default ]; then unset VY_CONV_FORCE; else export VY_CONV_FORCE=$f; fi
    python3 $R/tools/layer_profile.py --size $size --batch 1 --out /tmp/lp_${size}_$f.txt > /dev/null 2>&1
  done
  unset VY_CONV_FORCE
  python3 - $size > $O/${tag}_small_tiles_$size.txt <<'PY'
import sys
size = sys.argv[1]
def rows(f):
    out = []
    for ln in open('/tmp/lp_%s_%s.txt' % (size, f)):
        p = ln.split()
        if len(p) >= 5 and ('x' in p[1] or p[1] == '-'):
            try: out.append((p[0].split('|')[0], p[0], p[1], float(p[2]) * 1e3))
            except ValueError: pass
    return out
d, a, b = rows('default'), rows('32x32'), rows('32x64')
print('%-36s %-16s %9s %9s %9s   best' % ('launch (default tile)', 'w', 'default', '32x32', '32x64'))
tot = [0, 0, 0, 0]
for x, y, z in zip(d, a, b):
    assert x[0] == y[0] == z[0], (x, y, z)
    best = min(x[3], y[3], z[3])
    tot[0] += x[3]; tot[1] += y[3]; tot[2] += z[3]; tot[3] += best
    print('%-36s %-16s %9.1f %9.1f %9.1f   %s' % (x[1], x[2], x[3], y[3], z[3], 'default' if best == x[3] else ('32x32' if best == y[3] else '32x64')))
print('sum of launches (us): default %.1f, 32x32 %.1f, 32x64 %.1f, per-launch best %.1f' % tuple(tot))
PY
done
cat $O/${tag}_small_tiles_608.txt $O/${tag}_small_tiles_416.txt
