#!/bin/bash
# ablation sweep of the small-tile conv kernel on the three batch-1 3x3 layer shapes (see conv_small_probe.hip)
cd $(dirname $0)
for shape in "1 76 128 256 3" "1 19 512 1024 3" "1 38 256 512 3" "1 76 256 128 1" "4 38 256 512 3"; do
  for t in 32x32 32x64; do
    for a in 0 1 2 4 3 6 7; do VY_CONV_FORCE=$t ./conv_small_probe_$a $shape; done
  done
  VY_CONV_SMALL=0 ./conv_small_probe_0 $shape
done
