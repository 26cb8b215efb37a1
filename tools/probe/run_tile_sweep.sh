#!/bin/bash
# forced-tile sweep of the 32x32x2 conv kernel on small-batch 3x3 layer shapes (conv_small_probe_0 = plain launcher)
cd $(dirname $0)
for b in 1 2 4; do
for shape in "$b 76 128 256 3" "$b 38 256 512 3" "$b 19 512 1024 3" "$b 152 64 128 3" "$b 76 256 128 1" "$b 19 1024 512 1"; do
  for t in 64x64 128x64 128x128; do VY_CONV_FORCE=$t ./conv_small_probe_0 $shape; done
  ./conv_small_probe_0 $shape | sed 's/^/chosen: /'
done; done
