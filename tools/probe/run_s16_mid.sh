#!/bin/bash
# mid-size tiles of the 16x16x4 kernel against the 32x32x2 kernel on batch-1 / batch-2 layer shapes
cd $(dirname $0)
for shape in "1 76 128 256 3" "1 38 256 512 3" "1 19 512 1024 3" "1 152 64 128 3" "2 38 256 512 3" "2 19 512 1024 3"; do
  for t in 64x64 128x64; do VY_CONV_SMALL=0 VY_CONV_FORCE=$t ./conv_small_probe_0 $shape | sed 's/^/mfma32 /'; done
  for t in 64x64 96x64 64x96 96x96 128x64; do VY_CONV_S16=1 VY_CONV_FORCE=$t ./conv_small_probe_0 $shape | sed 's/^/s16    /'; done
done
