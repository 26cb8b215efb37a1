#!/bin/bash
# split-K sweep on small launches: forced (tile, k-split) against the cost model's own choice and the exact kernel
P=tools/probe/conv_split_probe
mkdir -p gpurun_out
{
for B in 1 2 4; do
  for shape in "19 512 1024 3" "38 256 512 3" "76 128 256 3" "152 64 128 3" "19 1024 512 1" "38 512 256 1" "76 256 128 1"; do
    echo "--- B=$B $shape"
    timeout 120 $P $B $shape 1 0 30 | grep -E "^\[|float64" | tail -2 | cut -c1-12,100-
    for f in 128x128x2 128x128x4 128x128x8 128x128x16 128x64x2 128x64x4 128x64x8 128x64x16; do
      VY_SPLIT_FORCE=$f timeout 120 $P $B $shape 1 0 30 | grep "^\[" | tail -1 | cut -c1-14,100- | sed "s/^/   /"
    done
  done
done
} 2>&1 | tee gpurun_out/split_ksplit_sweep.txt
