#!/bin/bash
# exact vs split over batch sizes on the three deep 3x3 shapes + a 1x1: where does the split kernel start to pay?
P=tools/probe/conv_split_probe
mkdir -p gpurun_out
{
for B in 1 2 4 8 16 32; do
  for shape in "19 512 1024 3" "38 256 512 3" "76 128 256 3" "152 64 128 3" "38 512 256 1" "19 1024 512 1"; do
    timeout 120 $P $B $shape 1 0 30 | grep "^conv" | tail -1 | cut -c1-40,60-
    VY_SPLIT_FORCE=128x64 timeout 120 $P $B $shape 1 0 30 | grep "^conv" | tail -1 | cut -c60- | sed 's/^/       128x64: /'
  done
done
} 2>&1 | tee gpurun_out/split_batch_sweep.txt
