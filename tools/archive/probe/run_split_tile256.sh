#!/bin/bash
# the 256 x 128 tile (8 waves, one block per CU) against 128 x 128 (4 waves, two blocks per CU) on the batch-64 shapes
P=tools/probe/conv_split_probe
mkdir -p gpurun_out
{
echo "== correctness (forced 256x128)"
VY_SPLIT_FORCE=256x128 timeout 120 $P 2 26 64 128 3 1 1 | tail -2
VY_SPLIT_FORCE=256x128 timeout 120 $P 2 27 64 256 3 2 | tail -2
VY_SPLIT_FORCE=256x128x3 timeout 120 $P 3 20 32 128 3 1 1 | tail -2
VY_SPLIT_FORCE=256x128x2 timeout 120 $P 2 19 256 128 1 | tail -2
for rep in 1 2; do
for f in 128x128 256x128; do
  echo "== $f"
  for shape in "64 76 128 256 3" "64 19 512 1024 3" "64 38 256 512 3" "64 152 64 128 3" "64 38 512 256 1" "64 19 1024 512 1" "16 52 128 256 3" "16 13 512 1024 3"; do
    VY_PROBE_SPLIT_ONLY=1 VY_SPLIT_FORCE=$f timeout 300 $P $shape 1 0 400 | grep "^\[" | tail -1 | cut -c1-60,118-
  done
done
done
} 2>&1 | tee gpurun_out/split_tile256.txt
