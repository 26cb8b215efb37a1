#!/bin/bash
# the 64x64 tile of the split conv on small launches: forced (tile, k-split) against the cost model's own choice
P=tools/probe/conv_split_probe
mkdir -p gpurun_out
{
echo "== correctness (forced 64x64)"
VY_SPLIT_FORCE=64x64 timeout 120 $P 2 26 64 128 3 1 1 | tail -2
VY_SPLIT_FORCE=64x64 timeout 120 $P 2 27 64 256 3 2 | tail -2
VY_SPLIT_FORCE=64x64x3 timeout 120 $P 3 20 32 64 3 1 1 | tail -2
VY_SPLIT_FORCE=64x64x2 timeout 120 $P 2 19 256 128 1 | tail -2
for B in 1 2 4; do
  for shape in "19 512 1024 3" "38 256 512 3" "76 128 256 3" "152 64 128 3" "304 32 64 3" "19 1024 512 1" "38 512 256 1" "76 256 128 1" "152 128 64 1"; do
    echo "--- B=$B $shape"
    timeout 120 $P $B $shape 1 0 30 | grep -E "^\[" | tail -1 | cut -c1-14,100-
    for f in 64x64x1 64x64x2 64x64x3 64x64x4 64x64x6 64x64x8; do
      VY_SPLIT_FORCE=$f timeout 120 $P $B $shape 1 0 30 | grep "^\[" | tail -1 | cut -c1-14,100- | sed "s/^/   /"
    done
  done
done
} 2>&1 | tee gpurun_out/split_tile64.txt
