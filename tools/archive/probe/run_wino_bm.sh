#!/bin/bash
# the product's Winograd kernel at 64 / 128 pairs per block (VY_WINO_BM) on the batch-64 shapes, small and odd shapes for the check
mkdir -p gpurun_out
{
for bm in 64 128; do
  export VY_WINO_BM=$bm
  echo "== VY_WINO_BM=$bm"
  for shape in "64 38 256 512 0" "64 76 128 256 0" "64 76 128 256 1" "64 19 512 1024 0" "64 152 64 128 0" "16 52 128 256 1" "8 38 256 512 0" "3 13 64 128 0" "2 19 128 256 1" "1 7 32 128 0"; do
    timeout 120 tools/probe/wino_abl_probe $shape 40 | head -2
  done
done
} 2>&1 | tee gpurun_out/wino_bm.txt
