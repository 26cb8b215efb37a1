#!/bin/bash
# is the Winograd kernel held by the power cap?  the same launches on all-zero operands (same instruction stream)
mkdir -p gpurun_out
{
for bm in 64 128; do
  export VY_WINO_BM=$bm
  echo "== VY_WINO_BM=$bm"
  for shape in "64 38 256 512 0" "64 76 128 256 1" "64 19 512 1024 0"; do
    timeout 120 tools/probe/wino_abl_probe $shape 40 | head -1
    VY_PROBE_ZERO=1 timeout 120 tools/probe/wino_abl_probe $shape 40 | head -1 | sed 's/^/zero  /'
    VY_PROBE_SMALLW=1 timeout 120 tools/probe/wino_abl_probe $shape 40 | head -1 | sed 's/^/bf16w /'
  done
done
} 2>&1 | tee gpurun_out/wino_zero.txt
