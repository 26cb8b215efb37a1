#!/bin/bash
# PROTOTYPE 1-D Winograd F(4, 3) (tools/probe/wino43_probe.hip) beside the product's F(2, 3) kernel (wino_abl_probe) on the
# batch-64 shapes; small and odd shapes for the check; all-zero operands for the stream's own rate
mkdir -p gpurun_out
{
for shape in "2 26 64 128 0" "1 19 32 64 1" "1 7 32 64 0" "3 13 64 128 1" "2 20 128 256 0"; do
  timeout 120 tools/probe/wino43_probe $shape 5 | head -2
done
for shape in "64 38 256 512 0" "64 76 128 256 0" "64 76 128 256 1" "64 19 512 1024 0" "64 152 64 128 0" "16 52 128 256 1"; do
  timeout 120 tools/probe/wino43_probe $shape 40 | head -2
  timeout 120 tools/probe/wino_abl_probe $shape 40 | head -1 | sed 's/^/   F(2,3) product: /'
done
for shape in "64 38 256 512 0" "64 19 512 1024 0"; do
  VY_PROBE_ZERO=1 timeout 120 tools/probe/wino43_probe $shape 40 | head -1 | sed 's/^/zero  /'
done
} 2>&1 | tee gpurun_out/wino43.txt
