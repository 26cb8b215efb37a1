#!/bin/bash
# the product's Winograd kernel and its k-loop ablations (tools/probe/build_wino_abl.sh) on the batch-64 shapes
mkdir -p gpurun_out
{
for shape in "64 38 256 512" "64 76 128 256"; do
  for abl in "" _1 _2 _3 _4 _7 _8 _16 _24; do
    timeout 120 tools/probe/wino_abl_probe$abl $shape 0 40 | head -2
  done
done
timeout 120 tools/probe/wino_abl_probe 64 76 128 256 1 40 | head -2
timeout 120 tools/probe/wino_abl_probe 64 19 512 1024 0 40 | head -2
timeout 120 tools/probe/wino_abl_probe 64 152 64 128 0 40 | head -2
timeout 120 tools/probe/wino_abl_probe 3 13 64 128 0 10 | head -2
} 2>&1 | tee gpurun_out/wino_abl.txt
