#!/bin/bash
# gpurun -- bash tools/probe/run_split_probe.sh   (binary built in-tree beforehand: tools/probe/build_split_probe.sh)
P=${P:-tools/probe/conv_split_probe}
mkdir -p gpurun_out
{
timeout 120 $P 2 26 64 128 3 1 1
timeout 300 $P 64 76 128 256 3
timeout 300 $P 64 76 128 256 3 1 1
timeout 300 $P 64 38 256 512 3
timeout 300 $P 64 19 512 1024 3
timeout 300 $P 64 152 64 128 3
timeout 300 $P 64 152 128 256 3 2
timeout 300 $P 16 52 128 256 3
timeout 300 $P 1 76 128 256 3
} 2>&1 | tee gpurun_out/split_probe.txt
