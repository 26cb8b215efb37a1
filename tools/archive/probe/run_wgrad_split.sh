#!/bin/bash
# split-fp32 weight gradient against the exact one: small correctness shapes, then the 416x416 batch-16 training shapes
P=tools/probe/wgrad_split_probe
mkdir -p gpurun_out
{
timeout 120 $P 2 8 32 128 3 1 2 3
timeout 120 $P 2 13 128 256 3 1 0 3
timeout 120 $P 2 16 64 128 3 2 0 3
timeout 120 $P 2 13 256 128 1 1 0 3
timeout 300 $P 16 52 128 256 3
timeout 300 $P 16 26 256 512 3
timeout 300 $P 16 13 512 1024 3
timeout 300 $P 16 104 64 128 3
timeout 300 $P 16 52 256 128 1
timeout 300 $P 16 104 128 256 3 2
} 2>&1 | tee gpurun_out/wgrad_split_probe.txt
