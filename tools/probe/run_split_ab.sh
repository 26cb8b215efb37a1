#!/bin/bash
# A/B of probe builds on the big shapes: gpurun -- bash tools/probe/run_split_ab.sh bin1 bin2 ...
mkdir -p gpurun_out
{
for rep in 1 2; do
for P in "$@"; do
echo "== $P"
timeout 300 $P 64 76 128 256 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 19 512 1024 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 152 64 128 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
done
done
P=$1
echo "== 64-channel layers, $P"
timeout 300 $P 64 304 32 64 3 1 1 20 | tail -3
VY_SPLIT_FORCE=128x64 timeout 300 $P 64 304 32 64 3 1 1 20 | tail -3
timeout 300 $P 64 608 32 64 3 2 0 20 | tail -3
VY_SPLIT_FORCE=128x64 timeout 300 $P 64 608 32 64 3 2 0 20 | tail -3
echo "== 1x1 layers, $P"
timeout 300 $P 64 76 256 128 1 1 0 40 | tail -3
timeout 300 $P 64 38 512 256 1 1 0 40 | tail -3
timeout 300 $P 64 19 1024 512 1 1 0 40 | tail -3
timeout 300 $P 64 152 128 64 1 1 0 40 | tail -3
} 2>&1 | tee gpurun_out/split_ab.txt
