#!/bin/bash
# A/B of probe builds on the big shapes: gpurun -- bash tools/probe/run_split_ab.sh bin1 bin2 ...
mkdir -p gpurun_out
{
for P in "$@"; do
echo "== correctness $P"
timeout 120 $P 2 26 64 128 3 1 1 | tail -2
timeout 120 $P 2 27 64 256 3 2 | tail -2
timeout 120 $P 3 20 32 64 3 1 1 | tail -2
timeout 120 $P 2 19 256 128 1 | tail -2
done
for rep in 1 2; do
for P in "$@"; do
echo "== $P"
timeout 300 $P 64 76 128 256 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 19 512 1024 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 152 64 128 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 304 32 64 3 1 1 20 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 64 38 512 256 1 1 0 40 | grep "^conv" | tail -1 | cut -c60-
timeout 300 $P 16 52 128 256 3 1 0 40 | grep "^conv" | tail -1 | cut -c60-
done
done
} 2>&1 | tee gpurun_out/split_ab.txt
