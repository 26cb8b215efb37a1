#!/bin/bash
# A/B of builds of the split weight gradient probe (e.g. staged LDS waits against one wait): run_wgrad_staged.sh bin1 bin2
mkdir -p gpurun_out
{
for rep in 1 2; do
for P in "$@"; do
echo "== $P"
timeout 120 $P 2 13 128 256 3 1 0 3 | tail -1
for shape in "16 52 128 256 3" "16 26 256 512 3" "16 13 512 1024 3" "16 104 64 128 3" "16 52 256 128 1" "16 104 128 256 3 2"; do
timeout 300 $P $shape | grep "^wgrad" | tail -1 | cut -c1-50,95-
done
done
done
} 2>&1 | tee gpurun_out/wgrad_staged.txt
