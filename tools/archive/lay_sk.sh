#!/bin/bash
# per-layer inference tables with stream-K off / on (gain threshold $1, default 0.01), 416x416 and 608x608 batch 64
R=$GRAFT_REPO_ROOT
g=${1:-0.01}
cd /tmp; export TMPDIR=/tmp
for s in 416 608; do for v in 0 1; do
VY_CONV_SK=$v VY_CONV_SK_GAIN=$g python3 $R/tools/layer_profile.py --size $s --out $R/gpurun_out/r03_lay_${s}_sk${v}.txt > /dev/null 2>&1
done; done
