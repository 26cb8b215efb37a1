#!/bin/bash
# the Winograd F(2,3) prototype (two block shapes) beside the product's split conv on the batch-64 3x3 shapes
mkdir -p gpurun_out
{
for P in tools/probe/conv_wino_probe tools/probe/conv_wino_probe_64 "v2"; do
  if [ "$P" = v2 ]; then export WINO_V2=1; P=tools/probe/conv_wino_probe_64; fi   # two xi per k-step
  timeout 60 $P 2 26 64 128 3 | tail -2
  timeout 60 $P 1 19 32 128 3 | tail -2
  for shape in "64 76 128 256" "64 38 256 512" "64 152 64 128" "64 19 512 1024" "16 52 128 256"; do timeout 120 $P $shape 40 | tail -2 | head -1; done
done
unset WINO_V2
for shape in "64 76 128 256 3" "64 38 256 512 3" "64 152 64 128 3" "64 19 512 1024 3" "16 52 128 256 3"; do
  VY_PROBE_SPLIT_ONLY=1 timeout 300 tools/probe/conv_split_probe $shape 1 0 40 | grep "^\[" | tail -1 | cut -c1-60,118-
done
} 2>&1 | tee gpurun_out/wino_probe.txt
