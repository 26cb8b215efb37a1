#!/bin/bash
# every (block tile, plain | stream-K) on the conv shapes of the three workloads, against the cost model's own choice
# (first line of each group).  usage (GPU box): tools/probe/run_model_sweep.sh > gpurun_out/r03_model_sweep.txt
cd $(dirname $0)
while read args; do
  echo "== $args"
  ./conv_tile_trace $args | head -1 | sed 's/^conv [^|]*|/  model choice:/'
  for t in 128x128 128x64 64x64; do
    VY_CONV_SK=0 VY_CONV_FORCE=$t ./conv_tile_trace $args | head -1 | sed 's/^conv [^|]*|/  /'
    VY_CONV_SK=1 VY_CONV_SK_GAIN=-100 VY_CONV_FORCE=$t ./conv_tile_trace $args | head -1 | grep "sk tiles" | sed 's/^conv [^|]*|/  /'
  done
done <<'LIST'
64 76 128 256 3 1 0 1
64 38 256 512 3 1 0 1
64 19 512 1024 3 1 0 1
64 152 64 128 3 1 0 1
64 52 128 256 3 1 0 1
64 26 256 512 3 1 0 1
64 13 512 1024 3 1 0 1
16 52 128 256 3 1 0 1
16 26 256 512 3 1 0 1
16 13 512 1024 3 1 0 1
16 13 1024 512 1 1 0 1
16 26 512 256 1 1 0 1
16 52 256 128 1 1 0 1
8 38 256 512 3 1 0 1
8 19 512 1024 3 1 0 1
4 38 256 512 3 1 0 1
4 19 512 1024 3 1 0 1
2 76 128 256 3 1 0 1
LIST
