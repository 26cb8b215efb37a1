#!/bin/bash
# shader clock and socket power while ONE split-conv shape runs back to back on a forced tile (~5 s per round, sampled in
# the second round): does the smaller tile's lower rate come with the same clock (structural) or a lower one (power)?
cd $(dirname $0)
for cfg in "128x128 5000" "128x64 4000" "256x64 4000"; do
  set -- $cfg
  for shape in "64 76 128 256 3" "64 19 512 1024 3"; do
    echo "== tile $1  shape $shape"
    VY_PROBE_SPLIT_ONLY=1 VY_SPLIT_FORCE=$1 ./conv_split_probe $shape 1 0 $2 > /tmp/clk_split.txt 2>&1 &
    L=$!
    sleep 8
    for s in 1 2 3; do
      /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed 's/.*: //' | tr '\n' ' '; echo
      sleep 1
    done
    wait $L
    grep "^\[" /tmp/clk_split.txt | cut -c1-14,90-
  done
done
