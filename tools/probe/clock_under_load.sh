#!/bin/bash
# shader clock and socket power while ONE conv shape is launched back to back for ~10 s (VY_TRACE_REPS), sampled once a
# second from the 4th second on; last line of each group: the launch's average time over the whole loop
cd $(dirname $0)
while read REPS ARGS; do
  echo "== $ARGS  ($REPS launches)"
  VY_TRACE_REPS=$REPS ./conv_tile_trace $ARGS > /tmp/clk_probe.txt 2>&1 &
  L=$!
  sleep 5
  for s in 1 2 3 4; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Current Socket" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 1
  done
  wait $L
  head -1 /tmp/clk_probe.txt | sed 's/^conv [^|]*|/  /'
done <<'LIST'
40000 64 76 256 128 1 1 0 1
7000 64 76 128 256 3 1 0 1
7000 64 19 512 1024 3 1 0 1
7000 64 76 128 256 3 1 0 0
6000 64 304 32 64 3 1 1 1
LIST
