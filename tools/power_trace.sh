#!/bin/bash
# Samples the GPU's package power and shader clock (rocm-smi) while a command runs: evidence for the power-management
# ceiling of the sustained fp32 MFMA rate.   usage: tools/power_trace.sh out.txt <command...>
out=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  p=$(rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')
  c=$(rocm-smi --showclocks 2>/dev/null | grep sclk | grep -o '([0-9]*Mhz)' | tr -d '()')
  echo "$(date +%s.%N | cut -c1-14) power_W=$p sclk=$c" >> "$out"
  sleep 0.2
done
wait $pid
