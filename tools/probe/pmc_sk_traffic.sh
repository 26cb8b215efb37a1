#!/bin/bash
# L2-miss traffic (FETCH_SIZE, counters only) of one conv shape as plain / stream-K launches on the same tile
cd $(dirname $0)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
run() { # tag env... 
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  ( export "$@"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$tag -- ./conv_tile_trace $ARGS > /dev/null 2>&1 )
  python3 - /tmp/pmc_$tag "$tag" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "conv_igemm" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("  %-12s %-62s %d launches  fetch %.0f MB per launch (x2 gfx950 correction applied)" % (sys.argv[2], k[:62], len(v), 2 * 1024 * sum(v) / len(v) / 1e6))
PY
}
for ARGS in "64 38 256 512 3 1 0 1" "64 19 512 1024 3 1 0 1" "64 76 128 256 3 1 0 1"; do
  export ARGS
  echo "== $ARGS"
  run plain128 VY_CONV_SK=0 VY_CONV_FORCE=128x128
  run sk128 VY_CONV_SK=1 VY_CONV_SK_GAIN=-100 VY_CONV_FORCE=128x128
  run plain128x64 VY_CONV_SK=0 VY_CONV_FORCE=128x64
done
