#!/bin/bash
# kernel-time totals of the training step, exact vs split conv mode (rocprofv3 --kernel-trace --stats, 5 steps each)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp
for mode in exact split_bf16x3_train; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trn_$mode -- python3 $R/bench.py --mode train --conv-mode $mode --steps 5 --warmup 2 --no-pmc --no-roofline > /dev/null 2>&1
  f=$(ls $O/trn_$mode/*/*kernel_stats.csv | tail -1)
  echo "== $mode"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-86s calls %5s  total %8.2f ms  avg %8.1f us" % (r["Name"][:86], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
print("TOTAL %.1f ms" % (tot / 1e6))
PY
done
