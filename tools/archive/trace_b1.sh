cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_b1_trace -- python3 $GRAFT_REPO_ROOT/tools/b1_trace.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
f = sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_b1_trace/*/*kernel_trace.csv"))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id"), r.get("Stream_Id")) for r in csv.DictReader(open(f))]
rows.sort()
# last forward: find last bn_fold
idx = [i for i, r in enumerate(rows) if "bn_fold" in r[2]][-1]
t0 = rows[idx][0]
for s, e, n, q, st in rows[idx:idx + 90]:
    print("%8.1f %8.1f  q%s s%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, st, n))
PY
