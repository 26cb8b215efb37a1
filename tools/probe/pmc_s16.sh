#!/bin/bash
# SQ counters of the 16x16x4 kernel vs the 32x32x2 kernel at the same 64x64 tile on the batch-1 19x19 layer
cd /tmp; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/probe
O=$GRAFT_REPO_ROOT/gpurun_out
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
for k in s16 m32; do
  if [ $k = s16 ]; then export VY_CONV_S16=1; else unset VY_CONV_S16; export VY_CONV_SMALL=0; fi
  export VY_CONV_FORCE=64x64
  rocprofv3 --pmc $C1 --output-format csv -d $O/pmc_${k}_1 -- $P/conv_small_probe_0 1 19 512 1024 3 > /dev/null 2>&1
  rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_${k}_2 -- $P/conv_small_probe_0 1 19 512 1024 3 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
for k in ("m32", "s16"):
    agg = collections.defaultdict(float); n = 0
    for p in (1, 2):
        f = glob.glob("%s/pmc_%s_%d/*/*counter_collection.csv" % (O, k, p))[0]
        disp = set()
        for r in csv.DictReader(open(f)):
            if "conv" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        n = len(disp)
    print(k, "launches", n)
    for c in sorted(agg): print("   %-28s %14.0f per launch" % (c, agg[c] / n))
PY
