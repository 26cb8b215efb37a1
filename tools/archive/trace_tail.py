"""Per-frame timeline of a batch-1 kernel trace (rocprofv3 --kernel-trace of tools/small_batch_latency.py --batches 1):
span and busy time of the last frames and the durations of the detection-stage kernels.
usage: python tools/trace_tail.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:28]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'sort_nms' in e[2]]
for a, b in list(zip(idx[-6:-1], idx[-5:])):
    fr = ev[a + 1:b + 1]
    print("frame: %d kernels, span %.1f us, busy %.1f us; tail: %s" % (
        len(fr), (fr[-1][1] - fr[0][0]) / 1e3, sum(e[1] - e[0] for e in fr) / 1e3,
        "  ".join("%s %.1f" % (e[2].split('(')[0], (e[1] - e[0]) / 1e3) for e in fr[-6:])))
