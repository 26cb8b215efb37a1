"""Kernel totals of ONE frame of a batch-1 kernel trace (rocprofv3 --kernel-trace of tools/small_batch_latency.py --batches 1).
usage: python tools/frame_kernels.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'sort_nms' in e[2]]
fr = ev[idx[-3] + 1:idx[-2] + 1]
d = collections.defaultdict(lambda: [0, 0.0])
for e in fr:
    d[e[2]][0] += 1
    d[e[2]][1] += (e[1] - e[0]) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print("%-62s n=%3d total %7.1f us avg %6.1f" % (k, v[0], v[1], v[1] / v[0]))
print("frame: %d kernels, span %.1f us" % (len(fr), (fr[-1][1] - fr[0][0]) / 1e3))
