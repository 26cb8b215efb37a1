"""Summarise the last training step of a rocprofv3 --kernel-trace CSV (per kernel class and per layer
shape: wgrad / dgrad / fwd durations with the FLOP rate where the grid identifies the layer).
usage: python tools/trace_step.py gpurun_out/tt0/runc/*_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('sgd_kernel')]
step = rows[idx[-2] + 1: idx[-1] + 1]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("queues", dict(collections.Counter(r['Queue_Id'] for r in step)))
print("span %.2f ms, sum %.2f ms, n=%d" % ((int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6,
                                         sum(dur(r) for r in step) / 1e3, len(step)))
agg = collections.OrderedDict()
for r in step:
    a = agg.setdefault(r['Kernel_Name'][:72], [0, 0.0]); a[0] += 1; a[1] += dur(r)
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-74s %4d %9.1f us" % (k, n, d))
if len(sys.argv) > 2:
    for r in step:
        print("%-60s grid %8s x %4s  %8.1f us" % (r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'], dur(r)))
