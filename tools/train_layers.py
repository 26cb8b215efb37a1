"""Per-layer table of the matrix-core launches of one training step: joins a rocprofv3 kernel trace with the launch
labels the library writes under VY_TRAIN_LABELS (kind, cell, FLOPs, GEMM dims) by launch order within each kernel
class.  usage: python tools/train_layers.py <kernel_trace.csv> <labels.txt> [--csv]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('sgd_kernel')]
step = rows[idx[-2] + 1: idx[-1] + 1]
labels = collections.defaultdict(list)
for line in open(sys.argv[2]):
    if line.startswith("#"):
        continue
    kind, name, fl, m, n, k = line.split()
    labels[kind].append((name, float(fl), int(float(m)), int(float(n)), int(float(k))))
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
cls = {"fwd": [], "dgrad": [], "wgrad": []}
for r in step:
    n = r['Kernel_Name']
    if n.startswith("void wgrad_kernel"):
        cls["wgrad"].append(r)
    elif n.startswith("void conv_igemm_kernel"):
        targs = n.split("(")[0].split("<")[1].rstrip(">").replace(" ", "").split(",")  # BM,BN,WM,WN,DGRAD[,NS]
        cls["dgrad" if targs[4] == "true" else "fwd"].append(r)
tot = {}
for kind in ("fwd", "dgrad", "wgrad"):
    assert len(cls[kind]) == len(labels[kind]), (kind, len(cls[kind]), len(labels[kind]))
    t_us = fl_sum = 0.0
    print("---- %s" % kind)
    for r, (name, fl, m, n, k) in zip(cls[kind], labels[kind]):
        d = dur(r)
        t_us += d
        fl_sum += fl
        tile = r['Kernel_Name'].split("<")[1].split(">")[0].replace(" ", "")
        print("%-28s M %7d N %5d K %5d  grid %6s x %-3s %-16s %8.1f us %6.1f TF" % (
            name, m, n, k, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), tile, d, fl / d / 1e6))
    print("%s total %.2f ms  %.1f TF" % (kind, t_us / 1e3, fl_sum / t_us / 1e6))
