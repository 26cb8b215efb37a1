"""Where does a training step's wall time go?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --mode train` and,
for the last steps, reports: the union of kernel intervals (device busy), the time no kernel runs (launch gaps and
dependencies), and per kernel family the exclusive time (alone on the device) vs time shared with another queue.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --mode train --steps 6 --warmup 2 --no-pmc
    python tools/step_timeline.py gpurun_out/tl
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?\(", name)
    if not m:
        return name[:40]
    base, targs = m.group(1), m.group(2) or ""
    if base == "conv_igemm_kernel":
        return "conv dgrad" if "true" in targs else "conv fwd"
    return base


def main(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    # a step ends with sgd_kernel: take the span between the 2nd last and the last one
    sgd = [i for i, r in enumerate(rows) if r[2] == "sgd_kernel"]
    if len(sgd) < 3:
        print("need >= 3 steps in the trace")
        return
    lo, hi = rows[sgd[-3]][1], rows[sgd[-1]][1]
    step = [r for r in rows if r[0] >= lo and r[1] <= hi]
    nsteps = 2
    ev = []
    for s, e, n, q in step:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    prev = lo
    busy = idle = 0
    excl = defaultdict(float)
    shared = defaultdict(float)
    gaps = []
    for t, d_, n in ev:
        dt = t - prev
        names = [k for k, v in active.items() if v > 0]
        total = sum(active.values())
        if total == 0:
            idle += dt
            if dt > 0:
                gaps.append(dt)
        else:
            busy += dt
            for k in names:
                (excl if total == 1 else shared)[k] += dt
        active[n] += d_
        prev = t
    idle += hi - prev
    span = hi - lo
    print("steps analysed: %d, %.3f ms per step; device busy %.1f %%, idle %.1f %% (%.3f ms per step in %d gaps, median %.1f us)"
          % (nsteps, span / nsteps / 1e6, 100.0 * busy / span, 100.0 * idle / span, idle / nsteps / 1e6, len(gaps) // nsteps,
             sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
    print("%-34s %10s %10s   (ms per step)" % ("kernel", "alone", "overlapped"))
    keys = sorted(set(excl) | set(shared), key=lambda k: -(excl[k] + shared[k]))
    for k in keys[:24]:
        print("%-34s %10.3f %10.3f" % (k, excl[k] / nsteps / 1e6, shared[k] / nsteps / 1e6))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tl")
