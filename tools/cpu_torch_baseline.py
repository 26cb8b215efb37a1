#!/usr/bin/env python3
"""torch-CPU forward of the yolo3_darknet53 conv graph as a CPU datapoint beside the bench line (BASELINE.json's metric asks
for "the reference's own MXNet CPU path timed on the same box's host cores (core count stated)"; mxnet cannot be installed
here, so this is an INDEPENDENT datapoint — oneDNN convolutions through torch — NOT MXNet and not the checker).

Run as a CHILD of bench.py (never touches a GPU) so that the OpenMP runtime starts with the settings a CPU benchmark
wants, which cannot be changed in a process that has already spun up its thread pool:
    OMP_NUM_THREADS = physical cores in this process's cpuset (SURVEY section 8d's recipe), OMP_PROC_BIND=close,
    OMP_PLACES=cores; one batch of 8 frames (the recipe's batch), channels_last, a warm-up pass at the SAME shape (oneDNN
    builds and caches its primitives and re-orders the weights on the first call of a shape — the round-5 figure timed that)
and prints ONE JSON line: frames/s, TFLOP/s, the threads actually used, and what bounds them on this host — CPU model,
logical / physical CPUs, the cpuset, the cgroup CPU quota, torch's CPU capability (the oneDNN ISA).

usage: cpu_torch_baseline.py PARAMS.npz FRAMES.npy [--classes 20] [--passes 3]
"""
import argparse
import json
import os
import sys
import time


def physical_cores(allowed):
    """(physical cores among the CPUs in `allowed`, logical CPUs in `allowed`)"""
    seen = set()
    try:
        for cpu in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            with open(base + "physical_package_id") as f:
                pkg = f.read().strip()
            with open(base + "core_id") as f:
                core = f.read().strip()
            seen.add((pkg, core))
    except OSError:
        return len(allowed), len(allowed)
    return max(1, len(seen)), len(allowed)


def cgroup_quota():
    """CPU bandwidth limit of this process's cgroup in CPUs (None: unlimited / unknown)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                     # cgroup v2
            q, p = f.read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:       # cgroup v1
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        return None if q <= 0 else q / float(p)
    except (OSError, ValueError):
        return None


def host_facts():
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    phys, logical = physical_cores(allowed)
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "cpu_count": os.cpu_count(), "cpuset": len(allowed), "physical_cores_in_cpuset": phys,
            "cfs_quota_cpus": cgroup_quota()}


def child_env(env=None):
    """The environment bench.py starts this script with: one thread per physical core of the cpuset (capped by a cgroup
    quota), bound to cores."""
    env = dict(os.environ if env is None else env)
    h = host_facts()
    n = h["physical_cores_in_cpuset"]
    if h["cfs_quota_cpus"]:
        n = max(1, min(n, int(h["cfs_quota_cpus"])))
    env.update(OMP_NUM_THREADS=str(n), MKL_NUM_THREADS=str(n), OMP_PROC_BIND="close", OMP_PLACES="cores",
               OMP_WAIT_POLICY="active")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def build_forward(params, classes):
    import torch
    import torch.nn.functional as F
    cl = torch.channels_last
    p = {k: (torch.from_numpy(v).contiguous(memory_format=cl) if v.ndim == 4 else torch.from_numpy(v)) for k, v in params.items()}

    def cell(t, pre, k, s):
        t = F.conv2d(t, p[pre + ".0.weight"], None, s, k // 2)
        t = F.batch_norm(t, p[pre + ".1.running_mean"], p[pre + ".1.running_var"], p[pre + ".1.gamma"], p[pre + ".1.beta"],
                         False, 0.9, 1e-5)
        return F.leaky_relu(t, 0.1)

    def forward(t):
        routes = []
        feats = [("c", 1)]
        for n in (1, 2, 8, 8, 4):
            feats += [("c", 2)] + [("b", 0)] * n
        for si, (lo, hi) in enumerate([(0, 15), (15, 24), (24, 29)]):
            for j, f in enumerate(feats[lo:hi]):
                pre = "stages.%d.%d" % (si, j)
                if f[0] == "c":
                    t = cell(t, pre, 3, f[1])
                else:
                    t = t + cell(cell(t, pre + ".body.0", 1, 1), pre + ".body.1", 3, 1)
            routes.append(t)
        outs, t = [], routes[2]
        for i in range(3):
            for j in range(5):
                t = cell(t, "yolo_blocks.%d.body.%d" % (i, j), 1 if j % 2 == 0 else 3, 1)
            tip = cell(t, "yolo_blocks.%d.tip" % i, 3, 1)
            outs.append(F.conv2d(tip, p["yolo_outputs.%d.prediction.weight" % i], p["yolo_outputs.%d.prediction.bias" % i]))
            if i == 2:
                break
            t = F.interpolate(cell(t, "transitions.%d" % i, 1, 1), scale_factor=2, mode="nearest")
            r = routes[1 - i]
            t = torch.cat([t[:, :, :r.shape[2], :r.shape[3]], r], 1)
        return outs
    return forward


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("params")
    ap.add_argument("frames")
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--gflop-per-frame", type=float, default=0.0)
    a = ap.parse_args()
    facts = host_facts()   # BEFORE the OpenMP runtime binds this thread to its core (OMP_PROC_BIND): the cpuset as inherited
    import numpy as np
    import torch
    params = dict(np.load(a.params))
    x = torch.from_numpy(np.load(a.frames)).contiguous(memory_format=torch.channels_last)
    fwd = build_forward(params, a.classes)
    with torch.no_grad():
        t0 = time.perf_counter()
        fwd(x)                                   # warm-up at the SAME shape: primitive creation + weight re-orders
        warm = time.perf_counter() - t0
        ts = []
        for _ in range(a.passes):
            t0 = time.perf_counter()
            fwd(x)
            ts.append(time.perf_counter() - t0)
    best = min(ts)
    n = int(x.shape[0])
    out = {"value": n / best, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "independent",
           "batch": n, "seconds_per_pass": ts, "warmup_pass_s": warm,
           "fp32_tflops": n / best * a.gflop_per_frame / 1e3 if a.gflop_per_frame else None,
           "host": dict(facts, torch_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                        onednn=torch.backends.mkldnn.is_available(),
                        omp={k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")}),
           "sample": "%d frames of the same batch as ONE batch (SURVEY 8d's recipe: batch 8, threads = physical cores), best of %d "
                     "passes after a warm-up pass at the same shape, through torch-CPU conv2d (oneDNN, channels_last) / batch_norm / "
                     "leaky_relu of the same 75-conv graph (no decode / NMS); an independent CPU datapoint, NOT MXNet and not "
                     "the checker" % (n, a.passes)}
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
