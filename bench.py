#!/usr/bin/env python3
"""bench.py — frames/sec of the yolo3_darknet53 hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 608] [--batch 64] [--classes 20]

One "step" = one inference pass of the whole path (stem + 74 fused conv launches + decode + NMS)
over one batch of synthetic frames that is already resident in HBM.  Default workload is
BASELINE.json configs[1]: 608x608, batch 64 per GPU, 20 classes, fp32, random-init weights
(videoyolo_amd.init 'synthetic', seed 233), N(0,1) frames.  With --gpus N there is one
process per GPU: either the driver starts them (torch.distributed.run) or, run plainly as
`python bench.py --gpus N`, this script starts its own N rank processes; frames are scattered, every rank runs its own batch
(weak scaling, no data-path collective — SURVEY §8e "Inference"), the timed region is bracketed
by barrier + synchronize and the max over ranks is reported.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline        the dominant kernel (the 128x128 implicit-GEMM conv) against the fp32 MFMA peak,
                  from HIP events recorded around every launch of one extra, un-timed pass
  cpu_baseline    the CPU oracle (oracle/, a port — NOT MXNet) timed on a bounded sample
  cpu_baseline_torch  torch-CPU forward of the same conv graph (independent datapoint, NOT MXNet)
  also_infer608_split  the SAME inference step in the opt-in conv mode 'split_bf16x3' (csrc/conv_split.hip: bf16 x 3, six
                  products, fp32 accumulate — NOT the parity path; tolerances in tests/test_gpu_split.py): frames/s,
                  conv rate against the fp32 roof and against the bf16 roof / 6, batch-1 latency, HBM traffic.
                  The headline `value` stays the exact-fp32 path.
  also_train416_split  BASELINE configs[2] in conv mode split_bf16x3_train (recorded forward, data gradients and weight
                  gradients on the bf16 matrix core); separately reported, like also_infer608_split
  also_train416   BASELINE configs[2] at the same N: training step 416x416, 16 frames per GPU (recorded forward +
                  backward + gradient all-reduce over RCCL, bucketed and overlapped + SGD), timed the same way
                  (barrier + synchronize, max over ranks) — with the forward / backward / exposed-all-reduce split,
                  the fraction of the fp32 MFMA roof and (N = 1) the HBM bytes per step
  also_syncbn608  (N > 1) BASELINE configs[4]: 608x608, 8 frames per GPU, the net built with
                  norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': N} (train_yolov3.py:350-354), plus the
                  246.5 MB gradient all-reduce timed alone and the fraction of it hidden behind backward
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# under the driver's `python -m torch.distributed.run ... bench.py` the ranks are NOT started by videoyolo_amd.launch:
# RCCL across processes needs dmabuf IPC on this host driver, and the HSA runtime reads this when the first GPU call
# initialises it — set it before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2516.6  # v_mfma_f32_32x32x16_bf16: 16x the fp32 rate (same guide); the split path spends
SPLIT_PRODUCTS = 6              # six bf16 products per fp32 multiply -> fp32-equivalent roof 2516.6 / 6 = 419.4
HBM_PEAK_GBS = 8000.0
# xGMI: 7 point-to-point links per GPU, ~153.6 GB/s each counting both directions (the task statement's figure) = 76.8 GB/s
# per direction.  What an all-reduce of S bytes over N fully connected GPUs can reach:
#   one ring over one link per neighbour          bus bandwidth <= 76.8 GB/s
#   reduce-scatter + all-gather over ALL N - 1 links at once (each GPU sends S / N to every peer, twice)
#                                                  bus bandwidth <= (N - 1) x 76.8 GB/s  (537.6 at N = 8)
XGMI_LINK_GBPS_PER_DIRECTION = 76.8
ROCPROF = "/opt/rocm/bin/rocprofv3"


def _short_kernel_name(name):
    """kernel name without its trailing argument list"""
    name = name.strip()
    if not name.endswith(")"):
        return name
    depth = 0
    for i in range(len(name) - 1, -1, -1):
        depth += name[i] == ")"
        depth -= name[i] == "("
        if depth == 0:
            return name[:i]
    return name


def _forward_tile_keys(names, bm, bn, streamk=False):
    """The counter file's names of the forward, double-buffered instances of a tile:
    conv_igemm_kernel<BM, BN, WM, WN, DGRAD = false, NS = 2, SK = streamk[, CK]> (older builds: 5, 6 or 7 arguments).  Two
    since round 6: the launches whose K is summed in runs have their own instance (CK = true)."""
    pre = "void conv_igemm_kernel<%s, %s," % (bm, bn)
    out = []
    for k in names:
        args = k[k.find("<") + 1:k.rfind(">")].replace(" ", "").split(",") if k.startswith(pre) else []
        if len(args) >= 5 and args[4] == "false" and (len(args) == 5 or args[5] == "2") \
                and (len(args) > 6 and args[6] == "true") == bool(streamk):
            out.append(k)
    return out


def _forward_tile_key(names, bm, bn, streamk=False):
    keys = _forward_tile_keys(names, bm, bn, streamk)
    return keys[0] if keys else None


_PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTRACER_", "HSA_TOOLS_", "ROCTX_")


def under_profiler(env=None):
    """True when this process runs inside rocprofv3 (or any HSA tool): its preloaded library has already
    initialised the GPU, and a nested `rocprofv3 --pmc ... -- python` started from here would be a launcher that
    re-execs under the outer profiler's LD_PRELOAD — the exec hop the pool forbids — and would mix counter
    collection with the outer run's tracing."""
    env = os.environ if env is None else env
    if any(k.startswith(_PROFILER_ENV_PREFIXES) for k in env):
        return True
    return any(t in env.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler"))


def clean_child_env(env):
    """The environment for a rocprofv3 child: no profiler settings inherited from whatever wraps this process."""
    out = {k: v for k, v in env.items() if not k.startswith(_PROFILER_ENV_PREFIXES)}
    pre = [t for t in out.get("LD_PRELOAD", "").split(":") if t and not any(
        w in t for w in ("rocprof", "roctracer", "rocprofiler"))]
    if pre:
        out["LD_PRELOAD"] = ":".join(pre)
    else:
        out.pop("LD_PRELOAD", None)
    return out


def measure_hbm_traffic(argv, steps_run):
    """HBM traffic of this very command, measured now: two extra child runs of bench.py under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only, no tracing — the
    recipe of MI355X_MICROARCH.md, section HBM), started BEFORE this process touches the GPU.  Counters are KiB per
    dispatch; gfx950 tallies the 128-B read requests of wide coalesced loads at 64 B, so FETCH_SIZE is doubled;
    WRITE_SIZE is taken as is.  Returns {kernel: {"launches", "fetch_bytes", "write_bytes", "hbm_bytes"}} per
    LAUNCH (mean over the dispatches of the run), plus "_per_step" = bytes of all library kernels per step."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not os.path.exists(ROCPROF):
        return None, "rocprofv3 not found"
    if under_profiler():
        return None, "this process is itself being profiled (LD_PRELOAD / ROCP* in the environment): no nested rocprofv3"
    out = {}
    tmp = tempfile.mkdtemp(prefix="vy_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            env = dict(clean_child_env(os.environ), VY_BENCH_CHILD="1", TMPDIR="/tmp")
            cmd = [ROCPROF, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + argv
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
            files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if p.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, p.returncode, p.stderr.decode()[-300:])
            agg, disp = {}, {}
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] != counter:
                    continue
                k = _short_kernel_name(r["Kernel_Name"])
                agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
                disp.setdefault(k, set()).add(r["Dispatch_Id"])
            for k, v in agg.items():
                e = out.setdefault(k, {"launches": len(disp[k]), "fetch_bytes": 0.0, "write_bytes": 0.0})
                if counter == "FETCH_SIZE":
                    e["fetch_bytes"] = 2.0 * v * 1024.0 / len(disp[k])
                else:
                    e["write_bytes"] = v * 1024.0 / len(disp[k])
    except Exception as e:  # the measurement is best effort: the bench line says so instead of failing
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per_step = 0.0
    for k, e in out.items():
        e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
        if not (k.startswith("__amd_rocclr") or "at::native" in k):
            per_step += e["hbm_bytes"] * e["launches"] / float(steps_run)
    out["_per_step"] = per_step
    return out, None


SPLIT_DTYPE = "f32 via bf16x3 (6 products), f32 acc"


_JSON_FD = None


class LegWatchdog(object):
    """A wall-clock budget per leg.  A leg of a multi-rank run that stops making progress (a collective whose peer never
    arrives, a fabric that is mis-set) would otherwise hold the whole run until the process group's timeout — and take the
    already measured headline with it, because the contract's ONE line is printed at the end.  Every rank arms the same
    budget at the start of every leg; when it runs out, rank 0 prints the line with everything measured so far, the current
    leg as {"timeout": true} and the legs not reached as absent, and every rank leaves (exit code 0 when the headline
    `value` is in the line, 4 otherwise).  Nothing is killed by pattern and nothing execs: the process ends itself."""

    def __init__(self, rank, budget_s, result):
        import threading
        self.rank, self.budget, self.result = rank, float(budget_s), result
        self.leg, self.deadline = None, None
        self.walls, self.t_leg, self.t0 = {}, None, time.time()   # wall-clock seconds per leg, for the line's `wall_s`
        self.lock = threading.Lock()
        if self.budget > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def _close(self, now):
        if self.leg is not None and self.t_leg is not None:
            self.walls[self.leg] = round(self.walls.get(self.leg, 0.0) + now - self.t_leg, 2)

    def arm(self, leg, scale=1.0):
        with self.lock:
            now = time.time()
            self._close(now)
            self.leg, self.deadline, self.t_leg = leg, now + self.budget * scale, now

    def disarm(self):
        with self.lock:
            self._close(time.time())
            self.leg, self.deadline, self.t_leg = None, None, None

    def wall_report(self):
        """{"since_start": seconds since the bench built its watchdog (after import torch + init), "legs": {leg: seconds}}:
        where the wall-clock time of THIS command went, so that the driver's own clock around the run can be reconciled"""
        with self.lock:
            return {"since_start": round(time.time() - self.t0, 2), "legs": dict(self.walls)}

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                leg, dl = self.leg, self.deadline
            if dl is not None and time.time() > dl:
                self.fire(leg)

    def fire(self, leg):
        sys.stderr.write("bench.py: leg %r exceeded its wall-clock budget of %.0f s on rank %d: reporting what was measured\n"
                         % (leg, self.budget, self.rank))
        if self.rank == 0:
            for _ in range(5):
                try:
                    out = dict(self.result)
                    out[leg if leg != "headline" else "headline_leg"] = {"timeout": True, "budget_s": self.budget}
                    out["aborted_after_timeout_of"] = leg
                    out["wall_s"] = {"since_start": round(time.time() - self.t0, 2), "legs": dict(self.walls)}
                    _emit(out)
                    break
                except RuntimeError:   # the main thread was adding to the dict: take another copy
                    time.sleep(0.05)
        os._exit(0 if "value" in self.result else 4)


def device_under_load(torch, dev, run_step, steps=4, period_s=0.02):
    """Clock, package power and temperature of the device WHILE the step runs: `steps` extra un-timed steps with a thread
    sampling torch.cuda.clock_rate / power_draw / temperature (amdsmi) every 20 ms.  Boxes of a pool differ by a few percent
    in what they sustain under this load (the same build read 945 - 987 frames/s during round 6); these figures let a reader
    tell a slower box from a slower build.  None when the runtime has no such query."""
    import threading
    idx = dev.index if dev.index is not None else 0
    try:
        torch.cuda.clock_rate(idx)
    except Exception as e:   # no amdsmi in this runtime
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:80])}
    clk, pwr, tmp, stop = [], [], [], threading.Event()

    def sample():
        while not stop.is_set():
            try:
                clk.append(torch.cuda.clock_rate(idx))
                pwr.append(torch.cuda.power_draw(idx))
                tmp.append(torch.cuda.temperature(idx))
            except Exception:
                return
            time.sleep(period_s)

    th = threading.Thread(target=sample, daemon=True)
    torch.cuda.synchronize(dev)
    th.start()
    for _ in range(steps):
        run_step()
    torch.cuda.synchronize(dev)
    stop.set()
    th.join(timeout=2.0)
    if not clk:
        return {"error": "no samples"}
    med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
    return {"sclk_mhz_median": med(clk), "sclk_mhz_min": min(clk), "sclk_mhz_max": max(clk), "power_median": med(pwr),
            "power_unit": "as torch.cuda.power_draw reports it (W on this image; mW upstream)", "temperature_c_max": max(tmp),
            "samples": len(clk), "steps": steps, "source": "torch.cuda.clock_rate / power_draw / temperature (amdsmi), sampled "
            "every %d ms during %d extra un-timed steps of the headline workload" % (int(period_s * 1e3), steps)}


def _emit(result):
    """the one line of the contract, on the process's REAL stdout (main() points fd 1 at stderr: library chatter)"""
    line = (json.dumps(result) + "\n").encode()
    sys.stdout.flush()
    if _JSON_FD is None:
        os.write(1, line)
    else:
        os.write(_JSON_FD, line)


def launch_table(net, x):
    """Median-of-3 per-launch table of one forward: [(name, ms, flops, algorithmic bytes)] (HIP events around every
    launch, on the launch stream) and the same aggregated per kernel instance {key: [launches, ms, flops, bytes]}."""
    passes = [net.profile(x) for _ in range(3)]
    med = []
    for j in range(len(passes[0])):
        ms = sorted(p[j][1] for p in passes)[1]
        med.append((passes[0][j][0], ms, passes[0][j][2], passes[0][j][3]))
    agg = {}
    for name, ms, fl, by in med:
        if "|split" in name:  # "<cell>|split<BM>x<BN>": the bf16 x 3 instance (conv_split.hip)
            key = "conv_split_kernel<%s>" % name.split("|split")[1]
        elif "|wino" in name:  # "<cell>|wino64x128": the same arithmetic as Winograd F(2, 3) (conv_wino.hip)
            key = "conv_wino_kernel<%s>" % name.split("|wino")[1]
        elif "|" in name:  # conv launches are reported as "<cell>|<BM>x<BN>"
            key = "conv_igemm_kernel<%s>" % name.split("|")[1]
        else:
            key = "stem_kernel" if name == "stages.0.0" else name
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += fl
        a[3] += by
    return med, agg


def batch1_latency(net, x1, torch, size, classes):
    """Single-frame latency (the reference's default detect call: batch_size 1, detect_yolo3.py:55,209-222): eager and
    as a replayed HIP graph."""
    lat = {}
    for label, hyb in (("eager", False), ("hip_graph", True)):
        net.hybridize(hyb)
        for _ in range(5):
            net(x1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            net(x1)
        torch.cuda.synchronize()
        lat[label + "_ms"] = 1e3 * (time.perf_counter() - t0) / 30
    net.hybridize(False)
    gf = FWD_GFLOP_PER_FRAME.get((size, classes))
    if gf:
        lat["frac_of_fp32_mfma_peak"] = gf / min(lat["eager_ms"], lat["hip_graph_ms"]) / FP32_MFMA_PEAK_TFLOPS
    return dict(lat, size=size, note="one frame resident in HBM -> 100 detection rows")


FWD_GFLOP_PER_FRAME = {(416, 20): 65.43, (608, 20): 139.76, (608, 30): 139.92}  # SURVEY.md 8(d) / BASELINE.md 3


def _barrier(dist, torch):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(dt, dist, torch, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def train_leg(vy, dev, dist, rank, world, size, batch, classes, steps, warmup, syncbn=False, overlap=True,
              split=True, allreduce_alone=False, conv_mode="exact"):
    """One training workload, timed like the headline: `warmup` un-timed steps, then exactly `steps` steps between
    barrier + synchronize, max over ranks.  One step = recorded forward (batch-statistics BatchNorm, targets, loss)
    + backward + gradient all-reduce (bucketed, overlapped with backward) + SGD: the call pattern of
    train_yolov3.py:623-634.  syncbn: the net is built the way train_yolov3.py:350-354 builds it
    (norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': world}); with one rank that is plain BatchNorm.
    EVERY rank runs this (it contains collectives); the returned dict is rank 0's view."""
    import torch
    from videoyolo_amd import autograd, targets
    cls_names = ["c%d" % i for i in range(classes)]
    from videoyolo_amd import parallel
    multi = world > 1 or parallel.collectives_active()  # (one rank with VY_FORCE_COLLECTIVES=1: the same code path)
    kw = dict(norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": world}) if (syncbn and multi) else {}
    net = vy.yolo3_darknet53(cls_names, pretrained_base=False, **kw)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx(dev)
    net.set_conv_mode(conv_mode)  # 'split_bf16x3': recorded forward + data gradients on the split kernel (weight gradients exact)
    g = torch.Generator(device="cpu").manual_seed(1233 + rank)
    x = torch.randn((batch, 3, size, size), generator=g, dtype=torch.float32).to(dev)
    gt_boxes, gt_ids = targets.synthetic_gt(batch, size, classes, m=8, seed=100 + rank)
    tg = targets.YOLOV3PrefetchTargetGenerator(classes)(size, size, gt_boxes, gt_ids)
    dv = [torch.as_tensor(t).to(dev) for t in (gt_boxes,) + tuple(tg)]
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    if multi and overlap:
        trainer.enable_overlap()
    global_batch = batch * world

    def step():
        with autograd.record():
            losses = net(x, *dv)
            autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
        trainer.step(global_batch)
        return losses

    for _ in range(warmup):
        step()
    _barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step()
    _barrier(dist, torch)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, torch, dev)
    fps = global_batch * steps / dt
    fwd_gflop = FWD_GFLOP_PER_FRAME.get((size, classes))
    out = {"frames_per_s": fps, "ms_per_step": 1e3 * dt / steps, "n_gpus": world, "per_gpu_batch": batch,
           "global_batch": global_batch, "size": size, "classes": classes, "steps": steps, "warmup": warmup,
           "batchnorm": "SyncBatchNorm(num_devices=%d) on the 6 layers that receive norm_layer" % world if kw else "per-device",
           "allreduce": ("bucketed, overlapped with backward" if overlap else "one all-reduce after backward") if multi else "none (1 rank)",
           "backend": dist.get_backend() if dist is not None else None,
           "loss_rank0": float(sum(l.sum() for l in losses).item() / batch)}
    if fwd_gflop:
        # whole timed step (forward + backward + exposed all-reduce + SGD) against the roof, per GPU
        out["whole_step_tflops_per_gpu"] = fps / world * 3 * fwd_gflop / 1e3
        out["frac"] = out["whole_step_tflops_per_gpu"] / FP32_MFMA_PEAK_TFLOPS
    if split:
        # extra un-timed steps split by events on the compute stream; EVERY rank runs them (collectives inside)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        rows = []
        for _ in range(3):
            ev[0].record()
            with autograd.record():
                losses = net(x, *dv)
                ev[1].record()
                autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
            ev[2].record()
            trainer.allreduce_grads()   # overlapped: waits for the buckets still in flight; else the whole all-reduce
            ev[3].record()
            trainer.update(global_batch)
            ev[4].record()
            torch.cuda.synchronize()
            rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
        fw, bw, ar, up = [sorted(r[i] for r in rows)[1] for i in range(4)]
        out.update(forward_ms=fw, backward_ms=bw, allreduce_exposed_ms=ar, sgd_ms=up)
        if fwd_gflop:
            fl = fwd_gflop * 1e9 * batch
            out.update(forward_tflops=fl / (fw * 1e-3) / 1e12, backward_tflops=2 * fl / (bw * 1e-3) / 1e12,
                       frac_forward_backward=3 * fl / ((fw + bw) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS)
    if allreduce_alone and multi:
        # the flat gradient buffer (246.5 MB) all-reduced with nothing else on the GPU
        nbytes = net._grads.numel() * 4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for i in range(6):
            _barrier(dist, torch)
            e0.record()
            dist.all_reduce(net._grads)
            e1.record()
            torch.cuda.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1))
        alone = sorted(ts)[len(ts) // 2]
        out["allreduce_alone_ms"] = alone
        out["allreduce_bytes"] = nbytes
        out["allreduce_busbw_GBps"] = 2.0 * (world - 1) / world * nbytes / (alone * 1e-3) / 1e9
        if world > 1:
            all_links = (world - 1) * XGMI_LINK_GBPS_PER_DIRECTION
            out["allreduce_expectation"] = {
                "all_links_rs_ag_busbw_GBps": all_links, "single_ring_busbw_GBps": XGMI_LINK_GBPS_PER_DIRECTION,
                "ms_at_all_links": 2.0 * (world - 1) / world * nbytes / (all_links * 1e9) * 1e3,
                "ms_at_single_ring": 2.0 * (world - 1) / world * nbytes / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3,
                "frac_of_all_links": out["allreduce_busbw_GBps"] / all_links,
                "note": "xGMI is point-to-point: %d links x %.1f GB/s per direction per GPU.  RCCL is left to choose (no NCCL_ALGO / "
                        "NCCL_PROTO override): above the single-ring figure it is using several links at once; the bucketed "
                        "all-reduce of the step hides behind backward either way (allreduce_overlap_fraction)"
                        % (world - 1, XGMI_LINK_GBPS_PER_DIRECTION)}
        if world == 1:
            out["allreduce_note"] = "one rank (VY_FORCE_COLLECTIVES): the collective runs through the backend but moves nothing between devices"
        if split:
            out["allreduce_overlap_fraction"] = max(0.0, min(1.0, 1.0 - out["allreduce_exposed_ms"] / alone))
    del trainer, net
    import gc
    gc.collect()  # the net's ctypes callbacks hold a reference cycle: collect before returning its buffers
    torch.cuda.empty_cache()
    return out


MULTISCALE_SIZES = tuple(32 * x for x in range(10, 20))  # train_yolov3.py:266 `x * 32 for x in range(10, 20)`: 320 ... 608


def multiscale_leg(vy, dev, dist, rank, world, batch, classes, interval, warmup, sizes=MULTISCALE_SIZES, seed=233,
                   conv_mode="exact"):
    """The reference's DEFAULT training mode (train_yolov3.py:258-271: `RandomTransformDataLoader(transform_fns, ...,
    interval=10)` unless --no_random_shape): one net, one trainer, the input side drawn from 320 ... 608 and changed every
    `interval` batches.  Here: every one of the ten sizes once, in a seeded random order (the same on every rank: the
    loader picks ONE transform per batch, all devices train the same size), `interval` steps each.  Timed like
    also_train416 — barrier + synchronize around the WHOLE sequence, max over ranks — so every re-plan (workspace bind,
    border zeroing, tile choices for the new shapes) is inside the clock; per size a synchronize on both sides of the
    re-plan and of its window gives `replan_ms` and the window's frames/s (rank 0's view).  The warm-up runs at the LARGEST
    size, so the workspace has its final extent before the clock starts (what the first epoch does for every later one)."""
    import random
    import torch
    from videoyolo_amd import autograd, targets, parallel
    multi = world > 1 or parallel.collectives_active()
    net = vy.yolo3_darknet53(["c%d" % i for i in range(classes)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx(dev)
    net.set_conv_mode(conv_mode)
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    if multi:
        trainer.enable_overlap()
    gb = batch * world
    order = list(sizes)
    random.Random(seed).shuffle(order)
    g = torch.Generator(device="cpu").manual_seed(4233 + rank)
    tgen = targets.YOLOV3PrefetchTargetGenerator(classes)
    data = {}
    for s in sizes:
        x = torch.randn((batch, 3, s, s), generator=g, dtype=torch.float32).to(dev)
        gt_boxes, gt_ids = targets.synthetic_gt(batch, s, classes, m=8, seed=100 + rank)
        tg = tgen(s, s, gt_boxes, gt_ids)
        data[s] = (x, [torch.as_tensor(t).to(dev) for t in (gt_boxes,) + tuple(tg)])

    def step(s):
        x, dv = data[s]
        with autograd.record():
            losses = net(x, *dv)
            autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
        trainer.step(gb)
        return losses

    for _ in range(max(1, warmup)):
        step(max(sizes))
    rows = []
    _barrier(dist, torch)
    t0 = time.perf_counter()
    for s in order:
        ta = time.perf_counter()
        with torch.cuda.device(dev):
            net._ensure_plan(batch, s, s, train=True)     # what the first step at a new size does anyway: timed alone
        torch.cuda.synchronize()
        tb = time.perf_counter()
        step(s)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(interval - 1):
            losses = step(s)
        torch.cuda.synchronize()
        td = time.perf_counter()
        rows.append((s, tb - ta, tc - tb, td - tc, td - ta, float(sum(l.sum() for l in losses).item() / batch)))
    _barrier(dist, torch)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, torch, dev)
    nsteps = interval * len(order)
    per_size = {}
    g416 = FWD_GFLOP_PER_FRAME[(416, 20)]
    for s, replan, first, rest, window, loss in rows:
        steady = rest / max(1, interval - 1)
        gf = g416 * (s / 416.0) ** 2   # every conv's M scales with the pixel count; prediction convs differ by < 0.1 %
        fps_w = gb * interval / window
        per_size[str(s)] = {
            "frames_per_s": fps_w, "steady_ms_per_step": 1e3 * steady, "replan_ms": 1e3 * replan,
            "first_step_extra_ms": 1e3 * (first - steady), "replan_share_of_window": replan / window,
            "frac_whole_step": gb / world / steady * 3 * gf / 1e3 / FP32_MFMA_PEAK_TFLOPS, "loss_rank0": loss}
    replan_total = sum(r[1] for r in rows)
    gf_total = sum(g416 * (s / 416.0) ** 2 * 3 * batch * interval for s in order)
    out = {"frames_per_s": gb * nsteps / dt, "ms_per_step": 1e3 * dt / nsteps, "n_gpus": world, "per_gpu_batch": batch,
           "global_batch": gb, "classes": classes, "sizes_in_order": order, "interval": interval, "steps": nsteps,
           "warmup": "%d step(s) at %d" % (max(1, warmup), max(sizes)), "dtype": "f32" if conv_mode == "exact" else SPLIT_DTYPE,
           "replan_ms_total": 1e3 * replan_total, "replan_share": replan_total / dt,
           "replan_ms_max": 1e3 * max(r[1] for r in rows),
           "frac": gf_total / dt / 1e3 / FP32_MFMA_PEAK_TFLOPS,
           "backend": dist.get_backend() if dist is not None else None, "per_size": per_size,
           "note": "timed region = the whole ten-window sequence incl. every re-plan; per-size figures are rank 0's, bracketed by "
                   "synchronize (20 extra synchronizes in %d steps); conv-only fractions and tile choices per size: "
                   "profiles/r06_multiscale_layers.txt (tools/multiscale_layers.sh)" % nsteps}
    del trainer, net, data
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def host_fed_leg(vy, dev, dist, rank, world, size, batch, classes, steps, warmup, src_hw, conv_mode="exact", obj_bias=0.0,
                 resident_fps=None):
    """The reference's detect loop (detect_yolo3.py:209-233) fed from the HOST: per step one clip batch of world x batch
    uint8 frames (src_hw, e.g. decoded 720p video) lies in host memory; every rank takes its slice
    (parallel.scatter_frames, even_split=False sizes), copies it in on a copy stream, resizes + normalises it on the GPU
    (csrc/preproc.hip, interp 9), runs the net, gathers the (B, 100, 6) rows of all ranks (RCCL all-gather; nothing at
    N = 1) and rank 0 copies them to the host — double-buffered (videoyolo_amd/stream.py): batch i + 1 goes in and batch
    i - 1 comes out while batch i computes.  Timed like the headline: warm-up batches, then exactly `steps` batches
    between barrier + synchronize, max over ranks; every result has been collected on the host when the clock stops."""
    import numpy as np
    import torch
    from videoyolo_amd import stream
    net = vy.yolo3_darknet53(["c%d" % i for i in range(classes)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233, obj_bias=obj_bias)
    net.collect_params().reset_ctx(dev)
    net.set_nms(0.45, 400, 100)
    net.set_conv_mode(conv_mode)
    h, w = src_hw
    gb = batch * world
    rng = np.random.default_rng(77)
    # The clip batch of a step = `world` x the same `batch` frames (random bytes drawn once, pinned: what a decoder's output
    # buffers are — copied in without a staging copy).  At N > 1 it is not materialised `world` times in every rank's host
    # memory (8 x 1.4 GB of 720p frames): _RepeatedClip answers parallel.scatter_frames' slice — any rank's share — with the
    # one pinned pool, which is exactly what the materialised tile would hold there.
    class _RepeatedClip(object):
        def __init__(self, pool, times):
            self.pool, self.times = pool, times

        def __len__(self):
            return len(self.pool) * self.times

        def __getitem__(self, sl):
            lo, hi, step = sl.indices(len(self))
            if step != 1 or hi - lo != len(self.pool) or lo % len(self.pool):
                raise IndexError("a rank's share of %d frames, not %s" % (len(self.pool), sl))
            return self.pool

    pools = [torch.from_numpy(rng.integers(0, 256, (batch, h, w, 3), dtype=np.uint8)).pin_memory() for _ in range(2)]
    clips = [p if world == 1 else _RepeatedClip(p, world) for p in pools]
    det = stream.HostFedDetector(net, gb, (h, w), size, depth=2, gather=True)
    kept = 0
    for i, out in enumerate(det.run(clips[i & 1] for i in range(warmup))):
        pass
    _barrier(dist, torch)
    t0 = time.perf_counter()
    for out in det.run(clips[i & 1] for i in range(steps)):
        if out is not None:
            kept = int((out[0] >= 0).sum())
    _barrier(dist, torch)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, torch, dev)
    fps = gb * steps / dt
    # the copies alone, on an idle GPU (this rank's slice in, the gathered rows out)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    cin, cout = [], []
    for _ in range(4):
        torch.cuda.synchronize()
        e0.record()
        det.dev_in[0].copy_(det.pin_in[0], non_blocking=True)
        e1.record()
        det.pin_out[0].copy_(det.dev_out[0], non_blocking=True)
        e2.record()
        torch.cuda.synchronize()
        cin.append(e0.elapsed_time(e1))
        cout.append(e1.elapsed_time(e2))
    leg = {"frames_per_s": fps, "ms_per_step": 1e3 * dt / steps, "n_gpus": world, "per_gpu_batch": batch, "global_batch": gb,
           "size": size, "classes": classes, "steps": steps, "warmup": warmup, "source_frames": "%dx%d uint8 (HWC)" % (h, w),
           "dtype": "f32" if conv_mode == "exact" else SPLIT_DTYPE, "kept_detections_last_batch": kept,
           "copy_in_alone_ms": sorted(cin)[len(cin) // 2], "copy_in_bytes": int(det.pin_in[0].numel()),
           "copy_in_GBps": det.pin_in[0].numel() / (sorted(cin)[len(cin) // 2] * 1e-3) / 1e9,
           "copy_out_alone_ms": sorted(cout)[len(cout) // 2], "copy_out_bytes": int(det.pin_out[0].numel() * 4),
           "pipeline": "uint8 frames in pinned host memory -> H2D (copy stream) -> resize + to_tensor + normalise -> net -> %s-> D2H (copy stream); "
                       "2 slots" % ("all-gather of the rows over %s " % dist.get_backend() if det.gather else "")}
    if resident_fps:
        # what feeding from the host costs per step beyond the resident step (pre-processing kernel + whatever of the copies
        # and of the host-side staging does not hide behind the previous batch's kernels)
        leg["resident_frames_per_s"] = resident_fps
        leg["vs_resident"] = fps / resident_fps
        leg["exposed_ms_per_step"] = 1e3 * dt / steps - 1e3 * gb / resident_fps
    del det, net
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return leg


def torch_cpu_child(params, frames, classes, gflop_per_frame):
    """The torch-CPU datapoint in a CHILD process (tools/cpu_torch_baseline.py): one thread per physical core of this
    process's cpuset, bound to cores, a warm-up pass at the same shape — settings an OpenMP runtime only takes at start-up.
    The child never touches a GPU.  Returns its JSON (or a note why it could not run): never fails the line."""
    import subprocess
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import cpu_torch_baseline as ctb
        with tempfile.TemporaryDirectory(prefix="vy_cpu_", dir="/tmp") as d:
            np.savez(os.path.join(d, "p.npz"), **params)
            np.save(os.path.join(d, "x.npy"), frames)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_torch_baseline.py"), os.path.join(d, "p.npz"),
                                os.path.join(d, "x.npy"), "--classes", str(classes), "--gflop-per-frame", str(gflop_per_frame)],
                               env=ctb.child_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        if p.returncode != 0:
            return {"value": None, "note": "child failed (rc %d): %s" % (p.returncode, p.stderr.decode()[-300:])}
        return json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        return {"value": None, "note": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step")
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--obj-bias", type=float, default=0.0,
                    help="added to the objectness biases (-5: trained-like sparse candidates)")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU-oracle sample (0: skip)")
    ap.add_argument("--cpu-torch-frames", type=int, default=8,
                    help="frames of the torch-CPU datapoint, as one batch (SURVEY 8d's recipe: batch 8, threads = physical cores)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="infer: BASELINE configs[1] headline + the training legs (default); train: configs[2] as the "
                         "headline (416x416, batch 16/GPU, fwd + bwd + gradient all-reduce + SGD)")
    ap.add_argument("--syncbn", action="store_true", help="train: SyncBatchNorm statistics all-reduce (configs[4])")
    ap.add_argument("--no-overlap", action="store_true", help="train: all-reduce after backward instead of bucketed")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1: initialise the process group anyway and send every broadcast / all-reduce / SyncBatchNorm "
                         "exchange of the training legs through the backend (VY_FORCE_COLLECTIVES=1): the RCCL code path on "
                         "one GPU; also_syncbn608 then runs at N = 1 too")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: all ranks use cuda:0 (with --backend gloo) to exercise the N>1 code path on one GPU")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 --pmc child runs that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency figure (N = 1 inference only)")
    ap.add_argument("--conv-mode", choices=["exact", "split_bf16x3", "split_bf16x3_train"], default="exact",
                    help="arithmetic of the HEADLINE inference step (default exact fp32: the parity path); the default "
                         "line reports the split mode beside it as also_infer608_split")
    ap.add_argument("--no-split-leg", action="store_true", help="skip also_infer608_split")
    ap.add_argument("--no-train-legs", action="store_true",
                    help="infer mode: skip also_train416 / also_syncbn608 (BASELINE configs[2] / [4])")
    ap.add_argument("--no-host-legs", action="store_true", help="skip also_hostfed608 / also_vid608 (host-fed pipelined legs)")
    ap.add_argument("--src-hw", default="720x1280", help="also_hostfed608: size of the uint8 source frames (HxW)")
    ap.add_argument("--vid-src-hw", default="360x640", help="also_vid608: size of the uint8 source frames (HxW)")
    ap.add_argument("--vid-batch", type=int, default=64, help="also_vid608: frames per GPU and clip batch")
    ap.add_argument("--vid-size", type=int, default=608, help="also_vid608: network input size (tests use small ones)")
    ap.add_argument("--train-steps", type=int, default=20, help="timed steps of each training leg (20: a 0.6 s window; 10 was short enough for one host hiccup to cost 10 %)")
    ap.add_argument("--train-size", type=int, default=416)
    ap.add_argument("--no-preflight", action="store_true",
                    help="N > 1: skip the collective pre-flight (5 barriers + 5 all-reduces of the 246.5 MB gradient buffer: sum "
                         "checked, bus bandwidth against the xGMI expectation; a failure exits 3 with a one-line reason)")
    ap.add_argument("--preflight-min-busbw", type=float, default=None,
                    help="fail the pre-flight below this all-reduce bus bandwidth (GB/s); default: report only")
    ap.add_argument("--leg-budget-s", type=float, default=600.0,
                    help="wall-clock budget per leg: a leg that exceeds it is reported as {\"timeout\": true} and the line is "
                         "printed with everything measured before it (0: no watchdog)")
    ap.add_argument("--no-multiscale-leg", action="store_true", help="skip also_train_multiscale (the reference's default training mode)")
    ap.add_argument("--multiscale-interval", type=int, default=10, help="steps per size (train_yolov3.py:270 interval=10)")
    ap.add_argument("--multiscale-sizes", default=",".join(str(v) for v in MULTISCALE_SIZES), help="tests use small ones")
    ap.add_argument("--train-batch", type=int, default=16, help="frames per GPU of also_train416")
    ap.add_argument("--syncbn-size", type=int, default=608)
    ap.add_argument("--syncbn-batch", type=int, default=8, help="frames per GPU of also_syncbn608")
    args = ap.parse_args()
    # `python bench.py --gpus N` (no rank environment): this process only starts the N ranks and waits —
    # it never touches a GPU and never execs (videoyolo_amd/launch.py).  Under torch.distributed.run the
    # rank environment is already there and this is skipped.
    from videoyolo_amd import launch
    if launch.needs_spawn(args.gpus):
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    # From here on this process is a rank (or the single process of N = 1).  The contract is ONE JSON line on stdout:
    # libraries write there too (gloo announces "[Gloo] Rank 0 is connected to ..." on stdout when a group is created —
    # also the host side group beside RCCL), so file descriptor 1 is pointed at stderr for the rest of the run and the
    # line goes out through a saved copy of the real stdout.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if args.mode == "train":
        if "--size" not in " ".join(sys.argv):
            args.size = 416
        if "--batch" not in " ".join(sys.argv):
            args.batch = 16
    legs = args.mode == "infer" and not args.no_train_legs
    # roofline.traffic is MEASURED by this run (not read from a committed file): at N = 1 child runs of the
    # same workloads under rocprofv3 --pmc, before this process touches the GPU
    traffic, traffic_note = None, "not measured (--no-pmc / --no-roofline / N > 1 / inside a profiler)"
    train_traffic, train_traffic_note = None, traffic_note
    split_traffic, split_traffic_note = None, traffic_note
    in_child = bool(os.environ.get("VY_BENCH_CHILD"))
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not in_child and not args.no_pmc and not args.no_roofline:
        quiet = ["--steps", "2", "--warmup", "1", "--cpu-frames", "0", "--no-roofline", "--no-pmc", "--no-latency",
                 "--no-train-legs", "--no-host-legs"]
        child = ["--mode", args.mode, "--size", str(args.size), "--batch", str(args.batch), "--classes",
                 str(args.classes), "--obj-bias", str(args.obj_bias), "--no-split-leg"] + quiet
        traffic, traffic_note = measure_hbm_traffic(child + ["--conv-mode", args.conv_mode], 3)
        if args.mode == "infer" and args.conv_mode == "exact" and not args.no_split_leg:
            split_traffic, split_traffic_note = measure_hbm_traffic(child + ["--conv-mode", "split_bf16x3"], 3)
        if legs:
            child = ["--mode", "train", "--size", str(args.train_size), "--batch", str(args.train_batch), "--classes",
                     str(args.classes)] + quiet
            train_traffic, train_traffic_note = measure_hbm_traffic(child, 3)

    import numpy as np
    import torch
    import videoyolo_amd as vy

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if args.force_collectives:
        os.environ["VY_FORCE_COLLECTIVES"] = "1"
    forced = os.environ.get("VY_FORCE_COLLECTIVES", "0") not in ("", "0")
    if world > 1 or forced:
        import torch.distributed as dist
        from videoyolo_amd import parallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.share_gpu:
            local = 0
        kw = {"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=parallel.group_timeout(), **kw)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # What a multi-rank number ran on, stated by the run itself (every rank contributes its device's identity), and a
    # pre-flight of the collectives before any leg: a mis-set fabric ends the run here, in seconds, with one line.
    collective = None
    if dist is not None:
        from videoyolo_amd import parallel
        # the first collectives of the run (an all_gather_object, then the pre-flight) under their own watchdog: a fabric that
        # hangs instead of failing ends the run with one line after at most 3 minutes, not after the group's timeout
        pre = LegWatchdog(rank, min(args.leg_budget_s, 180.0), {"n_gpus": world, "error": "the first collectives of the run did not return"})
        pre.arm("collective_preflight")
        collective = parallel.describe_group(dev)
        if not args.no_preflight:
            pf = parallel.preflight(dev, min_busbw_GBps=args.preflight_min_busbw if world > 1 else None)
            collective["preflight"] = pf
            if not pf["ok"]:
                if rank == 0:
                    sys.stderr.write("bench.py: collective pre-flight failed: %s\n" % pf["reason"])
                    _emit({"error": "collective pre-flight failed: " + pf["reason"], "n_gpus": world, "collective": collective})
                dist.destroy_process_group()
                sys.exit(3)
        pre.disarm()

    if args.mode == "train":
        return bench_train(args, vy, dev, dist, rank, world, traffic, traffic_note, collective)

    result = {}
    dog = LegWatchdog(rank, args.leg_budget_s, result)
    dog.arm("headline")
    classes = ["c%d" % i for i in range(args.classes)]
    net = vy.yolo3_darknet53(classes, pretrained_base=False)
    net.initialize(init="synthetic", seed=233, obj_bias=args.obj_bias)
    net.collect_params().reset_ctx(dev)
    net.set_nms(0.45, 400, 100)  # detect_yolo3.py:200; like the reference's detect() the net is not hybridized
    net.set_conv_mode(args.conv_mode)

    g = torch.Generator(device="cpu").manual_seed(233 + rank)
    x = torch.randn((args.batch, 3, args.size, args.size), generator=g, dtype=torch.float32).to(dev)

    for _ in range(args.warmup):
        net(x)
    _barrier(dist, torch)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = net(x)
    _barrier(dist, torch)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, torch, dev)
    frames = args.batch * world * args.steps
    fps = frames / dt

    result.update({
        "metric": "frames/sec, yolo3_darknet53 inference %dx%d" % (args.size, args.size),
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.conv_mode == "exact" else SPLIT_DTYPE, "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: Darknet-53 + 3-scale head + decode + NMS "
                               "inference, %dx%d, batch %d per GPU, %d classes, nms 0.45/topk 400/post 100"
                               % (args.size, args.size, args.batch, args.classes),
                   "per_gpu_batch": args.batch, "global_batch": args.batch * world, "size": args.size,
                   "classes": args.classes, "parallelism": "frame-scatter x%d" % world,
                   "kept_detections_rank0": int((out[0] >= 0).sum().item())},
    })
    if collective is not None:
        # (second gather: now every rank's net has bound its workspace, so the stream-K placement probe has a verdict)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "streamk_enabled": net.streamk_enabled()})
        for d, e in zip(collective["devices"], per_rank):
            d["streamk_enabled"] = e["streamk_enabled"]
        result["collective"] = collective

    dog.arm("roofline")
    if rank == 0 and not args.no_roofline:
        result["device_under_load"] = device_under_load(torch, dev, lambda: net(x))
        # per-launch HIP-event timing of extra passes (same stream the kernels run on)
        med, agg = launch_table(net, x)
        # the dominant kernel: the implicit-GEMM tile variant with the largest share of the step
        # (128x128 at the BASELINE shape; small test shapes fall back to the smaller tiles)
        conv_prefixes = ("conv_igemm_kernel", "conv_split_kernel", "conv_wino_kernel")  # (the last two only with --conv-mode split_bf16x3)
        dom = max((k for k in agg if k.startswith(conv_prefixes)), key=lambda k: agg[k][1])
        n, ms, fl, by = agg[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        total_ms = sum(a[1] for a in agg.values())
        total_fl = sum(a[2] for a in agg.values())
        result["roofline"] = {
            "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
            "kernel": dom, "launches_per_step": n, "avg_launch_ms": ms / n,
            "flops_per_launch_avg": fl / n, "kernel_share_of_step_time": ms / total_ms,
            "whole_step_tflops": total_fl / (total_ms * 1e-3) / 1e12,
            "by_kernel_ms": {k: round(a[1], 4) for k, a in agg.items()},
        }
        result["roofline"]["algorithmic_bytes_per_launch_avg"] = by / n
        # Comparable across rounds whatever the "dominant" instance is (it changed identity in round 3, when stream-K
        # split the 128x128 launches into two kernels): EVERY conv launch of the step, sum of FLOPs / sum of time.
        conv = [a for k, a in agg.items() if k.startswith(conv_prefixes)]
        conv_ms, conv_fl = sum(a[1] for a in conv), sum(a[2] for a in conv)
        result["roofline"]["frac_all_conv"] = conv_fl / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS
        result["roofline"]["all_conv"] = {"launches_per_step": sum(a[0] for a in conv), "ms_per_step": conv_ms,
                                          "tflops": conv_fl / (conv_ms * 1e-3) / 1e12}
        # the same tile's launches that ran as chain-preserving stream-K are a kernel of their own for rocprofv3
        # (conv_igemm_kernel<..., SK = true>): reported beside the dominant one, same definition of `achieved`
        twin = dom[:-3] + ">" if dom.endswith("sk>") else dom[:-1] + "sk>"
        if twin in agg:
            tn, tms, tfl, tby = agg[twin]
            result["roofline"]["same_tile_other_instance"] = {
                "kernel": twin, "launches_per_step": tn, "avg_launch_ms": tms / tn,
                "achieved": tfl / (tms * 1e-3) / 1e12, "frac": tfl / (tms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                "both_instances_frac": (fl + tfl) / ((ms + tms) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
        # HBM bytes per launch of the same kernel, measured by this run's two rocprofv3 --pmc child passes
        if traffic:
            label = dom.split("<")[1].rstrip(">")  # "<BM>x<BN>" or "<BM>x<BN>sk" (stream-K instance)
            tile = label[:-2].split("x") if label.endswith("sk") else label.split("x")
            keys = _forward_tile_keys(traffic, tile[0], tile[1], label.endswith("sk"))
            if keys:
                # (launch-weighted over the instances the label covers: the launches summed in runs are a kernel of their own)
                nl = float(sum(traffic[k]["launches"] for k in keys))
                avg = lambda f: sum(traffic[k][f] * traffic[k]["launches"] for k in keys) / nl   # noqa: E731
                key = keys[0]
                result["roofline"]["traffic"] = avg("hbm_bytes")
                result["roofline"]["traffic_detail"] = {
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (FETCH x2: gfx950)",
                    "kernels": {k[:72]: {"launches_in_child_run": traffic[k]["launches"], "hbm_bytes_per_launch": traffic[k]["hbm_bytes"]} for k in keys},
                    "fetch_bytes_per_launch": avg("fetch_bytes"), "write_bytes_per_launch": avg("write_bytes"),
                    "over_algorithmic": avg("hbm_bytes") / (by / n),
                    "whole_step_hbm_bytes": traffic["_per_step"],
                    "other_kernels_MB_per_launch": {k[:48]: round(v["hbm_bytes"] / 1e6, 2) for k, v in traffic.items()
                                                    if isinstance(v, dict) and k not in keys and "rocclr" not in k
                                                    and "at::native" not in k}}
        if result["roofline"]["traffic"] is None:
            result["roofline"]["traffic_note"] = traffic_note
        tail = agg.get("decode_nms")
        if tail:
            result["roofline"]["decode_nms"] = {
                "ms": tail[1], "algorithmic_GBps": tail[3] / (tail[1] * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS}

    dog.arm("latency_batch1")
    if rank == 0 and world == 1 and not args.no_latency:
        result["latency_batch1"] = batch1_latency(net, x[:1].contiguous(), torch, args.size, args.classes)

    if rank == 0 and world == 1 and not args.no_latency and args.size == 608:
        # BASELINE.json's metric names both frame sizes: the same batch at 416x416, timed the same way
        x4 = torch.randn((args.batch, 3, 416, 416), generator=g, dtype=torch.float32).to(dev)
        for _ in range(args.warmup):
            net(x4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x4)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t0
        fps4 = args.batch * args.steps / dt4
        gflop4 = FWD_GFLOP_PER_FRAME.get((416, args.classes))
        result["also_416"] = {"frames_per_s": fps4, "ms_per_step": 1e3 * dt4 / args.steps, "batch": args.batch,
                              "whole_step_tflops": None if gflop4 is None else fps4 * gflop4 / 1e3,
                              "frac_of_fp32_mfma_peak": None if gflop4 is None else fps4 * gflop4 / 1e3 / FP32_MFMA_PEAK_TFLOPS}
        # the reference's DEFAULT detect call: batch_size 1, data_shape 416 (detect_yolo3.py:55-57)
        result["latency_batch1_416"] = dict(batch1_latency(net, x4[:1].contiguous(), torch, 416, args.classes),
                                            note="one 416x416 frame resident in HBM -> 100 detection rows: the reference's default "
                                                 "detect call (detect_yolo3.py:55-57 batch_size 1, data_shape 416)")
        del x4

    dog.arm("also_infer%d_split" % args.size)
    if args.conv_mode == "exact" and not args.no_split_leg:
        # The same step in the opt-in split-fp32 conv mode, on every rank, timed the same way.  NOT the headline.
        net.set_conv_mode("split_bf16x3")
        for _ in range(args.warmup):
            net(x)
        _barrier(dist, torch)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            net(x)
        _barrier(dist, torch)
        sdt = _max_over_ranks(time.perf_counter() - t0, dist, torch, dev)
        sfps = args.batch * world * args.steps / sdt
        leg = {"frames_per_s": sfps, "ms_per_step": 1e3 * sdt / args.steps, "dtype": SPLIT_DTYPE,
               "speedup_over_exact": sfps / fps, "n_gpus": world,
               "workload": "the headline step (%dx%d, batch %d per GPU) with net.set_conv_mode('split_bf16x3'): every "
                           "conv+BN+leaky cell with cout %% 64 == 0 on v_mfma_f32_32x32x16_bf16 (six bf16 products per "
                           "fp32 multiply, fp32 accumulate; the long-K 3x3 stride-1 cells as Winograd F(2, 3) on the same "
                           "arithmetic, conv_wino.hip); stem, the 64->32 bottleneck, prediction convs, decode and NMS "
                           "as in the exact path.  Not bit-equal to the oracle: tests/test_gpu_split.py"
                           % (args.size, args.size, args.batch)}
        if rank == 0 and not args.no_roofline:
            smed, sagg = launch_table(net, x)
            sp = [a for k, a in sagg.items() if k.startswith("conv_split_kernel") or k.startswith("conv_wino_kernel")]
            ex = [a for k, a in sagg.items() if k.startswith("conv_igemm_kernel")]
            sp_ms, sp_fl = sum(a[1] for a in sp), sum(a[2] for a in sp)
            tot_ms, tot_fl = sum(a[1] for a in sagg.values()), sum(a[2] for a in sagg.values())
            eq = sp_fl / (sp_ms * 1e-3) / 1e12  # fp32-equivalent TFLOP/s of the split launches (FLOPs of the DIRECT conv)
            wino = [a for k, a in sagg.items() if k.startswith("conv_wino_kernel")]  # Winograd F(2, 3): 2/3 of the multiplications
            issued = (sp_fl - sum(a[2] for a in wino) / 3.0) * SPLIT_PRODUCTS / (sp_ms * 1e-3) / 1e12
            leg["roofline"] = {
                "bound": "mfma", "kernel": "conv_split_kernel + conv_wino_kernel (all instances)",
                "launches_per_step": sum(a[0] for a in sp),
                "ms_per_step": sp_ms, "achieved_fp32_equivalent": eq, "unit": "TFLOP/s",
                "frac_vs_fp32_mfma_peak_157": eq / FP32_MFMA_PEAK_TFLOPS,
                "frac_vs_bf16_peak_over_6_419": eq / (BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS),
                "winograd_launches": sum(a[0] for a in wino), "winograd_ms": sum(a[1] for a in wino),
                "winograd_note": "conv_wino.hip: the long-K 3x3 stride-1 cells as a 1-D Winograd F(2, 3) on the same split "
                                 "arithmetic (2/3 of the multiplications); FLOPs above are the direct conv's",
                "bf16_mfma_tflops": issued, "bf16_mfma_peak": BF16_MFMA_PEAK_TFLOPS,
                "exact_kernel_launches_left": sum(a[0] for a in ex), "exact_kernel_ms_left": sum(a[1] for a in ex),
                "whole_step_fp32_equivalent_tflops": tot_fl / (tot_ms * 1e-3) / 1e12,
                "kernel_share_of_step_time": sp_ms / tot_ms,
                "by_kernel_ms": {k: round(a[1], 4) for k, a in sagg.items()}, "traffic": None}
            if split_traffic:
                keys = [k for k in split_traffic if isinstance(split_traffic[k], dict) and ("conv_split_kernel" in k or "conv_wino_kernel" in k)]
                n_l = sum(split_traffic[k]["launches"] for k in keys)
                if n_l:
                    hb = sum(split_traffic[k]["hbm_bytes"] * split_traffic[k]["launches"] for k in keys) / n_l
                    alg = sum(a[3] for a in sp) / max(1, sum(a[0] for a in sp))
                    leg["roofline"]["traffic"] = hb
                    leg["roofline"]["traffic_detail"] = {
                        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs with --conv-mode split_bf16x3 (FETCH x2: "
                                  "gfx950); mean HBM bytes per conv_split_kernel / conv_wino_kernel launch",
                        "algorithmic_bytes_per_launch_avg": alg, "over_algorithmic": hb / alg if alg else None,
                        "whole_step_hbm_bytes": split_traffic["_per_step"]}
            if leg["roofline"]["traffic"] is None:
                leg["roofline"]["traffic_note"] = split_traffic_note
        if rank == 0 and world == 1 and not args.no_latency:
            leg["latency_batch1"] = batch1_latency(net, x[:1].contiguous(), torch, args.size, args.classes)
            if args.size == 608:  # BASELINE.json's metric names both frame sizes
                x4 = torch.randn((args.batch, 3, 416, 416), generator=g, dtype=torch.float32).to(dev)
                for _ in range(args.warmup):
                    net(x4)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    net(x4)
                torch.cuda.synchronize()
                fps4 = args.batch * args.steps / (time.perf_counter() - t0)
                leg["also_416"] = {"frames_per_s": fps4, "batch": args.batch,
                                   "speedup_over_exact": fps4 / result["also_416"]["frames_per_s"] if "also_416" in result else None}
                del x4
        result["also_infer%d_split" % args.size] = leg
        net.set_conv_mode("exact")

    dog.arm("cpu_baseline")
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        # CPU baseline: the oracle (a port of the same algorithm; NOT the reference's MXNet path,
        # which cannot be installed here) on a bounded sample of the same workload
        from oracle import yolo3_oracle as O
        params = {p.name: p.data() for p in net.collect_params().values()}
        orc = O.OracleYolo3(args.classes, params)
        try:  # one thread per physical core the host really grants (cpuset, cgroup quota): tools/cpu_torch_baseline.py
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import cpu_torch_baseline as ctb
            O.lib().vyo_set_num_threads(int(ctb.child_env()["OMP_NUM_THREADS"]))
        except Exception:
            pass
        xs = x[:args.cpu_frames].cpu().numpy()
        t0 = time.perf_counter()
        orc(xs)
        cdt = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": args.cpu_frames / cdt, "unit": "frames/s", "cores": int(O.lib().vyo_num_threads()),
            "kind": "port",
            "sample": "%d frames of the same %dx%d batch through oracle/ (C + OpenMP conv, numpy graph, "
                      "C NMS); MXNet itself is not installable here" % (args.cpu_frames, args.size, args.size)}
        try:  # what the host offers (the thread count above is what the OpenMP runtime of the port actually used)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import cpu_torch_baseline as ctb
            result["cpu_baseline"]["host"] = ctb.host_facts()
            gf = FWD_GFLOP_PER_FRAME.get((args.size, args.classes))
            if gf:
                result["cpu_baseline"]["fp32_tflops"] = args.cpu_frames / cdt * gf / 1e3
        except Exception:
            pass
        result["cpu_baseline_torch"] = torch_cpu_child(params, x[:max(1, min(args.cpu_torch_frames, args.batch))].cpu().numpy(),
                                                       args.classes, FWD_GFLOP_PER_FRAME.get((args.size, args.classes), 0.0))
        del orc, params

    if not args.no_host_legs:
        # host-fed, pipelined legs (every rank enters: the gather is a collective).  They build their own nets: release the
        # resident batch first
        hw = lambda t: tuple(int(v) for v in t.lower().split("x"))  # noqa: E731
        del x, out
        torch.cuda.empty_cache()
        dog.arm("also_hostfed%d" % args.size)
        result["also_hostfed%d" % args.size] = dict(
            host_fed_leg(vy, dev, dist, rank, world, args.size, args.batch, args.classes, args.steps, args.warmup, hw(args.src_hw),
                         conv_mode=args.conv_mode, obj_bias=args.obj_bias, resident_fps=fps),
            workload="the headline step fed from the host: decoded %s uint8 frames in pinned host memory instead of a resident "
                     "fp32 batch (detect_yolo3.py:209-233, transforms.py:316-350)" % args.src_hw)
        dog.arm("also_vid%d" % args.vid_size)
        result["also_vid%d" % args.vid_size] = dict(
            host_fed_leg(vy, dev, dist, rank, world, args.vid_size, args.vid_batch, 30, args.steps, args.warmup, hw(args.vid_src_hw),
                         conv_mode=args.conv_mode),
            workload="BASELINE.json configs[3]: ImageNet-VID-shape stream (30 classes), %dx%d, one host clip batch of %d x %d "
                     "frames (%s uint8) per step -> parallel.scatter_frames -> net -> parallel.gather_detections -> host"
                     % (args.vid_size, args.vid_size, world, args.vid_batch, args.vid_src_hw))
        x = out = None

    if legs:
        # BASELINE configs[2] (and [4] when N > 1) on the same ranks, timed the same way: all ranks enter.
        # The inference net's buffers are released first (each leg builds its own net).
        del net, x, out
        torch.cuda.empty_cache()
        dog.arm("also_train416")
        leg = train_leg(vy, dev, dist, rank, world, args.train_size, args.train_batch, args.classes,
                        args.train_steps, args.warmup, syncbn=False, overlap=True, allreduce_alone=True)
        leg["workload"] = ("BASELINE.json configs[2]: training step, VOC-shape synthetic (%d cls, 8 gt/img), %dx%d, "
                           "per-GPU batch %d, SGD(1e-3, 0.9, 5e-4), per-device BN, gradient all-reduce over %d rank(s)"
                           % (args.classes, args.train_size, args.train_size, args.train_batch, world))
        if train_traffic:
            leg["traffic"] = train_traffic["_per_step"]
            top = sorted(((k, v) for k, v in train_traffic.items() if isinstance(v, dict) and "rocclr" not in k and "at::native" not in k),
                         key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:6]
            leg["traffic_detail"] = {
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of `bench.py --mode train` (FETCH x2: gfx950); HBM bytes per step, all library kernels",
                "top_kernels_MB_per_step": {k[:56]: round(v["hbm_bytes"] * v["launches"] / 3.0 / 1e6, 1) for k, v in top}}
        else:
            leg["traffic"] = None
            leg["traffic_note"] = train_traffic_note
        result["also_train416"] = leg
        if not args.no_multiscale_leg:
            dog.arm("also_train_multiscale", scale=2.0)   # (100 steps at ten sizes)
            ms = multiscale_leg(vy, dev, dist, rank, world, args.train_batch, args.classes, args.multiscale_interval, args.warmup,
                                sizes=tuple(int(v) for v in args.multiscale_sizes.split(",")))
            ms["workload"] = ("the reference's DEFAULT training mode (train_yolov3.py:258-271, RandomTransformDataLoader interval=%d): "
                              "BASELINE.json configs[2]'s step with the input side changing through %s, per-GPU batch %d"
                              % (args.multiscale_interval, args.multiscale_sizes, args.train_batch))
            ms["vs_fixed_416"] = {"frames_per_s_416_window": ms["per_size"].get("416", {}).get("frames_per_s"),
                                  "also_train416_frames_per_s": leg["frames_per_s"]}
            result["also_train_multiscale"] = ms
        if not args.no_split_leg:
            # the same training step in conv mode split_bf16x3_train (forward, data and weight gradients on the bf16 matrix
            # core; NOT the parity path): separately reported, like also_infer608_split
            dog.arm("also_train416_split")
            sleg = train_leg(vy, dev, dist, rank, world, args.train_size, args.train_batch, args.classes,
                             args.train_steps, args.warmup, syncbn=False, overlap=True, allreduce_alone=False,
                             conv_mode="split_bf16x3_train")
            sleg["dtype"] = SPLIT_DTYPE + " (recorded forward, data gradients, weight gradients of the convs with cout % 128 == 0)"
            sleg["speedup_over_exact"] = sleg["frames_per_s"] / leg["frames_per_s"]
            sleg["workload"] = leg["workload"] + "; net.set_conv_mode('split_bf16x3_train')"
            result["also_train416_split"] = sleg   # named like also_train416 (the leg carries its size)
        if world > 1 or forced:
            dog.arm("also_syncbn608")
            leg = train_leg(vy, dev, dist, rank, world, args.syncbn_size, args.syncbn_batch, args.classes,
                            args.train_steps, args.warmup, syncbn=True, overlap=True, allreduce_alone=False)
            leg["workload"] = ("BASELINE.json configs[4]: SyncBN training step, %dx%d, per-GPU batch %d, net built with "
                               "norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': %d}; statistics + gradient all-reduce "
                               "over %d ranks" % (args.syncbn_size, args.syncbn_size, args.syncbn_batch, world, world))
            result["also_syncbn608"] = leg

    dog.disarm()
    result["wall_s"] = dog.wall_report()
    if rank == 0:
        _emit(result)
    if dist is not None:
        dist.barrier()  # rank 0 may still be in its un-timed measurement passes: leave together
        dist.destroy_process_group()


def bench_train(args, vy, dev, dist, rank, world, traffic=None, traffic_note=None, collective=None):
    """--mode train: BASELINE configs[2]/[4] as the headline line (see train_leg)."""
    dog = LegWatchdog(rank, args.leg_budget_s, {})
    dog.arm("headline")
    leg = train_leg(vy, dev, dist, rank, world, args.size, args.batch, args.classes, args.steps, args.warmup,
                    syncbn=args.syncbn, overlap=not args.no_overlap, split=not args.no_roofline,
                    allreduce_alone=not args.no_roofline, conv_mode=args.conv_mode)
    result = {
        "metric": "frames/sec, yolo3_darknet53 training %dx%d" % (args.size, args.size),
        "value": leg["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "conv_mode": args.conv_mode,
        "vs_baseline": None, "dtype": "f32" if args.conv_mode == "exact" else SPLIT_DTYPE, "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[%d]: training step, VOC-shape synthetic (%d cls, 8 gt/img), "
                               "%dx%d, per-GPU batch %d, SGD(1e-3, 0.9, 5e-4), %s BN, gradient all-reduce over %d rank(s)"
                               % (4 if args.syncbn else 2, args.classes, args.size, args.size, args.batch,
                                  "Sync" if args.syncbn else "per-device", world),
                   "per_gpu_batch": args.batch, "global_batch": leg["global_batch"], "size": args.size,
                   "classes": args.classes, "parallelism": "dp%d" % world, "loss_rank0": leg["loss_rank0"]},
    }
    if "forward_ms" in leg:
        result["step_split"] = {k: leg[k] for k in ("forward_ms", "backward_ms", "allreduce_exposed_ms", "sgd_ms",
                                                     "allreduce_alone_ms", "allreduce_overlap_fraction") if k in leg}
    if "forward_ms" in leg and "frac_forward_backward" in leg:
        fw, bw = leg["forward_ms"], leg["backward_ms"]
        result["roofline"] = {
            "bound": "mfma", "achieved": leg["frac_forward_backward"] * FP32_MFMA_PEAK_TFLOPS, "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": leg["frac_forward_backward"], "frac_whole_step": leg.get("frac"), "traffic": None,
            "kernel": "training step: conv_igemm (forward + dgrad) and wgrad_kernel; algorithmic FLOPs = 3 x forward",
            "forward_ms": fw, "backward_ms": bw, "allreduce_exposed_ms": leg["allreduce_exposed_ms"], "sgd_ms": leg["sgd_ms"],
            "allreduce_sgd_ms": leg["allreduce_exposed_ms"] + leg["sgd_ms"],
            "forward_tflops": leg["forward_tflops"], "backward_tflops": leg["backward_tflops"]}
        for k in ("allreduce_alone_ms", "allreduce_busbw_GBps", "allreduce_overlap_fraction"):
            if k in leg:
                result["roofline"][k] = leg[k]
        if traffic:
            # HBM bytes of ONE training step, all library kernels (measured: see measure_hbm_traffic)
            result["roofline"]["traffic"] = traffic["_per_step"]
            top = sorted(((k, v) for k, v in traffic.items() if isinstance(v, dict) and "rocclr" not in k and "at::native" not in k),
                         key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:8]
            result["roofline"]["traffic_detail"] = {
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command (FETCH x2: gfx950); bytes per step",
                "top_kernels_MB_per_step": {k[:56]: round(v["hbm_bytes"] * v["launches"] / 3.0 / 1e6, 1) for k, v in top}}
        else:
            result["roofline"]["traffic_note"] = traffic_note
    if collective is not None:
        result["collective"] = collective
    dog.disarm()
    result["wall_s"] = dog.wall_report()
    if rank == 0:
        _emit(result)
    if dist is not None:
        dist.barrier()  # rank 0 may still be in its un-timed measurement passes: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
