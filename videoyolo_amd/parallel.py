"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on
ROCm) / xGMI; ``gloo`` on CPU for the tests.  What shards:

  inference   frames are independent: ``scatter_frames`` gives each rank its slice
              (gluon.utils.split_and_load(..., even_split=False), detect_yolo3.py:211-213); no collective.
  training    data parallel: one sum all-reduce of the flat gradient buffer per step (246.5 MB fp32),
              bucketed and overlapped with the backward pass (heads, stage 2, 1, 0);
              SyncBatchNorm layers all-reduce their [2][C] statistics forward and backward.
"""
import os


def _dist():
    import torch.distributed as dist
    return dist


def is_initialized():
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def world_size():
    return _dist().get_world_size() if is_initialized() else 1


def rank():
    return _dist().get_rank() if is_initialized() else 0


def collectives_active():
    """True when the collectives of this module really run: a process group of more than one rank — or of ONE rank with
    VY_FORCE_COLLECTIVES=1, which sends every broadcast / all-reduce / SyncBatchNorm exchange of the training path through
    the backend anyway (an all-reduce over one rank is the identity).  That is how the RCCL code path — the float64
    [2][C] all-reduce from a ctypes callback, async bucket all-reduces on a side stream, the gloo side group beside an
    NCCL default group, ``device_id=`` initialisation — is executed on a one-GPU box (tests/test_gpu_rccl.py)."""
    if not is_initialized():
        return False
    return world_size() > 1 or os.environ.get("VY_FORCE_COLLECTIVES", "0") not in ("", "0")


_failed = None


def fail_group(exc):
    """A collective raised inside a library callback (or anywhere a peer would otherwise be left waiting): print the
    traceback, remember the failure and ABORT the process group's communicators, so that this rank's pending and later
    collectives error out at once instead of queueing behind a dead exchange.  The peers notice through the backend's
    own error handling (RCCL's watchdog sees the aborted communicator; gloo sees the closed sockets) or, at the latest,
    through the group timeout (``init_process_group``: VY_DIST_TIMEOUT_S, default 600 s — never the 30-minute library
    default) — and both launchers (videoyolo_amd.launch, torchrun) terminate the other ranks once this one exits
    non-zero, which it does because the library call that invoked the callback now returns an error."""
    global _failed
    import traceback
    traceback.print_exception(type(exc), exc, exc.__traceback__)
    if _failed is None:
        _failed = exc
        try:
            from torch.distributed import distributed_c10d as c10d
            if hasattr(c10d, "_abort_process_group"):
                c10d._abort_process_group()
        except Exception as e:  # pragma: no cover - best effort; the error return below still ends the step
            print("videoyolo_amd.parallel: abort of the process group failed: %s" % e)
    return 1


def failed():
    """The exception that made this rank abort its process group, or None."""
    return _failed


def group_timeout():
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("VY_DIST_TIMEOUT_S", "600")))


def init_process_group(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # see videoyolo_amd/__init__.py; effective while no GPU call has been made
    import torch
    dist = _dist()
    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    r, w = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=r, world_size=w, timeout=group_timeout(), **kw)


def split_sizes(n, parts):
    """even_split=False slice sizes: the first n % parts slices get one extra element."""
    base, extra = divmod(n, parts)
    return [base + (1 if i < extra else 0) for i in range(parts)]


def scatter_frames(batch, r=None, w=None):
    """This rank's slice of a frame batch along axis 0 (numpy or torch)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    sizes = split_sizes(len(batch), w)
    lo = sum(sizes[:r])
    return batch[lo:lo + sizes[r]]


def allreduce_(tensor):
    """In-place sum over all ranks (no-op without a process group)."""
    if collectives_active():
        _dist().all_reduce(tensor)
    return tensor


def broadcast_(tensor, src=0):
    """In-place broadcast from rank `src` (no-op without a process group)."""
    if collectives_active():
        _dist().broadcast(tensor, src=src)
    return tensor


_host_group = None


def host_group():
    """A process group whose collectives run on HOST tensors (gloo), for control-plane agreement that must not
    synchronise the GPU stream: the default group if it is gloo, else a gloo group created beside the RCCL one.
    Creating it is collective — ``Trainer.__init__`` (which every rank runs) does it; returns None before that."""
    return _host_group


def _new_gloo_group():
    """The gloo group beside an RCCL default group (its own function: the world-4 / world-8 tests make it fail on one rank)."""
    dist = _dist()
    # single node (the launch contract): gloo over loopback — the container's hostname may not resolve
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    return dist.new_group(backend="gloo", timeout=group_timeout())


def make_host_group(side_group=None):
    """Collective: every rank must call it (idempotent).  side_group: None = a gloo default group IS the host group and
    only an RCCL default group gets a gloo group beside it; True forces the side-group path on any backend (tests)."""
    global _host_group
    if not collectives_active() or _host_group is not None:
        return _host_group
    dist = _dist()
    if dist.get_backend() == "gloo" and not side_group:
        _host_group = dist.group.WORLD
    else:
        try:
            grp = _new_gloo_group()
        except Exception as e:  # no usable host interface for gloo on THIS rank
            import warnings
            warnings.warn("no gloo side group on rank %d (%s)" % (rank(), e))
            grp = None
        # Agree on the outcome: a rank-local failure (interface, environment, fd limit) would otherwise leave some ranks
        # all-reducing on gloo and the others on RCCL at the first recorded forward — a hang.  One all-reduce(MIN) of
        # "I have the group" on the DEFAULT group; if any rank failed, every rank uses the fallback.
        import torch
        ok = torch.tensor([1 if grp is not None else 0], dtype=torch.int32,
                          device="cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            _host_group = grp
        else:
            import warnings
            warnings.warn("gloo side group unavailable on at least one rank: control-plane agreement falls back to "
                          "one-element all-reduces on the default group (RCCL: with a host read-back per recorded forward)")
            _host_group = False
    return _host_group


def any_rank(flag, device=None):
    """True on every rank iff `flag` is true on at least one: all-reduce(MAX) of one integer.  Every rank must
    call it.  Runs on the host group when there is one (no GPU synchronisation); otherwise on a device tensor of
    the default group (one small RCCL all-reduce and a host read-back)."""
    if not collectives_active():
        return bool(flag)
    import torch
    dist = _dist()
    if _host_group:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_host_group)
        return bool(t.item())
    t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                     device=device if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())


def sync_replicas(net, src=0):
    """Make every rank's replica identical to rank `src`'s: one broadcast of the flat device parameter
    buffer (weights, gamma/beta, running statistics) — what ``Trainer(kvstore='local')`` gives the
    reference for free, because there ONE initialised copy is pushed to all devices
    (train_yolov3.py:428,494,527-530).  With one process per GPU each rank runs its own
    ``net.initialize()``, and an unseeded one draws different weights on every rank.  Collective — every
    rank must reach it: ``Trainer.__init__`` calls it unconditionally; the recorded forward calls
    ``sync_replicas_if_any_dirty`` (below), which first AGREES across ranks on whether anybody wrote parameters."""
    import torch
    if not collectives_active() or net._dev_params is None:
        return False
    broadcast_(net._dev_params.view(torch.float32), src)
    net._replicas_synced = True
    written = getattr(net, "_params_written", None)
    if written is not None:
        written()  # the device buffer was written behind the library's back (conv mode 'split_bf16x3' caches weight images)
    return True


def sync_replicas_if_any_dirty(net, src=0):
    """Recorded-forward hook: re-broadcast rank `src`'s parameters iff ANY rank wrote parameters since the last
    broadcast (``if rank == 0: net.load_parameters(...)`` after the Trainer exists is legal in the reference,
    where one process owns all devices).  The decision is an all-reduce(MAX) of the per-rank dirty flags, so all
    ranks take the same branch: a rank-local flag alone would send only the writing ranks into the broadcast and
    hang the others in their next collective."""
    if not collectives_active() or net._dev_params is None:
        return False
    make_host_group()  # first call on every rank = their first recorded forward (or Trainer()): collective, then cached
    if any_rank(not net._replicas_synced, net._device):
        return sync_replicas(net, src)
    return False


XGMI_LINK_GBPS_PER_DIRECTION = 76.8   # 7 point-to-point links per GPU, ~153.6 GB/s each counting both directions


def device_identity(device=None, extra=None):
    """What THIS rank runs on, as plain data: rank, local rank, host, pid, device ordinal, name, CUs, HBM bytes, PCI bus id,
    the environment that decides how ranks see each other's memory.  `extra`: caller's per-rank facts (e.g. the stream-K
    placement probe's verdict of its net)."""
    import socket
    import torch
    d = {"rank": rank(), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "host": socket.gethostname(), "pid": os.getpid(),
         "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
         "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))}
    if device is not None and torch.cuda.is_available() and str(device).startswith("cuda"):
        dev = torch.device(device)
        pr = torch.cuda.get_device_properties(dev)
        bus = None
        if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        d.update(device=dev.index if dev.index is not None else torch.cuda.current_device(), name=pr.name,
                 cus=int(pr.multi_processor_count), hbm_bytes=int(pr.total_memory), pci_bus_id=bus,
                 gcn_arch=getattr(pr, "gcnArchName", None), uuid=str(getattr(pr, "uuid", "")) or None)
    else:
        d.update(device=str(device) if device is not None else "cpu")
    if extra:
        d.update(extra)
    return d


def describe_group(device=None, extra=None):
    """Collective (every rank calls it): `{backend, world, rccl_version, ipc_mode, devices: [per-rank identity], ...}` —
    the proof of WHAT a multi-rank number ran on, gathered with all_gather_object.  `distinct_devices` counts the different
    (host, PCI bus id) pairs: N ranks on fewer than N devices is reported, not inferred.  Without a process group: the
    one-rank description."""
    import torch
    me = device_identity(device, extra)
    out = {"backend": None, "world": 1, "rccl_version": None, "ipc_mode": "dmabuf" if me["ipc_mode_legacy"] == "0" else
           "legacy (HSA_ENABLE_IPC_MODE_LEGACY=%s)" % me["ipc_mode_legacy"], "devices": [me]}
    if is_initialized():
        dist = _dist()
        out["backend"], out["world"] = dist.get_backend(), dist.get_world_size()
        if out["backend"] == "nccl":
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # pragma: no cover
                out["rccl_version"] = "unknown (%s)" % e
        got = [None] * out["world"]
        dist.all_gather_object(got, me)
        out["devices"] = got
    keys = {(d.get("host"), d.get("pci_bus_id") or d.get("uuid") or d.get("device")) for d in out["devices"]}
    out["distinct_devices"] = len(keys)
    out["ranks_in_order"] = [d.get("rank") for d in out["devices"]] == list(range(out["world"]))
    return out


def allreduce_expectation(nbytes, world):
    """What an all-reduce of nbytes over `world` fully connected GPUs can reach on xGMI: one ring uses
    one link per neighbour (bus bandwidth <= 76.8 GB/s); reduce-scatter + all-gather over all world - 1 links at once
    <= (world - 1) x 76.8."""
    if world < 2:
        return None
    f = 2.0 * (world - 1) / world * nbytes
    all_links = (world - 1) * XGMI_LINK_GBPS_PER_DIRECTION
    return {"all_links_rs_ag_busbw_GBps": all_links, "single_ring_busbw_GBps": XGMI_LINK_GBPS_PER_DIRECTION,
            "ms_at_all_links": f / (all_links * 1e9) * 1e3, "ms_at_single_ring": f / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3}


def preflight(device, nbytes=246_504_196, iters=5, min_busbw_GBps=None, max_barrier_ms=None, budget_s=5.0):
    """Collective pre-flight of a multi-rank run (every rank calls it; <= a few seconds): `iters` barriers (latency) and
    `iters` all-reduces of an nbytes fp32 buffer (the flat gradient buffer's size: 246.5 MB) — correctness of the sum, time,
    bus bandwidth against the xGMI expectation.  Returns a dict with "ok" and, when not ok, a one-line "reason" that is THE
    SAME on every rank (the verdict is computed from all-reduced numbers), so that all ranks leave together instead of
    some of them entering the legs.  A mis-set fabric shows up here in seconds, not as a hang in the first training step:
    the group's timeout bounds every collective (VY_DIST_TIMEOUT_S)."""
    import time
    import torch
    dist = _dist()
    w, r = world_size(), rank()
    on_gpu = str(device).startswith("cuda")
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)
    t_start = time.perf_counter()
    n = max(1, int(nbytes) // 4)
    buf = torch.empty(n, dtype=torch.float32, device=device)
    bar = []
    for _ in range(iters):
        sync()
        t0 = time.perf_counter()
        dist.barrier()
        sync()
        bar.append(1e3 * (time.perf_counter() - t0))
    ts, good = [], True
    for i in range(iters):
        buf.fill_(float(r + 1))
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(buf)
        sync()
        ts.append(1e3 * (time.perf_counter() - t0))
        want = w * (w + 1) / 2.0
        good = good and bool((buf[:: max(1, n // 4096)] == want).all().item()) and float(buf[-1].item()) == want
    # the verdict from numbers every rank shares: max over ranks of the median times, min of the correctness flags
    v = torch.tensor([sorted(ts)[len(ts) // 2], sorted(bar)[len(bar) // 2], 0.0 if good else 1.0, time.perf_counter() - t_start],
                     dtype=torch.float64, device=device if on_gpu else "cpu")
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    ar_ms, bar_ms, bad, took = (float(t) for t in v.tolist())
    busbw = 2.0 * (w - 1) / w * (n * 4) / (ar_ms * 1e-3) / 1e9 if w > 1 else None
    out = {"ok": True, "world": w, "backend": dist.get_backend(), "allreduce_bytes": n * 4, "allreduce_ms": ar_ms,
           "allreduce_busbw_GBps": busbw, "barrier_ms": bar_ms, "iters": iters, "seconds": took,
           "expectation": allreduce_expectation(n * 4, w) if on_gpu else None}
    if out["expectation"]:
        out["frac_of_all_links"] = busbw / out["expectation"]["all_links_rs_ag_busbw_GBps"]
        out["frac_of_single_ring"] = busbw / out["expectation"]["single_ring_busbw_GBps"]
    reasons = []
    if bad:
        reasons.append("all-reduce(sum) of rank + 1 over %d ranks did not give %g on every rank" % (w, w * (w + 1) / 2.0))
    if min_busbw_GBps is not None and busbw is not None and busbw < min_busbw_GBps:
        reasons.append("all-reduce bus bandwidth %.1f GB/s < floor %.1f GB/s (xGMI single ring would give %.1f): the ranks do not "
                       "talk over the fabric" % (busbw, min_busbw_GBps, XGMI_LINK_GBPS_PER_DIRECTION))
    if max_barrier_ms is not None and bar_ms > max_barrier_ms:
        reasons.append("barrier latency %.2f ms > %.2f ms" % (bar_ms, max_barrier_ms))
    if took > budget_s:
        out["over_budget"] = True   # reported, not fatal: the first collective of a communicator also builds its rings
    if reasons:
        out.update(ok=False, reason="; ".join(reasons))
    del buf
    return out


def gather_detections(ids, scores, bboxes, total=None):
    """Gather of per-rank (B_r, R, .) results (detect_yolo3.py:233 as_numpy concat): every rank gets the rows of the whole
    batch, in rank order.  ``total`` = frames of the whole batch when the caller knows it (the slices are then
    ``split_sizes(total, world)``, as ``scatter_frames`` cut them): no host-side size exchange, nothing waits for the
    GPU — the form a pipelined loop needs (videoyolo_amd/stream.py).  Without it the sizes are exchanged first."""
    import torch
    if not collectives_active():   # one rank (unless VY_FORCE_COLLECTIVES sends it through the backend anyway)
        return ids, scores, bboxes
    dist = _dist()
    w, r = world_size(), rank()
    packed = torch.cat([ids, scores, bboxes], dim=-1).contiguous()
    if total is not None:
        sizes = split_sizes(int(total), w)
        if sizes[r] != int(packed.shape[0]):
            # `total` and the slices must come from ONE decision every rank takes identically (HostFedDetector.submit checks
            # len(clip_batch) == global_batch on every rank before anything is scattered): then this raises on all ranks
            # together, before any collective, and the group stays usable.  A caller that passes a `total` only SOME ranks
            # disagree with has left the others in the all-gather; they end with the group timeout (VY_DIST_TIMEOUT_S)
            raise ValueError("rank %d holds %d frames of a batch of %d, expected %d" % (r, packed.shape[0], total, sizes[r]))
    else:
        sizes = [None] * w
        dist.all_gather_object(sizes, int(packed.shape[0]))
    if len(set(sizes)) == 1:
        full = torch.empty((w * sizes[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        dist.all_gather(list(full.split(sizes[0], 0)), packed)   # (views of one buffer: RCCL gathers in place)
    else:  # even_split=False with a remainder: one broadcast per rank
        outs = [packed if i == r else torch.empty((s,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
                for i, s in enumerate(sizes)]
        for i in range(w):
            dist.broadcast(outs[i], src=i)
        full = torch.cat(outs, 0)
    return full[..., 0:1], full[..., 1:2], full[..., 2:]


class SyncBatchNormHook(object):
    """Installs the statistics all-reduce for SyncBatchNorm(num_devices) (train_yolov3.py:352-354):
    the library calls back with a device pointer into the net's workspace and a count of doubles.
    ``yolo3_darknet53(..., norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': n})`` installs it by itself at
    the first recorded forward (model.YOLOV3._ensure_sync_bn); it survives re-planning (multi-scale training,
    train_yolov3.py:258-271): the workspace tensor is looked up per call, the callback registration lives in the
    library's per-net training state."""

    def __init__(self, net):
        import torch
        from . import _lib
        self.net = net
        self.calls = []   # doubles per statistics exchange of the CURRENT step, in call order (6 fwd + 6 bwd)

        def cb(user, ptr, count):
            try:
                ws = net._ws
                off = int(ptr) - ws.data_ptr()
                if off < 0 or off + 8 * count > ws.numel():
                    raise RuntimeError("SyncBatchNorm statistics pointer outside the bound workspace")
                view = ws[off:off + 8 * count].view(torch.float64)
                _dist().all_reduce(view)
                self.calls.append(int(count))
                return 0
            except Exception as e:  # surfaced as a library error on this rank; the group is aborted for the peers
                return fail_group(e)
        self._cb = _lib.ALLREDUCE_CB(cb)
        net._cb_keep.append(self._cb)
        net._sync_hook = self
        _lib.check(net._lib.vy_net_set_sync_bn(net._h, world_size(), self._cb, None))

    def begin_step(self):
        """Called by the recorded forward: the call log covers one step (it is instrumentation, not state)."""
        self.calls = []


def grad_bucket_table(net):
    """The gradient buckets of a backward pass, in the order the library reports them (csrc/train.hip: heads -> stages.2 ->
    stages.1 -> stages.0, the order backward finishes them): [(name, element offset, element count)] over the flat gradient
    buffer — host-side mirror of the library's bookkeeping, from the parameter table alone (no device needed).  Ranges are
    the contiguous extents of each group's TRAINABLE tensors (weight, gamma, beta, bias), ends rounded up to 64 elements
    (the 256-byte alignment of the parameter table).  Every rank derives the same table: the all-reduces of bucket i pair up
    across ranks by construction."""
    groups = [("heads", lambda n: not n.startswith("stages.")), ("stages.2", lambda n: n.startswith("stages.2.")),
              ("stages.1", lambda n: n.startswith("stages.1.")), ("stages.0", lambda n: n.startswith("stages.0."))]
    out = []
    for gname, pred in groups:
        lo, hi = None, 0
        for p in net.collect_params().values():
            if not pred(p.name) or p.name.rsplit(".", 1)[1] in ("running_mean", "running_var"):
                continue
            lo = p.offset if lo is None else min(lo, p.offset)
            hi = max(hi, p.offset + ((p.size + 63) & ~63))
        if lo is not None and hi > lo:
            out.append((gname, int(lo), int(hi - lo)))
    return out


class GradBucketOverlap(object):
    """Bucketed gradient all-reduce overlapped with backward: the library reports each finished
    contiguous gradient range; it is all-reduced on a side stream after an event on the compute
    stream, while the remaining backward kernels keep the compute stream busy."""

    def __init__(self, net):
        import torch
        from . import _lib
        self.net = net
        self.stream = torch.cuda.Stream(device=net._device) if net._device is not None else None
        self.pending = []
        self.launched = []       # (element offset, element count) of the buckets handed out since the last finish()
        self.last_launched = []  # the same for the step that finish() closed last (instrumentation)

        def cb(user, off, count):
            try:
                if not collectives_active():
                    return 0
                self.launched.append((int(off), int(count)))
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(net._device))
                with torch.cuda.stream(self.stream):
                    self.stream.wait_event(ev)
                    work = _dist().all_reduce(net._grads[off:off + count], async_op=True)
                self.pending.append(work)
                return 0
            except Exception as e:
                return fail_group(e)
        self._cb = _lib.GRAD_BUCKET_CB(cb)
        net._cb_keep.append(self._cb)
        _lib.check(net._lib.vy_net_set_grad_bucket_cb(net._h, self._cb, None))

    def remove(self):
        """Back to one all-reduce after backward: unregister the bucket callback."""
        from . import _lib
        self.finish()
        _lib.check(self.net._lib.vy_net_set_grad_bucket_cb(self.net._h, _lib.GRAD_BUCKET_CB(), None))

    def finish(self):
        import torch
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.launched:
            self.last_launched, self.launched = self.launched, []
        if self.stream is not None:
            torch.cuda.current_stream(self.net._device).wait_stream(self.stream)
