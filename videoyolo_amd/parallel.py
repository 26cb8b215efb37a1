"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` over RCCL (backend "nccl" on
ROCm) / xGMI; ``gloo`` on CPU for the tests.  What shards:

  inference   frames are independent: ``scatter_frames`` gives each rank its slice
              (gluon.utils.split_and_load(..., even_split=False), detect_yolo3.py:211-213); no collective.
  training    data parallel: one sum all-reduce of the flat gradient buffer per step (246.5 MB fp32),
              bucketed and overlapped with the backward pass (heads, stage 2, 1, 0);
              SyncBatchNorm layers all-reduce their [2][C] statistics forward and backward.
"""
import ctypes
import os

import numpy as np


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


def make_host_group():
    """Collective: every rank must call it (idempotent)."""
    global _host_group
    if not collectives_active() or _host_group is not None:
        return _host_group
    dist = _dist()
    if dist.get_backend() == "gloo":
        _host_group = dist.group.WORLD
    else:
        try:
            # single node (the launch contract): gloo over loopback — the container's hostname may not resolve
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            grp = dist.new_group(backend="gloo", timeout=group_timeout())
        except Exception as e:  # no usable host interface for gloo on THIS rank
            import warnings
            warnings.warn("no gloo side group on rank %d (%s)" % (rank(), e))
            grp = None
        # Agree on the outcome: a rank-local failure (interface, environment, fd limit) would otherwise leave some ranks
        # all-reducing on gloo and the others on RCCL at the first recorded forward — a hang.  One all-reduce(MIN) of
        # "I have the group" on the DEFAULT group; if any rank failed, every rank uses the fallback.
        import torch
        ok = torch.tensor([1 if grp is not None else 0], dtype=torch.int32,
                          device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            _host_group = grp
        else:
            import warnings
            warnings.warn("gloo side group unavailable on at least one rank: control-plane agreement falls back to "
                          "one-element RCCL all-reduces with a host read-back per recorded forward")
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
            # only THIS rank can see that its slice is wrong: the peers are about to enter the all-gather.  Abort the
            # group so that they error out at once instead of waiting for the timeout, then raise
            e = ValueError("rank %d holds %d frames of a batch of %d, expected %d" % (r, packed.shape[0], total, sizes[r]))
            if w > 1:
                fail_group(e)
            raise e
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
