"""The reference's detect loop as a pipeline (SURVEY.md 8f row 3, 8e "Inference").

``detect_yolo3.py:209-233`` does, per batch: ``split_and_load(batch, ctx_list, even_split=False)`` (host -> every GPU),
``net(x)`` per device, ``as_numpy`` of ``(ids, scores, bboxes)`` (every GPU -> host), one after the other — the copies
and the kernels of a batch never overlap, and neither do two batches.  Here one process drives one GPU, and a batch
goes through three HIP streams:

    host (pinned uint8 frames)  --copy-in stream-->  device uint8 -> resize + to_tensor + normalise (csrc/preproc.hip, one
                                                     kernel, on the copy stream too: it runs beside the previous batch's convs)
    device fp32 batch  --compute stream-->  net(x)  [-> all-gather of the (B, R, 6) rows over RCCL when there are several ranks]
    device rows   --copy-out stream-->  host (pinned fp32 rows)

with ``depth`` buffer slots (2: double-buffered), HIP events between the streams and NO host synchronisation except when
the caller asks for a finished batch: batch i + 1 is copied in and batch i - 1 copied out while batch i computes.
Nothing is computed on the host, and there is no fallback: without the HIP library every call raises.

Frames that already lie in PINNED host memory are copied in directly; pageable frames go through the slot's pinned staging
buffer first — one more host copy, and a slow one when done by a single thread (CPU stores into hipHostMalloc'ed memory:
tools/stream_soak.py reads 593 against 1 826 frames/s at 416 x 416, batch 16, where a batch computes in 8.7 ms), so a decoder
should own pinned output buffers.  `tools/stream_soak.py`: 1 200 batches over 2 and 3 slots, pinned and pageable, every
result bit-equal to the synchronous path (profiles/r05_stream_soak.txt).

Multi-GPU: frames shard (``parallel.scatter_frames``: this rank's slice of the clip batch, ``even_split=False`` sizes),
no data-path collective on the way in; the only exchange is the result gather (154 KB per 64-frame slice)."""
import ctypes

import numpy as np

from . import _lib, parallel
from .transforms import MEAN, STD


def _torch():
    import torch
    return torch


class HostFedDetector(object):
    """net: a ``yolo3_darknet53`` object on its device.  ``global_batch`` frames of ``src_hw`` uint8 pixels arrive per
    step (the same clip batch on every rank — a shared decoder / file; a rank touches only its slice), are resized to
    ``size`` and detected; ``gather`` brings every rank's rows to all ranks (rank 0 copies them to the host)."""

    def __init__(self, net, global_batch, src_hw, size, depth=2, gather=True, mean=MEAN, std=STD):
        torch = _torch()
        if net._device is None:
            raise RuntimeError("net.collect_params().reset_ctx(device) first")
        self.net, self.size, self.depth = net, int(size), int(depth)
        if self.depth < 1:
            raise ValueError("depth must be at least 1 slot, got %r" % (depth,))
        depth = self.depth
        self.dev = net._device
        self.world, self.rank = parallel.world_size(), parallel.rank()
        self.global_batch = int(global_batch)
        self.sizes = parallel.split_sizes(self.global_batch, self.world)
        self.b = self.sizes[self.rank]
        if self.b < 1:
            raise ValueError("rank %d gets no frame of a batch of %d" % (self.rank, self.global_batch))
        self.gather = bool(gather) and parallel.collectives_active()   # (> 1 rank, or one rank with VY_FORCE_COLLECTIVES)
        self.h, self.w = int(src_hw[0]), int(src_hw[1])
        self._mean, self._std = np.asarray(mean, np.float32), np.asarray(std, np.float32)
        self._lib = _lib.load()
        rows = net._out_rows()
        out_b = self.global_batch if self.gather else self.b
        with torch.cuda.device(self.dev):
            self.s_in, self.s_out = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
            self.pin_in = [torch.empty((self.b, self.h, self.w, 3), dtype=torch.uint8).pin_memory() for _ in range(depth)]
            self.dev_in = [torch.empty((self.b, self.h, self.w, 3), dtype=torch.uint8, device=self.dev) for _ in range(depth)]
            self.x = [torch.empty((self.b, 3, self.size, self.size), dtype=torch.float32, device=self.dev) for _ in range(depth)]
            self.dev_out = [torch.empty((out_b, rows, 6), dtype=torch.float32, device=self.dev) for _ in range(depth)]
            self.pin_out = [torch.empty((out_b, rows, 6), dtype=torch.float32).pin_memory() for _ in range(depth)]
            ev = lambda: [torch.cuda.Event() for _ in range(depth)]  # noqa: E731
            self.e_in, self.e_pre, self.e_cmp, self.e_out = ev(), ev(), ev(), ev()
        self._n = 0
        self._busy = [False] * depth
        self._src = [None] * depth

    # -- one batch -------------------------------------------------------------------------------------------------
    def submit(self, clip_batch):
        """Enqueue one clip batch: (global_batch, h, w, 3) uint8 on the host (numpy, or a torch CPU tensor — pinned: copied
        in directly; pageable: through this slot's pinned staging buffer).  Returns the slot to pass to ``result``.  Never
        waits for the GPU unless every slot is still in flight."""
        torch = _torch()
        k = self._n % self.depth
        if self._busy[k]:
            raise RuntimeError("slot %d still holds an uncollected batch: call result() before submitting %d more"
                               % (k, self.depth))
        # the checks every rank evaluates identically come FIRST: a wrong clip batch raises on all ranks together, before
        # any of them has entered the result gather (a one-sided raise would leave the others waiting in the collective)
        if len(clip_batch) != self.global_batch:
            raise ValueError("expected a clip batch of %d frames, got %d" % (self.global_batch, len(clip_batch)))
        dt = getattr(clip_batch, "dtype", None)
        if dt is not None and str(dt).replace("torch.", "") != "uint8":
            raise TypeError("frames must be uint8 (decoded images), got %s" % dt)
        mine = parallel.scatter_frames(clip_batch, self.rank, self.world)
        if tuple(mine.shape) != (self.b, self.h, self.w, 3):
            raise ValueError("expected %s uint8 frames for this rank, got %s" % ((self.b, self.h, self.w, 3), tuple(mine.shape)))
        src = mine if isinstance(mine, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mine))
        if src.dtype != torch.uint8:
            raise TypeError("frames must be uint8 (decoded images), got %s" % src.dtype)
        cur = torch.cuda.current_stream(self.dev)
        if src.is_pinned() and src.is_contiguous():
            # the frames already lie in pinned host memory (a decoder's output buffer): copied in straight from there.  The
            # caller must leave them alone until this batch's result has been collected
            staged = src
        else:
            if self._n >= self.depth:
                self.e_in[k].synchronize()        # the copy-in of this slot's previous batch has left the staging buffer
            self.pin_in[k].copy_(src)             # pageable memory: one host copy into this slot's pinned staging buffer
            staged = self.pin_in[k]
        self._src[k] = staged                     # (kept alive until the slot is reused)
        with torch.cuda.device(self.dev):
            with torch.cuda.stream(self.s_in):
                if self._n >= self.depth:
                    # slot k's previous batch: its pre-processing has read dev_in[k] (same stream: ordered) and its
                    # forward has read x[k] (the compute stream's e_cmp)
                    self.s_in.wait_event(self.e_cmp[k])
                self.dev_in[k].copy_(staged, non_blocking=True)
                self.e_in[k].record(self.s_in)
                # resize + to_tensor + normalise ON THE COPY STREAM: a bandwidth-bound kernel that runs beside the
                # previous batch's matrix-bound convs instead of in front of this batch's
                _lib.check(self._lib.vy_preprocess_resize_frames(
                    ctypes.c_void_p(self.dev_in[k].data_ptr()), self.h, self.w, ctypes.c_void_p(self.x[k].data_ptr()), self.b,
                    self.size, self.size, self._mean.ctypes.data_as(ctypes.c_void_p), self._std.ctypes.data_as(ctypes.c_void_p),
                    ctypes.c_void_p(self.s_in.cuda_stream)))
                self.e_pre[k].record(self.s_in)
            cur.wait_event(self.e_pre[k])
            if self._n >= self.depth:
                cur.wait_event(self.e_out[k])             # dev_out[k] has been copied out
            ids, scores, bboxes = self.net(self.x[k])
            if self.gather:
                g_ids, g_scores, g_bboxes = parallel.gather_detections(ids, scores, bboxes, total=self.global_batch)
                torch.cat([g_ids, g_scores, g_bboxes], dim=-1, out=self.dev_out[k])
            else:
                torch.cat([ids, scores, bboxes], dim=-1, out=self.dev_out[k])
            self.e_cmp[k].record(cur)
            if self.rank == 0 or not self.gather:
                with torch.cuda.stream(self.s_out):
                    self.s_out.wait_event(self.e_cmp[k])
                    self.pin_out[k].copy_(self.dev_out[k], non_blocking=True)
                    self.e_out[k].record(self.s_out)
            else:
                self.e_out[k].record(cur)
        self._busy[k] = True
        self._n += 1
        return k

    def result(self, slot):
        """Wait for the batch in ``slot`` and return (ids (B,R,1), scores (B,R,1), bboxes (B,R,4)) as numpy views of the
        pinned output buffer (valid until the slot is submitted again).  B = the whole clip batch on rank 0 when gathering
        (the other ranks get None), this rank's slice otherwise."""
        if not (0 <= int(slot) < self.depth) or not self._busy[slot]:
            raise RuntimeError("slot %r holds no submitted batch (never submitted, or its result was already collected)" % (slot,))
        self.e_out[slot].synchronize()
        self._busy[slot] = False
        if self.gather and self.rank != 0:
            return None
        a = self.pin_out[slot].numpy()
        return a[..., 0:1], a[..., 1:2], a[..., 2:6]

    # -- a stream of batches ---------------------------------------------------------------------------------------
    def run(self, batches):
        """for out in det.run(iterable of clip batches): the results in order, each yielded while later batches are
        already in flight (``depth - 1`` of them)."""
        pending = []
        for cb in batches:
            if len(pending) == self.depth:
                yield self.result(pending.pop(0))
            pending.append(self.submit(cb))
        for k in pending:
            yield self.result(k)
