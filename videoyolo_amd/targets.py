"""Prefetch target generation — the CPU-side stage that builds the five fixed target tensors the
training call consumes.

Host-side mirror of ``YOLOV3PrefetchTargetGenerator`` (models/definitions/yolo/yolo_target.py:13-148),
which the reference runs in DataLoader worker processes (train_yolov3.py:260-271,
models/definitions/yolo/transforms.py:185-197,259-277).  Same host language (Python/numpy), but
vectorised over the batch and the gt boxes instead of the reference's Python double loop; the
quirks that decide the numbers are kept: centre -> cell by int() truncation of the FLOAT64 value
`gtx / orig_width * width` (:115-116; an np.float32 scalar with Python ints is float64 under the
reference's pinned NumPy 1.x), log(max(w,1)/anchor) (:121-122), weight 2 - w*h/(W*H) (:123, fp32
product then float64), later gt boxes overwrite earlier ones in the same (cell, anchor) slot, and rows
after the first invalid (-1) gt box are ignored (:107-108).
"""
import numpy as np

ANCHORS = np.array([[116, 90], [156, 198], [373, 326], [30, 61], [62, 45], [59, 119],
                    [10, 13], [16, 30], [33, 23]], np.float32)  # stride 32, 16, 8 (wrappers.py:80-84 reversed)
STRIDES = (32, 16, 8)


def num_anchors(height, width):
    return sum(3 * (height // s) * (width // s) for s in STRIDES)


class YOLOV3PrefetchTargetGenerator(object):
    def __init__(self, num_class):
        self._num_class = num_class

    def __call__(self, height, width, gt_boxes, gt_ids, gt_mixratio=None, device=None):
        """gt_boxes (B,M,4) corner pixels, gt_ids (B,M,1), -1 rows = padding.
        numpy inputs and device=None: the DataLoader-worker variant below (numpy in, numpy out, like
        the reference); device tensors or device='cuda:N': the HIP kernel (`on_device`).
        Returns objectness (B,N,1), center_targets (B,N,2), scale_targets (B,N,2), weights (B,N,2),
        class_targets (B,N,C) in the order the training forward expects (stride 32,16,8; cell; anchor)."""
        if device is not None or _is_device_tensor(gt_boxes):
            return self.on_device(height, width, gt_boxes, gt_ids, gt_mixratio, device=device)
        gt_boxes = np.asarray(gt_boxes, np.float32)
        gt_ids = np.asarray(gt_ids, np.float32)
        B, M = gt_boxes.shape[:2]
        C = self._num_class
        N = num_anchors(height, width)
        obj = np.zeros((B, N, 1), np.float32)
        ctr = np.zeros((B, N, 2), np.float32)
        scl = np.zeros((B, N, 2), np.float32)
        wts = np.zeros((B, N, 2), np.float32)
        cls = np.full((B, N, C), -1, np.float32)
        if M == 0:
            return obj, ctr, scl, wts, cls
        gtw = gt_boxes[..., 2] - gt_boxes[..., 0]
        gth = gt_boxes[..., 3] - gt_boxes[..., 1]
        gtx = gt_boxes[..., 0] + gtw / np.float32(2)  # gluoncv BBoxCornerToCenter, fp32
        gty = gt_boxes[..., 1] + gth / np.float32(2)
        # best anchor by IoU of zero-centred boxes: inter = min(w)*min(h) (both centred at 0)
        iw = np.minimum(gtw[..., None], ANCHORS[:, 0])
        ih = np.minimum(gth[..., None], ANCHORS[:, 1])
        inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
        union = gtw[..., None] * gth[..., None] + ANCHORS[:, 0] * ANCHORS[:, 1] - inter
        iou = np.where(union > 0, inter / np.where(union > 0, union, 1), 0)
        match = iou.argmax(axis=-1)  # (B,M)
        valid = (gt_boxes >= 0).all(axis=-1)
        valid = np.logical_and.accumulate(valid, axis=1)  # stop at the first invalid row
        layer = match // 3
        stride = np.array(STRIDES)[layer]
        fw, fh = width // stride, height // stride
        # float64 division then truncation, as int(gtx / orig_width * width) does under NumPy 1.x
        fx = gtx.astype(np.float64) / float(width) * fw.astype(np.float64)
        fy = gty.astype(np.float64) / float(height) * fh.astype(np.float64)
        loc_x, loc_y = fx.astype(np.int64), fy.astype(np.int64)
        cells = np.array([0] + list(np.cumsum([(height // s) * (width // s) for s in STRIDES])))[:-1]
        base = np.array([0] + list(np.cumsum([3 * (height // s) * (width // s) for s in STRIDES])))[:-1]
        n_idx = base[layer] + (loc_y * fw + loc_x) * 3 + (match % 3)
        # a centre on the right / bottom image edge indexes past its scale's cells: the reference writes a
        # (cell, anchor) pair there that `_slice` (:139-148) discards
        valid &= (loc_y * fw + loc_x) < fw * fh
        del cells
        anc = ANCHORS[match]
        for b in range(B):  # in-order assignment so later boxes overwrite earlier ones
            for m in np.nonzero(valid[b])[0]:
                n = n_idx[b, m]
                ctr[b, n, 0] = fx[b, m] - loc_x[b, m]
                ctr[b, n, 1] = fy[b, m] - loc_y[b, m]
                for j, g in enumerate((gtw[b, m], gth[b, m])):
                    scl[b, n, j] = np.log(g / anc[b, m, j]) if g >= 1 else np.log(1.0 / float(anc[b, m, j]))
                wts[b, n, :] = 2.0 - float(gtw[b, m] * gth[b, m]) / width / height
                obj[b, n, 0] = gt_mixratio[b, m, 0] if gt_mixratio is not None else 1
                cls[b, n, :] = 0
                cls[b, n, int(gt_ids[b, m, 0])] = 1
        return obj, ctr, scl, wts, cls

    def on_device(self, height, width, gt_boxes, gt_ids, gt_mixratio=None, device=None):
        """The same five tensors built on the GPU by libvyolo's vy_prefetch_targets (csrc/targets.hip):
        a fill launch + one scatter launch for the whole batch; the outputs stay in HBM for the training
        call.  Inputs may be numpy arrays or device tensors; returns torch device tensors."""
        import ctypes
        import torch
        from . import _lib
        lib = _lib.load()
        dev = torch.device(device) if device is not None else gt_boxes.device
        gb = torch.as_tensor(gt_boxes, dtype=torch.float32).to(dev).contiguous()
        B, M = int(gb.shape[0]), int(gb.shape[1])
        gi = torch.as_tensor(gt_ids, dtype=torch.float32).to(dev).contiguous().reshape(B, M)
        gm = None
        if gt_mixratio is not None:
            gm = torch.as_tensor(gt_mixratio, dtype=torch.float32).to(dev).contiguous().reshape(B, M)
        C, N = self._num_class, num_anchors(height, width)
        out = [torch.empty((B, N, k), dtype=torch.float32, device=dev) for k in (1, 2, 2, 2, C)]

        def ptr(t):
            return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None

        with torch.cuda.device(dev):
            _lib.check(lib.vy_prefetch_targets(ptr(gb), ptr(gi), ptr(gm), B, M, int(height), int(width), C,
                                               *[ptr(t) for t in out],
                                               ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return tuple(out)


def _is_device_tensor(a):
    return type(a).__module__.startswith("torch") and getattr(a, "is_cuda", False)


def synthetic_gt(batch, size, num_class, m=8, seed=0):
    """M gt boxes / image, uniform centres, w,h ~ U(32,256) clipped to the image
    (scaled for small images), class ~ U{0..C-1}."""
    rng = np.random.default_rng(seed)
    lo, hi = (32, 256) if size >= 320 else (size / 8, size / 1.5)
    c = rng.uniform(0, size, (batch, m, 2))
    wh = rng.uniform(lo, hi, (batch, m, 2))
    x1y1 = np.clip(c - wh / 2, 0, size - 2)
    x2y2 = np.clip(c + wh / 2, x1y1 + 1, size - 1)
    boxes = np.concatenate([x1y1, x2y2], -1).astype(np.float32)
    ids = rng.integers(0, num_class, (batch, m, 1)).astype(np.float32)
    return boxes, ids
