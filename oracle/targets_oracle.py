"""CPU restatement of YOLOV3PrefetchTargetGenerator (models/definitions/yolo/yolo_target.py:13-148)
— the CPU-worker stage that builds the five fixed target tensors the training call consumes
(SURVEY.md §8f row 1).  TEST INFRASTRUCTURE ONLY.  Follows the reference loop literally, including
int() truncation (:115-116), max(gtw, 1) (:121), the weight 2 - w*h/(W*H) (:123) and the break at
the first invalid gt row (:107-108).  nd.contrib.box_iou is restated as corner IoU without +1
[UPSTREAM-RECALLED]; BBoxCornerToCenter (gluoncv.nn.bbox) as fp32 `xmin + width / 2` [UPSTREAM-RECALLED].

Scalar types.  The reference pins Python 3.6.3 (environment.yml:6), i.e. NumPy 1.x, where an
np.float32 SCALAR combined with a Python int is promoted to float64 (both operands are scalars, so
value-based casting does not apply).  Hence in the loop (:115-123)
  gtx / orig_width * width                  float64   -> int() truncates the float64 value
  gtx / orig_width * width - loc_x          float64   -> rounded to fp32 by the NDArray store
  max(gtw, 1) / np_anchors[match, 0]        float32 / float32 when gtw >= 1 (max returns gtw);
                                            int / float32 = float64 when 1 > gtw
  2.0 - gtw * gth / orig_width / orig_height   fp32 product, then float64
This container has NumPy 2.x (NEP 50: Python ints are weak, everything would stay fp32), so the
float64 steps are written out explicitly below."""
import numpy as np

from . import yolo3_oracle as O


def anchors_offsets_featmaps(height, width):
    """What the transform harvests from net(zeros) in train_mode (transforms.py:190-193,
    yolo3.py:1189-1192): per-scale anchors (1,1,3,2), offsets (1,HW,1,2), feature-map sizes —
    order stride 32, 16, 8."""
    anchors, offsets, fms = [], [], []
    for i in range(3):
        s = O.STRIDES[::-1][i]
        h, w = height // s, width // s
        anchors.append(np.array(O.ANCHORS[::-1][i], np.float32).reshape(1, 1, 3, 2))
        gx, gy = np.meshgrid(np.arange(w), np.arange(h))
        offsets.append(np.stack([gx, gy], -1).astype(np.float32).reshape(1, h * w, 1, 2))
        fms.append((h, w))
    return anchors, offsets, fms


def _iou_centered(anchor_wh, gt_wh):
    """box_iou of zero-centred anchor boxes (9,4) vs zero-centred gt boxes (M,4)."""
    a = np.concatenate([-0.5 * anchor_wh, 0.5 * anchor_wh], -1)[:, None]
    g = np.concatenate([-0.5 * gt_wh, 0.5 * gt_wh], -1)[None]
    iw = np.clip(np.minimum(a[..., 2], g[..., 2]) - np.maximum(a[..., 0], g[..., 0]), 0, None)
    ih = np.clip(np.minimum(a[..., 3], g[..., 3]) - np.maximum(a[..., 1], g[..., 1]), 0, None)
    inter = iw * ih
    ua = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1]) - inter
    return np.where(ua > 0, inter / np.where(ua > 0, ua, 1), 0).astype(np.float32)


def prefetch_targets(num_class, height, width, gt_boxes, gt_ids, gt_mixratio=None):
    """gt_boxes (B,M,4) corner px, gt_ids (B,M,1); -1 rows = padding.  Returns objectness (B,N,1),
    center_targets (B,N,2), scale_targets (B,N,2), weights (B,N,2), class_targets (B,N,C)."""
    anchors, offsets, fms = anchors_offsets_featmaps(height, width)
    all_anchors = np.concatenate([a.reshape(-1, 2) for a in anchors], 0)  # (9,2)
    num_anchors = np.cumsum([a.size // 2 for a in anchors])
    num_offsets = np.cumsum([o.size // 2 for o in offsets])
    _offsets = [0] + num_offsets.tolist()
    B, M = gt_boxes.shape[:2]
    ncell = int(num_offsets[-1])
    center_targets = np.zeros((B, ncell, 9, 2), np.float32)
    scale_targets = np.zeros_like(center_targets)
    weights = np.zeros_like(center_targets)
    objectness = np.zeros((B, ncell, 9, 1), np.float32)
    class_targets = np.full((B, ncell, 9, num_class), -1, np.float32)
    gt_boxes = np.asarray(gt_boxes, np.float32)
    gtw = gt_boxes[..., 2] - gt_boxes[..., 0]        # BBoxCornerToCenter, fp32 NDArray arithmetic
    gth = gt_boxes[..., 3] - gt_boxes[..., 1]
    gtx = gt_boxes[..., 0] + gtw / np.float32(2)
    gty = gt_boxes[..., 1] + gth / np.float32(2)
    valid = (gt_boxes >= 0).prod(axis=-1)
    for b in range(B):
        ious = _iou_centered(all_anchors, np.stack([gtw[b], gth[b]], -1))  # (9,M)
        matches = ious.argmax(axis=0)
        for m in range(M):
            if valid[b, m] < 1:
                break
            match = int(matches[m])
            nlayer = int(np.nonzero(num_anchors > match)[0][0])
            h, w = fms[nlayer]
            x, y, bw, bh = gtx[b, m], gty[b, m], gtw[b, m], gth[b, m]
            fx = float(x) / width * w      # float64 (see the module docstring)
            fy = float(y) / height * h
            loc_x = int(fx)
            loc_y = int(fy)
            index = _offsets[nlayer] + loc_y * w + loc_x
            center_targets[b, index, match, 0] = fx - loc_x
            center_targets[b, index, match, 1] = fy - loc_y
            aw, ah = all_anchors[match, 0], all_anchors[match, 1]
            scale_targets[b, index, match, 0] = np.log(bw / aw) if not 1 > bw else np.log(1 / float(aw))
            scale_targets[b, index, match, 1] = np.log(bh / ah) if not 1 > bh else np.log(1 / float(ah))
            weights[b, index, match, :] = 2.0 - float(bw * bh) / width / height
            objectness[b, index, match, 0] = gt_mixratio[b, m, 0] if gt_mixratio is not None else 1
            class_targets[b, index, match, :] = 0
            class_targets[b, index, match, int(gt_ids[b, m, 0])] = 1

    def _slice(x):  # yolo_target.py:139-148
        an = [0] + num_anchors.tolist()
        of = [0] + num_offsets.tolist()
        ret = []
        for i in range(3):
            y = x[:, of[i]:of[i + 1], an[i]:an[i + 1], :]
            ret.append(y.reshape(B, -1, y.shape[-1]))
        return np.concatenate(ret, 1)
    return tuple(_slice(t) for t in (objectness, center_targets, scale_targets, weights, class_targets))


def synthetic_gt(batch, size, num_class, m=8, seed=0, pad_to=None):
    """M gt boxes / image, uniform centres, w,h ~ U(32,256) clipped to the image
    (scaled down for small test images), class ~ U{0..C-1}; padded with -1 rows to pad_to."""
    rng = np.random.default_rng(seed)
    pad_to = pad_to or m
    boxes = np.full((batch, pad_to, 4), -1, np.float32)
    ids = np.full((batch, pad_to, 1), -1, np.float32)
    lo, hi = (32, 256) if size >= 320 else (size / 8, size / 1.5)
    for b in range(batch):
        n = m if pad_to == m else int(rng.integers(1, m + 1))
        c = rng.uniform(0, size, (n, 2))
        wh = rng.uniform(lo, hi, (n, 2))
        x1y1 = np.clip(c - wh / 2, 0, size - 2)
        x2y2 = np.clip(c + wh / 2, x1y1 + 1, size - 1)
        boxes[b, :n] = np.concatenate([x1y1, x2y2], 1)
        ids[b, :n, 0] = rng.integers(0, num_class, n)
    return boxes, ids
