"""PASCAL-VOC detection mAP for the detector's outputs.

Host-side mirror (numpy, like the reference) of ``VOCMApMetric`` — the area-under-curve AP both
drivers instantiate (train_yolov3.py:181, detect_yolo3.py:183) — and ``VOC07MApMetric`` (11-point AP),
metrics/pascalvoc.py:12-259,523-560 in /root/reference.  Same ``update(pred_bboxes, pred_labels,
pred_scores, gt_bboxes, gt_labels, gt_difficults)`` / ``get()`` / ``reset()`` contract and the same
conventions: label < 0 rows are padding, difficult ground truths are ignored (neither TP nor FP), a
ground truth can be matched once, IoU without the +1 pixel offset, classes with no ground truth give
NaN.  Unlike the hot path this row IS pinned by the reference itself: tests/golden/voc_metric_*.json
were produced by running the reference's own class (mxnet stubbed out) — see
tests/golden/make_voc_metric_golden.py.

Accepts numpy arrays, torch tensors (any device) or lists of them (one per device, concatenated
along the batch axis like utils/general.py:6-17 ``as_numpy``).
"""
import numpy as np


def _to_numpy(a):
    if isinstance(a, (list, tuple)):
        parts = [_to_numpy(x) for x in a]
        try:
            return np.concatenate(parts, axis=0)
        except ValueError:
            return np.array(parts)
    if hasattr(a, "detach"):  # torch tensor
        return a.detach().cpu().numpy()
    return np.asarray(a)


def pairwise_iou(a, b):
    """IoU of every box of a (N,4) with every box of b (M,4), corner format, no +1 offset."""
    lo = np.maximum(a[:, None, :2], b[None, :, :2])
    hi = np.minimum(a[:, None, 2:4], b[None, :, 2:4])
    inter = np.prod(hi - lo, axis=2) * (lo < hi).all(axis=2)
    area_a = np.prod(a[:, 2:4] - a[:, :2], axis=1)
    area_b = np.prod(b[:, 2:4] - b[:, :2], axis=1)
    return inter / (area_a[:, None] + area_b[None, :] - inter)


class VOCMApMetric(object):
    """Mean average precision, area under the monotone precision envelope (VOC 2010+ style)."""

    def __init__(self, iou_thresh=0.5, class_names=None, class_map=None):
        self.iou_thresh = iou_thresh
        self.class_names = list(class_names) if class_names is not None else None
        self.class_map = class_map
        self.name = 'VOCMeanAP' if class_names is None else self.class_names + ['mAP']
        self.reset()

    def reset(self):
        self._n_pos = {}     # class -> number of non-difficult ground truths
        self._scores = {}    # class -> list of detection scores
        self._flags = {}     # class -> list of +1 (TP) / 0 (FP) / -1 (ignored: matched a difficult gt)

    # ------------------------------------------------------------------ accumulation
    def update(self, pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, gt_difficults=None):
        pb, pl, ps, gb, gl = [_to_numpy(x) for x in (pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels)]
        gd = _to_numpy(gt_difficults) if gt_difficults is not None else None
        for i in range(len(pb)):
            self._update_image(pb[i], pl[i], ps[i], gb[i], gl[i], None if gd is None else gd[i])

    def _update_image(self, boxes, labels, scores, gboxes, glabels, gdiff):
        labels = np.asarray(labels).reshape(-1)
        keep = np.flatnonzero(labels >= 0)
        boxes, scores, labels = boxes[keep], np.asarray(scores).reshape(-1)[keep], labels[keep].astype(int)
        glabels = np.asarray(glabels).reshape(-1)
        if self.class_map is not None:
            glabels = np.array([self.class_map[int(g)] for g in glabels])
        gkeep = np.flatnonzero(glabels >= 0)
        gboxes, glabels = gboxes[gkeep], glabels[gkeep].astype(int)
        gdiff = np.zeros(len(gkeep)) if gdiff is None else np.asarray(gdiff).reshape(-1)[gkeep]
        for c in np.unique(np.concatenate([labels, glabels]).astype(int)):
            c = int(c)
            sel = labels == c
            order = scores[sel].argsort()[::-1]
            cb, cs = boxes[sel][order], scores[sel][order]
            gsel = glabels == c
            cg, cd = gboxes[gsel], gdiff[gsel]
            self._n_pos[c] = self._n_pos.get(c, 0) + int(np.logical_not(cd).sum())
            self._scores.setdefault(c, []).extend(cs)
            flags = self._flags.setdefault(c, [])
            if len(cb) == 0:
                continue
            if len(cg) == 0:
                flags.extend([0] * len(cb))
                continue
            iou = pairwise_iou(cb, cg)
            best = iou.argmax(axis=1)
            best[iou.max(axis=1) < self.iou_thresh] = -1
            taken = np.zeros(len(cg), dtype=bool)
            for g in best:  # detections in descending score order claim ground truths greedily
                if g < 0:
                    flags.append(0)
                elif cd[g]:
                    flags.append(-1)
                    taken[g] = True
                else:
                    flags.append(0 if taken[g] else 1)
                    taken[g] = True

    # ------------------------------------------------------------------ evaluation
    def _curves(self):
        n_cls = max(self._n_pos) + 1 if self._n_pos else 0
        rec, prec = [None] * n_cls, [None] * n_cls
        for c in self._n_pos:
            s = np.array(self._scores[c])
            f = np.array(self._flags[c], dtype=np.int32)[s.argsort()[::-1]]
            tp, fp = np.cumsum(f == 1), np.cumsum(f == 0)
            with np.errstate(divide='ignore', invalid='ignore'):
                prec[c] = tp / (fp + tp)
            if self._n_pos[c] > 0:
                rec[c] = tp / self._n_pos[c]
        return rec, prec

    def _average_precision(self, rec, prec):
        if rec is None or prec is None:
            return np.nan
        r = np.concatenate([[0.0], rec, [1.0]])
        p = np.concatenate([[0.0], np.nan_to_num(prec), [0.0]])
        p = np.maximum.accumulate(p[::-1])[::-1]          # monotone envelope from the right
        step = np.flatnonzero(r[1:] != r[:-1])
        return float(np.sum((r[step + 1] - r[step]) * p[step + 1]))

    def get(self):
        rec, prec = self._curves()
        aps = [self._average_precision(r, p) for r, p in zip(rec, prec)]
        with np.errstate(all='ignore'):
            mean_ap = float(np.nanmean(aps)) if len(aps) else float('nan')
        if self.class_names is None:
            return self.name, mean_ap
        n = len(self.class_names)
        per_class = [aps[c] if c < len(aps) and c in self._n_pos else float('nan') for c in range(n)]
        if self.class_map:
            per_class = [float('nan') if self.class_map[c] < 0 else
                         (aps[self.class_map[c]] if self.class_map[c] < len(aps) and self.class_map[c] in self._n_pos
                          else float('nan')) for c in range(n)]
        return list(self.name), per_class + [mean_ap]


class VOC07MApMetric(VOCMApMetric):
    """Mean AP with the VOC2007 11-point interpolation (max precision at recall >= 0, 0.1, ..., 1)."""

    def _average_precision(self, rec, prec):
        if rec is None or prec is None:
            return np.nan
        p = np.nan_to_num(prec)
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            hit = rec >= t
            ap += (np.max(p[hit]) if hit.any() else 0.0) / 11.0
        return float(ap)
