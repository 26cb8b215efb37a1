"""Generates tests/golden/voc_metric_golden.json by running THE REFERENCE's own metric classes
(/root/reference/metrics/pascalvoc.py: VOCMApMetric, VOC07MApMetric) on seeded random inputs.

Runs only in the build container (needs /root/reference).  mxnet and gluoncv are not installable, so
two throw-away stand-ins are injected before the import: `mxnet` (only `mx.nd.NDArray` for isinstance
checks and `mx.metric.EvalMetric` as a base class are touched) and `gluoncv.utils.bbox.bbox_iou`,
redirected to the reference's in-tree copy utils/bbox.py:11-40 (whose header says it was copied from
gluoncv).  The numeric body that produces the golden values is the reference's, unmodified.

    python tests/golden/make_voc_metric_golden.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    mx = types.ModuleType("mxnet")
    mx.nd = types.ModuleType("mxnet.nd")
    mx.nd.NDArray = type("NDArray", (), {})
    mx.metric = types.ModuleType("mxnet.metric")

    class EvalMetric(object):
        def __init__(self, name, **kwargs):
            self.name = name
    mx.metric.EvalMetric = EvalMetric
    sys.modules.update({"mxnet": mx, "mxnet.nd": mx.nd, "mxnet.metric": mx.metric})
    sys.path.insert(0, REF)
    from utils import bbox as ref_bbox
    g = types.ModuleType("gluoncv")
    g.utils = types.ModuleType("gluoncv.utils")
    g.utils.bbox = types.ModuleType("gluoncv.utils.bbox")
    g.utils.bbox.bbox_iou = ref_bbox.bbox_iou
    sys.modules.update({"gluoncv": g, "gluoncv.utils": g.utils, "gluoncv.utils.bbox": g.utils.bbox})


def make_case(rng, batch, n_pred, n_gt, n_cls, pad_frac=0.2, difficult=True, jitter=6.0):
    """Detections are jittered copies of ground truths plus clutter, -1 padded like the detector's."""
    gt_b = np.full((batch, n_gt, 4), -1.0, np.float32)
    gt_l = np.full((batch, n_gt, 1), -1.0, np.float32)
    gt_d = np.zeros((batch, n_gt, 1), np.float32)
    pr_b = np.full((batch, n_pred, 4), -1.0, np.float32)
    pr_l = np.full((batch, n_pred, 1), -1.0, np.float32)
    pr_s = np.full((batch, n_pred, 1), -1.0, np.float32)
    for b in range(batch):
        m = int(rng.integers(1, n_gt + 1))
        xy = rng.uniform(0, 300, (m, 2))
        wh = rng.uniform(20, 120, (m, 2))
        gt_b[b, :m] = np.concatenate([xy, xy + wh], 1)
        gt_l[b, :m, 0] = rng.integers(0, n_cls, m)
        if difficult:
            gt_d[b, :m, 0] = rng.random(m) < 0.2
        k = int(n_pred * (1 - pad_frac))
        src = rng.integers(0, m, k)
        noise = rng.normal(0, jitter, (k, 4))
        clutter = rng.random(k) < 0.3
        boxes = gt_b[b, src] + noise
        rb = rng.uniform(0, 300, (k, 2))
        boxes[clutter] = np.concatenate([rb, rb + rng.uniform(20, 120, (k, 2))], 1)[clutter]
        labels = gt_l[b, src, 0].copy()
        flip = rng.random(k) < 0.15
        labels[flip] = rng.integers(0, n_cls, int(flip.sum()))
        pr_b[b, :k], pr_l[b, :k, 0], pr_s[b, :k, 0] = boxes, labels, rng.random(k)
    return dict(pred_bboxes=pr_b, pred_labels=pr_l, pred_scores=pr_s, gt_bboxes=gt_b, gt_labels=gt_l,
                gt_difficults=gt_d)


def main():
    _install_stubs()
    from metrics.pascalvoc import VOCMApMetric, VOC07MApMetric
    rng = np.random.default_rng(20260233)
    cases = []
    specs = [dict(batch=4, n_pred=100, n_gt=8, n_cls=20), dict(batch=2, n_pred=30, n_gt=5, n_cls=3),
             dict(batch=3, n_pred=100, n_gt=12, n_cls=30, difficult=False), dict(batch=1, n_pred=10, n_gt=2, n_cls=5),
             dict(batch=6, n_pred=50, n_gt=6, n_cls=4, jitter=25.0)]
    for si, spec in enumerate(specs):
        names = ["class%d" % i for i in range(spec["n_cls"])]
        for use_names in (True, False):
            for iou in (0.5, 0.75):
                updates = [make_case(rng, **spec) for _ in range(2)]  # two update() calls accumulate
                out = {}
                for cls_name, cls in (("VOCMApMetric", VOCMApMetric), ("VOC07MApMetric", VOC07MApMetric)):
                    m = cls(iou_thresh=iou, class_names=names if use_names else None)
                    for u in updates:
                        m.update(u["pred_bboxes"], u["pred_labels"], u["pred_scores"], u["gt_bboxes"],
                                 u["gt_labels"], u["gt_difficults"] if spec.get("difficult", True) else None)
                    name, value = m.get()
                    out[cls_name] = dict(name=name, value=value if isinstance(value, list) else float(value))
                cases.append(dict(spec=spec, class_names=names if use_names else None, iou_thresh=iou,
                                  updates=[{k: v.tolist() for k, v in u.items()} for u in updates], expected=out))
    with open(os.path.join(HERE, "voc_metric_golden.json"), "w") as f:
        json.dump(cases, f)
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()
