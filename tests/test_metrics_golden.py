"""-m "not gpu": videoyolo_amd.metrics against golden vectors produced by THE REFERENCE's own
VOCMApMetric / VOC07MApMetric (tests/golden/make_voc_metric_golden.py runs /root/reference's
metrics/pascalvoc.py with mxnet stubbed out).  This row of SURVEY §8f is pinned by the reference."""
import json
import math
import os

import numpy as np
import pytest

from videoyolo_amd import metrics

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "voc_metric_golden.json")))


def _same(a, b):
    if isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b):
        return True
    return a == pytest.approx(b, abs=1e-12)


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_voc_metrics_match_reference_golden(idx):
    case = CASES[idx]
    for cls_name in ("VOCMApMetric", "VOC07MApMetric"):
        m = getattr(metrics, cls_name)(iou_thresh=case["iou_thresh"], class_names=case["class_names"])
        for u in case["updates"]:
            arrs = {k: np.array(v, np.float32) for k, v in u.items()}
            diff = arrs["gt_difficults"] if case["spec"].get("difficult", True) else None
            m.update(arrs["pred_bboxes"], arrs["pred_labels"], arrs["pred_scores"], arrs["gt_bboxes"],
                     arrs["gt_labels"], diff)
        name, value = m.get()
        exp = case["expected"][cls_name]
        assert name == exp["name"]
        if isinstance(exp["value"], list):
            assert len(value) == len(exp["value"])
            for a, b in zip(value, exp["value"]):
                assert _same(float(a), float(b)), (cls_name, value, exp["value"])
        else:
            assert _same(float(value), float(exp["value"]))


def test_reset_and_list_inputs():
    case = CASES[0]
    u = {k: np.array(v, np.float32) for k, v in case["updates"][0].items()}
    m = metrics.VOCMApMetric(iou_thresh=0.5, class_names=case["class_names"])
    # per-device lists are concatenated along the batch axis (utils/general.py:6-17)
    halves = {k: [v[:2], v[2:]] for k, v in u.items()}
    m.update(halves["pred_bboxes"], halves["pred_labels"], halves["pred_scores"], halves["gt_bboxes"],
             halves["gt_labels"], halves["gt_difficults"])
    a = m.get()
    m.reset()
    m.update(u["pred_bboxes"], u["pred_labels"], u["pred_scores"], u["gt_bboxes"], u["gt_labels"],
             u["gt_difficults"])
    b = m.get()
    assert a[0] == b[0]
    assert all(_same(float(x), float(y)) for x, y in zip(a[1], b[1]))
