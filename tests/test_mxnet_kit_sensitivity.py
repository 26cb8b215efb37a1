"""-m "not gpu": is the operator-level golden kit (tests/golden/mxnet_ops_kit.py) SENSITIVE to what it claims to decide?

The kit's cases are hand-built so that ONE run on a machine with mxnet decides every recalled semantic choice of
`contrib.box_nms`.  That only works if each alternative semantics actually changes the output of the case that names it.
This file holds a plain-Python box_nms with every such choice as a switch:

    strict_valid   score >  valid_thresh   (alternative: >=)
    strict_iou     iou   >  overlap_thresh (alternative: >=)
    tie_ascending  equal scores keep ascending input row order (alternative: descending row order)
    topk_first     the top-k cut is applied before suppression (alternative: after, survivors refill the cut)
    drop_background  rows with id -1 are removed before anything else (alternative recalled default: kept, id -1 is a class)
    plus_one       IoU computed with the +1 pixel convention (alternative to the corner format without it)

and checks (1) with the recalled defaults it reproduces the C oracle on EVERY box_nms / detect_heads case of the kit — an
independent restatement, sharing no code with oracle/ref_ops.c —, and (2) flipping one switch changes the output of exactly
the cases whose `decides` text claims that choice, and of no case that claims another.  So a red case in
tests/test_mxnet_ops.py on real goldens points at one switch.
"""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("mxnet_ops_kit", os.path.join(HERE, "golden", "mxnet_ops_kit.py"))
KIT = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(KIT)

DEFAULTS = dict(strict_valid=True, strict_iou=True, tie_ascending=True, topk_first=True, drop_background=False, plus_one=False)


def _iou(a, b, plus_one):
    one = np.float32(1.0 if plus_one else 0.0)
    iw = np.float32(min(a[2], b[2]) - max(a[0], b[0]) + one)
    ih = np.float32(min(a[3], b[3]) - max(a[1], b[1]) + one)
    if not (iw > 0 and ih > 0):
        return np.float32(0)
    inter = np.float32(iw * ih)
    aa = np.float32((a[2] - a[0] + one) * (a[3] - a[1] + one))
    ab = np.float32((b[2] - b[0] + one) * (b[3] - b[1] + one))
    uni = np.float32(np.float32(aa + ab) - inter)
    return np.float32(0) if uni <= 0 else np.float32(inter / uni)


def py_box_nms(data, overlap_thresh, valid_thresh, topk, force_suppress, strict_valid=True, strict_iou=True, tie_ascending=True,
               topk_first=True, drop_background=False, plus_one=False):
    data = np.asarray(data, np.float32)
    out = np.full_like(data, -1.0)
    vt, ot = np.float32(valid_thresh), np.float32(overlap_thresh)
    for b in range(data.shape[0]):
        rows = []
        for i, r in enumerate(data[b]):
            ok = (r[1] > vt) if strict_valid else (r[1] >= vt)
            if ok and not (drop_background and r[0] < 0):
                rows.append(i)
        rows.sort(key=lambda i: (-float(data[b, i, 1]), i if tie_ascending else -i))
        cut = topk if topk > 0 else len(rows)
        cand = rows[:cut] if topk_first else rows
        kept = []
        for i in cand:
            dead = False
            for j in kept:
                if not force_suppress and data[b, i, 0] != data[b, j, 0]:
                    continue
                v = _iou(data[b, j, 2:], data[b, i, 2:], plus_one)
                if (v > ot) if strict_iou else (v >= ot):
                    dead = True
                    break
            if not dead:
                kept.append(i)
        if not topk_first:
            kept = kept[:cut]
        for k, i in enumerate(kept):
            out[b, k] = data[b, i]
    return out


def _nms_inputs(case, oracle):
    """(rows, overlap, valid, topk, force) of a box_nms case, or of a detect_heads case (rows = the oracle's decode)."""
    p = case["params"]
    if case["op"] == "box_nms":
        return case["inputs"]["data"], p["overlap_thresh"], p["valid_thresh"], p["topk"], bool(p["force_suppress"])
    rows = oracle.run("detect_heads", case["inputs"], p)["rows"]
    return rows, p["nms_thresh"], 0.01, p["nms_topk"], False


NMS_CASES = [c for c in KIT.all_cases() if c["op"] in ("box_nms", "detect_heads")]


@pytest.mark.parametrize("case", NMS_CASES, ids=[c["name"] for c in NMS_CASES])
def test_plain_python_nms_with_the_recalled_semantics_equals_the_oracle(case):
    oracle = KIT.OracleOps()
    rows, ot, vt, topk, force = _nms_inputs(case, oracle)
    want = oracle.run(case["op"], case["inputs"], case["params"])["out"]
    got = py_box_nms(rows, ot, vt, topk, force, **DEFAULTS)
    assert np.array_equal(got, want), (case["name"], got[..., :2], want[..., :2])


# switch -> the kit cases that must change when it is flipped (and only cases that claim that decision may change)
EXPECT = {
    "strict_valid": {"nms_score_equals_valid_thresh"},
    "strict_iou": {"nms_iou_equals_overlap_thresh", "heads_iou_at_thresh"},
    "tie_ascending": {"nms_duplicate_scores_ab", "nms_duplicate_scores_ba", "nms_topk_cuts_through_tie", "heads_duplicate_scores",
                      "heads_topk_cuts_through_tie", "heads_same_box_three_classes", "heads_tie_across_scales"},
    "topk_first": {"nms_topk_before_suppression"},
    "drop_background": {"nms_background_id_rows"},
}
# cases allowed to change as a side effect of a switch they do not name (none today; listed so that an addition is deliberate)
ALSO_MAY = {"tie_ascending": {"nms_degenerate_boxes"}}   # two identical zero-area rows tie on nothing but order of equal boxes


@pytest.mark.parametrize("switch", sorted(EXPECT))
def test_every_recalled_choice_changes_the_case_that_names_it(switch):
    oracle = KIT.OracleOps()
    changed = set()
    for case in NMS_CASES:
        rows, ot, vt, topk, force = _nms_inputs(case, oracle)
        base = py_box_nms(rows, ot, vt, topk, force, **DEFAULTS)
        alt = py_box_nms(rows, ot, vt, topk, force, **dict(DEFAULTS, **{switch: not DEFAULTS[switch]}))
        if not np.array_equal(base, alt):
            changed.add(case["name"])
    missing = EXPECT[switch] - changed
    assert not missing, "flipping %s does not change %s: the kit would not notice that choice" % (switch, sorted(missing))
    extra = changed - EXPECT[switch] - ALSO_MAY.get(switch, set())
    assert not extra, "flipping %s also changes %s: a red run would not point at one choice" % (switch, sorted(extra))


def test_plus_one_iou_would_show_in_the_iou_cases():
    """The +1 pixel convention (gluoncv's numpy bbox_iou has an `offset` argument; box_nms is recalled to use none): it moves
    the IoU of the kit's dyadic pairs off their thresholds and the published docstring value off 0.1428."""
    ka = KIT.KNOWN_ANSWERS["box_iou_doc"]
    a, b = ka["lhs"][0], ka["rhs"][0]
    assert abs(float(_iou(a, b, False)) - 1.0 / 7.0) < 1e-7 and abs(float(_iou(a, b, True)) - 1.0 / 7.0) > 0.3
    case = next(c for c in NMS_CASES if c["name"] == "nms_iou_equals_overlap_thresh")
    rows, ot, vt, topk, force = _nms_inputs(case, KIT.OracleOps())
    assert not np.array_equal(py_box_nms(rows, ot, vt, topk, force, **DEFAULTS),
                              py_box_nms(rows, ot, vt, topk, force, **dict(DEFAULTS, plus_one=True)))


def test_running_variance_convention_would_show_in_the_cell_case():
    """`conv_bn_leaky_train_step` decides biased vs unbiased running variance: with 2 x 4 x 4 = 32 samples per channel the
    two conventions differ by var * 0.1 / 31 ~ 3e-3 of the running value — 300x the case's tolerance (1e-5)."""
    case = next(c for c in KIT.all_cases() if c["op"] == "conv_bn_leaky")
    ops = KIT.OracleOps()
    base = ops.run(case["op"], case["inputs"], case["params"])["running_var"]
    ops.TO.OracleYolo3Train.RUNNING_VAR_UNBIASED = True
    try:
        alt = ops.run(case["op"], case["inputs"], case["params"])["running_var"]
    finally:
        ops.TO.OracleYolo3Train.RUNNING_VAR_UNBIASED = False
    assert np.abs(alt - base).min() > 1e-4, (base, alt)


def test_loss_normalisation_would_show_in_the_loss_case():
    """`yolov3_loss_pos_ignore_neg`: a per-element MEAN instead of the recalled mean x count (= per-sample sum) is off by the
    anchor count (4) / channel count, far outside 1e-4."""
    case = next(c for c in KIT.all_cases() if c["op"] == "yolov3_loss")
    out = KIT.OracleOps().run(case["op"], case["inputs"], case["params"])
    n = case["inputs"]["objness"].shape[1]
    for k, v in out.items():
        assert np.all(np.abs(v - v / n) > 1e-3 * np.maximum(np.abs(v), 1e-3)) or np.all(v == 0), (k, v)
