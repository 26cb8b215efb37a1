"""Operator-level golden kit: one tiny case per semantic choice this repo had to RECALL about mxnet / gluoncv
(tagged [UPSTREAM-RECALLED] in oracle/ and SURVEY.md section 8), so that ONE run on a machine with mxnet decides every one
of them and a red test names the operator — not just "ids differ".

Used by
  tests/golden/make_mxnet_goldens.py   `capture_ops(backend)` runs every case through the backend's operators
                                       (MxnetBackend: the real ones; OracleBackend: this repo's oracle — plumbing only)
                                       and writes `mxnet_ops.npz`
  tests/test_mxnet_ops.py              replays the stored inputs through the oracle (CPU) and, where the C-ABI has a door
                                       at that level, through the HIP path (GPU), against the stored outputs
  tests/test_oracle_known_answers.py   the two PUBLISHED docstring examples of the dependency (`KNOWN_ANSWERS` below) run
                                       against the oracle today, without any fixture

A case = dict(name, op, decides, inputs {name: array}, params {name: scalar}).  `decides` names the recalled choice the
case settles and where it sits in the reference.  Inputs are hand-built: dyadic coordinates and exactly representable
scores wherever a comparison sits ON its threshold, so that no rounding of either implementation can move it.

Operators (the reference call site each one stands for):
  box_nms           F.contrib.box_nms                       models/definitions/yolo/yolo3.py:1198-1200
  detect_heads      net.yolo_outputs[i](pred) + concat + box_nms: the same call fed from hand-built LOGITS, which is the form the
                    HIP path can run too (vy_net_detect_heads)      yolo3.py:158-197,1195-1206
  box_iou           nd.contrib.box_iou                      models/definitions/yolo/yolo_target.py:92
  bbox_batch_iou    gluoncv.nn.bbox.BBoxBatchIOU            yolo_target.py:171,202
  dynamic_targets   YOLOV3DynamicTargetGeneratorSimple      yolo_target.py:173-205 (reference code over mxnet ops)
  target_merger     YOLOV3TargetMerger                      yolo_target.py:226-281
  prefetch_targets  YOLOV3PrefetchTargetGenerator           yolo_target.py:13-148, transforms.py:185-197
  yolov3_loss       gluoncv.loss.YOLOV3Loss                 yolo3.py:994,1187
  conv_bn_leaky     layers._conv2d cell, one recorded step  models/definitions/layers.py:63-70
  sgd               gluon.Trainer('sgd').step               train_yolov3.py:527-530,634
  imresize          gluoncv timage.imresize(interp=9)       models/definitions/yolo/transforms.py:325-327
"""
import numpy as np

F32 = np.float32

# ------------------------------------------------------------------------------------ published known answers
# Recalled from the operator documentation of mxnet.ndarray.contrib (the docstring examples of box_nms and box_iou).
# They are the only numbers in this repository that come from the dependency's own published text; everything else
# about the operators is recalled behaviour.  tests/test_oracle_known_answers.py runs the oracle against them.
KNOWN_ANSWERS = {
    "box_nms_doc": dict(
        data=np.array([[[0, 0.5, 0.1, 0.1, 0.2, 0.2], [1, 0.4, 0.1, 0.1, 0.2, 0.2],
                        [0, 0.3, 0.1, 0.1, 0.14, 0.14], [2, 0.6, 0.5, 0.5, 0.7, 0.8]]], F32),
        params=dict(overlap_thresh=0.1, valid_thresh=0.0, topk=-1, force_suppress=True),
        out=np.array([[[2, 0.6, 0.5, 0.5, 0.7, 0.8], [0, 0.5, 0.1, 0.1, 0.2, 0.2],
                       [-1, -1, -1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]]], F32)),
    "box_iou_doc": dict(
        lhs=np.array([[0.5, 0.5, 1.0, 1.0], [0.0, 0.0, 0.5, 0.5]], F32),
        rhs=np.array([[0.25, 0.25, 0.75, 0.75]], F32),
        out=np.array([[0.1428], [0.1428]], F32), atol=1e-4),   # the docstring prints four digits (1/7 = 0.142857 cut, not rounded)
}


def _rows(*rows):
    return np.array([list(rows)], F32)   # (1, n, 6)


def _nms_case(name, decides, data, overlap_thresh=0.5, valid_thresh=0.0, topk=-1, force_suppress=False):
    return dict(name=name, op="box_nms", decides=decides, inputs=dict(data=np.asarray(data, F32)),
                params=dict(overlap_thresh=float(overlap_thresh), valid_thresh=float(valid_thresh), topk=int(topk),
                            force_suppress=int(bool(force_suppress))))


def box_nms_cases():
    up, dn = np.nextafter(F32(0.25), F32(1)), np.nextafter(F32(0.25), F32(0))
    far = lambda i: [10.0 * i, 0, 10.0 * i + 4, 4]   # noqa: E731  (disjoint 4 x 4 boxes)
    out = []
    ka = KNOWN_ANSWERS["box_nms_doc"]
    out.append(_nms_case("nms_doc_example", "the published docstring example (known answer: KNOWN_ANSWERS['box_nms_doc'])",
                         ka["data"], **ka["params"]))
    out.append(_nms_case(
        "nms_score_equals_valid_thresh",
        "strictness of `score > valid_thresh` (oracle/ref_ops.c:196; the path uses valid_thresh=0.01, yolo3.py:1199): scores "
        "0.25 - ulp, 0.25, 0.25 + ulp, 0.5 against valid_thresh 0.25 on disjoint boxes",
        _rows([0, dn] + far(0), [0, 0.25] + far(1), [0, up] + far(2), [0, 0.5] + far(3)), valid_thresh=0.25))
    # IoU exactly 0.5: [0,0,2,2] (area 4) and [0,0,2,1] (area 2): inter 2, union 4.  Above: inter 2.125 / union 4; below: 1.875 / 4
    out.append(_nms_case(
        "nms_iou_equals_overlap_thresh",
        "strictness of `iou > overlap_thresh` (ref_ops.c:213): three same-class pairs with IoU 0.46875, 0.5 (== threshold), "
        "0.53125 — all dyadic, union 4",
        _rows([0, 0.95, 0, 0, 2, 2], [0, 0.90, 0, 0, 2, 1],            # IoU 0.5
              [0, 0.85, 10, 10, 12, 12], [0, 0.80, 10, 10, 12, 11.0625],  # 0.53125
              [0, 0.75, 20, 20, 22, 22], [0, 0.70, 20, 20, 22, 20.9375]),  # 0.46875
        overlap_thresh=0.5))
    for tag, order in (("ab", (0, 1, 2, 3)), ("ba", (1, 0, 3, 2))):
        base = [[0, 0.5] + far(0), [0, 0.5] + far(1),                   # equal scores, disjoint: output ORDER
                [0, 0.75, 50, 50, 54, 54], [0, 0.75, 50, 50, 54, 53]]  # equal scores, IoU 0.75: WHO survives
        out.append(_nms_case(
            "nms_duplicate_scores_" + tag,
            "order of rows with EQUAL scores (ref_ops.c:186-190: stable, ascending input row): a disjoint pair (output order) "
            "and an overlapping pair (which one suppresses the other), input order %s" % tag,
            _rows(*[base[i] for i in order]), overlap_thresh=0.5))
    out.append(_nms_case(
        "nms_topk_cuts_through_tie",
        "which of three rows with the same score pass a top-k cut that falls between them (ref_ops.c:203): scores "
        "0.9, 0.5, 0.5, 0.5, 0.1, topk 3, disjoint boxes",
        _rows([0, 0.5] + far(0), [0, 0.9] + far(1), [0, 0.5] + far(2), [0, 0.1] + far(3), [0, 0.5] + far(4)), topk=3))
    out.append(_nms_case(
        "nms_topk_before_suppression",
        "top-k is applied BEFORE suppression (ref_ops.c:203-216): A 0.9 and B 0.8 overlap, C 0.7 is disjoint, topk 2 -> C is "
        "not promoted into the freed slot",
        _rows([0, 0.9, 0, 0, 4, 4], [0, 0.8, 0, 0, 4, 3], [0, 0.7] + far(5)), topk=2))
    out.append(_nms_case(
        "nms_background_id_rows",
        "rows with id -1 (background_id default; ref_ops.c:176 'none excluded by id' — the path never produces them, ids are "
        "0..C-1, yolo3.py:194): kept? and do they suppress a same-box row of class 0?",
        _rows([-1, 0.9, 0, 0, 4, 4], [0, 0.8, 0, 0, 4, 4], [-1, 0.7] + far(3), [0, 0.6] + far(4))))
    for fs in (0, 1):
        out.append(_nms_case(
            "nms_force_suppress_%d" % fs,
            "per-class suppression (`force_suppress=False` at yolo3.py:1200; ref_ops.c:212): the same box under class 0 and "
            "class 1, force_suppress=%d" % fs,
            _rows([0, 0.9, 0, 0, 4, 4], [1, 0.8, 0, 0, 4, 4], [1, 0.7, 0, 0, 4, 3]), force_suppress=fs))
    out.append(_nms_case(
        "nms_two_images_one_empty",
        "images are independent and an image without a valid row is all -1 (ref_ops.c:219-222)",
        np.concatenate([_rows([0, 0.9, 0, 0, 4, 4], [0, 0.8, 0, 0, 4, 3], [1, 0.7] + far(2)),
                        _rows([0, 0.005] + far(0), [1, 0.0075] + far(1), [2, 0.0] + far(2))]), valid_thresh=0.01))   # (no score ON the
    # threshold here: tests/test_mxnet_kit_sensitivity.py found that a 0.01 in this case made it a second judge of strictness)
    out.append(_nms_case(
        "nms_degenerate_boxes",
        "zero-area and inverted boxes (include/vy_math.h vy_box_iou: no positive overlap -> 0, union <= 0 -> 0): two identical "
        "zero-area rows, an inverted box over a regular one",
        _rows([0, 0.9, 5, 5, 5, 5], [0, 0.8, 5, 5, 5, 5], [0, 0.7, 12, 12, 10, 10], [0, 0.6, 10, 10, 12, 12])))
    return out


# ------------------------------------------------------------------------------------ hand-built logits
HEAD_SIZE, HEAD_CLASSES = 64, 3   # 64 x 64 input: grids 2 x 2, 4 x 4, 8 x 8; N = 3 * 84 = 252 anchors, 756 rows


def blank_heads(batch=1, size=HEAD_SIZE, classes=HEAD_CLASSES):
    """Three prediction tensors (B, 3 * (5 + C), S/32 | S/16 | S/8, .) that decode to NO valid row: objectness logit -40
    (sigmoid 4e-18), everything else 0 (sigmoid(0) = 0.5 and exp(0) = 1 exactly: boxes sit on cell centres with their
    anchor's size — integer / half-integer pixels)."""
    heads = []
    for div in (32, 16, 8):
        h = np.zeros((batch, 3 * (5 + classes), size // div, size // div), F32)
        for a in range(3):
            h[:, a * (5 + classes) + 4] = -40.0
        heads.append(h)
    return heads


def put(heads, scale, y, x, anchor, cls_logits, classes=HEAD_CLASSES, b=0):
    """Make (scale, cell (y, x), anchor) a candidate: objectness logit +40 (sigmoid == 1.0f), class logits as given
    (missing classes -40), so that score(class c) = sigmoid(logit c) exactly."""
    p0 = anchor * (5 + classes)
    heads[scale][b, p0 + 4, y, x] = 40.0
    for c in range(classes):
        heads[scale][b, p0 + 5 + c, y, x] = cls_logits.get(c, -40.0)


def _heads_case(name, decides, heads, nms_thresh=0.45, nms_topk=400):
    return dict(name=name, op="detect_heads", decides=decides,
                inputs=dict(head0=heads[0], head1=heads[1], head2=heads[2]),
                params=dict(nms_thresh=float(nms_thresh), nms_topk=int(nms_topk), size=HEAD_SIZE, classes=HEAD_CLASSES))


def detect_heads_cases():
    out = []
    # stride 8, anchor 1 = 16 x 30: neighbours in x are 8 px apart: inter 8 * 30 = 240, union 480 + 480 - 240 = 720
    third = float(F32(240.0) / F32(720.0))
    for tag, thr in (("at", third), ("below", float(np.nextafter(F32(third), F32(0))))):
        h = blank_heads()
        put(h, 2, 3, 2, 1, {0: 2.0})
        put(h, 2, 3, 3, 1, {0: 1.0})
        out.append(_heads_case(
            "heads_iou_%s_thresh" % tag,
            "`iou > overlap_thresh` through the decode: two 16 x 30 boxes 8 px apart, IoU = 240 / 720 (integers: every fp32 "
            "formula gives the same quotient), nms_thresh %s that value" % ("==" if tag == "at" else "one ulp below"),
            h, nms_thresh=thr))
    h = blank_heads()
    put(h, 2, 1, 1, 0, {1: 0.5})      # disjoint pair, equal scores (10 x 13 boxes, 24 px apart)
    put(h, 2, 1, 4, 0, {1: 0.5})
    put(h, 2, 6, 2, 1, {0: 1.5})      # overlapping pair (IoU 1/3 > 0.3), equal scores: the earlier row must win
    put(h, 2, 6, 3, 1, {0: 1.5})
    out.append(_heads_case(
        "heads_duplicate_scores",
        "tie order through the decode (csrc/detect.hip: key = (score, -row)): equal logits in two cells -> bit-equal scores; "
        "a disjoint pair (output order) and an overlapping pair (survivor)", h, nms_thresh=0.3))
    h = blank_heads()
    put(h, 2, 0, 0, 0, {2: 3.0})
    for x in (2, 4, 6):
        put(h, 2, 4, x, 0, {2: 0.25})   # three equal scores, disjoint
    out.append(_heads_case(
        "heads_topk_cuts_through_tie",
        "nms_topk = 3 with scores s1 > s2 = s2 = s2: which two of the tied rows pass (yolo3.py:1199 topk=self.nms_topk; the "
        "radix select's index passes, detect.hip refine_kernel)", h, nms_topk=3))
    h = blank_heads()
    put(h, 1, 2, 1, 2, {0: 2.0, 1: 1.0, 2: 2.0})
    out.append(_heads_case(
        "heads_same_box_three_classes",
        "per-class suppression and the class-major row order (yolo3.py:191-197): ONE anchor scoring for three classes -> three "
        "identical boxes with ids 0, 1, 2, all kept; classes 0 and 2 tie on score", h))
    h = blank_heads()
    put(h, 0, 1, 1, 0, {1: 0.75})
    put(h, 1, 0, 3, 1, {1: 0.75})
    put(h, 2, 7, 0, 2, {1: 0.75})
    out.append(_heads_case(
        "heads_tie_across_scales",
        "concat order of the scales (yolo3.py:1195: stride 32, 16, 8): the same score at one anchor of each scale", h))
    return out


# ------------------------------------------------------------------------------------ IoU operators
IOU_LHS = np.array([[0.5, 0.5, 1.0, 1.0], [0.0, 0.0, 0.5, 0.5],      # the docstring example's lhs
                    [0, 0, 2, 2], [0, 0, 2, 2], [5, 5, 5, 5], [12, 12, 10, 10], [0, 0, 2, 2], [-1, -1, -1, -1]], F32)
IOU_RHS = np.array([[0.25, 0.25, 0.75, 0.75],                         # ... and its rhs
                    [0, 0, 2, 2],          # identical to lhs[2]: 1.0 (with BBoxBatchIOU's eps 1e-15 still 1.0 in fp32?)
                    [0, 0, 2, 1],          # IoU 0.5 with lhs[2]
                    [5, 5, 5, 5],          # zero area, identical to lhs[4]: 0 / 0
                    [10, 10, 12, 12],      # regular box under the inverted lhs[5]
                    [2, 0, 4, 2],          # touches lhs[2] on an edge: intersection width 0
                    [-1, -1, -1, -1]], F32)  # the batchify pad row (train_yolov3.py:254)


def iou_cases():
    return [
        dict(name="box_iou_pairs", op="box_iou",
             decides="nd.contrib.box_iou(format='corner') on the docstring pair (known answer 0.1428), identical, IoU 0.5, "
                     "zero-area, inverted, edge-touching and pad (-1) boxes: no +1, 0 where the union is not positive "
                     "(include/vy_math.h:124-136; used by the prefetch generator's anchor matching, yolo_target.py:92)",
             inputs=dict(lhs=IOU_LHS, rhs=IOU_RHS), params={}),
        dict(name="bbox_batch_iou_pairs", op="bbox_batch_iou",
             decides="gluoncv BBoxBatchIOU(fmt='corner', offset 0, eps 1e-15) on the same boxes as a batch of 2 "
                     "(oracle/ref_ops.c:228-247 `i / ((aa + ab) - i + 1e-15)`): identical boxes, zero area (0 / 1e-15), pad rows",
             inputs=dict(a=np.stack([IOU_LHS, IOU_LHS[::-1]]), b=np.stack([IOU_RHS, IOU_RHS[::-1]])), params={}),
    ]


# ------------------------------------------------------------------------------------ targets
def target_cases():
    # predicted boxes against one gt [0,0,2,2]: IoU 0.53125, 0.5 (== a dyadic ignore threshold), 0.46875, 0, and a pad row
    box_preds = np.array([[[0, 0, 2, 1.0625], [0, 0, 2, 1], [0, 0, 2, 0.9375], [10, 10, 12, 12], [0, 0, 2, 2], [0, 0, 1, 1]]], F32)
    gt = np.array([[[0, 0, 2, 2], [-1, -1, -1, -1]]], F32)
    cases = [dict(name="dynamic_targets_iou_at_thresh", op="dynamic_targets",
                  decides="`ious_max > ignore_iou_thresh` (yolo_target.py:204; oracle/yolo3_train_oracle.py merge_targets): IoU "
                          "0.53125 / 0.5 / 0.46875 against threshold 0.5, a -1 pad gt row contributes IoU 0",
                  inputs=dict(box_preds=box_preds, gt_boxes=gt), params=dict(num_class=3, ignore_iou_thresh=0.5))]
    n, c = box_preds.shape[1], 3
    obj_t = np.array([[[0], [0], [0], [1.0], [0.6], [0]]], F32)            # one positive, one mixup-weighted positive
    ctr = np.arange(n * 2, dtype=F32).reshape(1, n, 2) / 16
    scl = -np.arange(n * 2, dtype=F32).reshape(1, n, 2) / 8
    wts = np.full((1, n, 2), 1.5, F32)
    cls = np.full((1, n, c), -1, F32)
    cls[0, 3] = [0, 1, 0]
    cls[0, 4] = [1, 0, 0]
    for ls in (0, 1):
        cases.append(dict(
            name="target_merger_label_smooth_%d" % ls, op="target_merger",
            decides="YOLOV3TargetMerger (yolo_target.py:263-279): prefetched rows override dynamic ones where obj_t > 0, the "
                    "class mask, and label smoothing %s (min(1/C, 1/40), :272-278)" % ("ON" if ls else "off"),
            inputs=dict(box_preds=box_preds, gt_boxes=gt, obj_t=obj_t, centers_t=ctr, scales_t=scl, weights_t=wts, clas_t=cls),
            params=dict(num_class=c, ignore_iou_thresh=0.5, label_smooth=ls)))
    # prefetch generator: the float64-cell boxes of tests/conftest.py (centres on stride multiples where float64 and fp32
    # truncate to different cells), one box whose centre sits on the image edge, a 0.5-pixel-wide box (max(w, 1)), a pad row
    gt_boxes = np.array([[[225.0, 89.5, 255.0, 150.5], [232.0, 105.0, 248.0, 135.0], [115.0, 233.5, 125.0, 246.5],
                          [300.0, 300.0, 300.5, 340.0], [-1, -1, -1, -1]],
                         [[10.0, 20.0, 200.0, 380.0], [10.0, 20.0, 200.0, 380.0], [400.0, 380.0, 416.0, 416.0],
                          [-1, -1, -1, -1], [50.0, 50.0, 80.0, 90.0]]], F32)
    gt_ids = np.array([[[0], [1], [2], [3], [-1]], [[4], [5], [6], [-1], [7]]], F32)
    cases.append(dict(
        name="prefetch_targets_416", op="prefetch_targets",
        decides="YOLOV3PrefetchTargetGenerator (yolo_target.py:13-148) with the anchors / offsets / feature maps the net itself "
                "returns in train mode (transforms.py:185-197): float64 cell truncation (:115-116), log(max(w, 1) / anchor) "
                "(:121), weight 2 - wh / WH (:123), a later box overwriting an earlier one, rows after the first pad row ignored "
                "(:107-108)",
        inputs=dict(gt_boxes=gt_boxes, gt_ids=gt_ids), params=dict(num_class=20, size=416)))
    return cases


# ------------------------------------------------------------------------------------ loss, cell, sgd, resize
def loss_case():
    rs = np.random.RandomState(7)
    b, n, c = 2, 4, 3
    d = dict(objness=rs.randn(b, n, 1), box_centers=rs.randn(b, n, 2), box_scales=rs.randn(b, n, 2),
             cls_preds=rs.randn(b, n, c))
    d = {k: v.astype(F32) for k, v in d.items()}
    obj_t = np.array([[[1.0], [-1.0], [0.0], [0.6]], [[0.0], [0.0], [1.0], [-1.0]]], F32)  # positive, ignored, negative, mixup
    center_t = rs.rand(b, n, 2).astype(F32)
    scale_t = rs.randn(b, n, 2).astype(F32)
    weight_t = (1.0 + rs.rand(b, n, 2)).astype(F32)
    class_t = np.full((b, n, c), -1, F32)
    class_t[0, 0] = [0, 1, 0]
    class_t[0, 3] = [1, 0, 0]
    class_t[1, 2] = [0, 0, 1]
    class_mask = ((obj_t > 0) * (class_t >= 0)).astype(F32)
    d.update(objness_t=obj_t, center_t=center_t, scale_t=scale_t, weight_t=weight_t, class_t=class_t, class_mask=class_mask)
    return dict(name="yolov3_loss_pos_ignore_neg", op="yolov3_loss",
                decides="gluoncv.loss.YOLOV3Loss denominators and masks (oracle/yolo3_train_oracle.py loss(): every term is "
                        "mean(.) * count = a per-sample sum; hard objectness, ignore mask, weight_t * objness, class mask): rows = "
                        "positive / ignored (-1) / negative / mixup-weighted positive (0.6)",
                inputs=d, params={})


def cell_case():
    rs = np.random.RandomState(11)
    x = rs.randn(2, 4, 4, 4).astype(F32)
    return dict(name="conv_bn_leaky_train_step", op="conv_bn_leaky",
                decides="one recorded step of the reference's own `_conv2d(3, 1, 0, 1)` cell (layers.py:63-70): BatchNorm batch "
                        "statistics (biased variance), running_mean / running_var update with momentum 0.9 — BIASED or UNBIASED "
                        "variance stored (oracle RUNNING_VAR_UNBIASED = False) —, LeakyReLU(0.1), and the gradients of x, the "
                        "conv weight, gamma and beta for a given head gradient",
                inputs=dict(x=x, weight=(rs.randn(3, 4, 1, 1) * 0.5).astype(F32), gamma=(1 + 0.1 * rs.randn(3)).astype(F32),
                            beta=(0.1 * rs.randn(3)).astype(F32), running_mean=(0.1 * rs.randn(3)).astype(F32),
                            running_var=(1 + 0.1 * rs.rand(3)).astype(F32), dy=rs.randn(2, 3, 4, 4).astype(F32)),
                params=dict(kernel=1, stride=1, pad=0))


def sgd_cases():
    rs = np.random.RandomState(13)
    w, g1, g2 = [rs.randn(32).astype(F32) for _ in range(3)]
    out = []
    for name, wd_mult in (("sgd_momentum_wd", 1.0), ("sgd_momentum_no_wd", 0.0)):
        out.append(dict(
            name=name, op="sgd",
            decides="gluon.Trainer('sgd', {learning_rate 1e-3, wd 5e-4, momentum 0.9}).step(batch_size) twice "
                    "(train_yolov3.py:527-530,634; oracle sgd_step: g / batch_size, mom = 0.9 mom - lr (g + wd w), w += mom), "
                    "wd_mult %.0f (%s)" % (wd_mult, "default" if wd_mult else "--no_wd, train_yolov3.py:495-497"),
            inputs=dict(w=w, g1=g1, g2=g2), params=dict(lr=1e-3, momentum=0.9, wd=5e-4, batch_size=16, wd_mult=wd_mult)))
    return out


def imresize_cases():
    rs = np.random.RandomState(17)
    img = rs.randint(0, 256, (36, 48, 3)).astype(np.uint8)
    # smooth content too: interpolation differences show as structure, not noise
    yy, xx = np.mgrid[0:36, 0:48]
    img[..., 1] = ((np.sin(yy / 5.0) + np.cos(xx / 7.0) + 2) * 63).astype(np.uint8)
    out = []
    for tag, (nw, nh), what in (("shrink", (32, 32), "both sides shrink -> INTER_AREA"),
                                ("shrink_integer", (24, 18), "integer factor 2 x 2 area average"),
                                ("enlarge", (64, 64), "both sides grow -> INTER_CUBIC"),
                                ("mixed", (32, 64), "one shrinks, one grows -> INTER_LINEAR")):
        out.append(dict(name="imresize_" + tag, op="imresize",
                        decides="timage.imresize(img, w, h, interp=9) (transforms.py:325-327; oracle/resize_oracle.py restates "
                                "OpenCV's fixed-point paths): %s; 48 x 36 -> %d x %d" % (what, nw, nh),
                        inputs=dict(img=img), params=dict(width=nw, height=nh)))
    return out


def all_cases():
    cs = (box_nms_cases() + detect_heads_cases() + iou_cases() + target_cases() + [loss_case(), cell_case()] + sgd_cases()
          + imresize_cases())
    names = [c["name"] for c in cs]
    assert len(set(names)) == len(names)
    return cs


# ------------------------------------------------------------------------------------ this repo's oracle as a backend
class OracleOps(object):
    """Every operator of the kit through oracle/ (the CPU restatement).  The capture script's --from-oracle mode and the
    CPU tests use it; with mxnet's outputs in the fixture it is the thing under test."""

    def __init__(self):
        from oracle import yolo3_oracle as O, yolo3_train_oracle as TO, targets_oracle as T, resize_oracle as R
        self.O, self.TO, self.T, self.R = O, TO, T, R

    def run(self, op, inputs, params):
        return getattr(self, "op_" + op)(inputs, params)

    def op_box_nms(self, i, p):
        out, idx = self.O.box_nms(i["data"], p["overlap_thresh"], p["valid_thresh"], p["topk"], bool(p["force_suppress"]))
        return dict(out=out)

    def op_detect_heads(self, i, p):
        orc = self.O.OracleYolo3(int(p["classes"]), {})
        rows = orc.detections_from_heads([i["head0"], i["head1"], i["head2"]])
        out, idx = self.O.box_nms(rows, p["nms_thresh"], 0.01, int(p["nms_topk"]), False)   # yolo3.py:1198-1200
        return dict(rows=rows, out=out)

    def op_box_iou(self, i, p):
        # the oracle's batch IoU is BBoxBatchIOU's formula; box_iou (no eps, 0 on an empty union) is vy_math's vy_box_iou,
        # reached through box_nms's comparator only — restated here from include/vy_math.h:124-136 in fp32
        a, b = i["lhs"][:, None, :], i["rhs"][None, :, :]
        iw = np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0])
        ih = np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1])
        pos = (iw > 0) & (ih > 0)
        inter = (iw * ih).astype(F32)
        uni = ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])).astype(F32) - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = np.where(pos & (uni > 0), inter / uni, 0).astype(F32)
        return dict(out=iou)

    def op_bbox_batch_iou(self, i, p):
        return dict(out=self.O.batch_iou(i["a"], i["b"]))

    def _train(self, p):
        return self.TO.OracleYolo3Train(int(p["num_class"]), {}, ignore_iou_thresh=p["ignore_iou_thresh"],
                                        label_smooth=bool(p.get("label_smooth", 0)))

    def op_dynamic_targets(self, i, p):
        n = i["box_preds"].shape[1]
        c = int(p["num_class"])
        z = lambda k: np.zeros((1, n, k), F32)   # noqa: E731
        obj = self._train(p).merge_targets(i["box_preds"], i["gt_boxes"], z(1), z(2), z(2), z(2), np.full((1, n, c), -1, F32))[0]
        return dict(objness_t=obj)

    def op_target_merger(self, i, p):
        keys = ("objectness", "center_targets", "scale_targets", "weights", "class_targets", "class_mask")
        out = self._train(p).merge_targets(i["box_preds"], i["gt_boxes"], i["obj_t"], i["centers_t"], i["scales_t"],
                                           i["weights_t"], i["clas_t"])
        return dict(zip(keys, out))

    def op_prefetch_targets(self, i, p):
        keys = ("objectness", "center_targets", "scale_targets", "weights", "class_targets")
        s = int(p["size"])
        return dict(zip(keys, self.T.prefetch_targets(int(p["num_class"]), s, s, i["gt_boxes"], i["gt_ids"])))

    def op_yolov3_loss(self, i, p):
        orc = self.TO.OracleYolo3Train(i["cls_preds"].shape[-1], {})
        pr = dict(obj=i["objness"], xy=i["box_centers"], wh=i["box_scales"], cls=i["cls_preds"])
        tg = (i["objness_t"], i["center_t"], i["scale_t"], i["weight_t"], i["class_t"], i["class_mask"])
        losses, _ = orc.loss(pr, tg)
        return dict(zip(("obj_loss", "center_loss", "scale_loss", "cls_loss"), losses))

    def op_conv_bn_leaky(self, i, p):
        pre = "cell"
        prm = {pre + ".0.weight": i["weight"], pre + ".1.gamma": i["gamma"], pre + ".1.beta": i["beta"],
               pre + ".1.running_mean": i["running_mean"], pre + ".1.running_var": i["running_var"]}
        orc = self.TO.OracleYolo3Train(1, prm)
        orc.tape, orc.new_running, orc.new_running_dev = [], {}, []
        y = orc.cell(i["x"], pre, int(p["kernel"]), int(p["stride"]))
        grads = {}
        dx = orc._cell_bwd(orc.tape[-1], i["dy"], grads)
        return dict(y=y, running_mean=orc.new_running[pre + ".1.running_mean"], running_var=orc.new_running[pre + ".1.running_var"],
                    dx=dx, dweight=grads[pre + ".0.weight"], dgamma=grads[pre + ".1.gamma"], dbeta=grads[pre + ".1.beta"])

    def op_sgd(self, i, p):
        prm, mom = {"w": i["w"].copy()}, {}
        outs = {}
        for k, g in (("w1", i["g1"]), ("w2", i["g2"])):
            self.TO.sgd_step(prm, {"w": g}, mom, p["lr"], p["momentum"], p["wd"], p["batch_size"], wd_mult={"w": p["wd_mult"]})
            outs[k] = prm["w"].copy()
        return outs

    def op_imresize(self, i, p):
        return dict(out=self.R.imresize(i["img"], int(p["width"]), int(p["height"]), interp=9))


# ------------------------------------------------------------------------------------ the real operators
class MxnetOps(object):
    """Every operator of the kit through mxnet / gluoncv / the reference tree.  NEVER RUN in the build container (mxnet is
    not installable there); written against the public Gluon API the reference itself uses, each call next to the
    reference line it mirrors."""

    def __init__(self, mx, make_net):
        self.mx, self.make_net = mx, make_net
        self._nets = {}

    def nd(self, a):
        return self.mx.nd.array(np.asarray(a), dtype="float32")

    def run(self, op, inputs, params):
        return getattr(self, "op_" + op)(inputs, params)

    def net(self, classes, size):
        key = (classes, size)
        if key not in self._nets:
            net = self.make_net(["c%d" % i for i in range(classes)], pretrained_base=False, k=1)
            net.initialize()
            net(self.mx.nd.zeros((1, 3, size, size)))
            self._nets[key] = net
        return self._nets[key]

    def op_box_nms(self, i, p):
        out = self.mx.nd.contrib.box_nms(self.nd(i["data"]), overlap_thresh=p["overlap_thresh"], valid_thresh=p["valid_thresh"],
                                         topk=int(p["topk"]), id_index=0, score_index=1, coord_start=2,
                                         force_suppress=bool(p["force_suppress"]))       # yolo3.py:1198-1200
        return dict(out=out.asnumpy())

    def op_detect_heads(self, i, p):
        mx = self.mx
        net = self.net(int(p["classes"]), int(p["size"]))
        dets = [net.yolo_outputs[k](self.nd(i["head%d" % k])) for k in range(3)]          # yolo3.py:132-199, inference branch
        rows = mx.nd.concat(*dets, dim=1)                                                  # :1195
        out = mx.nd.contrib.box_nms(rows, overlap_thresh=p["nms_thresh"], valid_thresh=0.01, topk=int(p["nms_topk"]),
                                    id_index=0, score_index=1, coord_start=2, force_suppress=False)   # :1198-1200
        return dict(rows=rows.asnumpy(), out=out.asnumpy())

    def op_box_iou(self, i, p):
        return dict(out=self.mx.nd.contrib.box_iou(self.nd(i["lhs"]), self.nd(i["rhs"]), format="corner").asnumpy())

    def op_bbox_batch_iou(self, i, p):
        from gluoncv.nn.bbox import BBoxBatchIOU                                           # yolo_target.py:10,171
        return dict(out=BBoxBatchIOU()(self.nd(i["a"]), self.nd(i["b"])).asnumpy())

    def op_dynamic_targets(self, i, p):
        from models.definitions.yolo.yolo_target import YOLOV3DynamicTargetGeneratorSimple
        gen = YOLOV3DynamicTargetGeneratorSimple(int(p["num_class"]), p["ignore_iou_thresh"])
        return dict(objness_t=gen(self.nd(i["box_preds"]), self.nd(i["gt_boxes"]))[0].asnumpy())

    def op_target_merger(self, i, p):
        from models.definitions.yolo.yolo_target import YOLOV3TargetMerger
        m = YOLOV3TargetMerger(int(p["num_class"]), p["ignore_iou_thresh"])
        m._label_smooth = bool(p["label_smooth"])                                          # train_yolov3.py:499-500
        keys = ("objectness", "center_targets", "scale_targets", "weights", "class_targets", "class_mask")
        args = [self.nd(i[k]) for k in ("box_preds", "gt_boxes", "obj_t", "centers_t", "scales_t", "weights_t", "clas_t")]
        return dict(zip(keys, [t.asnumpy() for t in m(*args)]))

    def op_prefetch_targets(self, i, p):
        from mxnet import autograd
        from models.definitions.yolo.yolo_target import YOLOV3PrefetchTargetGenerator
        s = int(p["size"])
        net = self.net(int(p["num_class"]), s)
        fake = self.mx.nd.zeros((1, 3, s, s))
        with autograd.train_mode():                                                        # transforms.py:192-193
            _, anchors, offsets, feat_maps, _, _, _, _ = net(fake)
        gen = YOLOV3PrefetchTargetGenerator(num_class=int(p["num_class"]))
        keys = ("objectness", "center_targets", "scale_targets", "weights", "class_targets")
        out = gen(fake, feat_maps, anchors, offsets, self.nd(i["gt_boxes"]), self.nd(i["gt_ids"]), None)
        return dict(zip(keys, [t.asnumpy() for t in out]))

    def op_yolov3_loss(self, i, p):
        from gluoncv.loss import YOLOV3Loss                                                # yolo3.py:16,994
        order = ("objness", "box_centers", "box_scales", "cls_preds", "objness_t", "center_t", "scale_t", "weight_t",
                 "class_t", "class_mask")                                                  # the call at yolo3.py:1187
        out = YOLOV3Loss()(*[self.nd(i[k]) for k in order])
        return dict(zip(("obj_loss", "center_loss", "scale_loss", "cls_loss"), [t.asnumpy() for t in out]))

    def op_conv_bn_leaky(self, i, p):
        from mxnet import autograd
        from models.definitions.layers import _conv2d                                      # layers.py:63
        cell = _conv2d(i["weight"].shape[0], int(p["kernel"]), int(p["pad"]), int(p["stride"]))
        cell.initialize()
        x = self.nd(i["x"])
        cell(x)
        conv, bn = cell[0], cell[1]
        conv.weight.set_data(self.nd(i["weight"]))
        for k in ("gamma", "beta", "running_mean", "running_var"):
            getattr(bn, k).set_data(self.nd(i[k]))
        x.attach_grad()
        with autograd.record():
            y = cell(x)
        y.backward(self.nd(i["dy"]))
        return dict(y=y.asnumpy(), running_mean=bn.running_mean.data().asnumpy(), running_var=bn.running_var.data().asnumpy(),
                    dx=x.grad.asnumpy(), dweight=conv.weight.grad().asnumpy(), dgamma=bn.gamma.grad().asnumpy(),
                    dbeta=bn.beta.grad().asnumpy())

    def op_sgd(self, i, p):
        from mxnet import autograd, gluon
        w = gluon.Parameter("w", shape=i["w"].shape)
        w.initialize()
        w.set_data(self.nd(i["w"]))
        w.wd_mult = float(p["wd_mult"])                                                    # train_yolov3.py:496-497
        tr = gluon.Trainer([w], 'sgd', {'learning_rate': p["lr"], 'wd': p["wd"], 'momentum': p["momentum"]},
                           kvstore='local')                                                # :527-530
        outs = {}
        for k, g in (("w1", i["g1"]), ("w2", i["g2"])):
            with autograd.record():
                loss = (w.data() * self.nd(g)).sum()      # d loss / d w = g
            loss.backward()
            tr.step(int(p["batch_size"]))                                                  # :634
            outs[k] = w.data().asnumpy()
        return outs

    def op_imresize(self, i, p):
        from gluoncv.data.transforms import image as timage                               # transforms.py:7
        img = self.mx.nd.array(i["img"], dtype="uint8")
        return dict(out=timage.imresize(img, int(p["width"]), int(p["height"]), interp=9).asnumpy())   # :327
