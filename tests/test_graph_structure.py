"""-m "not gpu": the graph the planner executes (vy_net_num_convs / vy_net_conv_info / vy_net_param_info, host-only)
against tests/golden/graph_structure.json — the structure the REFERENCE's own constructors and hybrid_forward methods
build, recorded by tests/golden/make_graph_structure.py under a recording stand-in for mxnet / gluoncv (build container
only; the fixture is data: names, channel counts, constructor arguments, operator-call parameters).

STRUCTURE ONLY.  The fixture holds no number any mxnet operator computed, so this settles mechanically what a reading of
the reference could get wrong (SURVEY section 2 said 26 cells take SyncBatchNorm: it is 6) and ties rows a2-a6 / a13 of
SURVEY section 8 to executed reference code — it does NOT pin the oracle's arithmetic (DESIGN.md section 2 stays
"parity unpinned")."""
import ctypes
import inspect
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "graph_structure.json")) as f:
        return json.load(f)


def _net(ncls):
    from videoyolo_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(L.vy_net_create(ncls, ctypes.byref(h)))
    return L, h


def _convs(ncls):
    from videoyolo_amd import _lib
    L, h = _net(ncls)
    rows = []
    ci = _lib.ConvInfo()
    for i in range(L.vy_net_num_convs(h)):
        _lib.check(L.vy_net_conv_info(h, i, ctypes.byref(ci)))
        rows.append({f: (getattr(ci, f).decode() if f == "name" else int(getattr(ci, f))) for f, _ in _lib.ConvInfo._fields_})
    L.vy_net_destroy(h)
    return rows


def _params(ncls):
    from videoyolo_amd import _lib
    L, h = _net(ncls)
    rows = []
    pi = _lib.ParamInfo()
    for i in range(L.vy_net_num_params(h)):
        _lib.check(L.vy_net_param_info(h, i, ctypes.byref(pi)))
        rows.append((pi.name.decode(), int(pi.size), int(pi.trainable), [int(pi.shape[j]) for j in range(pi.ndim)], int(pi.backbone)))
    L.vy_net_destroy(h)
    return rows


@pytest.mark.parametrize("key,ncls", [("voc20", 20), ("vid30", 30)])
def test_conv_list_in_the_references_execution_order(golden, key, ncls):
    """75 convs; every row — cell name, cin (which Gluon infers at the first forward: recorded by running the reference's
    hybrid_forward methods), cout, kernel, stride, padding, bias or BatchNorm + LeakyReLU(0.1) — and the ORDER in which
    YOLOV3T.hybrid_forward (yolo3.py:1105-1177) reaches them."""
    ref, got = golden[key]["convs"], _convs(ncls)
    assert len(ref) == len(got) == 75
    for r, g in zip(ref, got):
        cell = r["name"][:-2] if r["norm"] is not None else r["name"]  # _conv2d: HybridSequential [Conv2D, norm, LeakyReLU]
        assert r["name"] == (cell + ".0" if r["norm"] is not None else cell)
        assert g["name"] == cell, (r["exec_index"], g["name"], cell)
        assert (g["cin"], g["cout"]) == (r["cin"], r["cout"]), cell
        assert r["kernel"] == [g["kernel"]] * 2 and r["stride"] == [g["stride"]] * 2 and r["pad"] == [g["pad"]] * 2, cell
        assert r["groups"] == 1
        assert bool(g["has_bn"]) == (r["norm"] is not None) == (not r["use_bias"]), cell
        if r["norm"] is not None:
            assert r["norm"]["name"] == cell + ".1"
            assert r["norm"]["kwargs"] == {"epsilon": 1e-05, "momentum": 0.9}, cell      # layers.py:68
            assert r["act"] == {"class": "LeakyReLU", "alpha": 0.1}, cell                 # layers.py:69
        else:
            assert r["act"] is None and cell.endswith(".prediction") and g["cout"] == 3 * (5 + ncls)   # yolo3.py:60-62
    assert sum(1 for g in got if g["has_bn"]) == 72


def test_constants_the_kernels_compile_in(golden):
    """eps / momentum / leaky slope of the recorded cells are the ones the library and the oracle use."""
    from oracle import yolo3_oracle as O
    import videoyolo_amd.model as M
    assert inspect.signature(O.bn_fold).parameters["eps"].default == 1e-5
    assert inspect.signature(O.bn_train).parameters["eps"].default == 1e-5
    src = open(os.path.join(os.path.dirname(HERE), "include", "vy_math.h")).read()
    assert "0.1f" in src  # vy_leaky
    assert M.YOLOV3._ANCHORS == tuple(tuple(int(a) for a in h["anchors"]) for h in golden["voc20"]["heads"])


@pytest.mark.parametrize("key,ncls", [("voc20", 20), ("vid30", 30)])
def test_stage_split_residuals_and_concat_fusion(golden, key, ncls):
    g = golden[key]
    got = _convs(ncls)
    by = {r["name"]: r for r in got}
    # wrappers.py:58: features[:15] / [15:24] / [24:] — a slice of a HybridSequential is a new one, indexed from 0
    assert g["stage_lengths"] == [15, 9, 5]
    for s, n in enumerate(g["stage_lengths"]):
        kids = {r["name"].split(".")[2] for r in got if r["name"].startswith("stages.%d." % s)}
        assert kids == {str(i) for i in range(n)}
    # darknet.py:40: x + body(x) — the addend rides in the epilogue of the block's second cell
    for r in got:
        assert bool(r["residual"]) == (r["name"].endswith(".body.1") and r["name"].startswith("stages.")), r["name"]
    assert sum(r["residual"] for r in got) == 23
    # yolo3.py:1167-1177: concat(slice_like(upsample(transition(x))), route_now) on axis 1, UPSAMPLED FIRST
    cats = [c for c in g["concats_axis1"] if c["channels"][0] is not None]
    assert [c["inputs"] for c in cats] == [["upsample(transitions.0.0)", "stages.1.8.body.1.0"],
                                           ["upsample(transitions.1.0)", "stages.0.14.body.1.0"]]
    for c in cats:
        up, route = by[c["inputs"][0][len("upsample("):-3]], by[c["inputs"][1][:-2]]
        total = sum(c["channels"])
        assert up["upsample"] == 2 and up["concat_offset"] == 0 and up["cout"] == c["channels"][0]
        assert route["concat_offset"] == c["channels"][0] and route["cout"] == c["channels"][1]   # written behind it
        assert up["out_channels_total"] == route["out_channels_total"] == total
    assert [r["name"] for r in got if r["upsample"] == 2] == ["transitions.0", "transitions.1"]
    # the recorded upsample is repeat(axis=-1) then repeat(axis=-2), both x2 (layers.py:11-20), cropped on axes (2, 3)
    top = [(o["op"], o["kwargs"]) for o in g["ops_top_level"]]
    assert top[:4] == [("x.repeat", {"axis": -1, "repeats": 2}), ("x.repeat", {"axis": -2, "repeats": 2}),
                       ("F.slice_like", {"axes": [2, 3]}), ("F.concat", {"dim": 1})]


def test_only_six_cells_take_the_passed_norm_layer(golden):
    """train_yolov3.py:350-354 passes SyncBatchNorm(num_devices) to yolo3_darknet53; wrappers.py:54 hands it to get_darknet
    and NOT to YOLOV3T (wrappers.py:101-103), and the residual blocks hard-wire BatchNorm (darknet.py:89-91): the stem and
    the five stride-2 convs are the only cells that exchange statistics.  (SURVEY section 2 said 26.)"""
    g = golden["voc20_syncbn8"]
    assert g["norm_classes"] == {"SyncBatchNorm": 6, "BatchNorm": 66}
    want = sorted(r["name"] + ".1" for r in _convs(20) if r["sync_bn"])
    assert want == g["cells_with_the_passed_norm_layer"] == ["stages.0.0.1", "stages.0.1.1", "stages.0.3.1", "stages.0.6.1",
                                                              "stages.1.0.1", "stages.2.0.1"]
    sync = [r for r in g["convs"] if r["norm"] and r["norm"]["class"] == "SyncBatchNorm"]
    assert all(r["norm"]["kwargs"] == {"epsilon": 1e-05, "momentum": 0.9, "num_devices": 8} for r in sync)
    assert [r["stride"] for r in sync] == [[1, 1]] + [[2, 2]] * 5
    # the graph itself does not depend on the norm layer
    strip = lambda rows: [{k: v for k, v in r.items() if k != "norm"} for r in rows]  # noqa: E731
    assert strip(g["convs"]) == strip(golden["voc20"]["convs"])


@pytest.mark.parametrize("key,ncls,count", [("voc20", 20, 61626049), ("vid30", 30, 61679899)])
def test_parameter_names_and_counts(golden, key, ncls, count):
    g = golden[key]
    rows = _params(ncls)
    assert g["trainable_parameters"] == count == sum(sz for _, sz, tr, _, _ in rows if tr)
    assert g["running_statistics"] == 52608 == sum(sz for _, sz, tr, _, _ in rows if not tr)
    # the structural names save_parameters writes (the file is a name -> array dict: order carries no meaning).  Gluon lists
    # them in child REGISTRATION order (stages, transitions, yolo_blocks, yolo_outputs: yolo3.py:1000-1008), the library in
    # execution order; inside one block the orders agree
    assert sorted(n for n, *_ in rows) == sorted(g["parameter_names"]) and len(rows) == len(g["parameter_names"]) == 366
    for top in ("stages.", "transitions.", "yolo_blocks.", "yolo_outputs."):
        sub = lambda names: sorted(n for n in names if n.startswith(top))  # noqa: E731
        assert sub(n for n, *_ in rows) == sub(g["parameter_names"])
    assert [n for n, *_ in rows if n.startswith("stages.")] == [n for n in g["parameter_names"] if n.startswith("stages.")]
    shapes = {n: shp for n, _, _, shp, _ in rows}
    for r in g["convs"]:
        assert shapes[r["name"] + ".weight"] == [r["cout"], r["cin"], r["kernel"][0], r["kernel"][1]]
        if r["use_bias"]:
            assert shapes[r["name"] + ".bias"] == [r["cout"]]
    # freeze_base (wrappers.py:55-57) reaches exactly the Darknet-53 tensors
    assert all(bool(bb) == n.startswith("stages.") for n, _, _, _, bb in rows)


def test_heads_and_the_detection_calls(golden):
    """anchors[::-1][i] / strides[::-1][i] per output layer (yolo3.py:1013-1014, wrappers.py:80-84), the offsets constant
    (x, y) meshgrid of alloc_size (yolo3.py:67-74), the box_nms call of yolo3.py:1198-1200 with the defaults of
    yolo3.py:959-963, and the slices that make (ids, scores, bboxes)."""
    import videoyolo_amd as vy
    from videoyolo_amd import targets
    from oracle import yolo3_oracle as O
    g = golden["voc20"]
    assert [h["stride"] for h in g["heads"]] == list(targets.STRIDES) == [32, 16, 8]
    assert np.array_equal(np.array([h["anchors"] for h in g["heads"]]).reshape(9, 2), targets.ANCHORS[[6, 7, 8, 3, 4, 5, 0, 1, 2]]) or \
        np.array_equal(np.array([h["anchors"] for h in g["heads"]]).reshape(9, 2), targets.ANCHORS)
    assert all(h["num_anchors"] == 3 and h["num_pred"] == 25 and h["offsets_shape"] == [1, 1, 128, 128, 2] for h in g["heads"])
    assert all(h["offsets_first_xy"] == [2, 1] for h in g["heads"])   # offsets[0, 0, y = 1, x = 2] = (x, y)
    net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
    consts = net.constants()
    for i, h in enumerate(g["heads"]):
        assert consts["yolo_outputs.%d.anchors" % i].reshape(-1).tolist() == h["anchors"]
        assert list(consts["yolo_outputs.%d.offsets" % i].shape) == h["offsets_shape"]
        assert consts["yolo_outputs.%d.offsets" % i][0, 0, 1, 2].tolist() == h["offsets_first_xy"]
    assert g["nms_defaults"] == {"nms_thresh": 0.45, "nms_topk": 400, "post_nms": 100}
    assert (net.nms_thresh, net.nms_topk, net.post_nms) == (0.45, 400, 100)
    ops = {o["op"]: o for o in g["ops_top_level"]}
    nms = ops["F.contrib.box_nms"]["kwargs"]
    assert nms == {"coord_start": 2, "force_suppress": False, "id_index": 0, "overlap_thresh": 0.45, "score_index": 1, "topk": 400,
                   "valid_thresh": 0.01}
    sig = inspect.signature(O.box_nms).parameters
    assert {k: sig[k].default for k in ("overlap_thresh", "valid_thresh", "topk", "force_suppress")} == \
        {"overlap_thresh": 0.45, "valid_thresh": 0.01, "topk": 400, "force_suppress": False}
    tail = [(o["op"], o["kwargs"]) for o in g["ops_top_level"][-4:]]
    assert tail == [("x.slice_axis", {"axis": 1, "begin": 0, "end": 100}), ("x.slice_axis", {"axis": -1, "begin": 0, "end": 1}),
                    ("x.slice_axis", {"axis": -1, "begin": 1, "end": 2}), ("x.slice_axis", {"axis": -1, "begin": 2, "end": None})]


@pytest.mark.parametrize("key,ncls", [("voc20", 20), ("vid30", 30)])
def test_decode_call_sequence_of_the_output_layer(golden, key, ncls):
    """The operator calls YOLOOutputV3.hybrid_forward makes in inference mode (yolo3.py:158-197), with their non-tensor
    arguments, as executed: the sequence oracle/yolo3_oracle.py's `output` restates literally and csrc/detect.hip fuses —
    (B, A*P, HW) -> (B, HW, A, P); xy / wh / obj / cls slices; class-major tiling; (C, B, HW, A, 6) -> (B, C*HW*A, 6)."""
    P = 5 + ncls
    want = [("x.reshape", [[0, 3 * P, -1]], {}), ("x.transpose", [], {"axes": [0, 2, 1]}), ("x.reshape", [[0, -1, 3, P]], {}),
            ("x.slice_axis", [], {"axis": -1, "begin": 0, "end": 2}), ("x.slice_axis", [], {"axis": -1, "begin": 2, "end": 4}),
            ("x.slice_axis", [], {"axis": -1, "begin": 4, "end": 5}), ("x.slice_axis", [], {"axis": -1, "begin": 5, "end": None}),
            ("F.slice_like", [], {"axes": [2, 3]}), ("x.reshape", [[1, -1, 1, 2]], {}),
            ("F.sigmoid", [], {}), ("F.broadcast_add", [], {}), ("F.exp", [], {}), ("F.broadcast_mul", [], {}),
            ("F.sigmoid", [], {}), ("F.sigmoid", [], {}), ("F.broadcast_mul", [], {}), ("F.concat", [], {"dim": -1}),
            ("F.tile", [], {"reps": [ncls, 1, 1, 1, 1]}), ("F.transpose", [], {"axes": [3, 0, 1, 2]}), ("x.expand_dims", [], {"axis": -1}),
            ("F.arange", [0, ncls], {}), ("x.reshape", [[0, 1, 1, 1, 1]], {}), ("F.broadcast_add", [], {}), ("F.concat", [], {"dim": -1}),
            ("x.transpose", [], {"axes": [1, 0, 2, 3, 4]}), ("F.reshape", [[0, -1, 6]], {})]
    got = [(o["op"], o["args"], o["kwargs"]) for o in golden[key]["ops_output_layer_0"]]
    assert got == want


@pytest.mark.parametrize("key,ncls", [("voc20", 20), ("vid30", 30)])
def test_training_call_structure(golden, key, ncls):
    """The RECORDED call net(x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t) (train_yolov3.py:625) as the
    reference executes it — which slice of the prediction goes to which argument of YOLOV3Loss (yolo3.py:1181-1187), what
    the target merger selects (yolo_target.py:259-281) and the dynamic generator computes (yolo_target.py:193-205) — and
    the 8-tuple of the train-mode non-recording call (yolo3.py:1189-1192).  Call structure only: what YOLOV3Loss and
    BBoxBatchIOU compute is not in the tree and is not recorded."""
    import videoyolo_amd as vy
    from oracle import yolo3_train_oracle as TO
    t = golden[key]["train"]
    assert t["returns"] == ["obj_loss", "center_loss", "scale_loss", "cls_loss"]
    assert t["loss_call_arg_tags"] == [
        "pred[-1:4:5]", "pred[-1:0:2]", "pred[-1:2:4]", "pred[-1:5:None]",                       # objness, centers, scales, classes
        "where((obj_t > 0) ? obj_t : ((batch_iou(pred[-1:0:2], gt_boxes) > 0.7) * -1))",       # ignore mask: -1 where max IoU > 0.7
        "where((obj_t > 0) ? centers_t : zeros)", "where((obj_t > 0) ? scales_t : zeros)", "where((obj_t > 0) ? weights_t : zeros)",
        "where((obj_t > 0) ? clas_t : (ones * -1))",
        "((obj_t > 0) * (where((obj_t > 0) ? clas_t : (ones * -1)) >= 0))"]                     # class_mask
    assert t["ctor_records"] == [{"ctor": "BBoxBatchIOU", "args": [], "kwargs": {}}, {"ctor": "YOLOV3Loss", "args": [], "kwargs": {}}]
    assert t["ignore_iou_thresh"] == 0.7 and t["label_smooth_default"] is False
    net = vy.yolo3_darknet53(["c%d" % i for i in range(ncls)], pretrained_base=False)
    assert net._ignore_iou_thresh == 0.7 and net._target_generator._label_smooth is False
    assert [(o["op"], o["args"], o["kwargs"]) for o in t["ops_dynamic_target"]][-2:] == \
        [("x.max", [], {"axis": -1, "keepdims": True}), ("x.>", [0.7], {})]
    merger = [(o["op"], o["args"], o["kwargs"]) for o in t["ops_target_merger"]]
    assert merger[:10] == [("x.>", [0], {}), ("F.where", [], {}), ("x.tile", [], {"reps": [2]}), ("F.where", [], {}), ("F.where", [], {}),
                           ("F.where", [], {}), ("x.tile", [], {"reps": [ncls]}), ("F.where", [], {}), ("x.tile", [], {"reps": [ncls]}),
                           ("x.>=", [0], {})]
    assert [m[0] for m in merger[10:]] == ["F.stop_gradient"] * 6
    # the oracle's restatement takes the same arguments in the same order and returns the same six targets
    assert list(inspect.signature(TO.OracleYolo3Train.merge_targets).parameters)[1:] == \
        ["box_preds", "gt_boxes", "obj_t", "centers_t", "scales_t", "weights_t", "clas_t"]
    # train-mode, not recording: (box_preds, [anchors] x 3, [offsets] x 3, [fake featmaps] x 3, centers, scales, objness, classes)
    assert t["train_mode_tuple"] == ["pred[-1:0:2]", ["const:anchors"] * 3, ["const:offsets"] * 3, ["zeros"] * 3,
                                     "pred[-1:0:2]", "pred[-1:2:4]", "pred[-1:4:5]", "pred[-1:5:None]"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/definitions"), reason="build container only: needs /root/reference")
def test_fixture_is_what_the_recorder_writes_today(golden, tmp_path):
    """Where the reference tree is present (the build container, never the GPU box) the recorder is re-run in a child process and
    must reproduce the committed fixture exactly: the fixture cannot drift from its generating script."""
    import subprocess
    import sys
    script = os.path.join(HERE, "golden", "make_graph_structure.py")
    # the script writes next to itself: run a copy that writes into tmp_path
    src = open(script).read().replace('OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_structure.json")',
                                      "OUT = %r" % str(tmp_path / "g.json"))
    copy = tmp_path / "make_graph_structure.py"
    copy.write_text(src)
    p = subprocess.run([sys.executable, str(copy)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    with open(tmp_path / "g.json") as f:
        fresh = json.load(f)
    assert fresh == golden


def _marked_net(classes):
    """a host-side net whose prediction convs hold marker values: row r of head i = 1000 (i + 1) + r (bias: + 0.5)"""
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(list(classes), pretrained_base=False)
    net.initialize(init="synthetic", seed=3)
    for i in range(3):
        w = net.collect_params()["yolo_outputs.%d.prediction.weight" % i]
        b = net.collect_params()["yolo_outputs.%d.prediction.bias" % i]
        rows = np.arange(w.shape[0], dtype=np.float32)
        w.set_data(np.broadcast_to((1000.0 * (i + 1) + rows).reshape(-1, 1, 1, 1), w.shape).astype(np.float32))
        b.set_data((1000.0 * (i + 1) + rows + 0.5).astype(np.float32))
    return net


def _src_rows(net):
    out = []
    for i in range(3):
        w = net.collect_params()["yolo_outputs.%d.prediction.weight" % i].data()
        b = net.collect_params()["yolo_outputs.%d.prediction.bias" % i].data()
        src = []
        for r in range(w.shape[0]):
            v = float(w[r, 0, 0, 0]) - 1000.0 * (i + 1)
            marked = abs(v - round(v)) < 1e-3 and 0 <= round(v) < 400 and np.all(w[r] == w[r, 0, 0, 0])
            if marked:
                assert abs(float(b[r]) - 0.5 - 1000.0 * (i + 1) - round(v)) < 1e-3, "weight and bias rows moved differently"
            else:
                assert abs(float(b[r])) < 100.0          # a fresh bias, too
            src.append(int(round(v)) if marked else -1)
        out.append(src)
    return out


def test_reset_class_moves_the_rows_the_reference_moves(golden):
    """`net.reset_class(classes, reuse_weights)` (train_yolov3.py:728-729) against the REFERENCE's own reset_class executed on
    marker values by the recorder (yolo3.py:1230-1302 + YOLOOutputV3.reset_class :76-129): for every form of `reuse_weights`
    its docstring lists — names, indices, mixed, a list, None, {} — every row of the three new prediction convs holds the same
    old row or is freshly initialised exactly where the reference leaves it so (note: the box / objectness rows are copied
    only when at least one class is re-used), the same warnings are issued and the same ValueErrors raised, message for message."""
    import warnings
    g = golden["reset_class"]
    old = g["old_classes"]
    for case in g["cases"]:
        reuse = case["reuse_weights"]
        if isinstance(reuse, list) and reuse and isinstance(reuse[0], list):
            reuse = {k: v for k, v in reuse}
        elif reuse == [] and case["label"] == "empty_dict":
            reuse = {}
        net = _marked_net(old)
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            net.reset_class(list(case["classes"]), reuse_weights=reuse)
        assert _src_rows(net) == case["src_rows"], case["label"]
        assert list(net.classes) == case["classes_after"] and net._target_generator._num_class == case["merger_num_class"]
        assert sorted(str(w.message) for w in wl) == sorted(case["warnings"]), case["label"]
    for e in g["errors"]:
        net = _marked_net(old)
        with pytest.raises(ValueError) as ei:
            net.reset_class(list(e["classes"]), reuse_weights={k: v for k, v in e["reuse_weights"]})
        assert str(ei.value) == e["message"]


def test_freeze_base_reaches_the_parameters_the_reference_freezes(golden):
    """yolo3_darknet53(..., freeze_base=True): wrappers.py:55-57 sets grad_req = 'null' on darknet.collect_params() — every
    tensor of the three stages, running statistics included — executed by the recorder; here: the same names."""
    import videoyolo_amd as vy
    g = golden["freeze_base"]
    net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False, freeze_base=True)
    null = sorted(p.name for p in net.collect_params().values() if p.grad_req == "null")
    write = sorted(p.name for p in net.collect_params().values() if p.grad_req != "null")
    # the running statistics of the HEADS are 'null' in gluon by construction (BatchNorm's running_mean / running_var have
    # grad_req 'null' always); the recorder's stand-in creates every parameter as 'write', so compare modulo those
    heads_running = [n for n in null if not n.startswith("stages.") and n.rsplit(".", 1)[-1].startswith("running_")]
    assert sorted(set(null) - set(heads_running)) == g["grad_req_null"]
    assert sorted(set(write) | set(heads_running)) == g["grad_req_write"]
    assert all(n.startswith("stages.") for n in g["grad_req_null"]) and len(g["grad_req_null"]) == 52 * 5


def test_pretrained_base_name_mapping(golden):
    """`pretrained_base=True` (three_darknet.py:262-264, wrappers.py:58): the classifier checkpoint's `features.<n>...` names
    against the detector's `stages.<s>.<j>...` names — in the reference the SAME parameter objects carry both, and the
    recorder read the pairs off by object identity; `output.*` (the dense layer) has no counterpart and is dropped."""
    from videoyolo_amd.model import darknet53_to_stage_names
    g = golden["backbone_names"]
    m = g["classifier_to_detector"]
    assert g["features_children"] == 29 and g["classifier_only"] == ["output.bias", "output.weight"] and len(m) == 52 * 5
    got = darknet53_to_stage_names({k: i for i, k in enumerate(list(m) + g["classifier_only"])})
    assert {v: k for k, v in got.items()} == {i: m[k] for i, k in enumerate(m)}
    with pytest.raises(ValueError):
        darknet53_to_stage_names({"darknetv30_conv0_weight": 0})
