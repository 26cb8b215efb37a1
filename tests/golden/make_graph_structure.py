#!/usr/bin/env python3
"""Records the STRUCTURE the reference's own constructors build for ``yolo3_darknet53`` and writes
tests/golden/graph_structure.json.  Build container only (reads /root/reference; nothing of it is copied).

What this is.  ``mxnet`` / ``gluoncv`` do not exist here, so the reference cannot compute anything.  But its model
DEFINITION is plain Python: ``models/definitions/yolo/wrappers.py:9-110`` calls ``get_darknet`` (three_darknet.py),
slices ``features[:15] / [15:24] / [24:]``, and builds ``YOLOV3T`` (yolo3.py:915-1054), which builds its detection
blocks, transitions and output layers through ``_conv2d`` (layers.py:63-70).  This script imports those files under a
RECORDING stand-in for the ``mxnet`` / ``gluoncv`` module trees: ``Block`` / ``HybridSequential`` keep their children by
name exactly as Gluon does (attribute name, or the running index inside a sequential — the structural names that
``save_parameters`` writes), ``Conv2D`` / ``BatchNorm`` / ``SyncBatchNorm`` / ``LeakyReLU`` record their constructor
arguments, and a forward pass with a shape-less tensor stand-in that only carries a CHANNEL COUNT runs the reference's
own ``hybrid_forward`` methods, so that every Conv2D learns its input channels (Gluon infers them the same way, at the
first forward) and the order in which the reference EXECUTES its convs, concats and the ``box_nms`` call is written
down together with the non-tensor arguments of every operator call.

What this is NOT.  No arithmetic of mxnet is reproduced or pretended: the stand-in tensors hold no values, no operator
computes anything, and nothing here says what ``Convolution``, ``BatchNorm``, ``box_nms`` or ``YOLOV3Loss`` return.
The fixture therefore pins rows a2-a6 / a13 of SURVEY section 8 (graph, channel plan, which cells take the passed
norm_layer, anchors / strides per head, concat order, operator call parameters, parameter counts) to the reference's
executed constructors instead of a reading of them — it does NOT lift the oracle from "parity unpinned" (DESIGN 2).

    python tests/golden/make_graph_structure.py            # writes tests/golden/graph_structure.json
"""
import collections
import contextlib
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_structure.json")


# ------------------------------------------------------------------------------------------------------------------
# shape-less tensor: carries a channel count (or None) and records nothing by itself
# ------------------------------------------------------------------------------------------------------------------
class T:
    def __init__(self, c=None, tag=None):
        self.c = c
        self.tag = tag  # provenance label for the concat record

    def _same(self, *a, **k):
        return T(self.c, self.tag)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def method(*a, **k):
            TRACE.op("x." + name, a, k)
            return T(self.c, self.tag)
        return method

    def slice_axis(self, *a, **k):  # the tag remembers the slice: how the loss call's arguments are told apart
        TRACE.op("x.slice_axis", a, k)
        return T(self.c, "%s[%s:%s:%s]" % (self.tag, k.get("axis"), k.get("begin"), k.get("end")))

    def _cmp(name):  # noqa: N805
        def f(self, other):
            TRACE.op("x." + name, (other,), {})
            return T(self.c, "(%s %s %s)" % (self.tag, name, other.tag if isinstance(other, T) else other))
        return f

    __gt__, __lt__, __ge__, __le__ = _cmp(">"), _cmp("<"), _cmp(">="), _cmp("<=")

    def repeat(self, *a, **k):  # _upsample (layers.py:11-20): repeat on W, then H
        TRACE.op("x.repeat", a, k)
        return T(self.c, "upsample(%s)" % self.tag if not str(self.tag).startswith("upsample(") else self.tag)

    __add__ = __radd__ = __sub__ = __rsub__ = __rmul__ = __truediv__ = __neg__ = _same

    def __mul__(self, other):  # (x * 0 is the reference's idiom for "a tensor shaped like x": the tag stays)
        if isinstance(other, T):
            return T(self.c, "(%s * %s)" % (self.tag, other.tag))
        return T(self.c, self.tag if other == 0 or str(self.tag).startswith(("stages.", "yolo_", "transitions.", "upsample(", "concat", "input"))
                 else "(%s * %s)" % (self.tag, other))

    def __iter__(self):
        raise TypeError("tensor stand-in is not iterable")


def _plain(v):
    """non-tensor arguments of an operator call, JSON-able"""
    if isinstance(v, T):
        return "<T>"
    if isinstance(v, (list, tuple)):
        return [_plain(u) for u in v]
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return repr(v)


class Trace:
    def __init__(self):
        self.reset()

    def reset(self):
        self.ops, self.convs, self.norms, self.acts, self.concats, self.stack = [], [], [], [], [], []
        self.misc = [m for m in getattr(self, "misc", []) if "ctor" in m]  # constructor records survive a reset

    def op(self, name, a, k):
        self.ops.append({"op": name, "in": self.stack[-1] if self.stack else "", "args": [_plain(v) for v in a if not isinstance(v, T)],
                         "kwargs": {kk: _plain(v) for kk, v in sorted(k.items())}})


TRACE = Trace()


class FNamespace:
    """stand-in for the ``F`` argument of hybrid_forward (mx.nd / mx.sym): every operator returns a tensor stand-in with
    the channel count of its first tensor argument; concat along axis 1 adds the counts"""

    def __init__(self, prefix="F."):
        self._p = prefix

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "contrib":
            return FNamespace(self._p + "contrib.")

        def fn(*a, **k):
            TRACE.op(self._p + name, a, k)
            ts = [v for v in a if isinstance(v, T)]
            if name == "concat":
                dim = k.get("dim", 1)
                rec = {"in": TRACE.stack[-1] if TRACE.stack else "", "dim": dim, "channels": [t.c for t in ts], "inputs": [t.tag for t in ts]}
                TRACE.concats.append(rec)
                import re
                tags = {re.sub(r"yolo_outputs\.\d\.prediction", "pred", str(t.tag)) for t in ts}  # the same slice of each scale
                tag = tags.pop() if len(tags) == 1 else "concat"
                if dim == 1 and all(t.c is not None for t in ts):
                    return T(sum(t.c for t in ts), tag)
                return T(None, tag)
            if name == "where":
                return T(None, "where(%s ? %s : %s)" % tuple(t.tag for t in ts[:3]))
            if name in ("zeros_like", "ones_like"):
                return T(None, name[:-5])
            return T(ts[0].c, ts[0].tag) if ts else T(None)
        return fn


F = FNamespace()


# ------------------------------------------------------------------------------------------------------------------
# Gluon's container semantics: children by attribute name / running index; parameters by name
# ------------------------------------------------------------------------------------------------------------------
class Param:
    def __init__(self, name, value=None, **kw):
        self.name, self.value, self.grad_req, self.kw = name, value, "write", kw
        self.wd_mult = self.lr_mult = 1.0
        self.shape = None      # set by the layer once its input channels are known
        self._data = None      # reset_class recording only: a numpy array; NaN = "freshly initialised, never written"

    def list_ctx(self):
        return ["cpu(0)"]

    def data(self, ctx=None):
        import numpy as np
        if self._data is None:
            self._data = np.full(self.shape, np.nan, np.float64)
        return self._data

    def set_data(self, value):
        import numpy as np
        self._data = np.array(value, np.float64)


class ParamDict(collections.OrderedDict):
    def get(self, name, **kw):  # noqa: A003
        if name not in self:
            self[name] = Param(name, **kw)
        return self[name]

    def get_constant(self, name, value=None):
        import numpy as np
        if name not in self:
            self[name] = Param(name, value=np.asarray(value))
            self[name].grad_req = "null"
        return self[name]

    def reset_ctx(self, ctx):
        pass


class Block:
    def __init__(self, prefix=None, params=None, **kw):
        object.__setattr__(self, "_children", collections.OrderedDict())
        object.__setattr__(self, "_reg_params", collections.OrderedDict())
        object.__setattr__(self, "params", ParamDict())
        object.__setattr__(self, "_prefix", prefix or "")
        object.__setattr__(self, "_extra_kw", kw)

    @property
    def prefix(self):
        return self._prefix

    def name_scope(self):
        return contextlib.nullcontext()

    def __setattr__(self, name, value):
        if isinstance(value, Block):
            self._children[name] = value
        elif isinstance(value, Param):
            self._reg_params[name] = value
        object.__setattr__(self, name, value)

    def register_child(self, block, name=None):
        self._children[str(len(self._children)) if name is None else name] = block

    def collect_params(self, select=None):
        out = ParamDict()
        for path, b in walk(self):
            for n, p in b.params.items():
                out[(path + "." if path else "") + n] = p
        return out

    def hybridize(self, *a, **k):
        pass

    def initialize(self, *a, **k):
        pass

    def _clear_cached_op(self):
        pass

    def __call__(self, *args):
        TRACE.stack.append(getattr(self, "_path", type(self).__name__))
        try:
            return self.forward(*args)
        finally:
            TRACE.stack.pop()

    def forward(self, *args):
        raise NotImplementedError(type(self).__name__)


class HybridBlock(Block):
    def forward(self, x, *args):
        params = {n: T(None, "const:" + n) for n in self._reg_params}
        return self.hybrid_forward(F, x, *args, **params)


class HybridSequential(HybridBlock):
    def add(self, *blocks):
        for b in blocks:
            self.register_child(b)

    def hybrid_forward(self, F, x):
        for b in self._children.values():
            x = b(x)
        return x

    def __getitem__(self, key):
        layers = list(self._children.values())[key]
        if isinstance(layers, list):
            net = type(self)(prefix=self._prefix)  # gluon: a slice is a NEW sequential, children re-indexed from 0
            net.add(*layers)
            return net
        return layers

    def __len__(self):
        return len(self._children)

    def __iter__(self):
        return iter(self._children.values())


Sequential = HybridSequential


def _pair(v):
    return [int(v), int(v)] if isinstance(v, int) else [int(u) for u in v]


class Conv2D(HybridBlock):
    def __init__(self, channels, kernel_size, strides=(1, 1), padding=(0, 0), dilation=(1, 1), groups=1, layout="NCHW",
                 activation=None, use_bias=True, weight_initializer=None, bias_initializer="zeros", in_channels=0, **kw):
        super().__init__(**kw)
        self.spec = {"cout": int(channels), "kernel": _pair(kernel_size), "stride": _pair(strides), "pad": _pair(padding),
                     "dilation": _pair(dilation), "groups": int(groups), "layout": layout, "activation": activation,
                     "use_bias": bool(use_bias), "in_channels_declared": int(in_channels)}
        self.weight = self.params.get("weight")
        if use_bias:
            self.bias = self.params.get("bias")
        if in_channels:
            self._set_shapes(int(in_channels))

    def _set_shapes(self, cin):
        self.weight.shape = (self.spec["cout"], cin) + tuple(self.spec["kernel"])
        if self.spec["use_bias"]:
            self.bias.shape = (self.spec["cout"],)

    def forward(self, x):
        if self.weight.shape is None and x.c is not None:
            self._set_shapes(x.c)                       # gluon's deferred shape inference, at the first forward
        rec = dict(self.spec, cin=x.c, name=getattr(self, "_path", "?"), exec_index=len(TRACE.convs))
        TRACE.convs.append(rec)
        return T(self.spec["cout"], self._path)


class _Norm(HybridBlock):
    def __init__(self, *a, **kw):
        super().__init__()
        self.ctor_args, self.ctor_kwargs = [_plain(v) for v in a], {k: _plain(v) for k, v in sorted(kw.items())}
        for n in ("gamma", "beta", "running_mean", "running_var"):
            setattr(self, n, self.params.get(n))

    def forward(self, x):
        TRACE.norms.append({"name": self._path, "class": type(self).__name__, "channels": x.c, "kwargs": self.ctor_kwargs,
                            "args": self.ctor_args})
        return T(x.c, x.tag)


class BatchNorm(_Norm):
    pass


class SyncBatchNorm(_Norm):
    pass


class Dense(HybridBlock):
    def __init__(self, units, **kw):
        super().__init__()
        self.units = int(units)
        self.weight = self.params.get("weight")
        self.bias = self.params.get("bias")

    def forward(self, x):
        return T(self.units, x.tag)


class LeakyReLU(HybridBlock):
    def __init__(self, alpha, **kw):
        super().__init__(**kw)
        self.alpha = float(alpha)

    def forward(self, x):
        TRACE.acts.append({"name": self._path, "class": "LeakyReLU", "alpha": self.alpha})
        return T(x.c, x.tag)


def walk(block, path=""):
    yield path, block
    for name, child in block._children.items():
        yield from walk(child, (path + "." if path else "") + name)


def name_paths(root):
    for path, b in walk(root):
        object.__setattr__(b, "_path", path)


# ------------------------------------------------------------------------------------------------------------------
# permissive module trees for `mxnet` and `gluoncv`: anything not given above is a subclassable, callable placeholder
# ------------------------------------------------------------------------------------------------------------------
class Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Anything()


class _Autograd(types.ModuleType):
    training = False
    recording = False

    def is_training(self):
        return self.training

    def is_recording(self):
        return self.recording

    def pause(self):
        return contextlib.nullcontext()


class BBoxBatchIOU(HybridBlock):
    def __init__(self, *a, **kw):
        super().__init__()
        TRACE.misc.append({"ctor": "BBoxBatchIOU", "args": [_plain(v) for v in a], "kwargs": {k: _plain(v) for k, v in kw.items()}})

    def forward(self, a, b):
        TRACE.op("BBoxBatchIOU()", (), {"a": a.tag, "b": b.tag})
        return T(None, "batch_iou(%s, %s)" % (a.tag, b.tag))


class YOLOV3Loss(HybridBlock):
    def __init__(self, *a, **kw):
        super().__init__()
        TRACE.misc.append({"ctor": "YOLOV3Loss", "args": [_plain(v) for v in a], "kwargs": {k: _plain(v) for k, v in kw.items()}})

    def forward(self, *args):
        TRACE.misc.append({"call": "YOLOV3Loss", "arg_tags": [t.tag for t in args]})
        return tuple(T(None, n) for n in ("obj_loss", "center_loss", "scale_loss", "cls_loss"))


KNOWN = {
    "gluoncv.nn.bbox": {"BBoxBatchIOU": BBoxBatchIOU},
    "gluoncv.loss": {"YOLOV3Loss": YOLOV3Loss},
    "mxnet.gluon": {"Block": Block, "HybridBlock": HybridBlock},
    "mxnet.gluon.nn": {"Block": Block, "HybridBlock": HybridBlock, "HybridSequential": HybridSequential, "Sequential": Sequential,
                       "Conv2D": Conv2D, "BatchNorm": BatchNorm, "LeakyReLU": LeakyReLU, "Dense": Dense},
    "mxnet.gluon.contrib.nn": {"SyncBatchNorm": SyncBatchNorm},
}


class _Permissive(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        cls = type(name, (Anything,), {"__module__": self.__name__})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("mxnet", "gluoncv")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name == "mxnet.autograd":
            m = _Autograd(spec.name)
        else:
            m = _Permissive(spec.name)
        m.__path__ = []
        for k, v in KNOWN.get(spec.name, {}).items():
            setattr(m, k, v)
        return m

    def exec_module(self, module):
        parent, _, leaf = module.__name__.rpartition(".")
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], leaf, module)


def install():
    sys.meta_path.insert(0, _Finder())
    import mxnet  # noqa: F401
    import mxnet.gluon  # noqa: F401
    import mxnet.gluon.nn  # noqa: F401
    import mxnet.gluon.contrib.nn  # noqa: F401
    import mxnet.autograd  # noqa: F401
    sys.path.insert(0, REF)


# ------------------------------------------------------------------------------------------------------------------
def record_freeze_base():
    """yolo3_darknet53(..., freeze_base=True) (wrappers.py:55-57): the parameters whose grad_req the reference sets to 'null'"""
    from models.definitions.yolo.wrappers import yolo3_darknet53
    net = yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False, k=1, freeze_base=True)
    name_paths(net)
    TRACE.reset()
    net(T(3, "input"))
    ps = net.collect_params()
    skip = lambda k: k.rsplit(".", 1)[-1].startswith(("anchor_", "offset_"))  # noqa: E731
    return {"grad_req_null": sorted(k for k, p in ps.items() if p.grad_req == "null" and not skip(k)),
            "grad_req_write": sorted(k for k, p in ps.items() if p.grad_req != "null" and not skip(k))}


def record_backbone_names():
    """pretrained_base: `get_darknet` loads the ImageNet classifier's file into a Darknet3D whose parameters are named
    `features.<n>...` / `output.*` (three_darknet.py:262-264, `ignore_extra`), and wrappers.py:58 re-homes its cells under
    `stages.<s>.<j>`.  The SAME parameter objects appear under both names: the mapping, by object identity."""
    import models.definitions.yolo.wrappers as wr
    captured = []
    orig = wr.get_darknet

    def capture(*a, **k):
        net = orig(*a, **k)
        captured.append(net)
        return net
    wr.get_darknet = capture
    try:
        net = wr.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False, k=1)
    finally:
        wr.get_darknet = orig
    dark = captured[0]
    by_id = {id(p): k for k, p in dark.collect_params().items()}
    mapping = {by_id[id(p)]: k for k, p in net.collect_params().items() if id(p) in by_id}
    return {"classifier_to_detector": dict(sorted(mapping.items(), key=lambda kv: (int(kv[0].split(".")[1]), kv[0]))),
            "classifier_only": sorted(k for k in by_id.values() if k not in mapping),
            "features_children": len(dark.features)}


def record(num_class, sync):
    """build yolo3_darknet53 as train_yolov3.py:347-360 / detect_yolo3.py:873-880 do and run one inference-mode forward"""
    import numpy as np
    from models.definitions.yolo.wrappers import yolo3_darknet53
    import mxnet
    classes = ["c%d" % i for i in range(num_class)]
    kw = dict(pretrained_base=False, k=1)
    if sync:
        kw.update(norm_layer=mxnet.gluon.contrib.nn.SyncBatchNorm, norm_kwargs={"num_devices": 8})
    net = yolo3_darknet53(classes, **kw)
    name_paths(net)
    TRACE.reset()
    mxnet.autograd.training = False
    out = net(T(3, "input"))
    assert isinstance(out, tuple) and len(out) == 3, "inference mode returns (ids, scores, bboxes)"
    convs, norms, acts = TRACE.convs, {n["name"]: n for n in TRACE.norms}, {a["name"]: a for a in TRACE.acts}
    rows = []
    for c in convs:
        cell = c["name"].rsplit(".", 1)[0] if c["name"].endswith(".0") else None  # _conv2d: [Conv2D, norm, LeakyReLU]
        norm = norms.get(cell + ".1") if cell else None
        act = acts.get(cell + ".2") if cell else None
        rows.append({"exec_index": c["exec_index"], "name": c["name"], "cin": c["cin"], "cout": c["cout"], "kernel": c["kernel"],
                     "stride": c["stride"], "pad": c["pad"], "use_bias": c["use_bias"], "groups": c["groups"],
                     "norm": None if norm is None else {"name": norm["name"], "class": norm["class"], "kwargs": norm["kwargs"]},
                     "act": None if act is None else {"class": act["class"], "alpha": act["alpha"]}})
    heads = []
    for i, o in enumerate(net.yolo_outputs):
        heads.append({"index": i, "anchors": np.asarray(o.anchors.value).reshape(-1).tolist(), "stride": int(o._stride),
                      "num_pred": int(o._num_pred), "num_anchors": int(o._num_anchors),
                      "offsets_shape": list(np.asarray(o.offsets.value).shape),
                      "offsets_first_xy": np.asarray(o.offsets.value)[0, 0, 1, 2].tolist()})  # (y = 1, x = 2) -> [2, 1]
    # ---- training: the recorded call (train_yolov3.py:623-626) and the train-mode non-recording call (transforms.py:192)
    infer_ops_top, infer_ops_out0, infer_concats = TRACE.ops, None, TRACE.concats
    infer_ops_out0 = [o for o in TRACE.ops if o["in"] == "yolo_outputs.0"]
    infer_ops_top = [o for o in TRACE.ops if o["in"] == ""]
    TRACE.reset()
    mxnet.autograd.training, mxnet.autograd.recording = True, True
    names = ("gt_boxes", "obj_t", "centers_t", "scales_t", "weights_t", "clas_t")
    losses = net(T(3, "input"), *[T(None, n) for n in names])
    loss_call = [m for m in TRACE.misc if m.get("call") == "YOLOV3Loss"]
    train = {
        "returns": [t.tag for t in losses],
        "loss_call_arg_tags": loss_call[0]["arg_tags"],
        "ignore_iou_thresh": net._target_generator._dynamic_target._ignore_iou_thresh,
        "label_smooth_default": net._target_generator._label_smooth,
        "ctor_records": [m for m in TRACE.misc if "ctor" in m][-2:],
        "ops_target_merger": [o for o in TRACE.ops if o["in"] == "_target_generator"],
        "ops_dynamic_target": [o for o in TRACE.ops if o["in"] == "_target_generator._dynamic_target"],
        "ops_output_layer_0_train": [o for o in TRACE.ops if o["in"] == "yolo_outputs.0"],
    }
    TRACE.reset()
    mxnet.autograd.training, mxnet.autograd.recording = True, False
    tup = net(T(3, "input"))
    train["train_mode_tuple"] = [([u.tag for u in t] if isinstance(t, list) else t.tag) for t in tup]
    mxnet.autograd.training = mxnet.autograd.recording = False

    weights = sum(r["cout"] * r["cin"] * r["kernel"][0] * r["kernel"][1] // r["groups"] for r in rows)
    biases = sum(r["cout"] for r in rows if r["use_bias"])
    bn_ch = sum(n["channels"] for n in TRACE.norms)
    params = collections.OrderedDict((k, v) for k, v in net.collect_params().items())
    return {
        "num_class": num_class,
        "norm_layer": "SyncBatchNorm(num_devices=8)" if sync else "BatchNorm (default)",
        "convs": rows,
        "stage_lengths": [len(s) for s in net.stages],
        "heads": heads,
        "concats_axis1": [c for c in infer_concats if c["dim"] == 1 and c["in"] == ""],
        "cells_with_the_passed_norm_layer": sorted(n["name"] for n in TRACE.norms if n["class"] == "SyncBatchNorm"),
        "norm_classes": collections.Counter(n["class"] for n in TRACE.norms),
        "trainable_parameters": weights + biases + 2 * bn_ch,
        "running_statistics": 2 * bn_ch,
        "parameter_names": [k for k in params if not k.rsplit(".", 1)[-1].startswith(("anchor_", "offset_"))],
        "nms_defaults": {"nms_thresh": net.nms_thresh, "nms_topk": net.nms_topk, "post_nms": net.post_nms},
        "ops_top_level": infer_ops_top,
        "ops_output_layer_0": infer_ops_out0,
        "train": train,
    }


def record_reset_class():
    """The reference's own reset_class (yolo3.py:1230-1302 + YOLOOutputV3.reset_class :76-129) EXECUTED on marker values:
    which row of the new prediction conv holds which old row (or stays freshly initialised), for the forms of
    `reuse_weights` its docstring lists.  Pure data movement: no operator arithmetic is involved."""
    import warnings
    import numpy as np
    from models.definitions.yolo.wrappers import yolo3_darknet53
    voc = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog", "horse",
           "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]

    def fresh_net():
        net = yolo3_darknet53(list(voc), pretrained_base=False, k=1)
        name_paths(net)
        TRACE.reset()
        net(T(3, "input"))                              # every Conv2D learns its input channels
        for i, o in enumerate(net.yolo_outputs):
            w, b = o.prediction.weight, o.prediction.bias
            rows = np.arange(w.shape[0], dtype=np.float64)
            w.set_data(np.broadcast_to((1000.0 * (i + 1) + rows).reshape(-1, 1, 1, 1), w.shape))   # marker = 1000 (head + 1) + row
            b.set_data(1000.0 * (i + 1) + rows + 0.5)
        return net

    def rows_of(net):
        out = []
        for i, o in enumerate(net.yolo_outputs):
            w, b = o.prediction.weight.data(), o.prediction.bias.data()
            assert w.shape[0] == 3 * (5 + len(net.classes)) and b.shape == (w.shape[0],)
            src_w = [(-1 if np.isnan(v) else int(round(v - 1000.0 * (i + 1)))) for v in w[:, 0, 0, 0]]
            src_b = [(-1 if np.isnan(v) else int(round(v - 0.5 - 1000.0 * (i + 1)))) for v in b]
            assert src_w == src_b and all(np.isnan(r).all() or (r == r.flat[0]).all() for r in w.reshape(w.shape[0], -1))
            out.append(src_w)
        return out

    cases = [
        ("names", ["person", "car", "dog"], {"person": "person", "car": "car", "dog": "dog"}),
        ("indices", ["person"], {0: 14}),
        ("mixed", ["cat", "person", "zebra"], {"person": 14, 0: "cat"}),
        ("list", ["bird", "cat", "zebra"], ["cat", "bird", "unicorn"]),
        ("none", ["a", "b"], None),
        ("empty_dict", ["a", "b"], {}),
        ("more_classes", voc + ["zebra", "unicorn"], voc),
        ("same_classes_permuted", voc[::-1], voc),
    ]
    out = {"marker": "new prediction row r of head i holds old row src[i][r] (weights and bias alike), -1 = freshly initialised",
           "old_classes": voc, "cases": []}
    for label, classes, reuse in cases:
        net = fresh_net()
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            net.reset_class(list(classes), reuse_weights=(dict(reuse) if isinstance(reuse, dict) else (list(reuse) if reuse is not None else None)))
        out["cases"].append({"label": label, "classes": list(classes),
                             "reuse_weights": ([[k, v] for k, v in reuse.items()] if isinstance(reuse, dict) else reuse),
                             "src_rows": rows_of(net), "warnings": [str(w.message) for w in wl],
                             "classes_after": list(net.classes),
                             "merger_num_class": net._target_generator._num_class})
    errors = []
    for classes, reuse in ((["person"], {"person": "unicorn"}), (["person"], {"unicorn": "person"}), (["person"], {0: 20}),
                           (["person"], {1: 14}), (["person"], {0: -1})):
        net = fresh_net()
        try:
            net.reset_class(list(classes), reuse_weights=dict(reuse))
            errors.append({"classes": classes, "reuse_weights": [[k, v] for k, v in reuse.items()], "raises": None})
        except ValueError as e:
            errors.append({"classes": classes, "reuse_weights": [[k, v] for k, v in reuse.items()], "raises": "ValueError", "message": str(e)})
    out["errors"] = errors
    return out


def main():
    install()
    doc = {
        "what": "STRUCTURE ONLY: recorded from the reference's own constructors and hybrid_forward methods under a "
                "recording stand-in for mxnet/gluoncv (tests/golden/make_graph_structure.py). No operator arithmetic is "
                "reproduced; this does not pin the oracle (DESIGN.md section 2).",
        "reference_files": ["models/definitions/yolo/wrappers.py", "models/definitions/darknet/three_darknet.py",
                            "models/definitions/darknet/darknet.py", "models/definitions/layers.py",
                            "models/definitions/yolo/yolo3.py"],
        "voc20": record(20, sync=False),
        "vid30": record(30, sync=False),
        "voc20_syncbn8": record(20, sync=True),
        "reset_class": record_reset_class(),
        "freeze_base": record_freeze_base(),
        "backbone_names": record_backbone_names(),
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=False)
        f.write("\n")
    for k in ("voc20", "vid30", "voc20_syncbn8"):
        d = doc[k]
        print(k, len(d["convs"]), "convs,", d["trainable_parameters"], "trainable,", dict(d["norm_classes"]),
              "stages", d["stage_lengths"])
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
