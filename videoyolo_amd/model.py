"""The model object: ``yolo3_darknet53(classes, ...)`` with the surface the reference's drivers use.

Host-side mirror (Python, like the reference) of the Gluon HybridBlock that
``models/definitions/yolo/wrappers.py:9-110`` returns (``YOLOV3T`` at k=1,
``models/definitions/yolo/yolo3.py:915-1302``), over the C-ABI of ``include/vyolo.h``.
All tensor work happens in libvyolo.so (hand-written HIP for gfx950); torch is used only for device
memory, streams and (in ``videoyolo_amd.parallel``) torch.distributed.

Call sites this surface serves (paths relative to /root/reference):
  construction      train_yolov3.py:350-360,384-392   detect_yolo3.py:873-880
  initialize        train_yolov3.py:428               detect_yolo3.py:885
  load/save params  train_yolov3.py:293-303,323-327   detect_yolo3.py:890
  collect_params    train_yolov3.py:494-497,527       detect_yolo3.py:199   wrappers.py:55-57
  hybridize         train_yolov3.py:441,586
  set_nms           train_yolov3.py:438               detect_yolo3.py:200
  reset_class       train_yolov3.py:728-729
  __call__          detect_yolo3.py:222 (inference)   train_yolov3.py:625 (recording)
"""
import copy
import ctypes
import os
import re
import warnings
from collections import OrderedDict

import numpy as np

from . import _lib, autograd, init as _init

_KINDS = ["weight", "gamma", "beta", "running_mean", "running_var", "bias"]


class BatchNorm:
    """Marker for ``norm_layer=BatchNorm`` (mxnet.gluon.nn.BatchNorm)."""


class SyncBatchNorm:
    """``norm_layer=gluon.contrib.nn.SyncBatchNorm`` with ``norm_kwargs={'num_devices': n}`` (train_yolov3.py:350-354).
    Passing this class to ``yolo3_darknet53`` makes the layers that receive ``norm_layer`` in the reference — the stem
    and the five stride-2 convs of the backbone (three_darknet.py:163-181; the residual blocks hard-code BatchNorm,
    darknet.py:89-91, and wrappers.py:101-103 does not forward it to the heads) — normalise with the statistics of
    the GLOBAL batch: their [2][C] sums are all-reduced over the process group, forward and backward.  With one
    process per GPU ``num_devices`` must equal the world size (``YOLOV3._ensure_sync_bn``)."""


class Parameter:
    """One row of ``collect_params()``: gluon.Parameter's attributes the drivers touch."""

    def __init__(self, net, index, info):
        self._net = net
        self.index = index
        self.name = info.name.decode()
        self.kind = _KINDS[info.kind]
        self.shape = tuple(info.shape[i] for i in range(info.ndim))
        self.size = int(info.size)
        self.offset = int(info.offset)
        self.trainable = bool(info.trainable)
        self.backbone = bool(info.backbone)
        self.grad_req = "write" if self.trainable else "null"
        self.wd_mult = 1.0
        self.lr_mult = 1.0

    def data(self, ctx=None):
        """Value in the reference layout (OIHW for conv weights), as numpy."""
        return self._net._get_param(self.index)

    def set_data(self, value):
        value = np.ascontiguousarray(value, dtype=np.float32)
        if value.shape != self.shape:
            raise ValueError("%s: shape %s does not match %s" % (self.name, value.shape, self.shape))
        self._net._set_param(self.index, value)

    def __repr__(self):
        return "Parameter %s (shape=%s, dtype=float32)" % (self.name, self.shape)


class ParameterDict(OrderedDict):
    def __init__(self, net, items=()):
        super().__init__(items)
        self._net = net

    def reset_ctx(self, ctx):
        self._net.reset_ctx(ctx)

    def setattr(self, name, value):
        for p in self.values():
            setattr(p, name, value)

    def zero_grad(self):
        """Gradients are overwritten (grad_req='write'), never accumulated, by every backward pass; the flat
        gradient buffer is cleared anyway so that a reader sees zeros, as after gluon's zero_grad."""
        if self._net._grads is not None:
            self._net._grads.zero_()


class _TargetGeneratorState:
    """Holder for ``net._target_generator._label_smooth`` (train_yolov3.py:500)."""

    def __init__(self, num_class, ignore_iou_thresh):
        self._num_class = num_class
        self._ignore_iou_thresh = ignore_iou_thresh
        self._label_smooth = False


def _torch():
    import torch
    return torch


class YOLOV3(object):
    """yolo3_darknet53 detector bound to one device / stream."""

    def __init__(self, classes, nms_thresh=0.45, nms_topk=400, post_nms=100, pos_iou_thresh=1.0,
                 ignore_iou_thresh=0.7, norm_layer=BatchNorm, norm_kwargs=None, alloc_size=(128, 128)):
        if pos_iou_thresh < 1:
            raise NotImplementedError(
                "pos_iou_thresh({}) < 1.0 is not implemented!".format(pos_iou_thresh))  # yolo3.py:992
        self._classes = list(classes)
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._lib.vy_net_create(len(self._classes), ctypes.byref(h)))
        self._h = h
        self.nms_thresh, self.nms_topk, self.post_nms = nms_thresh, nms_topk, post_nms
        self._ignore_iou_thresh = ignore_iou_thresh
        self._pos_iou_thresh = pos_iou_thresh
        self._norm_layer, self._norm_kwargs = norm_layer, norm_kwargs
        self._alloc_size = alloc_size
        self._target_generator = _TargetGeneratorState(len(self._classes), ignore_iou_thresh)
        self._params = ParameterDict(self)
        n = self._lib.vy_net_num_params(self._h)
        for i in range(n):
            info = _lib.ParamInfo()
            _lib.check(self._lib.vy_net_param_info(self._h, i, ctypes.byref(info)))
            p = Parameter(self, i, info)
            self._params[p.name] = p
        self._host = {}          # name -> numpy (reference layout) while not on a device
        self._dev_params = None  # torch uint8 tensor holding the device parameter buffer
        self._ws = None          # torch uint8 workspace
        self._plan = None        # (B, H, W, train?) the workspace is planned for
        self._grads = None       # torch float32 flat gradient buffer (device layout, training only)
        self._mom = None         # torch float32 flat SGD momentum buffer
        self._train_x = None     # image batch of the recorded forward (stem weight gradient)
        self._cb_keep = []       # ctypes callbacks kept alive
        self._sync_hook = None   # parallel.SyncBatchNormHook once installed (norm_layer=SyncBatchNorm or by hand)
        self._sync_bn_checked = False
        self._replicas_synced = False  # parameters broadcast from rank 0 since they were last written (parallel.sync_replicas)
        self._opts_sent = {}     # index -> (lr_mult, wd_mult, enabled) as last handed to the library
        self._device = None
        self._hybrid = False
        self._graphs = {}
        self._use_graphs = os.environ.get("VY_HIP_GRAPHS", "1") != "0"
        # batches at least this large run as two half-batches on two streams (two hardware queues);
        # 0 (default) disables.  Round 1: 838 vs 828 frames/s at 608x608 batch 64 (+1.2 %, +2.6 % at 416x416), the
        # second stream filling partly filled rounds.  Round 3, with those launches running as stream-K: 972 vs 984
        # (-1.2 %), 1990 vs 2049 at 416x416 (tools/archive/ab_two_stream.sh) — the holes are gone, the halves only cost
        self.two_stream_batch = int(os.environ.get("VY_TWO_STREAM_BATCH", "0"))
        self._twin = None
        _lib.check(self._lib.vy_net_set_nms(self._h, nms_thresh, nms_topk, post_nms))

    def __del__(self):
        try:
            if getattr(self, "_twin", None):
                self._lib.vy_net_destroy(self._twin["h"])
                if self._twin.get("raw_stream"):
                    self._lib.vy_stream_destroy(self._twin["raw_stream"])
                self._twin = None
            if getattr(self, "_h", None):
                self._lib.vy_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ properties
    @property
    def classes(self):
        """yolo3.py:1066-1074"""
        return self._classes

    @property
    def num_class(self):
        # The reference reads self._num_class, which is never assigned (yolo3.py:1057-1064), so
        # the property raises AttributeError there; here it returns the obvious value.
        return len(self._classes)

    def param_table(self):
        return [(p.name, p.shape) for p in self._params.values()]

    # ------------------------------------------------------------------ parameters
    def collect_params(self, select=None):
        if select is None:
            return self._params
        pat = re.compile(select)
        return ParameterDict(self, [(k, v) for k, v in self._params.items() if pat.match(k)])

    def initialize(self, init=None, ctx=None, force_reinit=False, seed=None, **kwargs):
        """``net.initialize()``: 'uniform' (gluon default) or 'synthetic' (videoyolo_amd.init)."""
        if self._host and not force_reinit and self._all_set():
            warnings.warn("parameters already initialized; use force_reinit=True")
        else:
            table = self.param_table()
            if init in (None, "uniform"):
                vals = _init.uniform_params(table, seed)
            elif init == "synthetic":
                vals = _init.synthetic_params(table, 233 if seed is None else seed, **kwargs)
            elif callable(init):
                vals = init(table)
            else:
                raise ValueError("unknown initializer %r" % (init,))
            for k, v in vals.items():
                self._params[k].set_data(v)
        if ctx is not None:
            self.reset_ctx(ctx)

    def _all_set(self):
        return self._dev_params is not None or len(self._host) == len(self._params)

    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _get_param(self, i):
        p = list(self._params.values())[i] if not isinstance(i, Parameter) else i
        if self._dev_params is None:
            if p.name not in self._host:
                raise RuntimeError("Parameter %s has not been initialized" % p.name)
            return self._host[p.name].copy()
        out = np.empty(p.shape, np.float32)
        torch = _torch()
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_param_get(self._h, p.index, out.ctypes.data_as(ctypes.c_void_p),
                                                  self._stream()))
        return out

    def _set_param(self, i, value):
        p = list(self._params.values())[i]
        self._replicas_synced = False
        if self._dev_params is None:
            self._host[p.name] = value.copy()
            return
        torch = _torch()
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_param_set(self._h, p.index, value.ctypes.data_as(ctypes.c_void_p),
                                                  self._stream()))
        self._params_written()

    def _params_written(self):
        """The device parameter buffer changed: the pre-split weight images of conv mode 'split_bf16x3' (of this
        handle and of the two-stream twin, which shares the buffer) are rebuilt by the next inference forward."""
        if getattr(self, "_conv_mode", "exact") == "exact":
            return
        _lib.check(self._lib.vy_net_invalidate_split_weights(self._h))
        tw = getattr(self, "_twin", None)
        if tw is not None:
            _lib.check(self._lib.vy_net_invalidate_split_weights(tw["h"]))
        self._graphs = {}  # a captured forward replays the old images

    def reset_ctx(self, ctx):
        """Move the parameters to a device (``net.collect_params().reset_ctx(ctx)``).  ctx: a
        torch.device / 'cuda:N' / int, or a one-element list of those (one net per device)."""
        torch = _torch()
        if isinstance(ctx, (list, tuple)):
            if len(ctx) != 1:
                raise ValueError("one YOLOV3 object drives one device; use videoyolo_amd.parallel for N devices")
            ctx = ctx[0]
        dev = torch.device("cuda", ctx) if isinstance(ctx, int) else torch.device(ctx)
        if dev.type != "cuda":
            raise RuntimeError("videoyolo_amd runs on an MI355X (cuda/hip device); there is no CPU path")
        if self._dev_params is not None and dev == self._device:
            return
        vals = {p.name: self._get_param(p.index) for p in self._params.values()} if self._all_set() else None
        if vals is None:
            raise RuntimeError("initialize() or load_parameters() before reset_ctx()")
        self._device = dev
        nbytes = self._lib.vy_net_param_bytes(self._h)
        self._dev_params = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(self._lib.vy_net_bind_params(self._h, ctypes.c_void_p(self._dev_params.data_ptr())))
        self._host = {}
        self._ws, self._plan = None, None
        self._grads = self._mom = None
        for i, p in enumerate(self._params.values()):
            self._set_param(i, vals[p.name])

    _CONST_RE = re.compile(r"^yolo_outputs\.([0-2])\.(anchors|offsets)$")
    _ANCHORS = ((116, 90, 156, 198, 373, 326), (30, 61, 62, 45, 59, 119), (10, 13, 16, 30, 33, 23))

    def constants(self):
        """The gluon Constants of the three YOLOOutputV3 blocks (yolo3.py:64-74).  Constants are Parameters,
        so ``net.save_parameters`` of the reference writes them and its ``load_parameters`` expects them:
        ``yolo_outputs.i.anchors`` (1,1,3,2) — ``anchors[::-1][i]`` of wrappers.py:80-84 — and
        ``yolo_outputs.i.offsets`` (1,1,alloc_h,alloc_w,2), the (x, y) meshgrid.  Here they are compiled into
        the library (net_internal.h kAnchors, the decode kernels' cell coordinates); this is their file image."""
        out = OrderedDict()
        ah, aw = self._alloc_size
        gx, gy = np.meshgrid(np.arange(aw), np.arange(ah))
        offsets = np.stack([gx, gy], -1).astype(np.float32)[None, None]
        for i in range(3):
            out["yolo_outputs.%d.anchors" % i] = np.array(self._ANCHORS[i], np.float32).reshape(1, 1, 3, 2)
            out["yolo_outputs.%d.offsets" % i] = offsets.copy()
        return out

    def save_parameters(self, filename, format=None):
        """Gluon structural names, reference layouts, including the anchors / offsets Constants — the key set
        of the reference's own ``net.save_parameters`` file.  Container: numpy .npz by default; mxnet's
        NDArray-dict layout with ``format='mxnet'`` (videoyolo_amd.mxparams: restated from memory,
        unverified against a real mxnet build — see INTEGRATION.md)."""
        arrays = OrderedDict()
        consts = self.constants()
        for p in self._params.values():
            m = re.match(r"^(yolo_outputs\.[0-2])\.prediction\.weight$", p.name)
            if m:  # gluon order: a block's own parameters (anchors, offsets) before its children's
                arrays[m.group(1) + ".anchors"] = consts[m.group(1) + ".anchors"]
                arrays[m.group(1) + ".offsets"] = consts[m.group(1) + ".offsets"]
            arrays[p.name] = self._get_param(p.index)
        if format == "mxnet":
            from . import mxparams
            mxparams.save(filename, arrays)
            return
        with open(filename, "wb") as f:
            np.savez(f, **arrays)

    def load_parameters(self, filename, ctx=None, allow_missing=False, ignore_extra=False):
        """Accepts both containers (sniffed from the first bytes: zip = .npz, 0x112 = mxnet list)."""
        with open(filename, "rb") as f:
            head = f.read(8)
        if head[:2] == b"PK":
            with np.load(filename) as z:
                loaded = {k: z[k] for k in z.files}
        else:
            from . import mxparams
            loaded = mxparams.load(filename)
        self.set_parameters(loaded, allow_missing=allow_missing, ignore_extra=ignore_extra)
        if ctx is not None:
            self.reset_ctx(ctx)

    def load_darknet53_backbone(self, filename):
        """``pretrained_base=True`` (three_darknet.py:262-264): load an ImageNet darknet53 checkpoint (gluoncv's
        ``darknet53-<hash>.params``, mxnet NDArray-dict layout, or an .npz with the same names) into the 52 backbone
        cells; the heads keep whatever ``initialize()`` gives them (wrappers.py builds them fresh)."""
        with open(filename, "rb") as f:
            head = f.read(2)
        if head == b"PK":
            with np.load(filename) as z:
                loaded = {k: z[k] for k in z.files}
        else:
            from . import mxparams
            loaded = mxparams.load(filename)
        mapped = darknet53_to_stage_names(loaded)
        want = [k for k, p in self._params.items() if p.backbone]
        missing = [k for k in want if k not in mapped]
        if missing:
            raise AssertionError("Parameter '%s' is missing in %s" % (missing[0], filename))
        self.set_parameters({k: mapped[k] for k in want}, allow_missing=True)

    def set_parameters(self, arrays, allow_missing=False, ignore_extra=False):
        """Load a {structural name: array} dict (reference layouts).  The anchors / offsets Constants of a
        reference checkpoint are checked against the built-in ones (a file made with other anchors cannot be
        run by this fixed-anchor path) and otherwise consumed silently; they may also be absent."""
        consts = None
        for k, v in arrays.items():
            m = self._CONST_RE.match(k)
            if not m:
                continue
            consts = consts or self.constants()
            v = np.asarray(v, np.float32)
            if m.group(2) == "anchors":
                if v.size != 6 or not np.array_equal(v.reshape(-1), consts[k].reshape(-1)):
                    raise ValueError("%s = %s differs from the yolo3_darknet53 anchors %s" %
                                     (k, v.reshape(-1).tolist(), consts[k].reshape(-1).tolist()))
            else:
                g = v.reshape(v.shape[-3:]) if v.ndim >= 3 else v
                want = consts[k][0, 0]
                hh, ww = min(g.shape[0], want.shape[0]), min(g.shape[1], want.shape[1])
                if g.ndim != 3 or g.shape[-1] != 2 or not np.array_equal(g[:hh, :ww], want[:hh, :ww]):
                    raise ValueError("%s is not the (x, y) cell-offset meshgrid of yolo3.py:67-74" % k)
        missing = [k for k in self._params if k not in arrays]
        extra = [k for k in arrays if k not in self._params and not self._CONST_RE.match(k)]
        if missing and not allow_missing:
            raise AssertionError("Parameter '%s' is missing in the file" % missing[0])
        if extra and not ignore_extra:
            raise AssertionError("Parameter '%s' loaded from the file is not present in the net" % extra[0])
        for k, v in arrays.items():
            if k in self._params:
                self._params[k].set_data(np.asarray(v, np.float32))

    # ------------------------------------------------------------------ configuration
    def hybridize(self, active=True, **kwargs):
        """``net.hybridize()`` (train_yolov3.py:441,586).  For inference the launch sequence of one
        forward (85 kernels + 2 memsets) is captured into a HIP graph per input shape and replayed:
        small batches are launch-bound, a replay costs one submission."""
        self._hybrid = bool(active)
        self._graphs = {}

    def _graph_forward(self, x, rows):
        """Capture-once / replay path of detect() when hybridized."""
        torch = _torch()
        key = (tuple(x.shape), rows, self.nms_thresh, self.nms_topk, self.post_nms)
        g = self._graphs.get(key)
        if g is None:
            b = x.shape[0]
            st = dict(x=torch.empty_like(x),
                      ids=torch.empty((b, rows, 1), dtype=torch.float32, device=self._device),
                      scores=torch.empty((b, rows, 1), dtype=torch.float32, device=self._device),
                      bboxes=torch.empty((b, rows, 4), dtype=torch.float32, device=self._device),
                      keep=torch.empty((b, rows), dtype=torch.int32, device=self._device))

            def launch():
                _lib.check(self._lib.vy_net_forward_infer(
                    self._h, ctypes.c_void_p(st["x"].data_ptr()), ctypes.c_void_p(st["ids"].data_ptr()),
                    ctypes.c_void_p(st["scores"].data_ptr()), ctypes.c_void_p(st["bboxes"].data_ptr()),
                    ctypes.c_void_p(st["keep"].data_ptr()), self._stream()))
            st["x"].copy_(x)
            launch()  # eager warm-up: one-time uploads happen outside the capture
            torch.cuda.synchronize(self._device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                launch()
            g = (graph, st)
            self._graphs = {key: g}  # one shape at a time (the workspace is planned for one shape)
        graph, st = g
        st["x"].copy_(x)
        graph.replay()
        return st["ids"].clone(), st["scores"].clone(), st["bboxes"].clone(), st["keep"].clone()

    def keep_activations(self, keep=True):
        """Parity taps: give every cell its own activation plane so that ``read_activation`` works after an inference
        forward.  By default inference planes are recycled by liveness (about a third of the workspace at 608x608
        batch 64), like the intermediates of the reference's hybridized graph; training plans always keep them."""
        _lib.check(self._lib.vy_net_set_keep_activations(self._h, int(bool(keep))))
        self._keep_activations = bool(keep)
        self._plan = None      # the plan changed: size and bind the workspace again at the next call
        self._graphs = {}

    def set_conv_mode(self, mode="exact"):
        """Arithmetic of the inference convolutions (no counterpart in the reference: mxnet picks its conv
        algorithm itself).  ``'exact'`` (default): fp32 fma chains, bit-identical to the CPU checker — the parity
        path.  ``'split_bf16x3'``: opt-in, the 3x3 cells run on the bf16 matrix core with every fp32 operand cut
        exactly into three bf16 numbers (six partial products, fp32 accumulation; include/vyolo.h
        ``vy_net_set_conv_mode``); training uses the exact kernels.  ``'split_bf16x3_train'``: training too — the
        recorded forward, the data gradients and the weight gradients (cout % 128 == 0) on the split kernels."""
        modes = {"exact": _lib.VY_CONV_EXACT_FP32, "split_bf16x3": _lib.VY_CONV_SPLIT_BF16X3,
                 "split_bf16x3_train": _lib.VY_CONV_SPLIT_BF16X3_TRAIN}
        if mode not in modes:
            raise ValueError("conv mode %r: expected one of %s" % (mode, sorted(modes)))
        _lib.check(self._lib.vy_net_set_conv_mode(self._h, modes[mode]))
        self._conv_mode = mode
        self._plan = None      # the plan changed: size and bind the workspace again at the next call
        self._graphs = {}
        tw = getattr(self, "_twin", None)
        if tw is not None:
            # the twin never trains: any split mode is the inference one there, as at its creation (detect_two_streams) —
            # its mode must not depend on whether set_conv_mode or the first two-stream call came first
            twin_mode = _lib.VY_CONV_EXACT_FP32 if mode == "exact" else _lib.VY_CONV_SPLIT_BF16X3
            _lib.check(self._lib.vy_net_set_conv_mode(tw["h"], twin_mode))
            tw["plan"] = None

    def set_nms(self, nms_thresh=0.45, nms_topk=400, post_nms=100):
        """yolo3.py:1208-1228"""
        self.nms_thresh, self.nms_topk, self.post_nms = nms_thresh, nms_topk, post_nms
        _lib.check(self._lib.vy_net_set_nms(self._h, nms_thresh, nms_topk, post_nms))

    def reset_class(self, classes, reuse_weights=None):
        """yolo3.py:1230-1302 + YOLOOutputV3.reset_class yolo3.py:76-129: new class list, fresh
        prediction convs, optionally re-using rows of the old predictors."""
        old_classes = self._classes
        classes = list(classes)

        def resolve(ref, names, which):
            """A class given by name or by index -> index into `names`; same errors as yolo3.py:1270-1286."""
            if isinstance(ref, str):
                if ref not in names:
                    raise ValueError("{} not found in {} class names {}".format(ref, which, names))
                return names.index(ref)
            if ref < 0 or ref >= len(names):
                raise ValueError("Index {} out of bounds for {} class names".format(ref, which))
            return ref

        if isinstance(reuse_weights, dict):      # {new class: old class}, names or indices on either side
            reuse_weights = {resolve(new, classes, "new"): resolve(old, old_classes, "old")
                             for new, old in reuse_weights.items()}
        elif isinstance(reuse_weights, list):    # names present in both lists keep their weights
            both = [n for n in reuse_weights if n in classes and n in old_classes]
            for n in reuse_weights:
                if n not in both:
                    warnings.warn("{} not found in old: {} or new class names: {}".format(n, old_classes, classes))
            reuse_weights = {classes.index(n): old_classes.index(n) for n in both}
        old_vals = {p.name: self._get_param(p.index) for p in self._params.values()}
        device = self._device
        fresh = YOLOV3(classes, self.nms_thresh, self.nms_topk, self.post_nms, self._pos_iou_thresh,
                       self._ignore_iou_thresh, self._norm_layer, self._norm_kwargs, self._alloc_size)
        new_vals = _init.uniform_params(fresh.param_table())  # prediction.initialize(), yolo3.py:110
        old_np, new_np = 5 + len(old_classes), 5 + len(classes)
        for name in new_vals:
            if "prediction" not in name:
                new_vals[name] = old_vals[name]
            elif reuse_weights:
                od, nd = old_vals[name], new_vals[name]
                for k, v in reuse_weights.items():
                    if k >= len(classes) or v >= len(old_classes):
                        warnings.warn("reuse mapping {}/{} -> {}/{} out of range".format(
                            k, len(classes), v, len(old_classes)))
                        continue
                    for a in range(3):
                        nd[5 + k + a * new_np] = od[5 + v + a * old_np]
                        nd[a * new_np:5 + a * new_np] = od[a * old_np:5 + a * old_np]
        # adopt the fresh object's state
        self._lib.vy_net_destroy(self._h)
        self.__dict__.update({k: v for k, v in fresh.__dict__.items()
                              if k in ("_h", "_params", "_classes", "_target_generator")})
        fresh._h = None
        for p in self._params.values():
            p._net = self
        self._params._net = self
        self._host, self._dev_params, self._ws, self._plan, self._device = {}, None, None, None, None
        self._grads = self._mom = None
        self._opts_sent = {}
        self._sync_hook, self._sync_bn_checked = None, False  # the new library handle has no callback yet
        if getattr(self, "_keep_activations", False):
            _lib.check(self._lib.vy_net_set_keep_activations(self._h, 1))
        if getattr(self, "_conv_mode", "exact") != "exact":
            self.set_conv_mode(self._conv_mode)
        for k, v in new_vals.items():
            self._params[k].set_data(v)
        if device is not None:
            self.reset_ctx(device)

    def __deepcopy__(self, memo):
        """``copy.deepcopy(net)`` (transforms.py:190): a host-side copy, not bound to a device."""
        twin = YOLOV3(self._classes, self.nms_thresh, self.nms_topk, self.post_nms, self._pos_iou_thresh,
                      self._ignore_iou_thresh, self._norm_layer, copy.deepcopy(self._norm_kwargs, memo),
                      self._alloc_size)
        if self._all_set():
            for p in self._params.values():
                twin._params[p.name].set_data(self._get_param(p.index))
        twin._target_generator._label_smooth = self._target_generator._label_smooth
        return twin

    # ------------------------------------------------------------------ execution
    def _ensure_plan(self, b, h, w, train=False):
        torch = _torch()
        if self._dev_params is None:
            raise RuntimeError("parameters are not on a device: call net.collect_params().reset_ctx(ctx)")
        if self._plan is not None and self._plan[:3] == (b, h, w) and (self._plan[3] or not train):
            return
        fn = self._lib.vy_net_train_workspace_bytes if train else self._lib.vy_net_workspace_bytes
        need = fn(self._h, b, h, w)
        if need == 0:
            raise _lib.VyError(-1, self._lib.vy_last_error().decode())
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device)
        if train:
            if self._grads is None:
                n = self._lib.vy_net_param_bytes(self._h) // 4
                self._grads = torch.zeros(n, dtype=torch.float32, device=self._device)
                self._mom = torch.zeros(n, dtype=torch.float32, device=self._device)
            _lib.check(self._lib.vy_net_bind_train(
                self._h, ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(), b, h, w,
                ctypes.c_void_p(self._grads.data_ptr()), ctypes.c_void_p(self._mom.data_ptr()), self._stream()))
        else:
            _lib.check(self._lib.vy_net_bind_workspace(self._h, ctypes.c_void_p(self._ws.data_ptr()),
                                                       self._ws.numel(), b, h, w, self._stream()))
        self._plan = (b, h, w, bool(train))
        self._graphs = {}

    def _out_rows(self):
        """Rows of the inference outputs: post_nms (or nms_topk) after box_nms; with nms_thresh outside (0,1)
        the reference returns the whole (B, N*C, 6) detection tensor (yolo3.py:1197-1202)."""
        if 0 < self.nms_thresh < 1:
            rows = self.post_nms if self.post_nms > 0 else self.nms_topk
            if rows > 0:
                return rows  # both "disabled": box_nms's un-sliced output, all N*C rows (yolo3.py:1198-1202)
        return int(self._lib.vy_net_num_anchors(self._h)) * len(self._classes)

    def _as_input(self, x):
        torch = _torch()
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x, np.float32))
        if self._device is None:
            raise RuntimeError("parameters are not on a device: call net.collect_params().reset_ctx(ctx)")
        x = x.to(device=self._device, dtype=torch.float32).contiguous()
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected (B,3,H,W) input, got %s" % (tuple(x.shape),))
        return x

    def __call__(self, x, *args, return_index=False):
        """Mode is selected by the autograd state, like YOLOV3T.hybrid_forward (yolo3.py:1179-1206)."""
        if autograd.is_training():
            if autograd.is_recording():
                if len(args) != 6:
                    raise ValueError("training call: net(x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)")
                return self.forward_train(x, *args)
            return self.forward_train_mode(x)
        if self.two_stream_batch and len(x) >= self.two_stream_batch and not getattr(self, "_hybrid", False):
            return self.detect_two_streams(x, return_index=return_index)
        return self.detect(x, return_index=return_index)

    def _dev(self, a):
        torch = _torch()
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.asarray(a, np.float32))
        return a.to(device=self._device, dtype=torch.float32).contiguous()

    def forward_train(self, x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t):
        """Recording branch (yolo3.py:1181-1187): the four (B,) losses.  The network gradient is
        produced by ``autograd.backward(...)`` / ``net.backward()`` afterwards."""
        torch = _torch()
        x = self._as_input(x)
        b, _, h, w = x.shape
        tg = [self._dev(t) for t in (gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)]
        m = int(tg[0].shape[1])
        from . import parallel
        # data parallel: all ranks train the same model.  Collective-safe: the ranks first agree (all-reduce MAX of
        # their dirty flags) whether anybody wrote parameters since the last broadcast (no-op single process).
        parallel.sync_replicas_if_any_dirty(self)
        with torch.cuda.device(self._device):
            self._ensure_plan(b, h, w, train=True)
            self._ensure_sync_bn()
            if self._sync_hook is not None:
                self._sync_hook.begin_step()
            n = self._lib.vy_net_num_anchors(self._h)
            c = len(self._classes)
            want = [(b, m, 4), (b, n, 1), (b, n, 2), (b, n, 2), (b, n, 2), (b, n, c)]
            for t, shp in zip(tg, want):
                if tuple(t.shape) != shp:
                    raise ValueError("target shape %s, expected %s" % (tuple(t.shape), shp))
            _lib.check(self._lib.vy_net_set_train_options(
                self._h, self._ignore_iou_thresh, int(bool(self._target_generator._label_smooth))))
            losses = torch.empty((4, b), dtype=torch.float32, device=self._device)
            p = [ctypes.c_void_p(t.data_ptr()) for t in tg]
            _lib.check(self._lib.vy_net_train_forward(
                self._h, ctypes.c_void_p(x.data_ptr()), p[0], m, p[1], p[2], p[3], p[4], p[5],
                ctypes.c_void_p(losses.data_ptr()), self._stream()))
        self._train_x = x
        autograd._register(self)
        return tuple(autograd.loss_vector(losses[i], self, i) for i in range(4))

    def _ensure_sync_bn(self):
        """Honour ``norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': n}`` (train_yolov3.py:350-354) at the first
        train-mode forward: with more than one rank install the statistics all-reduce
        (parallel.SyncBatchNormHook); ``num_devices`` must be the number of ranks — one process drives one GPU here,
        where the reference's single process drives ``len(ctx)``.  One rank: SyncBatchNorm over one device IS
        BatchNorm; warn once.  Plain ``BatchNorm`` (the default): nothing to do."""
        if self._sync_bn_checked:
            return
        self._sync_bn_checked = True
        if self._norm_layer is not SyncBatchNorm and not (
                isinstance(self._norm_layer, type) and self._norm_layer.__name__ == "SyncBatchNorm"):
            return
        from . import parallel
        world = parallel.world_size()
        nd = (self._norm_kwargs or {}).get("num_devices")
        if nd is not None and int(nd) != world:
            self._sync_bn_checked = False
            raise ValueError("norm_layer=SyncBatchNorm with num_devices=%s, but the process group has %d rank(s): one "
                             "process drives one GPU (start N ranks: torchrun / videoyolo_amd.launch)" % (nd, world))
        if world == 1 and not parallel.collectives_active():
            warnings.warn("norm_layer=SyncBatchNorm on a single device: identical to BatchNorm, no statistics exchange")
            return
        if self._sync_hook is None:
            parallel.SyncBatchNormHook(self)

    def backward(self):
        """autograd.backward(sum of the four losses) for this net (train_yolov3.py:631)."""
        torch = _torch()
        if self._train_x is None:
            raise RuntimeError("backward() without a recorded forward")
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_train_backward(self._h, ctypes.c_void_p(self._train_x.data_ptr()),
                                                       self._stream()))
        self._train_x = None

    def forward_train_mode(self, x):
        """``autograd.train_mode()`` without recording (transforms.py:190-193): the 8-tuple of
        yolo3.py:1189-1192 — (box_preds (B,N,4), [anchors (1,1,3,2)]*3, [offsets (1,HW,1,2)]*3,
        [fake feature maps (1,1,H,W)]*3, centers (B,N,2), scales (B,N,2), objness (B,N,1),
        class_pred (B,N,C)), scales in the order stride 32, 16, 8.  Items 0 and 4-7 are device tensors
        from one train-mode forward (BatchNorm on batch statistics); items 1-3 are constants of the input
        shape (numpy), which is all the reference's consumer reads."""
        torch = _torch()
        x = self._as_input(x)
        b, _, h, w = x.shape
        anchors, offsets, fms = [], [], []
        table = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]
        for i, s in enumerate((32, 16, 8)):
            hh, ww = h // s, w // s
            anchors.append(np.array(table[i], np.float32).reshape(1, 1, 3, 2))
            gx, gy = np.meshgrid(np.arange(ww), np.arange(hh))
            offsets.append(np.stack([gx, gy], -1).astype(np.float32).reshape(1, hh * ww, 1, 2))
            fms.append(np.zeros((1, 1, hh, ww), np.float32))
        with torch.cuda.device(self._device):
            self._ensure_plan(b, h, w, train=True)
            self._ensure_sync_bn()
            n = self._lib.vy_net_num_anchors(self._h)
            c = len(self._classes)
            outs = [torch.empty((b, n, k), dtype=torch.float32, device=self._device) for k in (4, 2, 2, 1, c)]
            _lib.check(self._lib.vy_net_train_mode_forward(
                self._h, ctypes.c_void_p(x.data_ptr()), *[ctypes.c_void_p(t.data_ptr()) for t in outs],
                self._stream()))
        return (outs[0], anchors, offsets, fms, outs[1], outs[2], outs[3], outs[4])

    def grad(self, name):
        """Gradient of parameter `name` in the reference layout (numpy)."""
        torch = _torch()
        p = self._params[name]
        out = np.empty(p.shape, np.float32)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_grad_get(self._h, p.index, out.ctypes.data_as(ctypes.c_void_p),
                                                 self._stream()))
        return out

    def read_grad_activation(self, name):
        """d(loss)/d(output of cell `name`) after backward(), NCHW (parity tap)."""
        torch = _torch()
        c, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.vy_net_read_activation(self._h, name.encode(), None, ctypes.byref(c),
                                                    ctypes.byref(h), ctypes.byref(w), None))
        out = torch.empty((self._plan[0], c.value, h.value, w.value), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_read_grad_activation(self._h, name.encode(),
                                                             ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def _sync_opts(self):
        """Parameter.lr_mult / wd_mult / grad_req -> the library's SGD segment table; only the rows that
        changed since the last call are sent (steady state: no calls, no table re-upload)."""
        for p in self._params.values():
            opt = (float(p.lr_mult), float(p.wd_mult), int(p.trainable and p.grad_req != 'null'))
            if self._opts_sent.get(p.index) != opt:
                _lib.check(self._lib.vy_net_param_set_opt(self._h, p.index, *opt))
                self._opts_sent[p.index] = opt

    def sgd_step(self, lr, momentum, wd, rescale_grad):
        torch = _torch()
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_sgd_step(self._h, lr, momentum, wd, rescale_grad, self._stream()))
        self._params_written()

    def detect(self, x, return_index=False):
        """Inference branch of YOLOV3T.hybrid_forward (yolo3.py:1194-1206): returns
        (ids (B,R,1), scores (B,R,1), bboxes (B,R,4)) fp32 device tensors, R = post_nms."""
        torch = _torch()
        x = self._as_input(x)
        b, _, h, w = x.shape
        with torch.cuda.device(self._device):
            replanned = self._plan is None or self._plan[:3] != (b, h, w)
            self._ensure_plan(b, h, w)
            rows = self._out_rows()
            if getattr(self, "_hybrid", False) and self._use_graphs:
                if replanned:
                    self._graphs = {}
                ids, scores, bboxes, keep = self._graph_forward(x, rows)
                return (ids, scores, bboxes, keep) if return_index else (ids, scores, bboxes)
            ids = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            scores = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            bboxes = torch.empty((b, rows, 4), dtype=torch.float32, device=self._device)
            keep = torch.empty((b, rows), dtype=torch.int32, device=self._device) if return_index else None
            _lib.check(self._lib.vy_net_forward_infer(
                self._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(bboxes.data_ptr()),
                ctypes.c_void_p(keep.data_ptr()) if keep is not None else None, self._stream()))
        if return_index:
            return ids, scores, bboxes, keep
        return ids, scores, bboxes

    def detect_heads(self, heads, size, return_index=False):
        """The detection tail alone (``vy_net_detect_heads``): ``heads`` = the three prediction-conv outputs
        (B, 3*(5+C), H_i, W_i) for strides 32, 16, 8 of a ``size`` = (height, width) input — what
        ``net.yolo_outputs[i](pred)`` (yolo3.py:132-199), the concat (:1195), ``box_nms`` and the slice (:1197-1206)
        make of them.  Honours ``set_nms`` like a forward."""
        torch = _torch()
        if self._device is None:
            raise RuntimeError("parameters are not on a device: call net.collect_params().reset_ctx(ctx)")
        h, w = (size, size) if isinstance(size, int) else size
        hs = [self._dev(t) for t in heads]
        if len(hs) != 3:
            raise ValueError("three head tensors (strides 32, 16, 8)")
        b = int(hs[0].shape[0])
        c = 3 * (5 + len(self._classes))
        for t, div in zip(hs, (32, 16, 8)):
            want = (b, c, -(-h // div), -(-w // div))
            if tuple(t.shape) != want:
                raise ValueError("head of stride %d: shape %s, expected %s" % (div, tuple(t.shape), want))
        with torch.cuda.device(self._device):
            self._ensure_plan(b, h, w)
            rows = self._out_rows()
            ids = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            scores = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            bboxes = torch.empty((b, rows, 4), dtype=torch.float32, device=self._device)
            keep = torch.empty((b, rows), dtype=torch.int32, device=self._device) if return_index else None
            _lib.check(self._lib.vy_net_detect_heads(
                self._h, *[ctypes.c_void_p(t.data_ptr()) for t in hs], ctypes.c_void_p(ids.data_ptr()),
                ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(bboxes.data_ptr()),
                ctypes.c_void_p(keep.data_ptr()) if keep is not None else None, self._stream()))
        return (ids, scores, bboxes, keep) if return_index else (ids, scores, bboxes)

    def detect_two_streams(self, x, return_index=False):
        """Frames are independent, so a large batch is run as two half-batches on two HIP streams (a twin
        vy_net that shares the parameter buffer, with its own workspace).  Every layer's launch covers a
        non-integer number of rounds of the 512 resident blocks; with two independent launch sequences in
        flight the last, partly filled round of one kernel overlaps with the other stream's kernels, and
        block prologues/epilogues overlap too.  Results are identical to detect()."""
        torch = _torch()
        x = self._as_input(x)
        b, _, h, w = x.shape
        if b < 2 or b % 2:
            return self.detect(x, return_index=return_index)
        hb = b // 2
        with torch.cuda.device(self._device):
            self._ensure_plan(hb, h, w)
            tw = getattr(self, "_twin", None)
            if tw is None or tw["dev"] is not self._dev_params:
                th = ctypes.c_void_p()
                _lib.check(self._lib.vy_net_create(len(self._classes), ctypes.byref(th)))
                _lib.check(self._lib.vy_net_bind_params(th, ctypes.c_void_p(self._dev_params.data_ptr())))
                if getattr(self, "_conv_mode", "exact") != "exact":
                    _lib.check(self._lib.vy_net_set_conv_mode(th, _lib.VY_CONV_SPLIT_BF16X3))  # (inference only: the twin never trains)
                # a stream of the library's own: torch's pooled streams may share the default stream's
                # hardware queue, in which case the two launch sequences would simply alternate
                sp = ctypes.c_void_p()
                _lib.check(self._lib.vy_stream_create(ctypes.byref(sp)))
                tw = self._twin = dict(h=th, dev=self._dev_params, ws=None, plan=None, raw_stream=sp,
                                       stream=torch.cuda.ExternalStream(sp.value, device=self._device))
            _lib.check(self._lib.vy_net_set_nms(tw["h"], self.nms_thresh, self.nms_topk, self.post_nms))
            cur = torch.cuda.current_stream(self._device)
            if tw["plan"] != (hb, h, w):
                need = self._lib.vy_net_workspace_bytes(tw["h"], hb, h, w)
                if tw["ws"] is None or tw["ws"].numel() < need:
                    tw["ws"] = torch.empty(need, dtype=torch.uint8, device=self._device)
                _lib.check(self._lib.vy_net_bind_workspace(tw["h"], ctypes.c_void_p(tw["ws"].data_ptr()),
                                                           tw["ws"].numel(), hb, h, w, ctypes.c_void_p(cur.cuda_stream)))
                tw["plan"] = (hb, h, w)
            rows = self._out_rows()
            ids = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            scores = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            bboxes = torch.empty((b, rows, 4), dtype=torch.float32, device=self._device)
            keep = torch.empty((b, rows), dtype=torch.int32, device=self._device) if return_index else None
            side = tw["stream"]
            side.wait_stream(cur)  # inputs, workspace zeroing and the output allocations are ordered before it
            for half, (hnd, st) in enumerate(((self._h, cur), (tw["h"], side))):
                lo = half * hb
                _lib.check(self._lib.vy_net_forward_infer(
                    hnd, ctypes.c_void_p(x[lo:].data_ptr()), ctypes.c_void_p(ids[lo:].data_ptr()),
                    ctypes.c_void_p(scores[lo:].data_ptr()), ctypes.c_void_p(bboxes[lo:].data_ptr()),
                    ctypes.c_void_p(keep[lo:].data_ptr()) if keep is not None else None,
                    ctypes.c_void_p(st.cuda_stream)))
            cur.wait_stream(side)
            for t in (x, ids, scores, bboxes) + ((keep,) if keep is not None else ()):
                t.record_stream(side)
        if return_index:
            return ids, scores, bboxes, keep
        return ids, scores, bboxes

    def streamk_enabled(self):
        """Did the bind-time placement probe (8 XCDs, workgroups dealt round-robin: csrc/conv_igemm.hip
        vy_sk_verify_topology) enable chain-preserving stream-K on this net's device?  None before a workspace is bound."""
        if self._plan is None:
            return None
        en = ctypes.c_int32(0)
        _lib.check(self._lib.vy_net_streamk_state(self._h, ctypes.byref(en), None, None))
        return bool(en.value)

    def read_head(self, i):
        """Prediction-conv output of head i (stride 32,16,8) of the last forward, NCHW."""
        torch = _torch()
        b, h, w = self._plan[:3]
        div = (32, 16, 8)[i]
        out = torch.empty((b, 3 * (5 + len(self._classes)), -(-h // div), -(-w // div)), dtype=torch.float32,
                          device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_read_head(self._h, i, ctypes.c_void_p(out.data_ptr()), self._stream()))
        return out

    def read_activation(self, name):
        """Output of cell `name` (e.g. 'stages.0.14.body.1') of the last forward, NCHW."""
        torch = _torch()
        c, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self._lib.vy_net_read_activation(self._h, name.encode(), None, ctypes.byref(c),
                                                    ctypes.byref(h), ctypes.byref(w), None))
        out = torch.empty((self._plan[0], c.value, h.value, w.value), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.vy_net_read_activation(self._h, name.encode(), ctypes.c_void_p(out.data_ptr()),
                                                        None, None, None, self._stream()))
        return out

    def profile(self, x):
        """One timed forward: list of (launch name, ms, flops, bytes) from HIP events around every
        kernel launch on the current stream."""
        torch = _torch()
        x = self._as_input(x)
        b, _, h, w = x.shape
        with torch.cuda.device(self._device):
            self._ensure_plan(b, h, w)
            rows = self._out_rows()
            ids = torch.empty((b, rows, 1), dtype=torch.float32, device=self._device)
            scores = torch.empty_like(ids)
            bboxes = torch.empty((b, rows, 4), dtype=torch.float32, device=self._device)
            cap = 256
            stats = (_lib.LaunchStat * cap)()
            n = ctypes.c_int32(cap)
            _lib.check(self._lib.vy_net_profile_infer(
                self._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(bboxes.data_ptr()), stats,
                ctypes.byref(n), self._stream()))
        return [(stats[i].name.decode(), float(stats[i].ms), float(stats[i].flops), float(stats[i].bytes))
                for i in range(n.value)]

    def summary(self, *inputs):
        """``net.summary(x)`` (train_yolov3.py:758): parameter table."""
        total = 0
        lines = []
        for p in self._params.values():
            lines.append("%-44s %-22s %10d" % (p.name, p.shape, p.size))
            if p.trainable:
                total += p.size
        lines.append("trainable parameters: %d" % total)
        print("\n".join(lines))


# the reference class names
YOLOV3T = YOLOV3


def _darknet_roots(root=None):
    """Where ``get_model_file('darknet53', root=...)`` would look: the reference's default root
    (three_darknet.py:234 ``models/definitions/darknet/weights``), gluoncv's cache, VY_MODEL_ROOT."""
    roots = [root] if root else []
    if os.environ.get("VY_MODEL_ROOT"):
        roots.append(os.environ["VY_MODEL_ROOT"])
    roots += [os.path.join("models", "definitions", "darknet", "weights"),
              os.path.join(os.path.expanduser("~"), ".mxnet", "models")]
    return roots


def find_darknet53_file(root=None):
    """``darknet53-<hash>.params`` (gluoncv's file name) or ``darknet53.params`` in the first root that has one."""
    import glob
    for r in _darknet_roots(root):
        hits = sorted(glob.glob(os.path.join(r, "darknet53-*.params"))) + glob.glob(os.path.join(r, "darknet53.params"))
        if hits:
            return hits[0]
    return None


def darknet53_to_stage_names(arrays):
    """Structural names of a DarknetV3 / Darknet3D checkpoint (``features.<n>...``, ``output.*``: what
    ``save_parameters`` of the ImageNet classifier writes) -> the detector's names: wrappers.py:58 slices
    ``features[:15] / [15:24] / [24:]`` into ``stages.0 / 1 / 2``.  The classifier's dense ``output`` layer is dropped
    (three_darknet.py:263 ``ignore_extra=return_features``).  Older prefix-style names are rejected."""
    out = {}
    for k, v in arrays.items():
        if k.startswith("output."):
            continue
        m = re.match(r"^features\.(\d+)\.(.+)$", k)
        if not m:
            raise ValueError("'%s' is not a structural DarknetV3 parameter name (features.<n>...): the file was not "
                             "written by save_parameters of the darknet53 classifier" % k)
        f = int(m.group(1))
        if f > 28:
            raise ValueError("features.%d: darknet53 has 29 feature cells" % f)
        si, j = (0, f) if f < 15 else ((1, f - 15) if f < 24 else (2, f - 24))
        out["stages.%d.%d.%s" % (si, j, m.group(2))] = v
    return out


def yolo3_darknet53(classes, pretrained_base=True, norm_layer=BatchNorm, norm_kwargs=None, freeze_base=False,
                    k=None, k_join_type=None, k_join_pos=None, block_conv_type='2', rnn_pos=None,
                    corr_pos=None, corr_d=None, motion_stream=None, add_type=None, agnostic=False,
                    new_model=False, hierarchical=(1, 1, 1, 1, 1), h_join_type=None, temporal=False,
                    t_out=False, **kwargs):
    """Drop-in for models/definitions/yolo/wrappers.py:9-110 on the default (k=1) branch.

    Only the arguments that select the hot path are honoured; a non-default value for any
    temporal / two-stream / hierarchical option raises NotImplementedError (out of scope).
    ``pretrained_base=True`` cannot be honoured offline (gluoncv model zoo download,
    three_darknet.py:262): it warns and leaves the backbone to ``initialize()`` / ``load_parameters``.
    """
    unsupported = {
        "k": k not in (None, 1), "k_join_type": k_join_type is not None, "k_join_pos": k_join_pos is not None,
        "block_conv_type": str(block_conv_type) != '2', "rnn_pos": rnn_pos is not None,
        "corr_pos": corr_pos is not None, "corr_d": corr_d is not None, "motion_stream": motion_stream is not None,
        "add_type": add_type is not None, "agnostic": bool(agnostic), "new_model": bool(new_model),
        "hierarchical": any(int(h) != 1 for h in hierarchical), "h_join_type": h_join_type is not None,
        "temporal": bool(temporal), "t_out": bool(t_out)}
    bad = [n for n, v in unsupported.items() if v]
    if bad:
        raise NotImplementedError(
            "yolo3_darknet53: option(s) %s select a temporal/two-stream research variant outside the "
            "MI355X hot path" % ", ".join(bad))
    root = kwargs.pop("root", None)
    net = YOLOV3(classes, norm_layer=norm_layer, norm_kwargs=norm_kwargs, **kwargs)
    if pretrained_base:
        # three_darknet.py:262-264: net.load_parameters(get_model_file('darknet53', tag=pretrained, root=root)).  There is
        # no model-zoo download here: the file must already be where gluoncv would have cached it.
        path = find_darknet53_file(root)
        if path is None:
            warnings.warn("pretrained_base=True: no darknet53-*.params under %s (gluoncv would download it; there is no "
                          "network access here) — backbone left to initialize() / load_parameters()" % (_darknet_roots(root),))
        else:
            net.load_darknet53_backbone(path)
    if freeze_base:  # wrappers.py:55-57
        for p in net.collect_params().values():
            if p.backbone:
                p.grad_req = 'null'
    return net
