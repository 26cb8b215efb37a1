"""CPU oracle for the yolo3_darknet53 hot path — numpy graph code over oracle/ref_ops.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and bench.py's
``cpu_baseline`` leg, as the checker.  Nothing under ``videoyolo_amd/`` imports it.

PARITY UNPINNED (see ref_ops.c header): mxnet/gluoncv are absent and unpinned and
the reference has no tests; this restates the reference's *graph* (which IS in the tree, cited
per function below) over restated operator semantics.

All citations are relative to /root/reference.  Layout is the reference's: NCHW / OIHW, fp32.
The graph code below deliberately mirrors the reference's reshape / transpose / concat sequence
so that element ORDER (anchor order, class-major detection order) is the reference's by
construction, not by derivation.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build():
    """Compile libvyoracle.so (gcc + OpenMP).  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvyoracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.vyo_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(_f32p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- operator wrappers
def conv2d(x, w, stride, pad, scale=None, shift=None, leaky=False):
    x, w = _c(x), _c(w)
    n, c, h, wd = x.shape
    o, ci, k, _ = w.shape
    assert ci == c
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = np.empty((n, o, ho, wo), np.float32)
    sc = _c(scale) if scale is not None else None
    sh = _c(shift) if shift is not None else None
    lib().vyo_conv2d(_p(x), n, c, h, wd, _p(w), o, k, stride, pad,
                     _p(sc) if sc is not None else None, _p(sh) if sh is not None else None,
                     int(bool(leaky)), _p(y))
    return y


def bn_fold(gamma, beta, mean, var, eps=1e-5):
    g, b, m, v = map(_c, (gamma, beta, mean, var))
    sc, sh = np.empty_like(g), np.empty_like(g)
    lib().vyo_bn_fold(_p(g), _p(b), _p(m), _p(v), ctypes.c_float(eps), g.size, _p(sc), _p(sh))
    return sc, sh


def bn_train(x, gamma, beta, eps=1e-5, leaky=True):
    x = _c(x)
    n, c, h, w = x.shape
    y = np.empty_like(x)
    mean, var = np.empty(c, np.float32), np.empty(c, np.float32)
    lib().vyo_bn_train(_p(x), n, c, h * w, _p(_c(gamma)), _p(_c(beta)), ctypes.c_float(eps),
                       int(bool(leaky)), _p(y), _p(mean), _p(var))
    return y, mean, var


def _ew(fn, x):
    x = _c(x)
    y = np.empty_like(x)
    getattr(lib(), fn)(_p(x), ctypes.c_size_t(x.size), _p(y))
    return y


def sigmoid(x):
    return _ew("vyo_sigmoid", x)


def exp(x):
    return _ew("vyo_exp", x)


def log(x):
    return _ew("vyo_log", x)


def box_nms(data, overlap_thresh=0.45, valid_thresh=0.01, topk=400, force_suppress=False):
    """contrib.box_nms as called at models/definitions/yolo/yolo3.py:1198-1200."""
    data = _c(data)
    b, n, six = data.shape
    assert six == 6
    out = np.empty_like(data)
    idx = np.empty((b, n), np.int32)
    lib().vyo_box_nms(_p(data), b, n, ctypes.c_float(overlap_thresh), ctypes.c_float(valid_thresh),
                      int(topk), int(bool(force_suppress)), _p(out), idx.ctypes.data_as(_i32p))
    return out, idx


def batch_iou(a, b):
    a, b = _c(a), _c(b)
    bb, n, _ = a.shape
    m = b.shape[1]
    out = np.empty((bb, n, m), np.float32)
    lib().vyo_batch_iou(_p(a), _p(b), bb, n, m, _p(out))
    return out


# ---------------------------------------------------------------- network description
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
STRIDES = [8, 16, 32]  # models/definitions/yolo/wrappers.py:80-84
DARKNET_LAYERS = [1, 2, 8, 8, 4]  # models/definitions/darknet/three_darknet.py:252
DARKNET_CHANNELS = [32, 64, 128, 256, 512, 1024]  # three_darknet.py:257
HEAD_CHANNELS = [512, 256, 128]  # wrappers.py:101


def _cell(prefix, cin, cout, k):
    """_conv2d cell (layers.py:63-70): Conv2D(no bias) + BatchNorm + LeakyReLU(0.1).
    Gluon structural names: child 0 = conv, child 1 = norm."""
    return [(prefix + ".0.weight", (cout, cin, k, k)),
            (prefix + ".1.gamma", (cout,)), (prefix + ".1.beta", (cout,)),
            (prefix + ".1.running_mean", (cout,)), (prefix + ".1.running_var", (cout,))]


def darknet_feature_cells():
    """Ordered list of Darknet3D.features entries (three_darknet.py:162-195, conv_types all 2).
    Each entry: ('conv', cin, cout, k, stride) or ('block', c) [c -> c/2 1x1 -> c 3x3, + x]."""
    feats = [("conv", 3, DARKNET_CHANNELS[0], 3, 1)]
    cin = DARKNET_CHANNELS[0]
    for nlayer, ch in zip(DARKNET_LAYERS, DARKNET_CHANNELS[1:]):
        feats.append(("conv", cin, ch, 3, 2))
        feats += [("block", ch)] * nlayer
        cin = ch
    return feats


STAGE_SLICES = [(0, 15), (15, 24), (24, 29)]  # wrappers.py:58 features[:15], [15:24], [24:]


def param_shapes(num_class):
    """[(structural name, shape)] in construction order — the names gluon's save_parameters
    would write for YOLOV3T (yolo3.py:1013-1054) with stages = slices of Darknet3D.features."""
    out = []
    feats = darknet_feature_cells()
    for si, (lo, hi) in enumerate(STAGE_SLICES):
        for j, f in enumerate(feats[lo:hi]):
            pre = "stages.%d.%d" % (si, j)
            if f[0] == "conv":
                out += _cell(pre, f[1], f[2], f[3])
            else:
                c = f[1]
                out += _cell(pre + ".body.0", c, c // 2, 1)
                out += _cell(pre + ".body.1", c // 2, c, 3)
    npred = 3 * (5 + num_class)
    head_in = [1024, 512 + 256, 256 + 128]
    for i, ch in enumerate(HEAD_CHANNELS):
        pre = "yolo_blocks.%d" % i
        cin = head_in[i]
        for j in range(5):  # yolo3.py:218-253 body: [1x1 c, 3x3 2c] x2, 1x1 c
            if j % 2 == 0:
                out += _cell("%s.body.%d" % (pre, j), cin, ch, 1)
                cin = ch
            else:
                out += _cell("%s.body.%d" % (pre, j), cin, ch * 2, 3)
                cin = ch * 2
        out += _cell(pre + ".tip", ch, ch * 2, 3)
        out += [("yolo_outputs.%d.prediction.weight" % i, (npred, ch * 2, 1, 1)),
                ("yolo_outputs.%d.prediction.bias" % i, (npred,))]
        if i > 0:
            pass
    for i, ch in enumerate(HEAD_CHANNELS[1:]):  # transitions: yolo3.py:1047-1054 (i > 0)
        out += _cell("transitions.%d" % i, ch * 2, ch, 1)
    return out


def synthetic_params(num_class, seed=233, bias_obj=0.0):
    """Deterministic synthetic parameters (no checkpoint exists):
    conv W ~ N(0, 2/(k*k*cin)) * 1.4, gamma ~ U(0.8,1.2), beta ~ N(0,0.1), running_mean ~ N(0,0.1),
    running_var ~ U(0.5,1.5), output bias ~ N(0, 0.5) (+ bias_obj on objectness rows)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes(num_class):
        leaf = name.rsplit(".", 1)[1]
        if leaf == "weight":
            fan = shape[1] * shape[2] * shape[3]
            p[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan) * 1.4).astype(np.float32)
        elif leaf == "gamma":
            p[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif leaf in ("beta", "running_mean"):
            p[name] = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == "running_var":
            p[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf == "bias":
            b = (rng.standard_normal(shape) * 0.5).astype(np.float32)
            b.reshape(3, -1)[:, 4] += bias_obj
            p[name] = b
    return p


# ---------------------------------------------------------------- the graph
class OracleYolo3:
    """yolo3_darknet53 at k=1 (YOLOV3T ≡ YOLOV3), inference mode, on the CPU.

    reference: wrappers.py:54-58,80-84,101-103; yolo3.py:1013-1054 (construction),
    yolo3.py:1076-1206 (hybrid_forward), yolo3.py:132-199 (YOLOOutputV3), layers.py:11-20,63-70.
    """

    def __init__(self, num_class, params, nms_thresh=0.45, nms_topk=400, post_nms=100):
        self.C = num_class
        self.p = params
        self.nms_thresh, self.nms_topk, self.post_nms = nms_thresh, nms_topk, post_nms
        self.taps = {}  # name -> activation, filled when keep=True

    # layers.py:63-70, eval mode
    def cell(self, x, pre, k, stride):
        sc, sh = bn_fold(self.p[pre + ".1.gamma"], self.p[pre + ".1.beta"],
                         self.p[pre + ".1.running_mean"], self.p[pre + ".1.running_var"])
        return conv2d(x, self.p[pre + ".0.weight"], stride, k // 2, sc, sh, leaky=True)

    # three_darknet.py:85-123 (conv_type 2)
    def block(self, x, pre):
        y = self.cell(x, pre + ".body.0", 1, 1)
        y = self.cell(y, pre + ".body.1", 3, 1)
        return y + x

    def stages(self, x):
        feats = darknet_feature_cells()
        routes = []
        for si, (lo, hi) in enumerate(STAGE_SLICES):
            for j, f in enumerate(feats[lo:hi]):
                pre = "stages.%d.%d" % (si, j)
                x = self.cell(x, pre, f[3], f[4]) if f[0] == "conv" else self.block(x, pre)
            routes.append(x)
        return routes

    # yolo3.py:256-263
    def det_block(self, x, i):
        pre = "yolo_blocks.%d" % i
        for j in range(5):
            x = self.cell(x, "%s.body.%d" % (pre, j), 1 if j % 2 == 0 else 3, 1)
        route = x
        tip = self.cell(route, pre + ".tip", 3, 1)
        return route, tip

    # yolo3.py:132-199, inference branch; returns (B, C*H*W*A, 6) class-major
    def output(self, tip, i, raw_only=False):
        C, A = self.C, 3
        P = 5 + C
        pred = conv2d(tip, self.p["yolo_outputs.%d.prediction.weight" % i], 1, 0,
                      None, self.p["yolo_outputs.%d.prediction.bias" % i], leaky=False)
        if raw_only:
            return pred
        return self.decode(pred, i)

    # yolo3.py:158-197: the decode of ONE scale's prediction-conv output `pred` (B, A*(5+C), H, W) — what
    # `net.yolo_outputs[i](pred)` returns outside autograd; i = 0, 1, 2 for strides 32, 16, 8
    def decode(self, pred, i):
        C, A = self.C, 3
        P = 5 + C
        pred = _c(pred)
        B, _, H, W = pred.shape
        anchors = np.array(ANCHORS[::-1][i], np.float32).reshape(1, 1, A, 2)  # yolo3.py:1013
        stride = np.float32(STRIDES[::-1][i])
        pred = pred.reshape(B, A * P, H * W)  # :158
        pred = pred.transpose(0, 2, 1).reshape(B, H * W, A, P)  # :160
        raw_xy, raw_wh = pred[..., 0:2], pred[..., 2:4]
        obj, cls = pred[..., 4:5], pred[..., 5:]
        gx, gy = np.meshgrid(np.arange(W), np.arange(H))  # :67-74 then slice_like :168
        offsets = np.stack([gx, gy], -1).astype(np.float32).reshape(1, H * W, 1, 2)
        centers = (sigmoid(raw_xy) + offsets) * stride  # :172
        scales = exp(raw_wh) * anchors  # :173
        conf = sigmoid(obj)  # :174
        score = sigmoid(cls) * conf  # :175
        wh = scales / np.float32(2.0)  # :176
        bbox = np.concatenate([centers - wh, centers + wh], -1)  # :177  (B,HW,A,4)
        bboxes = np.tile(bbox[None], (C, 1, 1, 1, 1))  # :191  (C,B,HW,A,4)
        scores = score.transpose(3, 0, 1, 2)[..., None]  # :192  (C,B,HW,A,1)
        ids = scores * 0 + np.arange(C, dtype=np.float32).reshape(C, 1, 1, 1, 1)  # :194
        det = np.concatenate([ids, scores, bboxes], -1)  # :195
        return det.transpose(1, 0, 2, 3, 4).reshape(B, -1, 6).astype(np.float32)  # :197

    def heads(self, routes, raw_only=False):
        """yolo3.py:1126-1177: deep -> shallow, transitions + upsample + concat."""
        x = routes[-1]
        outs = []
        for i in range(3):
            x, tip = self.det_block(x, i)
            outs.append(self.output(tip, i, raw_only))
            if i >= 2:
                break
            x = self.cell(x, "transitions.%d" % i, 1, 1)
            up = x.repeat(2, axis=-1).repeat(2, axis=-2)  # layers.py:20
            route_now = routes[::-1][i + 1]
            up = up[:, :, :route_now.shape[2], :route_now.shape[3]]  # slice_like, yolo3.py:1177
            x = np.concatenate([up, route_now], axis=1)
        return outs

    def detections_from_heads(self, heads):
        """yolo3.py:1195 on caller-supplied prediction-conv outputs (stride 32, 16, 8): (B, N*C, 6)."""
        return np.concatenate([self.decode(h, i) for i, h in enumerate(heads)], axis=1)

    def raw_heads(self, x):
        """Per-scale prediction-conv outputs (B, A*(5+C), H, W), order stride 32, 16, 8."""
        return self.heads(self.stages(_c(x)), raw_only=True)

    def detections(self, x):
        """Concatenated pre-NMS detections (B, N*C, 6), yolo3.py:1195."""
        return np.concatenate(self.heads(self.stages(_c(x))), axis=1)

    def nms(self, result):
        """yolo3.py:1197-1206.  Returns ids, scores, bboxes and the kept input-row indices."""
        idx = None
        if 0 < self.nms_thresh < 1:
            result, idx = box_nms(result, self.nms_thresh, 0.01, self.nms_topk, False)
            if self.post_nms > 0:
                result = result[:, :self.post_nms]
                idx = idx[:, :self.post_nms]
        return result[..., 0:1], result[..., 1:2], result[..., 2:], idx

    def __call__(self, x):
        return self.nms(self.detections(x))
