"""Independent cross-check of the oracle's graph + operators against a torch-CPU model of the same
network, written straight from the reference's layer definitions with torch.nn.functional ops.
This is NOT the reference (mxnet) — it is a second, independently written implementation whose
agreement to 1e-4 guards the restatement against transcription errors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import frames
from oracle import yolo3_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


class TorchYolo3:
    """yolo3_darknet53 (k=1) inference with un-fused BatchNorm, written against the reference:
    layers.py:63-70, three_darknet.py:85-123,162-195, yolo3.py:218-263,1013-1054,1105-1195."""

    def __init__(self, ncls, p):
        self.C, self.p = ncls, {k: _t(v) for k, v in p.items()}

    def cell(self, x, pre, k, s):
        p = self.p
        x = F.conv2d(x, p[pre + ".0.weight"], None, s, k // 2)
        x = F.batch_norm(x, p[pre + ".1.running_mean"], p[pre + ".1.running_var"], p[pre + ".1.gamma"],
                         p[pre + ".1.beta"], False, 0.9, 1e-5)
        return F.leaky_relu(x, 0.1)

    def forward_heads(self, x):
        x = _t(x)
        routes = []
        layers, chans = [1, 2, 8, 8, 4], [32, 64, 128, 256, 512, 1024]
        feats = [("c", 3, 1)]
        for n in layers:
            feats += [("c", 3, 2)] + [("b",)] * n
        bounds = [(0, 15), (15, 24), (24, 29)]
        for si, (lo, hi) in enumerate(bounds):
            for j, f in enumerate(feats[lo:hi]):
                pre = "stages.%d.%d" % (si, j)
                if f[0] == "c":
                    x = self.cell(x, pre, f[1], f[2])
                else:
                    x = x + self.cell(self.cell(x, pre + ".body.0", 1, 1), pre + ".body.1", 3, 1)
            routes.append(x)
        outs = []
        x = routes[2]
        for i in range(3):
            for j in range(5):
                x = self.cell(x, "yolo_blocks.%d.body.%d" % (i, j), 1 if j % 2 == 0 else 3, 1)
            tip = self.cell(x, "yolo_blocks.%d.tip" % i, 3, 1)
            outs.append(F.conv2d(tip, self.p["yolo_outputs.%d.prediction.weight" % i],
                                 self.p["yolo_outputs.%d.prediction.bias" % i]))
            if i == 2:
                break
            x = self.cell(x, "transitions.%d" % i, 1, 1)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            r = routes[1 - i]
            x = torch.cat([x[:, :, :r.shape[2], :r.shape[3]], r], 1)  # slice_like, yolo3.py:1177
        return outs

    def decode(self, pred, i):
        """(B, A*P, H, W) -> (B, C*H*W*A, 6), independent index arithmetic (no reshape chain)."""
        B, _, H, W = pred.shape
        A, C = 3, self.C
        P = 5 + C
        anchors = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]][i]
        stride = [32, 16, 8][i]
        pr = pred.view(B, A, P, H, W)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                                indexing="ij")
        cx = (torch.sigmoid(pr[:, :, 0]) + xs) * stride
        cy = (torch.sigmoid(pr[:, :, 1]) + ys) * stride
        aw = torch.tensor(anchors[0::2], dtype=torch.float32).view(1, A, 1, 1)
        ah = torch.tensor(anchors[1::2], dtype=torch.float32).view(1, A, 1, 1)
        w, h = torch.exp(pr[:, :, 2]) * aw, torch.exp(pr[:, :, 3]) * ah
        conf = torch.sigmoid(pr[:, :, 4])
        out = torch.empty(B, C, H, W, A, 6)
        for c in range(C):
            sc = torch.sigmoid(pr[:, :, 5 + c]) * conf  # (B,A,H,W)
            out[:, c, :, :, :, 0] = c
            out[:, c, :, :, :, 1] = sc.permute(0, 2, 3, 1)
            out[:, c, :, :, :, 2] = (cx - w / 2).permute(0, 2, 3, 1)
            out[:, c, :, :, :, 3] = (cy - h / 2).permute(0, 2, 3, 1)
            out[:, c, :, :, :, 4] = (cx + w / 2).permute(0, 2, 3, 1)
            out[:, c, :, :, :, 5] = (cy + h / 2).permute(0, 2, 3, 1)
        return out.reshape(B, -1, 6)


def _py_nms(d, thr, valid, topk):
    """Plain-Python per-class greedy NMS on one image (n,6): returns kept input rows."""
    order = [i for i in np.argsort(-d[:, 1], kind="stable") if d[i, 1] > valid][:topk]
    keep = []
    for i in order:
        ok = True
        for j in keep:
            if d[i, 0] != d[j, 0]:
                continue
            iw = min(d[i, 4], d[j, 4]) - max(d[i, 2], d[j, 2])
            ih = min(d[i, 5], d[j, 5]) - max(d[i, 3], d[j, 3])
            inter = max(iw, 0) * max(ih, 0)
            u = (d[i, 4] - d[i, 2]) * (d[i, 5] - d[i, 3]) + (d[j, 4] - d[j, 2]) * (d[j, 5] - d[j, 3]) - inter
            if u > 0 and inter / u > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


@pytest.fixture(scope="module")
def both(synth20):
    x = frames(2, 64)
    orc = O.OracleYolo3(20, synth20)
    tm = TorchYolo3(20, synth20)
    with torch.no_grad():
        th = tm.forward_heads(x)
    return x, orc, tm, th


def test_heads_agree_with_torch(both):
    x, orc, tm, th = both
    oh = orc.raw_heads(x)
    for a, b in zip(oh, th):
        assert a.shape == tuple(b.shape)
        np.testing.assert_allclose(a, b.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("hw", [(100, 136), (75, 93)])
def test_sizes_that_are_not_multiples_of_32(synth20, hw):
    """The reference crops the x2 upsample to the route it is concatenated with (slice_like, yolo3.py:1177) and a
    stride-2 3x3 pad-1 conv maps n rows to ceil(n / 2): heads of ceil(H / 32), ceil(H / 16), ceil(H / 8) rows.
    (100, 136) crops both upsamples on both axes; (75, 93) is odd at every level."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 3) + hw).astype(np.float32)
    orc, tm = O.OracleYolo3(20, synth20), TorchYolo3(20, synth20)
    with torch.no_grad():
        th = tm.forward_heads(x)
    oh = orc.raw_heads(x)
    for i, (a, b) in enumerate(zip(oh, th)):
        d = [32, 16, 8][i]
        assert a.shape == (1, 75, -(-hw[0] // d), -(-hw[1] // d)) == tuple(b.shape)
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-4, atol=1e-4)
    od = orc.detections(x)
    td = torch.cat([tm.decode(h, i) for i, h in enumerate(th)], 1).numpy()
    assert od.shape == td.shape
    np.testing.assert_allclose(od[..., 1], td[..., 1], rtol=0, atol=1e-4)
    np.testing.assert_allclose(od[..., 2:], td[..., 2:], rtol=1e-4, atol=1e-3)


def test_decode_layout_agrees_with_torch(both):
    x, orc, tm, th = both
    with torch.no_grad():
        td = torch.cat([tm.decode(h, i) for i, h in enumerate(th)], 1).numpy()
    od = orc.detections(x)
    assert od.shape == td.shape == (2, 20 * 3 * (4 + 16 + 64), 6)
    assert np.array_equal(od[..., 0], td[..., 0])
    np.testing.assert_allclose(od[..., 1], td[..., 1], rtol=0, atol=1e-5)
    np.testing.assert_allclose(od[..., 2:], td[..., 2:], rtol=1e-5, atol=1e-3)


def test_nms_agrees_with_python_reference(both):
    x, orc, tm, th = both
    det = orc.detections(x)
    ids, scores, bboxes, idx = orc.nms(det)
    for b in range(det.shape[0]):
        keep = _py_nms(det[b].astype(np.float64), 0.45, 0.01, 400)[:100]
        got = [int(i) for i in idx[b] if i >= 0]
        assert got == keep
    rng = np.random.default_rng(1)
    n = 300
    xy = rng.uniform(0, 80, (n, 2))
    wh = rng.uniform(5, 40, (n, 2))
    d = np.concatenate([rng.integers(0, 3, (n, 1)), rng.uniform(0, 1, (n, 1)), xy, xy + wh], 1).astype(np.float32)
    d[::17] = -1
    out, idx = O.box_nms(d[None], 0.45, 0.01, 100)
    assert [int(i) for i in idx[0] if i >= 0] == _py_nms(d.astype(np.float64), 0.45, 0.01, 100)


class TorchYolo3F64(TorchYolo3):
    """The same torch graph evaluated in float64: the closest thing to exact arithmetic available here."""

    def __init__(self, ncls, p):
        self.C, self.p = ncls, {k: _t(v).double() for k, v in p.items()}


@pytest.mark.parametrize("size", [416, 608])
def test_full_size_frame_agrees_with_torch_fp32_and_fp64(synth20, size, capsys):
    """BASELINE configs[0] (1 x 416 x 416, seed 233 = train_yolov3.py:135) and one 608 x 608 frame (the
    configs[1] shape): the oracle against the torch model in fp32 AND in float64.

    Bars (north_star: ids / kept rows exact, scores and coordinates 1e-4):
      heads   |oracle - torch| <= 1e-4 absolute (raw predictions are O(1); observed <= 2e-5)
      scores  <= 1e-4 absolute (observed <= 4e-6)
      boxes   <= 1e-4 x max(1, box width, box height): `exp(raw) * anchor` turns an absolute error of the raw
              prediction into one RELATIVE to the box extent, and with random weights boxes reach 1000 px —
              torch-fp32 and torch-float64 themselves differ by up to 5e-3 px there, so an absolute 1e-4
              between two independently ordered fp32 evaluations does not exist; relative to the extent the
              oracle sits at <= 3e-5
      NMS     kept rows identical between the oracle's box_nms and the plain-Python NMS on the same detections;
              the plain-Python NMS of torch-fp32's / torch-float64's own detection tensors keeps >= 90 of the same
              100 rows (last-bit score differences flip near-ties: that part is a sanity bound, not a parity claim)
    and the oracle must be about as close to float64 as torch's own fp32 path is (what the matrix-core
    summation order 0,4,1,5,2,6,3,7 costs): within 4x."""
    x = frames(1, size, seed=233)
    orc = O.OracleYolo3(20, synth20)
    oh = orc.raw_heads(x)
    with torch.no_grad():
        t32 = TorchYolo3(20, synth20)
        th = t32.forward_heads(x)
        th64 = TorchYolo3F64(20, synth20).forward_heads(x.astype(np.float64))
    worst = 0.0
    for i in range(3):
        a, b, c = oh[i], th[i].numpy(), th64[i].numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
        np.testing.assert_allclose(a, c, rtol=0, atol=1e-4)
        d_orc, d_t32 = np.abs(a - c).max(), np.abs(b - c).max()
        worst = max(worst, d_orc / d_t32)
        assert d_orc <= 4 * d_t32, (i, d_orc, d_t32)
    with torch.no_grad():
        td = torch.cat([t32.decode(h, i) for i, h in enumerate(th)], 1).numpy()
        td64 = torch.cat([t32.decode(h.float(), i) for i, h in enumerate(th64)], 1).numpy()
    od = orc.detections(x)
    n = 3 * sum((size // s) ** 2 for s in (32, 16, 8))
    assert od.shape == td.shape == (1, 20 * n, 6)
    assert np.array_equal(od[..., 0], td[..., 0])                         # class ids: the reference's row order
    for other in (td, td64):
        np.testing.assert_allclose(od[..., 1], other[..., 1], rtol=0, atol=1e-4)
        fin = np.isfinite(other[..., 2:]).all(-1) & np.isfinite(od[..., 2:]).all(-1)
        assert fin.mean() > 0.999
        ext = np.maximum(1.0, np.maximum(other[..., 4] - other[..., 2], other[..., 5] - other[..., 3]))[fin]
        err = np.abs(od[..., 2:] - other[..., 2:])[fin].max(-1)
        assert (err <= 1e-4 * ext).all(), float((err / ext).max())
    ids, scores, bboxes, idx = orc.nms(od)
    got = [int(i) for i in idx[0] if i >= 0]
    assert len(got) == 100
    assert got == _py_nms(od[0].astype(np.float64), 0.45, 0.01, 400)[:100]          # same detections: exact
    # NMS of torch's OWN detection tensors (scores / boxes differing in the last bits): the kept rows agree except
    # where a near-tie in score order or an IoU within rounding of the threshold flips a decision
    for det in (td, td64):
        other = _py_nms(det[0].astype(np.float64), 0.45, 0.01, 400)[:100]
        common = len(set(got) & set(other))
        assert common >= 90, common
    with capsys.disabled():
        print("\n[%d] oracle-vs-float64 / torch32-vs-float64 head distance ratio: %.2f" % (size, worst))
