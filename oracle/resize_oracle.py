"""CPU restatement of the frame resize in front of the network.  TEST INFRASTRUCTURE ONLY.

Reference call site: ``timage.imresize(src[i], self._width, self._height, interp=9)``
(models/definitions/yolo/transforms.py:325-327, gluoncv.data.transforms.image.imresize) on a uint8 HWC
frame; the result is a uint8 NDArray that goes to ``to_tensor`` + ``normalize`` (:331-334).

The arithmetic lives in third-party code that is absent here — gluoncv (unpinned) -> mxnet
``image.imresize`` (unpinned) -> OpenCV ``cv::resize`` — so this file is PARITY UNPINNED: it restates the
published algorithms from memory [UPSTREAM-RECALLED], and tests/test_resize_oracle.py cross-checks it
against independent implementations where their definitions coincide (torch bicubic A = -0.75 /
bilinear with half-pixel centres, PIL's BOX filter, exact block means).  What is restated:

  interp = 9 (gluoncv imresize -> mxnet.image._get_interp_method(9, (oh, ow, nh, nw))), OpenCV flag numbers:
      both sides enlarged -> 2 = cv2.INTER_CUBIC     both sides shrunk -> 3 = cv2.INTER_AREA
      anything else (mixed, or a side unchanged) -> 1 = cv2.INTER_LINEAR
  cv::resize on CV_8UC3 (modules/imgproc/src/resize.cpp):
      source coordinate of destination index d: f = (d + 0.5) * (ssize / dsize) - 0.5 as float,
      s = floor(f), f -= s (half-pixel centres)
      LINEAR   s < 0 -> (s, f) = (0, 0); s >= ssize-1 -> (ssize-1, 0); weights (1-f, f) -> short,
               round(w * 2048); rows: int sum of 2 taps; columns:
               uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)
      CUBIC    Keys kernel with A = -0.75 on taps s-1 .. s+2, tap indices clamped to [0, ssize-1];
               weights -> short, saturate(round(w * 2048)); rows: int sum of 4 taps; columns:
               saturate_uchar((sum of 4 + (1 << 21)) >> 22)
      AREA     integer scale factors: saturate_uchar(round_half_even(sum * (1.f / area))), except 2 x 2:
               (a + b + c + d + 2) >> 2 (the SIMD path of every x86 / aarch64 build);
               otherwise fractional pixel-area weights (computeResizeAreaTab) accumulated in float, rows first
Details that cannot be pinned from here: whether the OpenCV build contracts the float multiply-adds of the
AREA path, and the 2 x 2 special case on builds without SIMD.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def interp_method(oh, ow, nh, nw):
    """mxnet.image._get_interp_method(9, (oh, ow, nh, nw)) in OpenCV's numbering."""
    if nh > oh and nw > ow:
        return 2  # INTER_CUBIC
    if nh < oh and nw < ow:
        return 3  # INTER_AREA
    return 1      # INTER_LINEAR


def _sat_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)  # cvRound: to nearest even


def _src_coord(dsize, ssize):
    scale = float(ssize) / float(dsize)  # double, like cv::resize's scale_x
    f = ((np.arange(dsize, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32)


def _linear_tab(dsize, ssize):
    s, f = _src_coord(dsize, ssize)
    lo = s < 0
    f = np.where(lo, np.float32(0), f)
    s = np.where(lo, 0, s)
    hi = s >= ssize - 1
    f = np.where(hi, np.float32(0), f)
    s = np.where(hi, ssize - 1, s)
    idx = np.stack([s, np.minimum(s + 1, ssize - 1)], 1)
    w = np.stack([np.float32(1) - f, f], 1).astype(np.float32)
    return idx, _sat_short(w * np.float32(COEF_SCALE))


def _cubic_tab(dsize, ssize):
    s, x = _src_coord(dsize, ssize)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    w = np.stack([c0, c1, c2, c3], 1).astype(np.float32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, ssize - 1)
    return idx, _sat_short(w * np.float32(COEF_SCALE))


def resize_linear(img, nh, nw):
    h, w, _ = img.shape
    xi, xw = _linear_tab(nw, w)
    yi, yw = _linear_tab(nh, h)
    s = img.astype(np.int32)
    rows = s[:, xi[:, 0], :] * xw[None, :, 0, None] + s[:, xi[:, 1], :] * xw[None, :, 1, None]   # (h, nw, c) int
    s0, s1 = rows[yi[:, 0]], rows[yi[:, 1]]
    b0, b1 = yw[:, 0, None, None], yw[:, 1, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)  # uchar(): the value is within 0..255 by construction


def resize_cubic(img, nh, nw):
    h, w, _ = img.shape
    xi, xw = _cubic_tab(nw, w)
    yi, yw = _cubic_tab(nh, h)
    s = img.astype(np.int64)
    rows = sum(s[:, xi[:, k], :] * xw[None, :, k, None] for k in range(4))                     # (h, nw, c)
    val = sum(rows[yi[:, k]] * yw[:, k, None, None] for k in range(4))
    return np.clip((val + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS), 0, 255).astype(np.uint8)


def area_tab(ssize, dsize):
    """computeResizeAreaTab: list of (di, si, alpha float32) in OpenCV's order."""
    scale = float(ssize) / float(dsize)
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(img, nh, nw):
    h, w, c = img.shape
    sx, sy = float(w) / nw, float(h) / nh
    ix, iy = int(round(sx)), int(round(sy))
    if abs(sx - ix) < np.finfo(np.float64).eps and abs(sy - iy) < np.finfo(np.float64).eps:
        blocks = img[:nh * iy, :nw * ix].astype(np.int32).reshape(nh, iy, nw, ix, c).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((blocks + 2) >> 2).astype(np.uint8)
        scale = np.float32(1.0) / np.float32(ix * iy)
        return np.clip(np.rint(blocks.astype(np.float32) * scale), 0, 255).astype(np.uint8)
    xtab, ytab = area_tab(w, nw), area_tab(h, nh)
    s = img.astype(np.float32)

    def passes(tab, dsize):
        """The table regrouped by destination index: pass k holds the k-th entry of every destination cell (its
        source index and weight, weight 0 where the cell has fewer entries), so adding pass after pass performs,
        for each destination element, exactly the sequential float additions of OpenCV's loop over the table."""
        per = [[] for _ in range(dsize)]
        for di, si, alpha in tab:
            per[di].append((si, alpha))
        depth = max(len(p) for p in per)
        idx = np.zeros((depth, dsize), np.int64)
        wgt = np.zeros((depth, dsize), np.float32)
        for d, p in enumerate(per):
            for k, (si, alpha) in enumerate(p):
                idx[k, d], wgt[k, d] = si, alpha
        return idx, wgt, np.array([len(p) for p in per])

    xi, xw, _ = passes(xtab, nw)
    buf = np.zeros((h, nw, c), np.float32)                 # every source row, reduced along x
    for k in range(xi.shape[0]):
        buf = buf + s[:, xi[k], :] * xw[k][None, :, None]
    yi, yw, ycnt = passes(ytab, nh)
    out = yw[0][:, None, None] * buf[yi[0]]                # first contributing row: sum = beta * buf
    for k in range(1, yi.shape[0]):
        live = (k < ycnt)[:, None, None]
        out = np.where(live, out + yw[k][:, None, None] * buf[yi[k]], out)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def imresize(img, nw, nh, interp=9):
    """gluoncv imresize(src, w, h, interp=9) on one (h, w, 3) uint8 frame."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    m = interp_method(h, w, nh, nw) if interp == 9 else interp
    if (h, w) == (nh, nw):
        return img.copy()           # cv::resize: same size is a copy
    return {1: resize_linear, 2: resize_cubic, 3: resize_area}[m](img, nh, nw)


MEAN = np.array((0.485, 0.456, 0.406), np.float32)
STD = np.array((0.229, 0.224, 0.225), np.float32)


def inference_transform(frames, width, height, mean=MEAN, std=STD):
    """YOLO3VideoInferenceTransform.__call__ (transforms.py:316-350) for (k, h, w, 3) uint8 frames:
    resize -> to_tensor (HWC uint8 -> CHW float / 255) -> normalize."""
    frames = np.asarray(frames)
    if frames.ndim == 3:
        frames = frames[None]
    out = np.stack([imresize(f, width, height) for f in frames])
    x = out.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)), out
