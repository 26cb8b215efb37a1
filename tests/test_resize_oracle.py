"""-m "not gpu": the resize oracle (oracle/resize_oracle.py, a from-memory restatement of gluoncv imresize(interp=9)
-> OpenCV cv::resize on uint8 frames; PARITY UNPINNED) against independent implementations where their
definitions coincide:

  CUBIC   torch.nn.functional.interpolate(mode='bicubic', align_corners=False): Keys A = -0.75, half-pixel
          centres, clamped taps — the same continuous definition; OpenCV quantises the weights to 11 bits,
          so agreement is +-1 grey level
  LINEAR  torch bilinear (align_corners=False), +-1 level
  AREA    exact block means (and PIL's BOX filter) for integer factors; the overlap-length definition of area
          resampling in float64 otherwise, +-1 level
plus hand cases of the method selection and of the fixed-point rounding rules."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resize_oracle as R


def _img(h, w, seed=0):
    rng = np.random.default_rng(seed)
    # smooth + noise: exercises ties less pathologically than pure noise, still full range
    y, x = np.mgrid[0:h, 0:w]
    base = 127 + 100 * np.sin(x / 7.0)[..., None] * np.cos(y / 5.0)[..., None] * np.array([1, 0.5, -1])
    return np.clip(base + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)


def _torch_resize(img, nh, nw, mode):
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    o = F.interpolate(t, size=(nh, nw), mode=mode, align_corners=False)
    return o[0].permute(1, 2, 0).numpy()


def test_interp_9_selects_by_direction():
    assert R.interp_method(480, 640, 608, 800) == 2       # both enlarged: cubic
    assert R.interp_method(720, 1280, 416, 416) == 3      # both shrunk: area
    assert R.interp_method(300, 800, 416, 416) == 1       # mixed: bilinear
    assert R.interp_method(416, 500, 416, 416) == 1       # one side unchanged: bilinear


@pytest.mark.parametrize("src,dst", [((37, 53), (64, 96)), ((120, 160), (416, 416)), ((200, 301), (608, 608))])
def test_cubic_agrees_with_torch_bicubic(src, dst):
    img = _img(*src, seed=1)
    got = R.imresize(img, dst[1], dst[0]).astype(np.float64)
    ref = np.clip(np.rint(_torch_resize(img, dst[0], dst[1], "bicubic")), 0, 255)
    d = np.abs(got - ref)
    assert d.max() <= 1 and d.mean() < 0.05, (d.max(), d.mean())


@pytest.mark.parametrize("src,dst", [((50, 200), (96, 96)), ((96, 64), (64, 128))])
def test_linear_agrees_with_torch_bilinear(src, dst):
    img = _img(*src, seed=2)
    assert R.interp_method(src[0], src[1], dst[0], dst[1]) == 1
    got = R.imresize(img, dst[1], dst[0]).astype(np.float64)
    ref = _torch_resize(img, dst[0], dst[1], "bilinear")
    d = np.abs(got - ref)
    assert d.max() <= 1.0 and d.mean() < 0.35, (d.max(), d.mean())   # the >> 4, >> 16 steps truncate


def test_area_integer_factors_are_block_means():
    img = _img(96, 144, seed=3)
    got = R.imresize(img, 48, 32)                          # 3 x 3 blocks
    blocks = img.astype(np.float64).reshape(32, 3, 48, 3, 3).mean(axis=(1, 3))
    assert np.abs(got - blocks).max() <= 0.5 + 1e-6
    got2 = R.imresize(img, 72, 48)                         # 2 x 2: (a + b + c + d + 2) >> 2, ties round UP
    s = img.astype(np.int64).reshape(48, 2, 72, 2, 3).sum(axis=(1, 3))
    assert np.array_equal(got2, (s + 2) >> 2)
    tie = np.zeros((2, 2, 3), np.uint8)
    tie[0, 0] = 2                                          # sum 2 -> 0.5: the 2 x 2 path gives 1
    assert R.imresize(tie, 1, 1)[0, 0, 0] == 1
    tie3 = np.zeros((3, 3, 3), np.uint8)
    tie3[0, 0] = 9
    tie3[1, 1] = 4                                         # sum 13 / 9 = 1.44 -> 1
    assert R.imresize(tie3, 1, 1)[0, 0, 0] == 1


def _overlap_matrix(ssize, dsize):
    """w[d, s] = |[s, s+1) intersect [d*scale, (d+1)*scale)| / scale: the definition of area resampling."""
    scale = ssize / dsize
    d = np.arange(dsize)[:, None] * scale
    s = np.arange(ssize)[None, :]
    return np.clip(np.minimum(s + 1, d + scale) - np.maximum(s, d), 0, None) / scale


@pytest.mark.parametrize("src,dst", [((480, 640), (416, 416)), ((97, 131), (64, 96)), ((720, 1280), (608, 608))])
def test_area_agrees_with_the_overlap_definition(src, dst):
    """Fractional scale factors: destination pixel = mean of the source over its (real-valued) footprint, written
    as two overlap-length matrices in float64 — an independent formulation of what computeResizeAreaTab encodes
    (which also drops overlaps below 1e-3 of a pixel: hence +-1 level, not exact)."""
    img = _img(*src, seed=4)
    got = R.imresize(img, dst[1], dst[0]).astype(np.float64)
    wy, wx = _overlap_matrix(src[0], dst[0]), _overlap_matrix(src[1], dst[1])
    ref = np.tensordot(wy, np.tensordot(img.astype(np.float64), wx, axes=([1], [1])), axes=([1], [0]))  # (nh, c, nw)
    ref = ref.transpose(0, 2, 1)
    d = np.abs(got - ref)
    assert d.max() <= 0.5 + 2e-2 and np.abs(got - np.rint(ref)).mean() < 0.01, (d.max(), np.abs(got - np.rint(ref)).mean())
    # the fractional-area weights of every destination cell sum to 1
    for ssize, dsize in ((src[0], dst[0]), (src[1], dst[1])):
        acc = np.zeros(dsize)
        for di, si, a in R.area_tab(ssize, dsize):
            assert 0 <= si < ssize
            acc[di] += a
        np.testing.assert_allclose(acc, 1.0, atol=2e-3)


def test_area_integer_factor_agrees_with_pil_box():
    from PIL import Image
    img = _img(96, 144, seed=5)
    got = R.imresize(img, 48, 32).astype(np.int64)         # 3 x 3
    ref = np.asarray(Image.fromarray(img).resize((48, 32), Image.BOX)).astype(np.int64)
    assert np.abs(got - ref).max() <= 1


def test_fixed_point_rules_by_hand():
    """2 -> 4 columns, linear... is cubic (enlarge); take the linear path explicitly: one row [0, 200] to 4
    columns.  f = (d + .5) * .5 - .5 = -.25, .25, .75, 1.25 -> (s, f) = (0,0) (0,.25) (0,.75) (1,0);
    weights * 2048 = (2048,0) (1536,512) (512,1536) (2048,0); rows = 0, 102400, 307200, 409600;
    one source row twice (same clamp in y): b = (2048, 0): ((2048 * (S >> 4)) >> 16 + 2) >> 2 = 0, 50, 150, 200."""
    img = np.zeros((1, 2, 3), np.uint8)
    img[0, 1] = 200
    out = R.resize_linear(img, 1, 4)
    assert out[0, :, 0].tolist() == [0, 50, 150, 200]
    # cubic with A = -0.75 at x = 0.25: weights (-0.10546875, 0.87890625, 0.26171875, -0.03515625) * 2048 =
    # (-216, 1800, 536, -72); taps clamp at the border
    idx, w = R._cubic_tab(4, 2)
    assert w[1].tolist() == [-216, 1800, 536, -72] and idx[1].tolist() == [0, 0, 1, 1]
    assert w.sum(axis=1).tolist() == [2048] * 4
    flat = np.full((5, 7, 3), 93, np.uint8)                      # constants survive every method exactly
    for nh, nw in ((11, 13), (3, 4), (5, 20), (2, 3)):
        assert (R.imresize(flat, nw, nh) == 93).all()


def test_inference_transform_matches_the_formula():
    frames = np.stack([_img(60, 80, seed=s) for s in range(2)])
    x, resized = R.inference_transform(frames, 96, 96)
    assert x.shape == (2, 3, 96, 96) and resized.shape == (2, 96, 96, 3)
    want = (resized[1, 5, 7, 2] / np.float32(255.0) - np.float32(0.406)) / np.float32(0.225)
    assert x[1, 2, 5, 7] == np.float32(want)
