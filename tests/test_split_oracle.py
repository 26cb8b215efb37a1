"""CPU: the restatement of the split-fp32 arithmetic (oracle/split_oracle.py) against its own exactness properties and
against float64 — what pins the checker that tests/test_gpu_split.py::test_kernel_computes_the_six_products uses."""
import numpy as np

from oracle import split_oracle as S


def _vals(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32) * np.exp2(rng.integers(-12, 12, n)).astype(np.float32)
    x[:4] = [0.0, 1.0, -1.0, 3.0e-5]
    return x


def test_bf16_rne_hand_cases():
    one = np.float32(1.0)
    ulp = np.float32(2.0 ** -7)          # bf16 spacing at 1.0
    assert S.bf16_rne(one + ulp / 2)[()] == one                  # tie -> even (1.0 has an even significand)
    assert S.bf16_rne(one + ulp * 1.5)[()] == one + 2 * ulp      # tie -> even (upwards)
    assert S.bf16_rne(one + ulp * 0.51)[()] == one + ulp
    assert S.bf16_rne(np.float32(-3.0))[()] == np.float32(-3.0)
    v = S.bf16_rne(_vals(4096, 1))
    assert (v.view(np.uint32) & 0xFFFF == 0).all()               # only the upper 16 bits are set


def test_three_planes_are_exact():
    x = _vals(200_000, 2)
    h, m, l = S.split3(x)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()
    for p in (h, m, l):
        assert (p.view(np.uint32) & 0xFFFF == 0).all()


def test_six_products_are_the_product_to_2_pow_minus_24():
    x, w = _vals(100_000, 3), _vals(100_000, 4)
    hx, mx, lx = [p.astype(np.float64) for p in S.split3(x)]
    hw, mw, lw = [p.astype(np.float64) for p in S.split3(w)]
    planes_x, planes_w = (hx, mx, lx), (hw, mw, lw)
    six = sum(planes_x[a] * planes_w[b] for a, b in S.SIX)
    exact = x.astype(np.float64) * w.astype(np.float64)
    nz = exact != 0
    assert (np.abs(six - exact)[nz] <= np.abs(exact[nz]) * 2.0 ** -24).all()
    five = sum(planes_x[a] * planes_w[b] for a, b in S.FIVE)
    assert np.abs(five - exact)[nz].max() > 100 * np.abs(six - exact)[nz].max()      # a dropped product is visible


def test_conv_reference_against_float64():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 32, 9, 9)).astype(np.float32)
    w = (rng.standard_normal((64, 32, 3, 3)) / 17.0).astype(np.float32)
    import torch
    import torch.nn.functional as F
    exact = F.conv2d(torch.from_numpy(x.astype(np.float64)), torch.from_numpy(w.astype(np.float64)), None, 1, 1).numpy()
    ref = S.conv_split_ref(x, w, 1, 1)
    scale = S.abs_product_sum(x, w, 1, 1)
    assert (np.abs(ref - exact) <= scale * 2.0 ** -24).all()
    assert np.abs(S.conv_split_ref(x, w, 1, 1, S.FIVE) - exact).max() > 50 * np.abs(ref - exact).max()
