"""-m gpu: the device prefetch target generator (csrc/targets.hip through vy_prefetch_targets) against
the CPU restatement of the reference loop (oracle/targets_oracle.py, yolo_target.py:31-148) and a
hand-derived case.  Bar: every tensor bit-identical except the two log() scale targets (device:
include/vy_math.h log, oracle: numpy's) which must agree to 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device(C, size, gt, ids, mix=None):
    from videoyolo_amd import targets
    out = targets.YOLOV3PrefetchTargetGenerator(C)(size, size, gt, ids, mix, device="cuda:0")
    return [t.cpu().numpy() for t in out]


def _check(got, want):
    names = ("objectness", "centers", "scales", "weights", "classes")
    for n, g, w in zip(names, got, want):
        assert g.shape == w.shape, n
        if n == "scales":
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-6, err_msg=n)
        else:
            assert np.array_equal(g, w), n


@pytest.mark.parametrize("size,C,B,m,pad", [(64, 3, 3, 6, 9), (416, 20, 16, 8, 12), (608, 30, 4, 40, 50)])
def test_device_targets_match_the_reference_loop(size, C, B, m, pad):
    from oracle import targets_oracle as T
    rng = np.random.default_rng(size)
    gt, ids = T.synthetic_gt(B, size, C, m=m, seed=size, pad_to=pad)
    gt[1, 2] = gt[1, 1]                # same (cell, anchor) slot twice: the later box wins every field
    ids[1, 2] = (ids[1, 1] + 1) % C
    mix = rng.uniform(0.3, 1.0, (B, pad, 1)).astype(np.float32)
    _check(_device(C, size, gt, ids, mix), T.prefetch_targets(C, size, size, gt, ids, mix))
    _check(_device(C, size, gt, ids), T.prefetch_targets(C, size, size, gt, ids))


def test_device_targets_edge_cases():
    from oracle import targets_oracle as T
    C, size = 5, 96
    # no gt rows at all, only padding, and a valid row AFTER a padded one (ignored: the loop breaks)
    empty = np.zeros((2, 0, 4), np.float32), np.zeros((2, 0, 1), np.float32)
    _check(_device(C, size, *empty), T.prefetch_targets(C, size, size, *empty))
    gt = np.full((2, 4, 4), -1, np.float32)
    ids = np.full((2, 4, 1), -1, np.float32)
    _check(_device(C, size, gt, ids), T.prefetch_targets(C, size, size, gt, ids))
    gt[0, 0] = [4, 4, 30, 50]
    ids[0, 0] = 2
    gt[0, 2] = [40, 40, 90, 90]        # after the padded row 1: must not be written
    ids[0, 2] = 1
    gt[1, 0] = [0, 0, 95, 95]          # box of the whole image: stride-32 anchors
    ids[1, 0] = 4
    gt[1, 1] = [10, 10, 10.5, 10.25]   # sub-pixel box: max(w, 1) in the scale target
    ids[1, 1] = 0
    got = _device(C, size, gt, ids)
    _check(got, T.prefetch_targets(C, size, size, gt, ids))
    assert got[0][0].sum() == 1 and got[0][1].sum() == 2


def test_device_cell_index_follows_float64_promotion():
    """Centres where int(fp32 quotient) != int(float64 quotient): the kernel must pick the float64 cell
    (yolo_target.py:115-119 under the reference's NumPy 1.x scalar promotion) — an index, so exact."""
    from conftest import check_float64_cell_case, float64_cell_case
    from oracle import targets_oracle as T
    size, gt, ids, expected = float64_cell_case()
    got = _device(4, size, gt, ids)
    check_float64_cell_case(got, expected)
    _check(got, T.prefetch_targets(4, size, size, gt, ids))


def test_device_targets_by_hand():
    """One gt box (8,8)-(40,24), class 1, in a 64x64 image: w=32, h=16 -> best zero-centred IoU among
    the nine anchors is (33,23), the last one -> stride-8 scale (8x8 cells), anchor slot 2; centre
    (24,16) -> cell (x=3, y=2), tx = ty = 0; scale = log(32/33), log(16/23); weight = 2 - 512/4096."""
    gt = np.full((1, 2, 4), -1, np.float32)
    gt[0, 0] = [8, 8, 40, 24]
    ids = np.full((1, 2, 1), -1, np.float32)
    ids[0, 0] = 1
    obj, ctr, scl, wt, cls = _device(3, 64, gt, ids)
    n = 3 * (4 + 16 + 64)
    idx = 3 * (4 + 16) + (2 * 8 + 3) * 3 + 2
    assert obj.shape == (1, n, 1) and obj.sum() == 1 and obj[0, idx, 0] == 1
    assert np.array_equal(ctr[0, idx], [0, 0])
    assert np.allclose(scl[0, idx], np.log([32 / 33, 16 / 23]), atol=1e-6)
    assert np.array_equal(wt[0, idx], [1.875, 1.875]) and list(cls[0, idx]) == [0, 1, 0]
    assert (cls[0, np.arange(n) != idx] == -1).all() and (scl[0, np.arange(n) != idx] == 0).all()


def test_device_targets_feed_the_training_call():
    """Targets built on the device go straight into net(x, ...) under autograd.record(): same losses as
    with the host-built targets."""
    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import autograd, targets
    C, B, S = 4, 2, 64
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.initialize(init="synthetic", seed=3)
    net.collect_params().reset_ctx("cuda:0")
    x = torch.randn((B, 3, S, S), device="cuda:0")
    gt, ids = targets.synthetic_gt(B, S, C, m=3, seed=5)
    gen = targets.YOLOV3PrefetchTargetGenerator(C)
    host = gen(S, S, gt, ids)
    dev = gen(S, S, torch.as_tensor(gt).cuda(), torch.as_tensor(ids).cuda())
    with autograd.record():
        l_host = [t.cpu().numpy() for t in net(x, gt, *host)]
    with autograd.record():
        l_dev = [t.cpu().numpy() for t in net(x, gt, *dev)]
    for a, b in zip(l_host, l_dev):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4)
