"""-m gpu: the OPT-IN split-fp32 conv mode (`net.set_conv_mode('split_bf16x3')`, csrc/conv_split.hip: every fp32
operand cut exactly into three bf16 numbers, six partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation)
against the CPU oracle and against float64.

The default mode ('exact') is the parity path and is bit-identical to the oracle (tests/test_gpu_parity.py).  The split
mode is NOT: the matrix core sums a 16-channel group in its own tree and three of the nine partial products (< 2^-25 of
the product) are left out.  The bars here are therefore tolerances, and each is written next to what was observed:

  raw head tensors   |split - oracle| <= HEAD_TOL   (observed: see test output; the oracle's own distance to float64
                                                     is 8e-6 at 416 / 608, tests/test_oracle_vs_torch.py)
  scores             <= 1e-4 absolute               (north_star's bar)
  boxes              <= 1e-4 * max(1, w, h)         (the bar tests/test_oracle_vs_torch.py uses between independently
                                                     ordered fp32 evaluations: exp(raw) * anchor makes the error of a
                                                     coordinate relative to the box extent)
  NMS kept rows      identical to the oracle's, except where two candidates' ORACLE scores lie within NEAR_TIE = 5e-6 of
                     each other (twice the largest score deviation of the split path, itself 40x inside north_star's 1e-4):
                     those may come out in the other order.  No arithmetic that is not bit-equal to the oracle's can promise
                     the order of two scores closer than its own error.  `_check_keep` does not wave such cases through:
                     the SET of kept rows per image must be identical, every displaced row's oracle score must lie within
                     NEAR_TIE of the row the oracle has in that slot, at most 4 slots per fixture may differ, and every
                     exception is printed.  Observed (the list moves with the summation order, i.e. with the tile /
                     k-split the cost model picks): 1 x 416 x 416, seed 233: rows 10505 / 11591 (scores 0.885419488 /
                     0.885419548, one ulp apart) swap output slots 84 / 85; 30 classes, nms (0.6, 1000, 300): rows 6298 /
                     9541 (3.6e-7 apart) swap.  Everything else — 12 fixtures — identical.
"""
import os

import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_supported_launch_on_the_split_kernel(monkeypatch):
    """The product sends a launch to the split kernel only where its cost model predicts a gain (large launches: the
    test shapes here are far too small).  The parity tests force every supported launch onto it."""
    monkeypatch.setenv("VY_SPLIT_ALWAYS", "1")

HEAD_TOL = 3e-5
TOL = 1e-4
NEAR_TIE = 5e-6   # twice the largest |score - oracle score| seen on any fixture (2.6e-6); north_star's score bar is 1e-4


def _check_keep(keep, r_keep, det, label, capsys=None):
    """Kept rows equal to the oracle's up to near-ties (module docstring).  det: the oracle's (B, N*C, 6) detections.
    Returns the exceptions [(image, slot, row here, row in the oracle, score gap)]."""
    exc = []
    for b in range(keep.shape[0]):
        assert sorted(keep[b].tolist()) == sorted(r_keep[b].tolist()), "%s: image %d keeps a different SET of rows" % (label, b)
        for j in np.nonzero(keep[b] != r_keep[b])[0]:
            g, w = int(keep[b, j]), int(r_keep[b, j])
            gap = abs(float(det[b, g, 1]) - float(det[b, w, 1]))
            assert gap <= NEAR_TIE, "%s: image %d slot %d: rows %d / %d are %.3e apart — not a near-tie" % (label, b, j, g, w, gap)
            exc.append((b, int(j), g, w, gap))
    if exc and capsys is not None:
        with capsys.disabled():
            print("\n[%s] near-tie exceptions (image, slot, row, oracle's row, oracle score gap): %s" % (label, exc))
    return exc


def _net(classes, params, mode="split_bf16x3", **kw):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(classes, pretrained_base=False, **kw)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    net.set_conv_mode(mode)
    return net


def _oracle(params, ncls=20, **kw):
    from oracle import yolo3_oracle as O
    return O.OracleYolo3(ncls, params, **kw)


def _split_launches(net, x):
    """names of the launches of one profiled forward"""
    return [r[0] for r in net.profile(x)]


def test_split_mode_really_runs_the_bf16_kernel(voc_classes, synth20):
    """The mode must change which kernel runs (70 of the 75 convs: all but the stem, the 64 -> 32 bottleneck and the three
    prediction convs), and must not be bit-equal to the exact path by accident of a silent fallback."""
    x = frames(2, 96)
    net = _net(voc_classes, synth20)
    names = _split_launches(net, x)
    n_split = sum("|split" in n for n in names)
    assert n_split == 70, names
    net(x)
    h_split = [net.read_head(i).cpu().numpy() for i in range(3)]
    net.set_conv_mode("exact")
    assert not any("|split" in n for n in _split_launches(net, x))
    net(x)
    h_exact = [net.read_head(i).cpu().numpy() for i in range(3)]
    ref = _oracle(synth20).raw_heads(x)
    for i in range(3):
        assert np.array_equal(h_exact[i], ref[i])          # back on the parity path: bit-exact again
        assert not np.array_equal(h_split[i], ref[i])      # the split path is a different summation
        assert np.abs(h_split[i] - ref[i]).max() <= HEAD_TOL


@pytest.mark.parametrize("batch,size", [(2, 96), (1, 64), (3, 160), (1, 416)])
def test_split_heads_close_to_oracle(voc_classes, synth20, batch, size, capsys):
    x = frames(batch, size)
    net = _net(voc_classes, synth20)
    net(x)
    ref = _oracle(synth20).raw_heads(x)
    worst = 0.0
    for i in range(3):
        got = net.read_head(i).cpu().numpy()
        assert got.shape == ref[i].shape and np.isfinite(got).all()
        worst = max(worst, float(np.abs(got - ref[i]).max()))
    with capsys.disabled():
        print("\n[split %dx%d^2] max |head - oracle| = %.3e (bar %.0e)" % (batch, size, worst, HEAD_TOL))
    assert worst <= HEAD_TOL


@pytest.mark.parametrize("name", ["stages.0.3", "stages.0.4.body.1", "stages.0.14.body.1", "stages.1.0", "stages.1.8.body.1",
                                  "stages.2.4.body.1", "yolo_blocks.0.tip", "yolo_blocks.1.body.1", "yolo_blocks.2.tip"])
def test_split_intermediate_cells(voc_classes, synth20, name):
    """Layer taps of cells the split kernel computes: 3x3 stride 1 / stride 2, + residual, into a concat plane."""
    x = frames(2, 64, seed=7)
    net = _net(voc_classes, synth20)
    net.keep_activations()
    net(x)
    got = net.read_activation(name).cpu().numpy()
    net.set_conv_mode("exact")
    net.keep_activations()
    net(x)
    want = net.read_activation(name).cpu().numpy()   # == the oracle's tap, tests/test_gpu_parity.py
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= HEAD_TOL * max(1.0, float(np.abs(want).max()))
    assert not np.array_equal(got, want)


@pytest.mark.parametrize("name,src,k,stride", [("stages.0.1", "stages.0.0", 3, 2), ("stages.1.0", "stages.0.14.body.1", 3, 2),
                                               ("yolo_blocks.0.tip", "yolo_blocks.0.body.4", 3, 1),
                                               ("yolo_blocks.0.body.2", "yolo_blocks.0.body.1", 1, 1)])
def test_kernel_computes_the_six_products(voc_classes, synth20, name, src, k, stride, capsys):
    """One cell against oracle/split_oracle.py: the cell's actual fp32 input (the tap of the cell before it) is cut into
    three bf16 planes on the CPU, the SIX partial products are summed in float64, affine + leaky applied — the kernel may
    differ from that only by the rounding of its fp32 accumulation (bar: 2^-22 of sum |x w| per output, observed far
    inside).  The same reference with ONE product left out (h_x l_w) must lie well outside what the kernel delivers:
    the test would notice a missing product.  K = 288, 1024, 2304, 4608; stride 1 and 2; 3x3 and 1x1."""
    from oracle import split_oracle as S
    from oracle import yolo3_oracle as O
    x = frames(2, 64, seed=7)
    net = _net(voc_classes, synth20)
    net.keep_activations()
    net(x)
    a_in = net.read_activation(src).cpu().numpy()
    got = net.read_activation(name).cpu().numpy().astype(np.float64)
    w = synth20[name + ".0.weight"]
    sc, sh = O.bn_fold(synth20[name + ".1.gamma"], synth20[name + ".1.beta"], synth20[name + ".1.running_mean"],
                       synth20[name + ".1.running_var"])
    sc, sh = sc.astype(np.float64).reshape(1, -1, 1, 1), sh.astype(np.float64).reshape(1, -1, 1, 1)

    def finish(z):
        y = z * sc + sh
        return np.where(y > 0, y, 0.1 * y)
    six = finish(S.conv_split_ref(a_in, w, stride, k // 2))
    five = finish(S.conv_split_ref(a_in, w, stride, k // 2, S.FIVE))
    bound = S.abs_product_sum(a_in, w, stride, k // 2) * np.abs(sc) * 2.0 ** -22 + 1e-6
    err6, err5 = np.abs(got - six), np.abs(got - five)
    with capsys.disabled():
        print("\n[%s K=%d] max |kernel - six products| %.2e (bound min %.2e) | against five products %.2e" %
              (name, w.shape[1] * k * k, err6.max(), bound.min(), err5.max()))
    assert (err6 <= bound).all()
    assert err5.max() > 4 * err6.max()


@pytest.mark.parametrize("batch,h,w", [(2, 96, 96), (3, 160, 160), (1, 128, 224), (1, 416, 416)])
def test_winograd_cells_forced_everywhere(voc_classes, synth20, batch, h, w, monkeypatch, capsys):
    """conv_wino.hip — the long-K 3x3 stride-1 cells as a 1-D Winograd F(2, 3) on the same split arithmetic — is chosen by
    the product only for launches of >= 1024 blocks; VY_SPLIT_WINO=2 sends EVERY supported cell through it (the 31 cells
    of 128 output channels and more, residual adds included, at widths 3 ... 52: odd widths exercise the pair without a
    second pixel).  Heads and layer taps within the split mode's bars, kept rows up to near-ties; =0 switches it off."""
    monkeypatch.setenv("VY_SPLIT_WINO", "2")
    rng = np.random.default_rng(h * 1000 + w)
    x = rng.standard_normal((batch, 3, h, w)).astype(np.float32)
    net = _net(voc_classes, synth20)
    names = _split_launches(net, x)
    assert sum("|wino" in n for n in names) == 31, [n for n in names if "|wino" in n]
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    orc = _oracle(synth20)
    ref = orc.raw_heads(x)
    worst = max(float(np.abs(net.read_head(i).cpu().numpy() - ref[i]).max()) for i in range(3))
    with capsys.disabled():
        print("\n[winograd %dx%dx%d] max |head - oracle| = %.3e (bar %.0e)" % (batch, h, w, worst, HEAD_TOL))
    assert worst <= HEAD_TOL
    r = orc(x)
    exc = _check_keep(keep, r[3], orc.detections(x), "winograd %dx%dx%d" % (batch, h, w), capsys)
    assert len(exc) <= 4
    np.testing.assert_allclose(np.sort(scores, 1), np.sort(r[1], 1), rtol=0, atol=TOL)
    monkeypatch.setenv("VY_SPLIT_WINO", "0")
    assert not any("|wino" in n for n in _split_launches(net, x))


@pytest.mark.parametrize("batch,size,obj_bias", [(2, 96, 0.0), (2, 128, -3.0), (1, 416, 0.0), (2, 160, -5.0)])
def test_split_detections_match_oracle(voc_classes, batch, size, obj_bias, capsys):
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(20), seed=233, obj_bias=obj_bias)
    x = frames(batch, size)
    net = _net(voc_classes, params)
    net.set_nms(0.45, 400, 100)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r_ids, r_scores, r_bboxes, r_keep = _oracle(params)(x)
    same = np.array_equal(keep, r_keep)
    with capsys.disabled():
        print("\n[split det %dx%d^2 bias %g] kept rows identical: %s; max |score diff| %.2e" %
              (batch, size, obj_bias, same, float(np.abs(scores - r_scores).max())))
    exc = _check_keep(keep, r_keep, _oracle(params).detections(x), "det %dx%d bias %g" % (batch, size, obj_bias), capsys)
    if exc:  # rows in swapped slots: compare slot-independent quantities only
        np.testing.assert_allclose(np.sort(scores, 1), np.sort(r_scores, 1), rtol=0, atol=TOL)
        assert len(exc) <= 4
        return
    assert np.array_equal(ids, r_ids)
    np.testing.assert_allclose(scores, r_scores, rtol=0, atol=TOL)
    fin = np.isfinite(r_bboxes)
    assert np.array_equal(np.isfinite(bboxes), fin)
    ext = np.maximum(1.0, np.maximum(r_bboxes[..., 2] - r_bboxes[..., 0], r_bboxes[..., 3] - r_bboxes[..., 1]))
    err = np.where(fin, np.abs(bboxes - r_bboxes), 0.0).max(-1)
    assert (err <= TOL * ext).all(), float((err / ext).max())


def test_default_policy_keeps_small_launches_exact(voc_classes, synth20, monkeypatch):
    """Without VY_SPLIT_ALWAYS the per-launch cost model decides (csrc/conv_cost_model.h): of a single 416 x 416 frame's
    70 eligible launches only those that pay go to the split kernel (the long-K ones as split-K, "k<S>" in the label;
    the short 1x1 launches stay on the exact kernel's 64 x 64 tiles), 16 frames send nearly all of them.  Either way the
    result obeys the bars."""
    monkeypatch.delenv("VY_SPLIT_ALWAYS")
    net = _net(voc_classes, synth20)
    n1 = sum("|split" in n or "|wino" in n for n in _split_launches(net, frames(1, 416)))
    x16 = frames(16, 416, seed=3)
    n16 = sum("|split" in n or "|wino" in n for n in _split_launches(net, x16))
    assert n1 < n16 and n1 <= 45 and n16 >= 50, (n1, n16)
    assert any("k" in n.split("|split")[1] for n in _split_launches(net, frames(1, 416)) if "|split" in n)   # split-K is in use
    net(x16[:2])
    ref = _oracle(synth20).raw_heads(x16[:2])
    for i in range(3):
        assert np.abs(net.read_head(i).cpu().numpy() - ref[i]).max() <= HEAD_TOL


def test_split_30_classes_and_nms_settings(capsys):
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(30)]
    params = init.synthetic_params(O.param_shapes(30), seed=5)
    x = frames(2, 96, seed=3)
    net = _net(classes, params)
    for thr, topk, post in [(0.45, 400, 100), (0.3, 50, 20), (0.6, 1000, 300)]:
        net.set_nms(thr, topk, post)
        ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
        orc = _oracle(params, 30, nms_thresh=thr, nms_topk=topk, post_nms=post)
        r = orc(x)
        exc = _check_keep(keep, r[3], orc.detections(x), "30 classes nms %s" % ((thr, topk, post),), capsys)
        assert len(exc) <= 4
        if not exc:
            assert np.array_equal(ids, r[0])
            np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)


@pytest.mark.parametrize("size", [416, 608])
def test_split_distance_to_float64(synth20, voc_classes, size, capsys):
    """One frame at BASELINE's sizes: both conv modes against the float64 torch model of tests/test_oracle_vs_torch.py.
    The split path must be about as close to float64 as the exact fp32 chain is (bar: within 3x of it, and 1e-4)."""
    import torch
    from test_oracle_vs_torch import TorchYolo3F64
    x = frames(1, size, seed=233)
    with torch.no_grad():
        h64 = [t.numpy() for t in TorchYolo3F64(20, synth20).forward_heads(x.astype(np.float64))]
    dist = {}
    for mode in ("exact", "split_bf16x3"):
        net = _net(voc_classes, synth20, mode=mode)
        net(x)
        dist[mode] = max(float(np.abs(net.read_head(i).cpu().numpy() - h64[i]).max()) for i in range(3))
    with capsys.disabled():
        print("\n[%d^2] max |head - float64|: exact fp32 chain %.3e, split bf16x3 %.3e" %
              (size, dist["exact"], dist["split_bf16x3"]))
    assert dist["split_bf16x3"] <= 1e-4
    assert dist["split_bf16x3"] <= 3 * dist["exact"]


def test_split_weights_follow_parameter_writes(voc_classes, synth20):
    """The pre-split weight images are a cache of the parameter buffer: set_data / load must invalidate it."""
    x = frames(1, 64, seed=11)
    net = _net(voc_classes, synth20)
    net(x)
    h0 = net.read_head(0).cpu().numpy()
    p = net.collect_params()["stages.2.4.body.1.0.weight"]
    w = p.data()
    p.set_data(w * 0.5)
    net(x)
    h1 = net.read_head(0).cpu().numpy()
    assert np.abs(h1 - h0).max() > 1e-3          # the new weights were used
    p.set_data(w)
    net(x)
    assert np.array_equal(net.read_head(0).cpu().numpy(), h0)   # deterministic: same images, same result


def test_split_hybridized_and_two_streams(voc_classes, synth20):
    """hipGraph replay and the two-stream twin run the split kernels too, with identical results."""
    x = frames(4, 96, seed=5)
    net = _net(voc_classes, synth20)
    a = [t.cpu().numpy() for t in net(x, return_index=True)]
    b = [t.cpu().numpy() for t in net.detect_two_streams(x, return_index=True)]
    net.hybridize()
    net(x)
    c = [t.cpu().numpy() for t in net(x, return_index=True)]
    for u, v, w in zip(a, b, c):
        assert np.array_equal(u, v) and np.array_equal(u, w)


def _train_case(C, B, S):
    from videoyolo_amd import init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(C), seed=11)
    x = frames(B, S, seed=5)
    gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=3, seed=2, pad_to=5)
    tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
    return params, x, gt_boxes, tg


def _step(net, x, gt_boxes, tg):
    from videoyolo_amd import autograd
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    grads = {p.name: net.grad(p.name) for p in net.collect_params().values() if p.trainable}
    return [l.cpu().numpy() for l in losses], grads


def _rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


@pytest.mark.parametrize("C,B,S", [(20, 2, 256), (4, 2, 128)])
def test_split_training_step_within_the_steps_own_sensitivity(C, B, S, capsys):
    """conv mode 'split_bf16x3' in TRAINING: the recorded forward (raw conv + per-tile batch statistics in double) and
    the data gradients run on the split kernel; weight gradients stay exact.  Losses: 1e-4 against the oracle, as for the
    exact path.  Gradients need a different yardstick than the exact path's 2e-3 of the tensor's max: that bar is met
    there only because the exact forward is BIT-equal to the oracle's, so every LeakyReLU branch (|y| within 1e-5 of the
    kink: ~25 per million activations) and every ignore-mask decision comes out the same.  Any forward that differs in
    the last bits flips a few of them, each flip changes a local gradient by 90 %, and the change spreads upstream: the
    EXACT path itself, fed the same frames scaled by (1 + 2^-22), moves its own parameter gradients by 1-2e-2 in relative
    L2 and ~1e-1 of a tensor's max (measured: tools/archive/dbg_split_train.py).  So: the split step's distance to the oracle must
    stay within 3x the distance the exact path's own one-ulp-perturbed step has — per tensor class, in relative L2."""
    import videoyolo_amd as vy
    from oracle import yolo3_train_oracle as TO
    params, x, gt_boxes, tg = _train_case(C, B, S)
    orc = TO.OracleYolo3Train(C, dict(params))
    ref_losses = orc.forward_train(x, gt_boxes, *tg)
    ref_grads = orc.backward()
    classes = ["c%d" % i for i in range(C)]
    l_split, g_split = _step(_net(classes, params, mode="split_bf16x3_train"), x, gt_boxes, tg)
    for got, want in zip(l_split, ref_losses):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    exact = _net(classes, params, mode="exact")
    _, g_exact = _step(exact, x, gt_boxes, tg)
    _, g_pert = _step(exact, (x * np.float32(1.0 + 2.0 ** -22)).astype(np.float32), gt_boxes, tg)
    d_exact = max(_rel_l2(g_exact[k], ref_grads[k]) for k in ref_grads)
    d_pert = max(_rel_l2(g_pert[k], ref_grads[k]) for k in ref_grads)
    d_split = max(_rel_l2(g_split[k], ref_grads[k]) for k in ref_grads)
    with capsys.disabled():
        print("\n[train C=%d B=%d %d^2] worst relative L2 of a parameter gradient against the oracle: exact %.2e | exact on "
              "frames x (1 + 2^-22) %.2e | split %.2e" % (C, B, S, d_exact, d_pert, d_split))
    assert d_exact < 2e-3                      # the parity path itself
    assert d_split <= 3 * d_pert + 1e-3        # the split step is as far from the oracle as a one-ulp change of the input is
    assert d_split < 0.1


@pytest.mark.parametrize("which,env", [("data gradients", {"VY_SPLIT_TRAIN": "3", "VY_SPLIT_WGRAD": "0"}),
                                       ("weight gradients", {"VY_SPLIT_TRAIN": "0", "VY_SPLIT_WGRAD": "1"})])
def test_split_gradients_alone_meet_the_exact_bars(which, env):
    """Own process (the switches are read once): forward EXACT — bit-equal to the oracle, so no branch flips — and ONE
    kind of gradient on its split kernel.  VY_SPLIT_TRAIN=3: the data gradients (conv_split.hip: [k = cout][n = cin] weight
    images, flipped taps, the four parity classes of the stride-2 convs, cout = 3 (5 + C) zero-padded to 32 for the
    prediction convs, accumulate into a skip gradient).  VY_SPLIT_WGRAD=1 with VY_SPLIT_TRAIN=0: the weight gradients
    (wgrad_split.hip: both operands split in registers, transposed LDS reads, split-K slabs) of every conv with
    cout % 128 == 0.  Then the exact path's own bars hold: losses 1e-4, every gradient within 2e-3 of its tensor's max."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import test_gpu_split as t\n"
        "from oracle import yolo3_train_oracle as TO\n"
        "worst = 0.0\n"
        "for C, B, S in [(20, 2, 96), (80, 1, 96), (4, 2, 64)]:\n"
        "    params, x, gt, tg = t._train_case(C, B, S)\n"
        "    orc = TO.OracleYolo3Train(C, dict(params)); rl = orc.forward_train(x, gt, *tg); rg = orc.backward()\n"
        "    l, g = t._step(t._net(['c%%d' %% i for i in range(C)], params, mode='split_bf16x3_train'), x, gt, tg)\n"
        "    assert all(np.allclose(a, b, rtol=1e-4, atol=1e-4) for a, b in zip(l, rl))\n"
        "    worst = max(worst, max(float(np.abs(g[k] - rg[k]).max() / (np.abs(rg[k]).max() + 1e-6)) for k in rg))\n"
        "print('RESULT', json.dumps({'worst': worst}))\n") % (here, os.path.dirname(here))
    env = dict(os.environ, VY_SPLIT_ALWAYS="1", **env)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert r["worst"] < 2e-3, (which, r)


def test_split_gradients_under_the_products_routing_at_a_realistic_shape(tmp_path):
    """the tight gradient bar used to be applied only with VY_SPLIT_ALWAYS=1 on shapes of at most 96 pixels.
    Here the PRODUCT's cost-model routing decides (no VY_SPLIT_ALWAYS) at 416 x 416, 16 frames, 20 classes (BASELINE configs[2] per GPU): forward exact
    (VY_SPLIT_TRAIN=3: bit-equal activations, no branch flips), data gradients and weight gradients on the split kernels
    wherever the models send them — and the step's label log (VY_TRAIN_LABELS, '# via' lines) must show that the launches
    this is about really ran: data gradients on the 256 x 64 tile and data gradients as k-split launches.  Reference: the
    same step in the exact mode on the GPU (itself held to the oracle by tests/test_gpu_fullsize.py); bar: every gradient
    within 2e-3 of its tensor's maximum, as for the exact path against the oracle."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    labels = str(tmp_path / "labels.txt")
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import test_gpu_split as t\n"
        "C, B, S = 20, 16, 416\n"
        "params, x, gt, tg = t._train_case(C, B, S)\n"
        "cls = ['c%%d' %% i for i in range(C)]\n"
        "ls, gs = t._step(t._net(cls, params, mode='split_bf16x3_train'), x, gt, tg)\n"   # first step of the process: the one the label log covers
        "le, ge = t._step(t._net(cls, params, mode='exact'), x, gt, tg)\n"
        "assert all(np.array_equal(a, b) for a, b in zip(ls, le)), 'forward is exact in both: the losses must be bit-equal'\n"
        "worst = max((float(np.abs(gs[k] - ge[k]).max() / (np.abs(ge[k]).max() + 1e-6)), k) for k in ge)\n"
        "differ = sum(1 for k in ge if not np.array_equal(gs[k], ge[k]))\n"
        "print('RESULT', json.dumps({'worst': worst[0], 'where': worst[1], 'differ': differ, 'n': len(ge)}))\n") % (here, os.path.dirname(here))
    env = {k: v for k, v in os.environ.items() if k != "VY_SPLIT_ALWAYS"}
    env.update(VY_SPLIT_TRAIN="3", VY_SPLIT_WGRAD="1", VY_TRAIN_LABELS=labels)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    lines = open(labels).read().splitlines()
    via_dgrad = [lines[i + 1] for i, ln in enumerate(lines[:-1]) if ln.startswith("dgrad ") and lines[i + 1].startswith("# via")]
    assert any(v.startswith("# via split 256x64") for v in via_dgrad), "no data gradient ran on the 256x64 split tile"
    assert any(v.startswith("# via split") and int(v.rsplit("k", 1)[1]) > 1 for v in via_dgrad), "no data gradient ran as a k-split launch"
    assert any(v == "# via exact" for v in via_dgrad)          # the routing really is per launch
    assert r["differ"] > r["n"] // 2, r                          # the split kernels did produce the gradients
    assert r["worst"] < 2e-3, r


def test_split_training_equals_exact_training_when_switched_off(voc_classes, synth20, monkeypatch):
    """The mode switch itself: 'split_bf16x3' (inference only) and 'exact' train bit-identically to a net that never
    left the exact mode; 'split_bf16x3_train' really is a different summation."""
    from videoyolo_amd import autograd
    from oracle import targets_oracle as T
    x = frames(2, 64, seed=9)
    gt_boxes, gt_ids = T.synthetic_gt(2, 64, 20, m=3, seed=2, pad_to=5)
    tg = T.prefetch_targets(20, 64, 64, gt_boxes, gt_ids)

    def step(net):
        with autograd.record():
            l = net(x, gt_boxes, *tg)
            autograd.backward([l[0] + l[1] + l[2] + l[3]])
        return [t.cpu().numpy() for t in l], net._grads.cpu().numpy().copy()
    a = _net(voc_classes, synth20, mode="exact")
    la, ga = step(a)
    b = _net(voc_classes, synth20, mode="split_bf16x3_train")
    lb, gb = step(b)
    b.set_conv_mode("split_bf16x3")         # the inference-only mode trains on the exact kernels
    ld, gd = step(b)
    assert all(np.array_equal(u, v) for u, v in zip(la, ld)) and np.array_equal(ga, gd)
    b.set_conv_mode("exact")
    lc, gc = step(b)
    assert all(np.array_equal(u, v) for u, v in zip(la, lc)) and np.array_equal(ga, gc)
    assert not np.array_equal(ga, gb)       # the split step really was a different summation


def test_recorded_forward_then_inference_keeps_the_winograd_images(voc_classes, monkeypatch):
    """in conv mode 'split_bf16x3_train' the recorded forward rebuilds the forward / data-gradient images in
    one launch (train.hip: refresh_split_images) and used to clear the flag that also gates the Winograd image sets, which
    only the inference forward rebuilds — a recorded forward followed by `net(x)` with no parameter write in between
    (validation under record, gradient accumulation) then ran conv_wino_kernel on images that were never written.  The
    Winograd sets have their own flag now (net_internal.h: wino_dirty).  Same plan for both calls (model.py reuses a
    training plan for inference at the same size); every supported cell forced through the Winograd kernel."""
    monkeypatch.setenv("VY_SPLIT_WINO", "2")
    C, B, S = 20, 2, 128
    params, x, gt_boxes, tg = _train_case(C, B, S)
    net = _net(voc_classes, params, mode="split_bf16x3_train")
    _step(net, x, gt_boxes, tg)                      # recorded forward + backward; no optimizer step
    assert any("wino" in n for n in _split_launches(net, x)), "no Winograd launch in the inference pass"
    got = [t.cpu().numpy() for t in net(x, return_index=True)]
    heads = [net.read_head(i).cpu().numpy() for i in range(3)]
    # (the recorded forward moved the running statistics: the fresh net gets the parameters as they are NOW)
    fresh = _net(voc_classes, {p.name: p.data() for p in net.collect_params().values()}, mode="split_bf16x3")
    want = [t.cpu().numpy() for t in fresh(x, return_index=True)]
    for i in range(3):
        assert np.array_equal(heads[i], fresh.read_head(i).cpu().numpy()), "head %d differs after a recorded forward" % i
    for u, v in zip(got, want):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("ncls,batch,obj_bias", [(20, 64, 0.0), (30, 32, -4.0)])
def test_split_inference_full_size(ncls, batch, obj_bias, monkeypatch, capsys):
    """BASELINE configs[1] / configs[3] shapes (608 x 608, batch 64 / 30 classes batch 32) in the split conv mode under the
    PRODUCT's policy (no VY_SPLIT_ALWAYS): run-to-run bit-reproducible, frame order does not matter, box_nms invariants on
    every frame, and two frames against the oracle — heads within HEAD_TOL, kept rows up to near-ties."""
    import torch
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    from test_gpu_fullsize import _check_nms_invariants
    monkeypatch.delenv("VY_SPLIT_ALWAYS")
    classes = ["c%d" % i for i in range(ncls)]
    params = init.synthetic_params(O.param_shapes(ncls), seed=233, obj_bias=obj_bias)
    net = _net(classes, params)
    x = torch.as_tensor(frames(batch, 608, seed=7)).cuda()
    names = _split_launches(net, x)
    assert sum("|split" in n or "|wino" in n for n in names) >= 67   # (a marginal 1x1 launch may stay on the exact kernel: the model decides)
    assert sum("|wino" in n for n in names) >= 20                    # the long-K 3x3 stride-1 cells: Winograd F(2, 3) (conv_wino.hip)
    out = [t.clone() for t in net(x, return_index=True)]
    again = net(x, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip(out, again)), "not reproducible run to run"
    heads = [net.read_head(i).cpu().numpy() for i in range(3)]   # (of the forward just run: before the reversed batch)
    perm = torch.arange(batch - 1, -1, -1, device=x.device)
    rev = net(x[perm].contiguous(), return_index=True)
    assert all(torch.equal(a[perm], b) for a, b in zip(out, rev)), "frame order changes the results"
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in out]
    _check_nms_invariants(ids, scores, bboxes, ncls, 0.45, 100)
    orc = O.OracleYolo3(ncls, params)
    for j in (0, batch - 1):
        xj = x[j:j + 1].cpu().numpy()
        ref = orc.raw_heads(xj)
        worst = max(float(np.abs(heads[i][j:j + 1] - ref[i]).max()) for i in range(3))
        assert worst <= HEAD_TOL, (j, worst)
        r = orc(xj)
        exc = _check_keep(keep[j:j + 1], r[3], orc.detections(xj), "608 batch %d frame %d" % (batch, j), capsys)
        assert len(exc) <= 4
        np.testing.assert_allclose(np.sort(scores[j:j + 1], 1), np.sort(r[1], 1), rtol=0, atol=TOL)
