"""-m gpu: BASELINE.json's FULL sizes (configs[1] 608x608 batch 64 inference, configs[3] 30 classes,
configs[2] 416x416 batch 16 training), where running the CPU oracle over the whole batch would take
minutes.  Parity is carried to these sizes by
  * one frame of the full batch checked against the oracle directly (a 608x608 frame is ~2 s of CPU),
  * frame independence: a frame's detections inside the 64-batch == the same frame run alone, bit for
    bit (so every frame is tied to the single-frame path the small-size tests pin to the oracle),
  * the invariants box_nms guarantees (yolo3.py:1197-1206): scores descending, -1 padding contiguous,
    score > valid_thresh, no kept same-class pair above the overlap threshold, <= post_nms rows,
  * run-to-run bit-reproducibility (no atomics anywhere),
and for training by a central finite difference of the loss along the gradient direction at full size
(independent of the oracle), determinism, and the closed-form SGD update on all 61.6 M parameters."""
import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu


def _net(classes, params):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(classes, pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    net.set_nms(0.45, 400, 100)
    return net


def _iou(a, b):
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None] - inter, 1e-30)


def _check_nms_invariants(ids, scores, bboxes, ncls, thresh, post):
    B = ids.shape[0]
    assert ids.shape == (B, post, 1) and scores.shape == (B, post, 1) and bboxes.shape == (B, post, 4)
    for b in range(B):
        valid = ids[b, :, 0] >= 0
        n = int(valid.sum())
        assert valid[:n].all() and not valid[n:].any(), "padding must be contiguous at the end"
        assert (scores[b, n:] == -1).all() and (bboxes[b, n:] == -1).all()
        s = scores[b, :n, 0]
        assert (np.diff(s) <= 0).all(), "scores must be descending"
        assert (s > 0.01).all()
        c = ids[b, :n, 0]
        assert (c == np.round(c)).all() and (c < ncls).all()
        iou = _iou(bboxes[b, :n], bboxes[b, :n])
        same = c[:, None] == c[None, :]
        np.fill_diagonal(same, False)
        assert not (same & (iou > thresh + 1e-6)).any(), "two kept boxes of one class overlap above the threshold"


@pytest.mark.parametrize("ncls,batch,obj_bias", [(20, 64, 0.0), (20, 64, -4.0), (30, 32, 0.0)])
def test_inference_full_size(ncls, batch, obj_bias):
    import torch
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(ncls)]
    params = init.synthetic_params(O.param_shapes(ncls), seed=233, obj_bias=obj_bias)
    net = _net(classes, params)
    x = torch.as_tensor(frames(batch, 608, seed=7)).cuda()
    out = [t.clone() for t in net(x, return_index=True)]
    again = net(x, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip(out, again)), "not reproducible run to run"
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in out]
    _check_nms_invariants(ids, scores, bboxes, ncls, 0.45, 100)
    assert (ids >= 0).sum() > batch  # the synthetic weights do produce detections
    # frame independence: frames of the batch run alone (and as a pair at another batch position)
    for i in (0, batch // 2 + 1, batch - 1):
        alone = [t.cpu().numpy() for t in net(x[i:i + 1], return_index=True)]
        for full, one in zip((ids, scores, bboxes, keep), alone):
            assert np.array_equal(full[i:i + 1], one), "frame %d differs between batch %d and batch 1" % (i, batch)
    perm = torch.arange(batch - 1, -1, -1, device=x.device)
    rev = net(x[perm].contiguous(), return_index=True)
    assert all(torch.equal(a[perm], b) for a, b in zip(out, rev)), "frame order changes the results"
    # one full-size frame against the oracle itself
    j = batch - 1
    r_ids, r_scores, r_bboxes, r_keep = O.OracleYolo3(ncls, params)(x[j:j + 1].cpu().numpy())
    assert np.array_equal(keep[j:j + 1], r_keep), "NMS kept-row indices differ from the oracle"
    assert np.array_equal(ids[j:j + 1], r_ids)
    np.testing.assert_allclose(scores[j:j + 1], r_scores, rtol=0, atol=1e-4)
    np.testing.assert_allclose(bboxes[j:j + 1], r_bboxes, rtol=0, atol=1e-4)


def test_inference_large_odd_size():
    """A video-sized frame batch whose height and width are odd (609 x 611): ceil-sized feature maps, the cropped x2
    upsample on both axes, the stem's unaligned-row path, the big conv tiles on ragged geometry.  Frame independence
    inside the batch + one frame against the oracle (heads bit for bit, detections)."""
    import torch
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(20)]
    params = init.synthetic_params(O.param_shapes(20), seed=233, obj_bias=-4.0)
    net = _net(classes, params)
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.standard_normal((9, 3, 609, 611)).astype(np.float32)).cuda()
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    _check_nms_invariants(ids, scores, bboxes, 20, 0.45, 100)
    j = 5
    alone = [t.cpu().numpy() for t in net(x[j:j + 1], return_index=True)]
    for full, one in zip((ids, scores, bboxes, keep), alone):
        assert np.array_equal(full[j:j + 1], one)
    orc = O.OracleYolo3(20, params)
    xj = x[j:j + 1].cpu().numpy()
    ref = orc.raw_heads(xj)
    for i in range(3):  # the heads of the single-frame run just made
        got = net.read_head(i).cpu().numpy()
        assert got.shape == ref[i].shape == (1, 75, -(-609 // [32, 16, 8][i]), -(-611 // [32, 16, 8][i]))
        assert np.array_equal(got, ref[i])
    r_ids, r_scores, r_bboxes, r_keep = orc(xj)
    assert np.array_equal(keep[j:j + 1], r_keep) and np.array_equal(ids[j:j + 1], r_ids)
    np.testing.assert_allclose(scores[j:j + 1], r_scores, rtol=0, atol=1e-4)
    fin = np.isfinite(r_bboxes)
    np.testing.assert_allclose(bboxes[j:j + 1][fin], r_bboxes[fin], rtol=0, atol=1e-4)


def test_training_full_size_step():
    """416x416, batch 16, 20 classes (configs[2]): determinism, a directional finite difference of the
    summed loss against <grad, d>, and the SGD update in closed form."""
    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import autograd, targets
    C, B, S = 20, 16, 416
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx("cuda:0")
    x = torch.as_tensor(frames(B, S, seed=3)).cuda()
    gt, gid = targets.synthetic_gt(B, S, C, m=8, seed=1)
    tg = targets.YOLOV3PrefetchTargetGenerator(C)(S, S, gt, gid, device="cuda:0")

    def loss_and_grads(with_grads=True):
        with autograd.record():
            losses = net(x, gt, *tg)
        total = float(sum(l.double().sum().item() for l in losses))
        if not with_grads:
            return total, None
        autograd.backward(losses)
        return total, net._grads.clone()

    # running statistics move with every recorded forward; they do not enter the train-mode loss
    l0, g0 = loss_and_grads()
    l1, g1 = loss_and_grads()
    assert l0 == l1 and torch.equal(g0, g1), "training step is not bit-reproducible"
    assert np.isfinite(l0) and torch.isfinite(g0).all()

    # directional derivative along d = g / |g|: L(w + eps d) - L(w - eps d) = 2 eps |g| + O(eps^3)
    wp = net._dev_params.view(torch.float32)
    assert wp.numel() == g0.numel()
    w0 = wp.clone()
    gn = float(g0.double().norm().item())
    assert gn > 0
    d = (g0.double() / gn).float()
    eps = 0.01 * abs(l0) / gn          # a 1 % change of the loss in each direction
    try:
        wp.copy_(w0 + eps * d)
        lp, _ = loss_and_grads(False)
        wp.copy_(w0 - eps * d)
        lm, _ = loss_and_grads(False)
    finally:
        wp.copy_(w0)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - gn) <= 0.05 * gn, "finite difference %.6g vs |grad| %.6g (eps %.3g)" % (fd, gn, eps)

    # SGD in closed form on every parameter (train_yolov3.py:527-530,634): momentum buffer starts at 0
    l2, g2 = loss_and_grads()
    assert torch.equal(g2, g0)
    lr, mom, wd = 1e-3, 0.9, 5e-4
    before = wp.clone()
    net.sgd_step(lr, mom, wd, 1.0 / B)
    mask = torch.zeros(wp.numel(), dtype=torch.bool, device="cuda:0")
    for p in net.collect_params().values():
        if p.trainable:
            mask[p.offset:p.offset + p.size] = True
    assert int(mask.sum().item()) == 61626049  # trainable parameters at 20 classes
    want = before - lr * (g0 / B + wd * before)
    diff = (wp - want).abs()[mask]
    scale = before.abs()[mask].clamp_min(1e-3)
    assert float((diff / scale).max().item()) < 1e-5, "SGD update deviates from w - lr*(g/B + wd*w)"
    assert torch.equal(wp[~mask], before[~mask]), "non-trainable slots (running stats, padding) must not move"


def test_training_step_at_416_against_the_oracle():
    """One recorded step at the configs[2] frame size (416 x 416, VOC classes, 2 frames — what the CPU oracle
    finishes in about half a minute on the host cores): the four losses per sample at 1e-4 and EVERY parameter
    gradient within 2e-3 of its tensor's maximum, against the oracle's hand-derived backward — the full-size
    counterpart of tests/test_gpu_train_parity.py (which stops at 96 x 96)."""
    import videoyolo_amd as vy
    from videoyolo_amd import autograd, init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    from oracle import yolo3_train_oracle as TO
    C, B, S = 20, 2, 416
    params = init.synthetic_params(O.param_shapes(C), seed=233)
    x = frames(B, S, seed=3)
    gt, gid = T.synthetic_gt(B, S, C, m=8, seed=1, pad_to=10)
    tg = T.prefetch_targets(C, S, S, gt, gid)
    orc = TO.OracleYolo3Train(C, dict(params))
    ref_losses = orc.forward_train(x, gt, *tg)
    ref_grads = orc.backward()
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    with autograd.record():
        losses = net(x, gt, *tg)
        autograd.backward(losses)
    for got, want in zip(losses, ref_losses):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    worst = ("", 0.0)
    for name, want in ref_grads.items():
        got = net.grad(name)
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-6)
        if err > worst[1]:
            worst = (name, err)
        assert err < 2e-3, (name, err)
    assert len(ref_grads) == 72 * 3 + 3 * 2   # (weight, gamma, beta) of the 72 cells + (weight, bias) of the 3 prediction convs
    print("416 x 416: worst gradient mismatch", worst)


def test_stream_k_launches_equal_plain_launches():
    """416x416 batch 16 (configs[2]'s shape) is where the forward launches of 676 / 680 / 688 tiles sit — 2.6 rounds of
    the 256 CUs — and the library runs them as chain-preserving stream-K launches by default (conv_igemm.hip).  The same
    seeded network and batch in two processes, VY_CONV_SK=1 and =0: the default run must really contain stream-K
    launches, the other none, and heads, detections, losses and all 61.6 M gradients must be the same BYTES."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    got = {}
    for sk, extra in (("1", []), ("0", []), ("graph", ["graph"])):
        p = subprocess.run([sys.executable, os.path.join(here, "sk_digest_worker.py"), "416", "16"] + extra,
                           env=dict(os.environ, VY_CONV_SK="0" if sk == "0" else "1"), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=900, universal_newlines=True)
        assert p.returncode == 0, p.stdout[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("DIGEST ")][-1]
        got[sk] = json.loads(line[len("DIGEST "):])
    assert len(got["1"]["sk_launches"]) >= 10, got["1"]["sk_launches"]
    assert got["0"]["sk_launches"] == []
    assert got["1"]["conv_launches"] == got["0"]["conv_launches"] == 74
    assert got["1"]["infer"] == got["0"]["infer"], "stream-K launches changed the inference results"
    assert got["1"]["train"] == got["0"]["train"], "stream-K launches changed the training step"
    # the same launches captured into a hipGraph as the very first thing the process does with them, replayed twice
    assert got["graph"]["sk_launches"] == got["1"]["sk_launches"]
    assert got["graph"]["infer"] == got["1"]["infer"], "hipGraph replay of stream-K launches differs from eager"
