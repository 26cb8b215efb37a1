"""-m gpu: one training step of the HIP path (through the C-ABI) against the CPU training oracle:
the four per-sample losses, every parameter gradient, the BatchNorm running statistics and the
SGD-updated parameters.  Tolerances: losses 1e-4 (north_star), gradients 2e-3 of each tensor's
max |g| (fp32 MFMA reductions over up to 10^5 pixels vs the oracle's double accumulation)."""
import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu


def _setup(C, B, S, seed=11, m=3, pad_to=5):
    from videoyolo_amd import init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(C), seed=seed)
    x = frames(B, S, seed=5)
    gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=m, seed=2, pad_to=pad_to)
    tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
    return params, x, gt_boxes, tg


def _net(C, params):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    return net


# (1, 1, 64): a single class and a single frame (prediction planes 18 -> 32 channels, batch statistics over one
# image); (80, 1, 96): COCO-sized heads (255 -> 256 channels)
@pytest.mark.parametrize("C,B,S", [(4, 2, 64), (20, 2, 96), (1, 1, 64), (80, 1, 96)])
def test_train_step_matches_oracle(C, B, S):
    import videoyolo_amd as vy
    from videoyolo_amd import autograd
    from oracle import yolo3_train_oracle as TO
    params, x, gt_boxes, tg = _setup(C, B, S)
    orc = TO.OracleYolo3Train(C, dict(params))
    ref_losses = orc.forward_train(x, gt_boxes, *tg)
    ref_grads = orc.backward()

    net = _net(C, params)
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        total = losses[0] + losses[1] + losses[2] + losses[3]
        autograd.backward([total])
    for got, want in zip(losses, ref_losses):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert float(sum(l.sum() for l in losses)) > 0

    worst = ("", 0.0)
    for name, want in ref_grads.items():
        got = net.grad(name)
        assert got.shape == want.shape
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-6)
        if err > worst[1]:
            worst = (name, err)
        assert err < 2e-3, (name, err)
    print("worst gradient mismatch:", worst)

    # running statistics (momentum 0.9, biased batch variance)
    for name, want in orc.new_running.items():
        got = net.collect_params()[name].data()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)

    # trainer.step(batch_size): mom = 0.9*mom - lr*(g/bs + wd*w); w += mom
    p_ref = {k: v.copy() for k, v in params.items()}
    mom = {}
    TO.sgd_step(p_ref, ref_grads, mom, 1e-3, 0.9, 5e-4, B)
    trainer.step(B)
    for name in ("stages.0.0.0.weight", "stages.1.4.body.1.0.weight", "yolo_blocks.1.tip.1.gamma",
                 "yolo_outputs.2.prediction.bias", "transitions.0.0.weight"):
        got = net.collect_params()[name].data()
        # the gradient bar above (2e-3 of the tensor's max) carried through lr / batch, plus fp32 rounding of w
        atol = 2e-6 + 2e-3 * 1e-3 * float(np.abs(ref_grads[name]).max()) / B
        np.testing.assert_allclose(got, p_ref[name], rtol=0, atol=atol)
        assert np.abs(got - params[name]).max() > 0


@pytest.mark.parametrize("C,B,H,W", [(5, 2, 64, 128), (3, 1, 160, 96), (3, 1, 96, 32), (2, 3, 128, 32)])
def test_train_step_on_rectangular_frames(C, B, H, W):
    """Height != width (multiples of 32): per-axis geometry in the target kernels, the weight-gradient pixel tables,
    the stride-2 data gradients and the x2-summed transition gradients.  Losses and every gradient against the oracle.
    The 32-wide shapes are the ones whose BatchNorm-backward partial rows (one per image-row chunk) outnumber the
    64-pixel chunks the scratch region used to be sized by: an overflow there lands in the weight-gradient slabs."""
    from videoyolo_amd import autograd, init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    from oracle import yolo3_train_oracle as TO
    params = init.synthetic_params(O.param_shapes(C), seed=23)
    rng = np.random.default_rng(H + W)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    gt_boxes, gt_ids = T.synthetic_gt(B, min(H, W), C, m=3, seed=4, pad_to=4)
    tg = T.prefetch_targets(C, H, W, gt_boxes, gt_ids)
    orc = TO.OracleYolo3Train(C, dict(params))
    ref_losses = orc.forward_train(x, gt_boxes, *tg)
    ref_grads = orc.backward()
    net = _net(C, params)
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    for got, want in zip(losses, ref_losses):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    for name, want in ref_grads.items():
        got = net.grad(name)
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-6)
        assert err < 2e-3, (name, err)


@pytest.mark.parametrize("thresh", [0.3, 0.9])
def test_ignore_iou_thresh_reaches_the_dynamic_targets(thresh):
    """YOLOV3T(ignore_iou_thresh=...) (yolo3.py:962) decides which unmatched predictions are ignored by the objectness
    loss (yolo_target.py:204): a low and a high threshold against the oracle built with the same value — and the two
    objectness losses differ, so the option is not a no-op."""
    import videoyolo_amd as vy
    from videoyolo_amd import autograd
    from oracle import yolo3_train_oracle as TO
    C, B, S = 4, 2, 96
    params, x, gt_boxes, tg = _setup(C, B, S, m=4)
    # raise the box-size predictions so that many predicted boxes overlap a ground-truth box
    params = dict(params)
    for i in range(3):
        b = params["yolo_outputs.%d.prediction.bias" % i].copy().reshape(3, 5 + C)
        b[:, 2:4] = 0.7
        params["yolo_outputs.%d.prediction.bias" % i] = b.reshape(-1)
    ref = TO.OracleYolo3Train(C, dict(params), ignore_iou_thresh=thresh).forward_train(x, gt_boxes, *tg)
    base = TO.OracleYolo3Train(C, dict(params), ignore_iou_thresh=0.7).forward_train(x, gt_boxes, *tg)
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False, ignore_iou_thresh=thresh)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
    for got, want in zip(losses, ref):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    assert np.abs(ref[0] - base[0]).max() > 1e-3, "the constructed case does not exercise the threshold"


def test_train_options_and_inference_after_training():
    """label smoothing + no_wd / frozen backbone switches, then net(x) in inference mode on the same
    object (validate() after an epoch, train_yolov3.py:434-441)."""
    import videoyolo_amd as vy
    from videoyolo_amd import autograd
    from oracle import yolo3_train_oracle as TO
    from oracle import yolo3_oracle as O
    C, B, S = 3, 2, 64
    params, x, gt_boxes, tg = _setup(C, B, S, seed=4)
    orc = TO.OracleYolo3Train(C, dict(params), label_smooth=True)
    ref_losses = orc.forward_train(x, gt_boxes, *tg)
    ref_grads = orc.backward()
    net = _net(C, params)
    net._target_generator._label_smooth = True
    for p in net.collect_params('.*beta|.*gamma|.*bias').values():
        p.wd_mult = 0.0
    for p in net.collect_params().values():
        if p.backbone:
            p.grad_req = 'null'
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-2, 'wd': 5e-4, 'momentum': 0.9})
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        autograd.backward([sum(losses)])
    for got, want in zip(losses, ref_losses):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    trainer.step(B)
    frozen = net.collect_params()["stages.0.3.0.weight"].data()
    assert np.array_equal(frozen, params["stages.0.3.0.weight"])
    g = ref_grads["yolo_blocks.0.tip.1.gamma"] / B
    want = params["yolo_blocks.0.tip.1.gamma"] - 1e-2 * g          # wd_mult = 0
    np.testing.assert_allclose(net.collect_params()["yolo_blocks.0.tip.1.gamma"].data(), want, atol=2e-6)
    # inference on the updated parameters == oracle inference on the same parameters
    newp = {k: p.data() for k, p in net.collect_params().items()}
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = O.OracleYolo3(C, newp)(x)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])


@pytest.mark.parametrize("C,B,S", [(4, 2, 64), (20, 3, 96)])
def test_train_mode_without_recording(C, B, S):
    """net(x) under autograd.train_mode() without record(): the 8-tuple of yolo3.py:1189-1192 with real
    values — decoded boxes and raw centre / scale / objectness / class predictions of a forward on batch
    statistics — against the oracle's train-mode forward; anchors / offsets / fake feature maps as the
    prefetch target generator consumes them (transforms.py:190-197); running statistics move as in any
    is_training BatchNorm forward."""
    from videoyolo_amd import autograd
    from oracle import targets_oracle as T
    from oracle import yolo3_train_oracle as TO
    params, x, _, _ = _setup(C, B, S)
    orc = TO.OracleYolo3Train(C, dict(params))
    want = orc.split_preds(orc.forward_raw(x))
    net = _net(C, params)
    with autograd.train_mode():
        out = net(x)
    assert len(out) == 8
    box, anchors, offsets, fms, ctr, scl, obj, cls = out
    for got, key in ((ctr, "xy"), (scl, "wh"), (obj, "obj"), (cls, "cls")):
        got = got.cpu().numpy()
        assert got.shape == want[key].shape
        assert np.array_equal(got, want[key]), key       # raw predictions: the conv stack is order-reproducible
    box = box.cpu().numpy()
    fin = np.isfinite(want["box"])
    np.testing.assert_allclose(box[fin], want["box"][fin], rtol=0, atol=1e-4)
    r_anchors, r_offsets, r_fms = T.anchors_offsets_featmaps(S, S)
    for i in range(3):
        assert np.array_equal(anchors[i], r_anchors[i]) and np.array_equal(offsets[i], r_offsets[i])
        assert fms[i].shape == (1, 1) + tuple(r_fms[i])
    for name, w in orc.new_running.items():
        np.testing.assert_allclose(net.collect_params()[name].data(), w, rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError):
        autograd.backward([])                                # nothing was recorded
    # the recorded call still works on the same object afterwards
    gt_boxes, tg = _setup(C, B, S)[2:]
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        autograd.backward(losses)
    assert all(np.isfinite(l.cpu().numpy()).all() for l in losses)


def test_backward_rejects_heads_that_are_not_the_sum_of_the_four_losses():
    """autograd.backward(obj + center + scale + cls) is the one pattern the reference uses (train_yolov3.py:626-631);
    a subset or a scaled loss must raise instead of silently back-propagating the full sum."""
    from videoyolo_amd import autograd
    C, B, S = 3, 2, 64
    params, x, gt_boxes, tg = _setup(C, B, S)
    net = _net(C, params)
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        with pytest.raises(NotImplementedError):
            autograd.backward([losses[0]])
        with pytest.raises(NotImplementedError):
            autograd.backward([losses[0] * 2 + losses[1] + losses[2] + losses[3]])
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])   # still pending: now it runs
    net.collect_params().zero_grad()
    assert float(np.abs(net.grad("stages.0.0.0.weight")).max()) == 0.0


def test_syncbn_plumbing_with_simulated_ranks():
    """SyncBatchNorm path on one GPU: the statistics callback is driven with a stand-in for the RCCL
    all-reduce that doubles the [2][C] sums, i.e. two ranks holding the same frames.  Batch mean and
    variance are then unchanged, so losses and gradients must equal the per-device-BN run exactly;
    the callback must fire for the 6 SyncBN layers, forward and backward (12 calls)."""
    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import _lib, autograd
    C, B, S = 3, 2, 64
    params, x, gt_boxes, tg = _setup(C, B, S, seed=9)

    def run(sync):
        net = _net(C, params)
        calls = []
        if sync:
            def cb(user, ptr, count):
                off = int(ptr) - net._ws.data_ptr()
                view = net._ws[off:off + 8 * count].view(torch.float64)
                view.mul_(2.0)
                calls.append(int(count))
                return 0
            keep = _lib.ALLREDUCE_CB(cb)
            net._cb_keep.append(keep)
            with autograd.record():      # plan + bind first: the callback needs the workspace
                net(x, gt_boxes, *tg)
            autograd._get().tape = []
            _lib.check(net._lib.vy_net_set_sync_bn(net._h, 2, keep, None))
        with autograd.record():
            losses = net(x, gt_boxes, *tg)
            autograd.backward([sum(losses)])
        grads = {n: net.grad(n) for n in ("stages.0.0.0.weight", "stages.0.1.1.gamma", "stages.1.0.1.beta",
                                          "stages.2.0.0.weight", "yolo_blocks.0.body.0.0.weight")}
        return [l.cpu().numpy() for l in losses], grads, calls

    l0, g0, _ = run(False)
    l1, g1, calls = run(True)
    assert len(calls) == 12 and sorted(set(calls)) == [64, 128, 256, 512, 1024, 2048]
    for a, b in zip(l0, l1):
        assert np.array_equal(a, b)
    for k in g0:
        np.testing.assert_allclose(g1[k], g0[k], rtol=1e-5, atol=1e-7)


def test_overfit_one_batch_and_checkpoint_roundtrip(tmp_path):
    """End to end: 25 SGD steps on one fixed batch drive the loss down (the whole forward/backward/
    update chain is wired with the right signs), the step is bit-reproducible run to run (no float
    atomics), and save_parameters -> load_parameters -> inference gives identical detections
    (train_yolov3.py:289-329 resume path; optimizer state is not saved there either)."""
    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import autograd, targets
    C, B, S = 3, 4, 96
    params, x, _, _ = _setup(C, B, S, seed=21)
    gt_boxes, gt_ids = targets.synthetic_gt(B, S, C, m=3, seed=5)
    tg = targets.YOLOV3PrefetchTargetGenerator(C)(S, S, gt_boxes, gt_ids)

    def run(steps):
        net = _net(C, params)
        trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 2e-3, 'wd': 5e-4, 'momentum': 0.9})
        hist = []
        for _ in range(steps):
            with autograd.record():
                losses = net(x, gt_boxes, *tg)
                autograd.backward([sum(losses)])
            trainer.step(B)
            hist.append(float(sum(l.sum() for l in losses)))
        return net, hist

    net, hist = run(25)
    assert all(np.isfinite(hist))
    assert hist[-1] < 0.6 * hist[0], hist
    _, hist2 = run(25)
    assert hist == hist2, "training step is not bit-reproducible"
    f = str(tmp_path / "ckpt_0025.params")
    net.save_parameters(f)
    twin = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    twin.load_parameters(f, ctx="cuda:0")
    a = net(x, return_index=True)
    b = twin(x, return_index=True)
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_reset_class_on_device():
    """net.reset_class(...) after reset_ctx (train_yolov3.py:728-729): predictors are rebuilt on the
    device, reused rows carry over, inference runs with the new class count."""
    C, B, S = 5, 2, 64
    params, x, _, _ = _setup(C, B, S, seed=31)
    net = _net(C, params)
    old = net.collect_params()["yolo_outputs.2.prediction.weight"].data()
    net.reset_class(["c3", "new"], reuse_weights={"c3": "c3"})
    new = net.collect_params()["yolo_outputs.2.prediction.weight"].data()
    assert new.shape == (21, 256, 1, 1)
    for a in range(3):
        assert np.array_equal(new[a * 7 + 5], old[a * 10 + 5 + 3])
        assert np.array_equal(new[a * 7:a * 7 + 5], old[a * 10:a * 10 + 5])
    ids, scores, bboxes = net(x)
    assert ids.shape == (B, 100, 1)
    assert float(ids.max()) <= 1.0


def test_image_without_ground_truth_boxes():
    """A batch in which one image has NO object (every gt row is the -1 padding, all prefetch targets zero) and the
    other has one: the empty image's centre / scale / class losses are exactly 0, its objectness loss is the pure
    background term, and losses + gradients of the whole step match the oracle."""
    from videoyolo_amd import autograd, init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    from oracle import yolo3_train_oracle as TO
    C, B, S = 4, 2, 96
    params = init.synthetic_params(O.param_shapes(C), seed=41)
    x = frames(B, S, seed=12)
    gt_boxes = np.full((B, 3, 4), -1, np.float32)
    gt_ids = np.full((B, 3, 1), -1, np.float32)
    gt_boxes[1, 0] = [20, 30, 70, 80]
    gt_ids[1, 0, 0] = 2
    tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
    assert float(np.abs(tg[0][0]).sum()) == 0.0 and float(tg[0][1].sum()) > 0
    orc = TO.OracleYolo3Train(C, dict(params))
    ref_losses = orc.forward_train(x, gt_boxes, *tg)
    ref_grads = orc.backward()
    net = _net(C, params)
    with autograd.record():
        losses = net(x, gt_boxes, *tg)
        autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    got = [l.cpu().numpy() for l in losses]
    for g, w in zip(got, ref_losses):
        np.testing.assert_allclose(g, w, rtol=1e-4, atol=1e-4)
    assert got[1][0] == 0.0 and got[2][0] == 0.0 and got[3][0] == 0.0 and got[0][0] > 0.0
    for name, want in ref_grads.items():
        err = np.abs(net.grad(name) - want).max() / (np.abs(want).max() + 1e-6)
        assert err < 2e-3, (name, err)
