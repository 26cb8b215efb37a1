"""-m gpu: the HIP path (through the C-ABI, via videoyolo_amd.model) against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): NMS box indices / class ids bit-exact;
bbox coords within 1e-4.  Because the conv kernels reproduce the oracle's fma order and
decode uses include/vy_math.h on both sides, every comparison below is in fact exact."""
import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu

TOL = 1e-4  # fp32 tolerance stated by north_star for coords/scores


def _net(classes, params, **kw):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(classes, pretrained_base=False, **kw)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    return net


def _oracle(params, ncls=20, **kw):
    from oracle import yolo3_oracle as O
    return O.OracleYolo3(ncls, params, **kw)


@pytest.mark.parametrize("batch,size", [(2, 96), (1, 64), (3, 160)])
def test_heads_bit_exact(voc_classes, synth20, batch, size):
    x = frames(batch, size)
    net = _net(voc_classes, synth20)
    net(x)
    ref = _oracle(synth20).raw_heads(x)
    for i in range(3):
        got = net.read_head(i).cpu().numpy()
        assert got.shape == ref[i].shape
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, ref[i], rtol=0, atol=TOL)
        assert np.array_equal(got, ref[i]), "head %d: max |diff| %g" % (i, np.abs(got - ref[i]).max())


@pytest.mark.parametrize("name", ["stages.0.0", "stages.0.1", "stages.0.2.body.0", "stages.0.2.body.1",
                                  "stages.0.14.body.1", "stages.1.8.body.1", "stages.2.4.body.1",
                                  "yolo_blocks.0.tip", "transitions.0", "yolo_blocks.1.body.0",
                                  "yolo_blocks.2.tip"])
def test_intermediate_cells(voc_classes, synth20, name):
    """Layer-by-layer taps: stem, stride-2 conv, 1x1, 3x3+residual, concat-fused routes, upsample."""
    from oracle import yolo3_oracle as O
    x = frames(2, 64, seed=7)
    net = _net(voc_classes, synth20)
    net.keep_activations()        # inference planes are recycled by default: taps need every cell's own plane
    net(x)
    orc = _oracle(synth20)
    taps = {}
    orig = orc.cell

    def cell(xx, pre, k, s):
        y = orig(xx, pre, k, s)
        taps[pre] = y
        return y
    orc.cell = cell
    orig_block = orc.block

    def block(xx, pre):
        y = orig_block(xx, pre)
        taps[pre + ".body.1"] = y  # the HIP path fuses the residual add into body.1's epilogue
        return y
    orc.block = block
    orc.raw_heads(x)
    got = net.read_activation(name).cpu().numpy()
    want = taps[name]
    if name.startswith("transitions"):
        want = want.repeat(2, axis=-1).repeat(2, axis=-2)  # stored x2-replicated (layers.py:20)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "max |diff| %g" % np.abs(got - want).max()


@pytest.mark.parametrize("batch,size,obj_bias", [(2, 96, 0.0), (2, 128, -3.0), (1, 416, 0.0)])
def test_detections_match_oracle(voc_classes, batch, size, obj_bias):
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(20), seed=233, obj_bias=obj_bias)
    x = frames(batch, size)
    net = _net(voc_classes, params)
    net.set_nms(0.45, 400, 100)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r_ids, r_scores, r_bboxes, r_keep = _oracle(params)(x)
    assert ids.shape == (batch, 100, 1) and scores.shape == (batch, 100, 1) and bboxes.shape == (batch, 100, 4)
    assert np.array_equal(keep, r_keep), "NMS kept-row indices differ"
    assert np.array_equal(ids, r_ids), "class ids differ"
    np.testing.assert_allclose(scores, r_scores, rtol=0, atol=TOL)
    fin = np.isfinite(r_bboxes)
    assert np.array_equal(np.isfinite(bboxes), fin)
    np.testing.assert_allclose(bboxes[fin], r_bboxes[fin], rtol=0, atol=TOL)
    assert (ids >= 0).sum() > 0


def test_nms_settings_and_30_classes():
    """set_nms(…) variants + the ImageNet-VID class count (config 4: 30 classes)."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(30)]
    params = init.synthetic_params(O.param_shapes(30), seed=5)
    x = frames(2, 96, seed=3)
    net = _net(classes, params)
    for thr, topk, post in [(0.45, 400, 100), (0.3, 50, 20), (0.6, 1000, 300), (0.45, 7, 100)]:
        net.set_nms(thr, topk, post)
        ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
        r = _oracle(params, 30, nms_thresh=thr, nms_topk=topk, post_nms=post)(x)
        assert np.array_equal(keep, r[3]), (thr, topk, post)
        assert np.array_equal(ids, r[0])
        np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)


# (200, 1, 64): prediction vectors of 615 -> 640 floats: the decode pass stages 12 instead of 32 pixels per block
@pytest.mark.parametrize("ncls,batch,size", [(1, 3, 64), (80, 2, 96), (3, 5, 128), (200, 1, 64)])
def test_class_counts_and_odd_batches(ncls, batch, size):
    """1 class (18 prediction channels), 80 classes (255, COCO) and odd batch sizes: head tensors,
    kept-row indices, ids, scores and boxes against the oracle."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(ncls)]
    params = init.synthetic_params(O.param_shapes(ncls), seed=17 + ncls)
    x = frames(batch, size, seed=ncls)
    net = _net(classes, params)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    orc = _oracle(params, ncls)
    r = orc(x)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)
    heads = orc.raw_heads(x)
    for i in range(3):
        assert np.array_equal(net.read_head(i).cpu().numpy(), heads[i])


def _random_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        out.append((int(rng.integers(1, 41)), int(rng.integers(1, 6)), int(rng.integers(32, 161)), int(rng.integers(32, 161)),
                    float(rng.choice([0.3, 0.45, 0.6])), int(rng.choice([1, 17, 100, 400, 1000])), int(rng.choice([1, 10, 100])),
                    float(rng.choice([0.0, -2.0, -4.0]))))
    return out


@pytest.mark.parametrize("ncls,batch,h,w,thr,topk,post,obj_bias", _random_configs(14, 2024))
def test_random_configurations(ncls, batch, h, w, thr, topk, post, obj_bias):
    """Seeded random draws over class count (1..40), batch (1..5), ANY height / width in [32, 160], NMS threshold, topk,
    post_nms and objectness bias: heads bit for bit, detections and kept rows like the CPU checker."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(ncls)]
    params = init.synthetic_params(O.param_shapes(ncls), seed=1000 + ncls, obj_bias=obj_bias)
    rng = np.random.default_rng(h * 1000 + w)
    x = rng.standard_normal((batch, 3, h, w)).astype(np.float32)
    net = _net(classes, params)
    net.set_nms(thr, topk, post)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    orc = O.OracleYolo3(ncls, params, nms_thresh=thr, nms_topk=topk, post_nms=post)
    heads = orc.raw_heads(x)
    for i in range(3):
        assert np.array_equal(net.read_head(i).cpu().numpy(), heads[i]), "head %d" % i
    r = orc(x)
    assert ids.shape == r[0].shape == (batch, post, 1)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)


@pytest.mark.parametrize("ncls,obj_bias", [(1, 0.0), (3, 0.0), (2, -2.0)])
def test_topk_around_word_boundaries(ncls, obj_bias):
    """sort_nms keeps its pair mask in 32-candidate words, counts ranks in slices of the candidate list that depend on the
    next power of two, and falls back to a serial pass above 416 candidates: nms_topk on and around every one of those
    boundaries (and post_nms below / above the survivors), one class (everything may suppress everything) and several.
    The oracle's detection tensor is computed once per case; box_nms per setting."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["c%d" % i for i in range(ncls)]
    params = init.synthetic_params(O.param_shapes(ncls), seed=77 + ncls, obj_bias=obj_bias)
    x = frames(2, 96, seed=5)
    net = _net(classes, params)
    orc = O.OracleYolo3(ncls, params)
    det = orc.detections(x)
    for topk, post, thr in [(1, 1, 0.45), (2, 2, 0.45), (3, 100, 0.3), (31, 100, 0.45), (32, 100, 0.45), (33, 100, 0.45),
                            (63, 10, 0.6), (64, 100, 0.45), (65, 100, 0.45), (127, 100, 0.45), (128, 100, 0.3), (129, 200, 0.45),
                            (255, 100, 0.45), (256, 300, 0.45), (257, 100, 0.6), (384, 100, 0.45), (415, 100, 0.45),
                            (416, 416, 0.45), (417, 100, 0.45), (512, 100, 0.45), (1023, 100, 0.3), (1024, 500, 0.45)]:
        net.set_nms(thr, topk, post)
        ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
        orc.nms_thresh, orc.nms_topk, orc.post_nms = thr, topk, post
        r = orc.nms(det)
        assert np.array_equal(keep, r[3]), (topk, post, thr)
        assert np.array_equal(ids, r[0]), (topk, post, thr)
        np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
        fin = np.isfinite(r[2])
        np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)


def test_ties_and_empty():
    """Degenerate inputs: all-zero weights give every candidate the SAME score (0.25): the order
    must fall back to the reference row index; a very negative objectness bias leaves no valid
    candidate: all rows -1."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["a", "b", "c"]
    table = O.param_shapes(3)
    params = init.uniform_params(table, seed=0)
    for k in params:
        if k.endswith("weight"):
            params[k] = np.zeros_like(params[k])
    x = frames(1, 64)
    net = _net(classes, params)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = _oracle(params, 3)(x)
    assert np.array_equal(keep, r[3])
    assert np.array_equal(ids, r[0])
    assert np.array_equal(scores, r[1])
    for i in range(3):
        params["yolo_outputs.%d.prediction.bias" % i].reshape(3, -1)[:, 4] = -30.0
    net = _net(classes, params)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    assert (ids == -1).all() and (scores == -1).all() and (bboxes == -1).all() and (keep == -1).all()


def test_preprocess_and_postprocess(voc_classes, synth20):
    """to_tensor + normalize kernel is bit-equal to the numpy formula (transforms.py:331-334); the
    post-processing helpers reproduce detect_yolo3.py:226,256-265,327-330."""
    import torch
    from videoyolo_amd import transforms
    rng = np.random.default_rng(0)
    frames_u8 = rng.integers(0, 256, (3, 96, 96, 3), dtype=np.uint8)
    t = transforms.YOLO3VideoInferenceTransform(96, 96)
    x = t(frames_u8)
    want = ((frames_u8.astype(np.float32) / np.float32(255.0) - np.array(transforms.MEAN, np.float32))
            / np.array(transforms.STD, np.float32)).transpose(0, 3, 1, 2)
    assert x.shape == (3, 3, 96, 96)
    assert np.array_equal(x.cpu().numpy(), want)
    with pytest.raises(ValueError):
        t(frames_u8[..., :2])                        # not 3 channels (other sizes are resized: test_gpu_resize.py)
    net = _net(voc_classes, synth20)
    ids, scores, bboxes = net(x)
    rows = transforms.postprocess(ids, scores, bboxes, 96)
    assert len(rows) == 3
    i_np, b_np = ids.cpu().numpy(), np.clip(bboxes.cpu().numpy(), 0, 96)
    for i, r in enumerate(rows):
        k = int((i_np[i] >= 0).sum())
        assert r.shape == (k, 6) and (r[:, 2:] >= 0).all() and (r[:, 2:] <= 1).all()
        np.testing.assert_allclose(r[:, 2:], b_np[i, :k] / 96.0)
    line = transforms.prediction_lines("a/b.jpg", rows[0][:1])[0]
    assert line.startswith("a/b.jpg,%d," % int(rows[0][0, 0])) and line.count(",") == 6


def test_hybridize_replays_a_hip_graph(voc_classes, synth20):
    """net.hybridize(): the forward is captured once per shape and replayed; results are identical to
    the eager path, set_nms() and a new input shape invalidate the capture (yolo3.py:1225)."""
    import torch
    x1, x2 = frames(2, 96, seed=1), frames(2, 96, seed=2)
    net = _net(voc_classes, synth20)
    eager1 = [t.clone() for t in net(x1, return_index=True)]
    eager2 = [t.clone() for t in net(x2, return_index=True)]
    net.hybridize()
    for _ in range(2):
        g1 = net(x1, return_index=True)
        g2 = net(x2, return_index=True)
        assert all(torch.equal(a, b) for a, b in zip(eager1, g1))
        assert all(torch.equal(a, b) for a, b in zip(eager2, g2))
    assert len(net._graphs) == 1
    net.set_nms(0.3, 50, 20)
    ids, _, _ = net(x1)
    assert ids.shape == (2, 20, 1)
    x3 = frames(1, 64, seed=3)
    ids, _, _ = net(x3)
    assert ids.shape == (1, 20, 1) and len(net._graphs) == 1


def test_non_square_input(voc_classes, synth20):
    """H != W (multiples of 32): planner, concat planes and decode offsets are per-axis."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 3, 96, 160)).astype(np.float32)
    net = _net(voc_classes, synth20)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = _oracle(synth20)(x)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)
    for i in range(3):
        assert np.array_equal(net.read_head(i).cpu().numpy(), _oracle(synth20).raw_heads(x)[i])


@pytest.mark.parametrize("shape", [(2, 3, 100, 136), (1, 3, 75, 93), (1, 3, 333, 250), (3, 3, 33, 47)])
def test_sizes_that_are_not_multiples_of_32(voc_classes, synth20, shape):
    """The reference takes any input size: stride-2 convs give ceil(n / 2) rows and the x2 upsample is cropped to
    the route (slice_like, yolo3.py:1177).  Heads bit-exact, detections and kept rows like the CPU checker."""
    rng = np.random.default_rng(shape[2])
    x = rng.standard_normal(shape).astype(np.float32)
    net = _net(voc_classes, synth20)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    orc = _oracle(synth20)
    ref = orc.raw_heads(x)
    for i in range(3):
        got = net.read_head(i).cpu().numpy()
        assert got.shape == ref[i].shape == (shape[0], 75, -(-shape[2] // [32, 16, 8][i]), -(-shape[3] // [32, 16, 8][i]))
        assert np.array_equal(got, ref[i]), "head %d: max |diff| %g" % (i, np.abs(got - ref[i]).max())
    r = orc(x)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)
    # the borders of every plane are still zero: a second, multiple-of-32 shape on the same handle stays exact
    x2 = frames(1, 64)
    net(x2)
    for i in range(3):
        assert np.array_equal(net.read_head(i).cpu().numpy(), orc.raw_heads(x2)[i])


def test_training_keeps_multiples_of_32(voc_classes, synth20):
    from videoyolo_amd import autograd, _lib
    net = _net(voc_classes, synth20)
    x = np.zeros((1, 3, 100, 136), np.float32)
    with autograd.train_mode():
        with pytest.raises(_lib.VyError, match="multiples of 32"):
            net(x)


def test_two_stream_batch_split_is_identical(voc_classes, synth20):
    """Large batches run as two half-batches on two HIP streams (twin handle sharing the parameters):
    same results as the single-stream path, also after the parameters or NMS settings change."""
    import torch
    x = frames(6, 96, seed=4)
    net = _net(voc_classes, synth20)
    net.two_stream_batch = 0
    one = [t.clone() for t in net(x, return_index=True)]
    net.two_stream_batch = 4
    for _ in range(2):
        two = net(x, return_index=True)
        assert all(torch.equal(a, b) for a, b in zip(one, two))
    net.set_nms(0.3, 100, 40)
    two = net(x, return_index=True)
    net.two_stream_batch = 0
    one = net(x, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip(one, two))
    odd = net(x[:5])                      # odd batch: single-stream path
    assert odd[0].shape == (5, 40, 1)


def test_api_surface_on_device(voc_classes, synth20, tmp_path):
    """The object keeps working through the call patterns of the reference's scripts: changing batch and
    image size between calls (replanning), hybridize() + set_nms() invalidation, deepcopy of a device net
    (transforms.py:190), save -> load round trip, alternating training-mode and inference calls."""
    import copy
    import torch
    from videoyolo_amd import autograd
    net = _net(voc_classes, synth20)
    xa, xb = frames(2, 96, seed=1), frames(3, 128, seed=2)
    ra = [t.clone() for t in net(xa, return_index=True)]
    rb = [t.clone() for t in net(xb, return_index=True)]
    assert all(torch.equal(p, q) for p, q in zip(ra, net(xa, return_index=True)))      # back to the first plan
    net.hybridize()
    assert all(torch.equal(p, q) for p, q in zip(rb, net(xb, return_index=True)))
    assert all(torch.equal(p, q) for p, q in zip(rb, net(xb, return_index=True)))      # graph replay
    net.set_nms(0.45, 400, 50)                                                          # invalidates the graph
    assert net(xb)[0].shape == (3, 50, 1)
    net.set_nms(0.45, 400, 100)
    assert all(torch.equal(p, q) for p, q in zip(ra, net(xa, return_index=True)))
    net.hybridize(False)

    twin = copy.deepcopy(net)                          # host-side copy with the same parameters
    twin.collect_params().reset_ctx("cuda:0")
    assert all(torch.equal(p, q) for p, q in zip(ra, twin(xa, return_index=True)))

    path = str(tmp_path / "w.params")
    net.save_parameters(path)
    other = _net(voc_classes, {k: v * 0 + 0.5 for k, v in synth20.items()})
    other.load_parameters(path)
    assert all(torch.equal(p, q) for p, q in zip(ra, other(xa, return_index=True)))

    # train-mode non-recording call on a deep copy, as the DataLoader transform does (transforms.py:190-193);
    # values: tests/test_gpu_train_parity.py::test_train_mode_without_recording
    fake = copy.deepcopy(net)
    fake.collect_params().reset_ctx("cuda:0")
    with autograd.train_mode():
        out = fake(np.zeros((1, 3, 96, 96), np.float32))
    assert len(out) == 8 and out[1][0].shape == (1, 1, 3, 2) and out[0].shape == (1, 3 * (9 + 36 + 144), 4)
    assert all(torch.equal(p, q) for p, q in zip(ra, net(xa, return_index=True)))      # the net itself is unaffected


def test_nms_disabled_returns_the_detection_tensor(voc_classes, synth20):
    """set_nms(nms_thresh >= 1): the reference skips box_nms and the post_nms slice (yolo3.py:1197-1202) and
    returns the (B, N*C, 6) detection tensor itself, class-major per scale."""
    x = frames(2, 96, seed=9)
    net = _net(voc_classes, synth20)
    net.set_nms(1.0, 400, 100)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r_ids, r_scores, r_bboxes, _ = _oracle(synth20, nms_thresh=1.0)(x)
    n = 3 * (3 * 3 + 6 * 6 + 12 * 12) * 20
    assert ids.shape == (2, n, 1) and scores.shape == (2, n, 1) and bboxes.shape == (2, n, 4)
    assert np.array_equal(ids, r_ids) and np.array_equal(scores, r_scores) and np.array_equal(bboxes, r_bboxes)
    assert np.array_equal(keep, np.broadcast_to(np.arange(n, dtype=np.int32), (2, n)))
    net.set_nms(0.45, 400, 100)                       # and back: the usual 100 rows
    assert net(x)[0].shape == (2, 100, 1)


@pytest.mark.parametrize("size,post,obj_bias", [(96, 100, 0.0), (96, 700, 0.0), (224, 300, 0.0), (96, 100, -3.5)])
def test_nms_topk_disabled_uses_every_valid_candidate(voc_classes, size, post, obj_bias):
    """set_nms(nms_topk=-1) ("use -1 to disable", yolo3.py:1208-1228): box_nms sorts and suppresses ALL valid
    candidates (11 340 per image at 96 x 96 / 20 classes, 61 740 at 224 x 224 — far beyond one 1024-entry chunk)
    and the first post_nms survivors come back: kept rows exact against the oracle's unbounded NMS.  The
    sparse case (objectness bias -3.5) has fewer valid candidates than one chunk."""
    from videoyolo_amd import _lib, init
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(20), seed=233, obj_bias=obj_bias)
    x = frames(2, size, seed=size + post)
    net = _net(voc_classes, params)
    net.set_nms(0.45, -1, post)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = _oracle(params, nms_topk=-1, post_nms=post)(x)
    assert ids.shape == (2, post, 1)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)
    assert (keep >= 0).sum() > 10
    # and it differs from the top-400 result whenever the 400 best do not already give `post` survivors
    net.set_nms(0.45, 400, post)
    k400 = net(x, return_index=True)[3].cpu().numpy()
    if (k400 >= 0).sum() < (keep >= 0).sum():
        assert np.array_equal(k400[k400 >= 0], keep[:, :k400.shape[1]][k400 >= 0])   # a prefix of the unbounded run


@pytest.mark.parametrize("topk,post", [(-1, -1), (2000, -1), (-1, 1500), (3000, 2000)])
def test_nms_settings_with_long_outputs(voc_classes, topk, post):
    """The set_nms combinations that used to return VY_ERR_UNSUPPORTED (yolo3.py:1208-1228 accepts them all): a chunked
    nms_topk (<= 0: every valid candidate; > 1024) together with an output of more than 1024 rows (post_nms <= 0: no
    slice; post_nms > 1024).  The kept rows are then read back from the output instead of one workgroup's LDS.
    Rows: post_nms, else nms_topk, else all N*C rows of box_nms's output; kept rows exact against the oracle, whose
    un-sliced output has N*C rows of which everything past ours is -1 filler."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(20), seed=233)
    x = frames(2, 96, seed=41)
    net = _net(voc_classes, params)
    net.set_nms(0.45, topk, post)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = _oracle(params, nms_topk=topk, post_nms=post)(x)
    n_cand = 3 * (3 * 3 + 6 * 6 + 12 * 12) * 20
    rows = post if post > 0 else (topk if topk > 0 else n_cand)
    assert ids.shape == (2, rows, 1) and bboxes.shape == (2, rows, 4)
    assert r[3].shape[1] >= rows and (r[3][:, rows:] == -1).all()        # nothing of the oracle's result is cut off
    assert np.array_equal(keep, r[3][:, :rows]) and np.array_equal(ids, r[0][:, :rows])
    np.testing.assert_allclose(scores, r[1][:, :rows], rtol=0, atol=TOL)
    fin = np.isfinite(r[2][:, :rows])
    np.testing.assert_allclose(bboxes[fin], r[2][:, :rows][fin], rtol=0, atol=TOL)
    assert (keep >= 0).sum(1).min() > 1024 or topk > 0, (keep >= 0).sum(1)   # the unbounded runs keep more than one chunk


@pytest.mark.parametrize("topk,post", [(1025, 100), (2000, 400), (3000, 1024), (5000, 60)])
def test_nms_topk_above_one_chunk(voc_classes, topk, post):
    """set_nms(nms_topk > 1024) (yolo3.py:1208-1228 accepts any): the topk best valid candidates are consumed in
    1024-entry chunks (the last one shortened to the cap).  Kept rows exact against the oracle's box_nms(topk)."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    params = init.synthetic_params(O.param_shapes(20), seed=233)
    x = frames(2, 128, seed=topk)
    net = _net(voc_classes, params)
    net.set_nms(0.45, topk, post)
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
    r = _oracle(params, nms_topk=topk, post_nms=post)(x)
    assert ids.shape == (2, post, 1)
    assert np.array_equal(keep, r[3]) and np.array_equal(ids, r[0])
    np.testing.assert_allclose(scores, r[1], rtol=0, atol=TOL)
    fin = np.isfinite(r[2])
    np.testing.assert_allclose(bboxes[fin], r[2][fin], rtol=0, atol=TOL)
    # the cap is honoured: every kept row is one of the topk best-scored valid candidates of its image
    det = _oracle(params).detections(x)
    for b in range(2):
        sc = det[b, :, 1]
        order = np.argsort(-sc, kind="stable")
        best = set(order[:topk][sc[order[:topk]] > 0.01].tolist())
        assert set(keep[b][keep[b] >= 0].tolist()) <= best


def test_denormal_range_is_kept():
    """Products and sums in the fp32 subnormal range (weights 1e-30, inputs 1e-9): the matrix-core fma chain
    keeps them exactly like the CPU checker's fmaf chain — no flush to zero."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    classes = ["a", "b", "c"]
    params = init.synthetic_params(O.param_shapes(3), seed=1)
    params["stages.0.0.0.weight"] = (params["stages.0.0.0.weight"] * 1e-30).astype(np.float32)
    params["stages.0.0.1.gamma"][:] = 1
    params["stages.0.0.1.beta"][:] = 0
    params["stages.0.0.1.running_mean"][:] = 0
    params["stages.0.0.1.running_var"][:] = 1
    x = (np.random.default_rng(0).standard_normal((1, 3, 64, 64)) * 1e-9).astype(np.float32)
    net = _net(classes, params)
    net.keep_activations()
    net(x)
    orc = _oracle(params, 3)
    taps = {}
    cell0 = orc.cell

    def cell(xx, pre, *a, **k):
        y = cell0(xx, pre, *a, **k)
        taps[pre] = y
        return y
    orc.cell = cell
    orc.raw_heads(x)
    want = taps["stages.0.0"]
    got = net.read_activation("stages.0.0").cpu().numpy()
    sub = (np.abs(want) < np.finfo(np.float32).tiny) & (want != 0)
    assert sub.sum() > 1000, "the case must actually exercise subnormals"
    assert np.array_equal(got, want)


@pytest.mark.parametrize("size,batch", [(96, 3), (75, 2), (416, 1)])
def test_recycled_planes_change_nothing_but_the_workspace(voc_classes, synth20, size, batch):
    """Default inference plan (planes recycled by liveness) vs vy_net_set_keep_activations(1): identical heads and
    detections bit for bit, a workspace of little more than half the size, and read_activation refused in the recycled plan."""
    import torch
    from videoyolo_amd import _lib
    x = frames(batch, size, seed=size)
    lean = _net(voc_classes, synth20)
    a = [t.clone() for t in lean(x, return_index=True)]
    ha = [lean.read_head(i).clone() for i in range(3)]
    keep = _net(voc_classes, synth20)
    keep.keep_activations()
    b = keep(x, return_index=True)
    hb = [keep.read_head(i) for i in range(3)]
    assert all(torch.equal(p, q) for p, q in zip(a, b)) and all(torch.equal(p, q) for p, q in zip(ha, hb))
    fixed = (32 << 20) + 8192          # the stream-K scratch both plans carry (kernels.h VY_SK_PARTIAL_BYTES + flags)
    assert lean._ws.numel() - fixed < 0.6 * (keep._ws.numel() - fixed), (lean._ws.numel(), keep._ws.numel())
    with pytest.raises(_lib.VyError, match="recycled"):
        lean.read_activation("stages.0.2.body.1")
    keep.read_activation("stages.0.2.body.1")
    # a second forward on the recycled plan (stale interiors from the first one) and a switch of mode on the same object
    a2 = lean(x, return_index=True)
    assert all(torch.equal(p, q) for p, q in zip(a, a2))
    lean.keep_activations()
    a3 = lean(x, return_index=True)
    assert all(torch.equal(p, q) for p, q in zip(a, a3))
    lean.read_activation("yolo_blocks.1.tip")


@pytest.mark.parametrize("tile", ["32x32", "32x64"])
def test_small_tile_kernel_stays_bit_exact(tile):
    """conv_small.hip (16x16 wave tiles on v_mfma_f32_16x16x4_f32) is off by default — it measured slower than the
    32x32 tiles it was built to beat (profiles/r03_negative_results.txt) — but it stays in the library and must stay
    bit-exact: the head / layer-tap / odd-size tests of this file in a child process that forces every forward conv
    launch through it (the tile override is read once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, VY_CONV_FORCE=tile)
    here = os.path.abspath(__file__)
    p = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                        "heads_bit_exact or intermediate_cells or not_multiples_of_32"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout


@pytest.mark.parametrize("slots,plain", [("13", False), ("24", False), ("0", True)])
def test_stream_k_hand_off_stays_bit_exact(slots, plain):
    """The chain-preserving stream-K instances of conv_igemm.hip (a tile started by one block, its accumulators handed
    over through a write-through slab + flag, finished by the next block with the SAME fma chain) serve the forward
    launches whose last round of the CUs is poorly filled — none of this file's small shapes.  VY_CONV_SK_SLOTS=n runs
    EVERY conv launch of more than n tiles (data gradients included) on n blocks: 13 = five XCD groups of two blocks and
    three of one, 24 = three per group — whole tiles, heads, tails and the hand-off on the small shapes.  The third
    case is the other switch position: VY_CONV_SK=0, plain launches only.  Heads, layer taps, odd sizes, one training
    step, all against the oracle."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, VY_CONV_SK="0") if plain else dict(os.environ, VY_CONV_SK="1", VY_CONV_SK_SLOTS=slots)
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"),
                        os.path.join(here, "test_gpu_train_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "heads_bit_exact or intermediate_cells or not_multiples_of_32 or train_step_matches_oracle"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200, universal_newlines=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout


@pytest.mark.parametrize("env", [
    {"VY_CONV_KSPLIT": "0"},                                             # the runs one after the other in ONE workgroup (parked chains)
    {"VY_CONV_KSPLIT": "0", "VY_CONV_FORCE": "128x128"},                 # ... on every tile of the kernel
    {"VY_CONV_KSPLIT": "0", "VY_CONV_FORCE": "128x64"},
    {"VY_CONV_KSPLIT": "0", "VY_CONV_FORCE": "64x64"},
    {"VY_CONV_SK": "1", "VY_CONV_SK_SLOTS": "5", "VY_CONV_FORCE": "128x128"},   # stream-K pieces: parked chains handed from the
    {"VY_CONV_SK": "1", "VY_CONV_SK_SLOTS": "13", "VY_CONV_FORCE": "64x64"},    # block that starts a tile to the one that finishes it
    {"VY_CONV_SK": "0"},                                                 # no stream-K scratch at all: plain launches, parked chains
], ids=lambda e: ",".join("%s=%s" % (k[8:], v) for k, v in e.items()))
def test_runs_of_k_are_bit_exact_in_every_execution_form(env):
    """The pinned summation order cuts K >= 4096 into 4 runs (include/vy_math.h vy_conv_k_chunks).  A launch executes
    them in one of three forms — split-K (one workgroup per run; the default wherever the tiles leave half the chip empty:
    every shape of this file), one workgroup running them in turn with the finished chains parked in scratch, or stream-K
    pieces of that — and all three must give the oracle's bits.  The default run of this file covers split-K; this test forces
    the other forms (and every tile of the kernel) in child processes."""
    import os
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    p = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                        "heads_bit_exact or intermediate_cells or not_multiples_of_32"],
                       env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                       universal_newlines=True)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout


def test_split_k_launches_are_chosen_for_one_frame(voc_classes, synth20):
    """One 416 x 416 frame (the reference's default detect call, detect_yolo3.py:55-57): the 13 x 13 cells on 512 channels
    (48 tiles of 64 x 64) go out as split-K launches of 4 workgroups per tile; at batch 16 none does (the tiles fill the chip)."""
    net = _net(voc_classes, synth20)
    names = [r[0] for r in net.profile(frames(1, 416, seed=1))]
    assert sum(1 for n in names if n.endswith("ks4")) == 8 and not any(n.endswith("ks2") for n in names), names
    assert "stages.2.1.body.1|64x64ks4" in names and "yolo_blocks.0.tip|64x64ks4" in names
    names16 = [r[0] for r in net.profile(frames(16, 416, seed=1))]
    assert not any("ks" in n.split("|")[-1] for n in names16)


@pytest.mark.parametrize("topk,post", [(400, 100), (-1, 100), (3000, 50)])
def test_no_valid_candidate_at_all(voc_classes, topk, post):
    """Objectness biases of -30: every score is far below valid_thresh = 0.01, so box_nms has nothing to sort — all
    three NMS paths (radix-select + bitonic, the unbounded chunked kernel, the capped chunked kernel) must return rows
    of -1 only, like the oracle, and must not touch stale state of a previous call with thousands of candidates."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    table = O.param_shapes(20)
    x = frames(2, 96, seed=3)
    busy = _net(voc_classes, init.synthetic_params(table, seed=233))
    busy.set_nms(0.45, topk, post)
    first = busy(x, return_index=True)
    assert int((first[3] >= 0).sum()) > 0
    params = init.synthetic_params(table, seed=233, obj_bias=-30.0)
    busy.set_parameters(params)                                  # same net object, same workspace: now silent
    ids, scores, bboxes, keep = [t.cpu().numpy() for t in busy(x, return_index=True)]
    r = _oracle(params, nms_topk=topk, post_nms=post)(x)
    assert (r[3] < 0).all(), "the constructed case must have no valid candidate"
    assert (keep == -1).all() and (ids == -1).all() and (scores == -1).all() and (bboxes == -1).all()
    assert ids.shape == (2, post, 1) and bboxes.shape == (2, post, 4)
