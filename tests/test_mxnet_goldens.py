"""Parity against the REFERENCE ITSELF: golden vectors captured from mxnet + gluoncv running
/root/reference's `yolo3_darknet53` (tests/golden/make_mxnet_goldens.py).

The fixtures `tests/golden/mxnet_*.npz` DO NOT EXIST YET — mxnet cannot be installed in the build container — so
every test here skips, and SURVEY.md §8(c) stays "parity unpinned".  The day somebody runs the capture script on a
machine with mxnet, these tests light up: `-m "not gpu"` checks the CPU oracle (the thing every GPU parity test is
measured against) and `-m gpu` checks the HIP path through the C-ABI, both against the reference's own numbers.
Tolerances are north_star's: NMS ids / kept rows exact, scores, boxes and losses 1e-4 (boxes relative to the box
extent: `exp(raw) * anchor` makes the error scale with the box), gradients 2e-3 of each tensor's max.

VY_MXNET_GOLDEN_DIR points the tests at another directory — used once to exercise this file's plumbing with
fixtures written by `make_mxnet_goldens.py --from-oracle` (source tag "oracle-selfcheck": they pin nothing).
"""
import importlib.util
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.environ.get("VY_MXNET_GOLDEN_DIR", os.path.join(HERE, "golden"))
_spec = importlib.util.spec_from_file_location("make_mxnet_goldens", os.path.join(HERE, "golden", "make_mxnet_goldens.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)   # frames(), sample_idx(), VOC: the capture script's own input generators


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("%s not captured yet: run tests/golden/make_mxnet_goldens.py where mxnet + gluoncv import "
                    "(parity stays unpinned until then)" % name)
    z = dict(np.load(path, allow_pickle=False))
    src = str(z["meta/source"])
    if src != "mxnet" and "VY_MXNET_GOLDEN_DIR" not in os.environ:
        pytest.fail("%s was not captured from mxnet (source=%s): not a golden" % (path, src))
    return z


def _params(z, classes=20):
    """This repo's synthetic parameters, proven identical to the ones the capture used (names, shapes, CRC32s)."""
    from videoyolo_amd import init
    from oracle import yolo3_oracle as O
    table = O.param_shapes(classes)
    names = [str(n) for n in z["meta/param_names"]]
    shapes = {n: tuple(int(v) for v in str(s).split(",")) for n, s in zip(names, z["meta/param_shapes"])}
    assert sorted(names) == sorted(k for k, _ in table), "structural parameter names differ from the reference net's"
    for k, shp in table:
        assert shapes[k] == tuple(shp), (k, shapes[k], shp)
    params = init.synthetic_params(table, seed=233)
    crcs = dict(zip(names, z["meta/param_crc"]))
    for k, v in params.items():
        assert (zlib.crc32(np.ascontiguousarray(v).tobytes()) & 0xFFFFFFFF) == int(crcs[k]), \
            "parameter %s regenerated with different bits than the capture used (numpy version?)" % k
    return params


def _input(z):
    size, seed = int(z["in/size"]), int(z["in/seed"])
    b = int(z["in/batch"]) if "in/batch" in z else 1
    x = G.frames(b, size, seed)
    assert (zlib.crc32(x.tobytes()) & 0xFFFFFFFF) == int(z["meta/crc_x"])
    return x


def _sampled(z, key, arr, atol=None, rel_to_max=None):
    arr = np.ascontiguousarray(arr, np.float32)
    assert tuple(z[key + "/shape"]) == arr.shape, (key, tuple(z[key + "/shape"]), arr.shape)
    idx = G.sample_idx(arr.size, int(z[key + "/idx_seed"]), int(z[key + "/n"]))
    got, want = arr.reshape(-1)[idx], z[key + "/values"]
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), key
    tol = atol if atol is not None else rel_to_max * (float(z[key + "/absmax"]) + 1e-6)
    np.testing.assert_allclose(got[fin], want[fin], rtol=0, atol=tol, err_msg=key)
    f = arr.reshape(-1)[np.isfinite(arr.reshape(-1))].astype(np.float64)
    n = max(f.size, 1)
    assert abs(f.sum() - float(z[key + "/sum"])) <= tol * n, key           # nothing outside the sample drifted either


def _check_inference(z, heads, prenms, nms_rows, ids, scores, bboxes):
    for i in range(3):
        if "head%d" % i in z:
            np.testing.assert_allclose(heads[i], z["head%d" % i], rtol=0, atol=1e-4)
        else:
            _sampled(z, "head%d" % i, heads[i], atol=1e-4)
    if prenms is not None:
        # rows are [id, score, x1, y1, x2, y2]: ids exact, scores 1e-4, boxes 1e-4 of the box extent
        want_shape = tuple(z["prenms/shape"])
        assert prenms.shape == want_shape
        idx = G.sample_idx(prenms.size, int(z["prenms/idx_seed"]), int(z["prenms/n"]))
        got, want = prenms.reshape(-1)[idx], z["prenms/values"]
        col = idx % 6
        assert np.array_equal(got[col == 0], want[col == 0])
        np.testing.assert_allclose(got[col == 1], want[col == 1], rtol=0, atol=1e-4)
        box = col >= 2
        np.testing.assert_allclose(got[box], want[box], rtol=1e-4, atol=1e-4)
    want = z["nms/first_rows"]
    assert bool(z["nms/rest_all_minus_one"])
    if nms_rows is not None:
        k = want.shape[1]
        assert np.array_equal(nms_rows[:, :k, 0], want[:, :, 0]), "class ids of the box_nms survivors differ"
        np.testing.assert_allclose(nms_rows[:, :k, 1], want[:, :, 1], rtol=0, atol=1e-4)
        np.testing.assert_allclose(nms_rows[:, :k, 2:], want[:, :, 2:], rtol=1e-4, atol=1e-4)
        assert (nms_rows[:, k:] == -1).all()
    assert np.array_equal(ids, z["ids"]), "ids of the 100 returned rows differ from mxnet's"
    np.testing.assert_allclose(scores, z["scores"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(bboxes, z["bboxes"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("size", [416, 608])
def test_oracle_inference_matches_mxnet(size):
    """BASELINE configs[0] (and one 608 x 608 frame): the CPU checker against the reference's own outputs."""
    from oracle import yolo3_oracle as O
    z = _load("mxnet_infer_%d.npz" % size)
    params, x = _params(z), _input(z)
    orc = O.OracleYolo3(20, params)
    heads = orc.raw_heads(x)
    prenms = orc.detections(x)
    full, _ = O.box_nms(prenms, 0.45, 0.01, 400)
    ids, scores, bboxes, _ = orc(x)
    assert full.shape[1] == int(z["nms/total_rows"])
    _check_inference(z, heads, prenms, full, ids, scores, bboxes)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [416, 608])
def test_hip_inference_matches_mxnet(size):
    import videoyolo_amd as vy
    z = _load("mxnet_infer_%d.npz" % size)
    params, x = _params(z), _input(z)
    net = vy.yolo3_darknet53(G.VOC, pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    ids, scores, bboxes = [t.cpu().numpy() for t in net(x)]
    heads = [net.read_head(i).cpu().numpy() for i in range(3)]
    net.set_nms(-1, 400, -1)                                               # the detection tensor itself (yolo3.py:1197)
    prenms = np.concatenate([t.cpu().numpy() for t in net(x)], -1)
    _check_inference(z, heads, prenms, None, ids, scores, bboxes)


def _check_training(z, losses, grad_of, running_of, updated_of):
    np.testing.assert_allclose(losses, z["losses"], rtol=1e-4, atol=1e-4)
    names = sorted(k[len("grad/"):-len("/shape")] for k in z if k.startswith("grad/") and k.endswith("/shape"))
    assert len(names) == 222
    for n in names:
        _sampled(z, "grad/" + n, grad_of(n), rel_to_max=2e-3)
    for k in z:
        if k.startswith("running/"):
            np.testing.assert_allclose(running_of(k[len("running/"):]), z[k], rtol=1e-4, atol=1e-5, err_msg=k)
    for n in names:
        g = float(z["grad/" + n + "/absmax"])
        _sampled(z, "updated/" + n, updated_of(n), atol=2e-6 + 2e-3 * 1e-3 * g / int(z["in/batch"]))


def _train_inputs(z):
    return [z["in/gt_boxes"]] + [z["in/" + k] for k in ("obj_t", "centers_t", "scales_t", "weights_t", "clas_t")]


def test_oracle_training_step_matches_mxnet():
    """One recorded step (2 x 96 x 96): YOLOV3Loss, all 222 gradients, BatchNorm running statistics (this is where
    biased-vs-unbiased running variance gets decided) and SGD(momentum, wd) after trainer.step(2)."""
    from oracle import yolo3_train_oracle as TO
    z = _load("mxnet_train_96.npz")
    params, x = _params(z), _input(z)
    orc = TO.OracleYolo3Train(20, dict(params))
    losses = np.stack(orc.forward_train(x, *_train_inputs(z)))
    grads = orc.backward()
    upd = {k: v.copy() for k, v in params.items()}
    TO.sgd_step(upd, grads, {}, 1e-3, 0.9, 5e-4, x.shape[0])
    _check_training(z, losses, grads.__getitem__, orc.new_running.__getitem__, upd.__getitem__)


@pytest.mark.gpu
def test_hip_training_step_matches_mxnet():
    import videoyolo_amd as vy
    from videoyolo_amd import autograd
    z = _load("mxnet_train_96.npz")
    params, x = _params(z), _input(z)
    net = vy.yolo3_darknet53(G.VOC, pretrained_base=False)
    net.set_parameters(params)
    net.collect_params().reset_ctx("cuda:0")
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9}, kvstore='local')
    with autograd.record():
        ls = net(x, *_train_inputs(z))
        autograd.backward([ls[0] + ls[1] + ls[2] + ls[3]])
    losses = np.stack([l.cpu().numpy() for l in ls])
    grads = {p.name: net.grad(p.name) for p in net.collect_params().values() if p.trainable}
    running = {p.name: p.data() for p in net.collect_params().values() if "running" in p.name}
    trainer.step(x.shape[0])
    upd = {k: net.collect_params()[k].data() for k in grads}
    _check_training(z, losses, grads.__getitem__, running.__getitem__, upd.__getitem__)


def test_mxparams_reads_a_real_mxnet_file():
    """videoyolo_amd/mxparams.py against a file written by mxnet's own save_parameters (f4)."""
    from videoyolo_amd import mxparams
    path = os.path.join(GOLD, "mxnet_tiny.params")
    want = os.path.join(GOLD, "mxnet_tiny_params_expected.npz")
    if not (os.path.exists(path) and os.path.exists(want)):
        pytest.skip("mxnet_tiny.params not captured yet (tests/golden/make_mxnet_goldens.py)")
    got, exp = mxparams.load(path), dict(np.load(want))
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert got[k].dtype == exp[k].dtype and np.array_equal(got[k], exp[k]), k
