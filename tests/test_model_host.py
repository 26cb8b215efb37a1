"""-m "not gpu": the Python model object's host logic — the Gluon surface the reference's drivers
call — without touching a device."""
import copy
import os

import numpy as np
import pytest

import videoyolo_amd as vy
from videoyolo_amd import autograd


def _net(classes, **kw):
    return vy.yolo3_darknet53(classes, pretrained_base=False, **kw)


def test_constructor_contract(voc_classes):
    net = _net(voc_classes)
    assert net.classes == voc_classes and net.num_class == 20
    assert (net.nms_thresh, net.nms_topk, net.post_nms) == (0.45, 400, 100)   # yolo3.py:959-963
    for bad in [dict(k=3), dict(motion_stream="flownet"), dict(temporal=True), dict(block_conv_type='3'),
                dict(new_model=True), dict(agnostic=True), dict(hierarchical=[3, 1, 1, 1, 1])]:
        with pytest.raises(NotImplementedError):
            _net(voc_classes, **bad)
    with pytest.raises(NotImplementedError):
        vy.YOLOV3(voc_classes, pos_iou_thresh=0.5)                             # yolo3.py:990-992
    _net(voc_classes, k=1, norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": 8})
    with pytest.warns(UserWarning):
        vy.yolo3_darknet53(voc_classes, pretrained_base=True)


def test_collect_params_select_and_freeze(voc_classes):
    net = _net(voc_classes, freeze_base=True)
    ps = net.collect_params()
    assert len(ps) == 366
    sel = net.collect_params('.*beta|.*gamma|.*bias')                          # train_yolov3.py:496
    assert len(sel) == 72 * 2 + 3
    for p in sel.values():
        p.wd_mult = 0.0
    assert ps["stages.0.0.1.gamma"].wd_mult == 0.0 and ps["stages.0.0.0.weight"].wd_mult == 1.0
    frozen = [p for p in ps.values() if p.grad_req == 'null']
    assert all(p.name.startswith("stages.") or not p.trainable for p in frozen)
    assert ps["stages.2.4.body.1.0.weight"].grad_req == 'null'                # wrappers.py:55-57
    assert ps["yolo_blocks.0.body.0.0.weight"].grad_req == 'write'
    assert ps["stages.0.0.1.running_mean"].grad_req == 'null'


def test_initialize_save_load_roundtrip(voc_classes, tmp_path):
    net = _net(voc_classes)
    with pytest.raises(RuntimeError):
        net.collect_params()["stages.0.0.0.weight"].data()
    net.initialize()
    w = net.collect_params()["stages.0.0.0.weight"].data()
    assert w.shape == (32, 3, 3, 3) and np.abs(w).max() <= 0.07
    assert np.all(net.collect_params()["stages.0.0.1.gamma"].data() == 1)
    net.initialize(init="synthetic", force_reinit=True, seed=1)
    f = str(tmp_path / "yolo3_darknet53_voc_0000.params")
    net.save_parameters(f)
    twin = _net(voc_classes)
    twin.load_parameters(f)
    for k, p in net.collect_params().items():
        assert np.array_equal(p.data(), twin.collect_params()[k].data())
    other = _net(voc_classes[:3])
    with pytest.raises(ValueError):
        other.load_parameters(f)  # prediction shapes differ
    with pytest.raises(AssertionError):
        twin.set_parameters({"stages.0.0.0.weight": w})  # missing others
    twin.set_parameters({"stages.0.0.0.weight": w * 2}, allow_missing=True)
    assert np.array_equal(twin.collect_params()["stages.0.0.0.weight"].data(), w * 2)


def test_reset_class_reuses_rows(voc_classes):
    """yolo3.py:76-129: new[1+4+k + a*P_new] = old[1+4+v + a*P_old]; box/objectness rows copied."""
    net = _net(voc_classes)
    net.initialize(init="synthetic", seed=3)
    old_w = net.collect_params()["yolo_outputs.1.prediction.weight"].data()
    old_b = net.collect_params()["yolo_outputs.1.prediction.bias"].data()
    keep_backbone = net.collect_params()["stages.1.3.body.0.0.weight"].data()
    net.reset_class(["person", "mug"], reuse_weights={"person": "person"})    # person = VOC index 14
    assert net.classes == ["person", "mug"]
    new_w = net.collect_params()["yolo_outputs.1.prediction.weight"].data()
    new_b = net.collect_params()["yolo_outputs.1.prediction.bias"].data()
    assert new_w.shape == (3 * 7, 512, 1, 1)
    for a in range(3):
        assert np.array_equal(new_w[a * 7:a * 7 + 5], old_w[a * 25:a * 25 + 5])
        assert np.array_equal(new_w[a * 7 + 5], old_w[a * 25 + 5 + 14])
        assert np.array_equal(new_b[a * 7 + 5], old_b[a * 25 + 5 + 14])
    assert np.array_equal(net.collect_params()["stages.1.3.body.0.0.weight"].data(), keep_backbone)
    with pytest.raises(ValueError):
        net.reset_class(["x"], reuse_weights={"x": "nonexistent"})
    with pytest.raises(ValueError):
        net.reset_class(["x"], reuse_weights={0: 99})
    net.reset_class(["person"], reuse_weights=["person"])
    assert net.collect_params()["yolo_outputs.0.prediction.bias"].shape == (18,)


def test_deepcopy_is_host_side_and_independent(voc_classes):
    net = _net(voc_classes[:4])
    net.initialize(init="synthetic", seed=2)
    net._target_generator._label_smooth = True                                  # train_yolov3.py:500
    twin = copy.deepcopy(net)                                                   # transforms.py:190
    assert twin._target_generator._label_smooth is True
    a = net.collect_params()["transitions.0.0.weight"]
    b = twin.collect_params()["transitions.0.0.weight"]
    assert np.array_equal(a.data(), b.data())
    b.set_data(b.data() * 0)
    assert np.abs(a.data()).sum() > 0


def test_modes_and_errors_without_device(voc_classes):
    net = _net(voc_classes[:2])
    net.initialize()
    net.hybridize()
    net.set_nms(nms_thresh=0.3, nms_topk=200, post_nms=50)
    assert (net.nms_thresh, net.nms_topk, net.post_nms) == (0.3, 200, 50)
    assert not autograd.is_training() and not autograd.is_recording()
    with autograd.record():
        assert autograd.is_training() and autograd.is_recording()
        with autograd.pause():
            assert not autograd.is_recording() and not autograd.is_training()
    with autograd.train_mode():
        assert autograd.is_training() and not autograd.is_recording()
    with pytest.raises(RuntimeError, match="not on a device"):
        net(np.zeros((1, 3, 64, 64), np.float32))
    with pytest.raises(RuntimeError):
        net.collect_params().reset_ctx("cpu")                                   # no CPU path


def test_mxnet_params_container_roundtrip(voc_classes, tmp_path):
    """save_parameters(format='mxnet') / load_parameters sniffing.  The container layout is restated
    from memory (videoyolo_amd/mxparams.py, UNVERIFIED against mxnet): this only proves the reader and the
    writer agree with each other and with the documented byte layout of the header."""
    import struct
    net = _net(voc_classes[:3])
    net.initialize(init="synthetic", seed=7)
    f = str(tmp_path / "net.params")
    net.save_parameters(f, format="mxnet")
    raw = open(f, "rb").read()
    assert struct.unpack_from("<QQQ", raw, 0) == (0x112, 0, 366 + 6)          # 366 tensors + 3 x (anchors, offsets)
    assert struct.unpack_from("<IiI", raw, 24) == (0xF993FAC9, 0, 4)          # first tensor: V2, dense, 4-D
    assert struct.unpack_from("<4q", raw, 36) == (32, 3, 3, 3)                  # stages.0.0.0.weight OIHW
    twin = _net(voc_classes[:3])
    twin.load_parameters(f)
    for k, p in net.collect_params().items():
        assert np.array_equal(p.data(), twin.collect_params()[k].data())
    from videoyolo_amd import mxparams
    d = mxparams.load(f)
    assert list(d)[0] == "stages.0.0.0.weight" and d["yolo_outputs.2.prediction.bias"].shape == (24,)


def test_checkpoints_carry_the_anchor_and_offset_constants(voc_classes, tmp_path):
    """YOLOOutputV3 registers `anchors` and `offsets` as gluon Constants (yolo3.py:64-74); Constants are
    Parameters, so the reference's save_parameters writes `yolo_outputs.{0,1,2}.anchors/.offsets` and its
    load_parameters (train_yolov3.py:323-327, detect_yolo3.py:890: default ignore_extra=False,
    allow_missing=False) expects them.  Both directions must work with exactly that call."""
    from videoyolo_amd import mxparams
    net = _net(voc_classes[:2])
    net.initialize(init="synthetic", seed=3)
    for fmt, name in ((None, "a.params"), ("mxnet", "b.params")):
        f = str(tmp_path / name)
        net.save_parameters(f, format=fmt)
        d = mxparams.load(f) if fmt else dict(np.load(f))
        keys = list(d)
        assert d["yolo_outputs.0.anchors"].shape == (1, 1, 3, 2) and d["yolo_outputs.2.offsets"].shape == (1, 1, 128, 128, 2)
        assert d["yolo_outputs.0.anchors"].reshape(-1).tolist() == [116, 90, 156, 198, 373, 326]     # anchors[::-1][0]
        assert d["yolo_outputs.2.anchors"].reshape(-1).tolist() == [10, 13, 16, 30, 33, 23]
        assert d["yolo_outputs.1.offsets"][0, 0, 5, 7].tolist() == [7, 5]                             # (x, y)
        assert keys.index("yolo_outputs.1.anchors") < keys.index("yolo_outputs.1.prediction.weight")
        twin = _net(voc_classes[:2])
        twin.load_parameters(f)                                   # the reference's call: no ignore_extra
        for k, p in net.collect_params().items():
            assert np.array_equal(p.data(), twin.collect_params()[k].data())
    # a file made with other anchors is refused, not silently run with the built-in ones
    d = dict(np.load(str(tmp_path / "a.params")))
    d["yolo_outputs.1.anchors"] = d["yolo_outputs.1.anchors"] * 2
    with pytest.raises(ValueError):
        _net(voc_classes[:2]).set_parameters(d)
    # files without the constants (this package's round-1 .npz) still load
    d = {k: v for k, v in d.items() if not k.endswith(("anchors", "offsets"))}
    _net(voc_classes[:2]).set_parameters(d)
    # int64 offsets (newer mxnet keeps the numpy dtype of the meshgrid) are accepted too
    d["yolo_outputs.0.offsets"] = net.constants()["yolo_outputs.0.offsets"].astype(np.int64)
    _net(voc_classes[:2]).set_parameters(d)


def test_lr_schedule_of_the_training_script():
    """The schedule train_yolov3.py:517-530 builds: linear warm-up from 0 to lr over warmup_epochs, then
    'step' decay at the given epochs — values derived by hand (4 iterations per epoch, lr 0.1, 2 warm-up
    epochs, 6 more epochs with x0.1 steps after 2 and 4 of them)."""
    from videoyolo_amd import LRScheduler, LRSequential
    n = 4
    sched = LRSequential([
        LRScheduler('linear', base_lr=0, target_lr=0.1, nepochs=2, iters_per_epoch=n),
        LRScheduler('step', base_lr=0.1, nepochs=6, iters_per_epoch=n, step_epoch=[2, 4], step_factor=0.1, power=2)])
    # warm-up: 8 updates, lr = 0.1 * t / 7
    assert sched(0) == 0.0
    assert abs(sched(7) - 0.1) < 1e-12 and abs(sched(3) - 0.1 * 3 / 7) < 1e-12
    # step part starts at update 8 (its own T = num_update - 8): boundaries at T = 8 and 16
    assert abs(sched(8) - 0.1) < 1e-12 and abs(sched(15) - 0.1) < 1e-12
    assert abs(sched(16) - 0.01) < 1e-12 and abs(sched(23) - 0.01) < 1e-12
    assert abs(sched(24) - 0.001) < 1e-12
    assert abs(sched(10 ** 6) - 0.001) < 1e-12            # clamps at the end of the last scheduler
    # the other modes at their end points and midpoint (niters = 11 -> N = 10)
    for mode, mid in (('linear', 0.5), ('poly', 0.25), ('cosine', 0.5)):
        s = LRScheduler(mode, base_lr=1.0, target_lr=0.0, niters=11, power=2)
        assert abs(s(0) - 1.0) < 1e-12 and abs(s(10)) < 1e-12 and abs(s(5) - mid) < 1e-12
    assert LRScheduler('constant', base_lr=0.3, niters=5)(3) == 0.3
    import pytest as _pt
    with _pt.raises(ValueError):
        LRScheduler('step', base_lr=0.1, niters=10)
    with _pt.raises(ValueError):
        LRScheduler('exp')


def test_trainer_uses_the_scheduler(voc_classes):
    import videoyolo_amd as vy
    net = _net(voc_classes[:2])
    sched = vy.LRScheduler('linear', base_lr=1.0, target_lr=0.0, niters=11)
    tr = vy.Trainer(net.collect_params(), 'sgd', {'wd': 5e-4, 'momentum': 0.9, 'lr_scheduler': sched})
    assert tr.learning_rate == 1.0
    with pytest.raises(UserWarning):
        tr.set_learning_rate(0.5)


def test_fastdiv_formula_is_exact():
    """The conv kernel's row tables divide by the feature-map width / height with one multiply-high
    (csrc/conv_igemm.hip make_fastdiv / fd_div): the same integer arithmetic, checked here against // for every
    divisor a plan can produce and n across the 32-bit range."""
    def make(d):
        l = 0
        while (1 << l) < d:
            l += 1
        return ((((1 << l) - d) << 32) // d + 1) & 0xffffffff, min(l, 1), max(l - 1, 0)

    def div(n, f):
        m, s1, s2 = f
        t = (m * n) >> 32
        return ((t + (((n - t) & 0xffffffff) >> s1)) & 0xffffffff) >> s2

    rng = np.random.default_rng(0)
    for d in list(range(1, 300)) + [304, 416, 608, 1024, 2048, 4095, 4096]:
        f = make(d)
        ns = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 ** 31 - 1, 2 ** 32 - 1] + [int(v) for v in rng.integers(0, 2 ** 32, 200)]
        for n in ns:
            assert div(n, f) == n // d, (n, d)


def test_syncbatchnorm_constructor_argument_is_a_contract(voc_classes):
    """norm_layer=SyncBatchNorm, norm_kwargs={'num_devices': n} (train_yolov3.py:350-354) is honoured at the first
    train-mode forward: num_devices must be the number of ranks (one process per GPU); on a single rank it is
    BatchNorm and says so once; the default BatchNorm needs nothing."""
    net = _net(voc_classes[:2], norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": 8})
    with pytest.raises(ValueError, match="num_devices=8"):
        net._ensure_sync_bn()                      # no process group here: world size 1 != 8
    with pytest.raises(ValueError):
        net._ensure_sync_bn()                      # and it keeps raising (not latched by the first attempt)
    one = _net(voc_classes[:2], norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": 1})
    with pytest.warns(UserWarning, match="single device"):
        one._ensure_sync_bn()
    assert one._sync_hook is None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        one._ensure_sync_bn()                      # warned once
        _net(voc_classes[:2])._ensure_sync_bn()    # plain BatchNorm: silent


def test_trainer_rejects_options_it_would_otherwise_swallow(voc_classes):
    net = _net(voc_classes[:2])
    with pytest.raises(NotImplementedError):
        vy.Trainer(net.collect_params(), 'sgd', {}, compression_params={"type": "2bit"})
    with pytest.raises(NotImplementedError):
        vy.Trainer(net.collect_params(), 'sgd', {}, update_on_kvstore=True)
    with pytest.raises(ValueError):
        vy.Trainer(net.collect_params(), 'sgd', {"clip_gradient": 1.0})
    vy.Trainer(net.collect_params(), 'sgd', {'wd': 5e-4, 'momentum': 0.9}, kvstore='local')   # train_yolov3.py:527-530


def test_record_nested_in_train_mode_is_an_outermost_recording():
    """autograd.record() inside train_mode() / pause() still starts a fresh tape (only record scopes nest)."""
    s = autograd._get()
    with autograd.train_mode():
        with autograd.record():
            s.tape.append("stale")
    with autograd.train_mode():
        with autograd.pause():
            with autograd.record():
                assert s.tape == []
                with autograd.record():
                    s.tape.append("inner")
                assert s.tape == ["inner"]           # a nested record() does not clear
    assert not autograd.is_recording() and not autograd.is_training()


def test_pretrained_base_loads_a_darknet53_checkpoint_when_one_is_present(voc_classes, tmp_path):
    """pretrained_base=True (three_darknet.py:262-264): gluoncv's darknet53-<hash>.params — save_parameters of the
    ImageNet classifier, names features.<n>... + output.* — found under `root`, read with the mxnet-layout reader,
    renamed features[:15] / [15:24] / [24:] -> stages.0 / 1 / 2 (wrappers.py:58); the heads are untouched; without a
    file it only warns (there is no download here)."""
    from videoyolo_amd import init, mxparams
    ref = _net(voc_classes[:2])
    table = [(k, p.shape) for k, p in ref.collect_params().items() if p.backbone]
    assert len(table) == 52 * 5
    vals = init.synthetic_params(table, seed=5)
    ckpt = {}
    for k, v in vals.items():
        si, j, rest = k.split(".", 3)[1:]
        f = int(j) + (0, 15, 24)[int(si)]
        ckpt["features.%d.%s" % (f, rest)] = v
    ckpt["output.weight"] = np.zeros((1000, 1024), np.float32)      # the classifier's dense layer: dropped
    ckpt["output.bias"] = np.zeros((1000,), np.float32)
    mxparams.save(str(tmp_path / "darknet53-2189ea49.params"), ckpt)
    net = vy.yolo3_darknet53(voc_classes[:2], pretrained_base=True, root=str(tmp_path))
    for k, v in vals.items():
        assert np.array_equal(net.collect_params()[k].data(), v), k
    with pytest.raises(RuntimeError):
        net.collect_params()["yolo_blocks.0.body.0.0.weight"].data()   # heads: still to be initialised
    assert vy.model.darknet53_to_stage_names({"features.14.body.1.0.weight": 1, "features.15.0.weight": 2,
                                              "features.28.body.0.1.gamma": 3, "output.bias": 4}) == \
        {"stages.0.14.body.1.0.weight": 1, "stages.1.0.0.weight": 2, "stages.2.4.body.0.1.gamma": 3}
    with pytest.raises(ValueError):
        vy.model.darknet53_to_stage_names({"darknetv30_conv0_weight": 1})
    bad = dict(ckpt)
    del bad["features.3.0.weight"]
    mxparams.save(str(tmp_path / "darknet53.params"), bad)
    os.remove(str(tmp_path / "darknet53-2189ea49.params"))
    with pytest.raises(AssertionError, match="missing"):
        vy.yolo3_darknet53(voc_classes[:2], pretrained_base=True, root=str(tmp_path))
    os.remove(str(tmp_path / "darknet53.params"))
    with pytest.warns(UserWarning, match="no darknet53"):
        vy.yolo3_darknet53(voc_classes[:2], pretrained_base=True, root=str(tmp_path))


def test_host_fed_detector_refuses_a_net_without_a_device(voc_classes):
    """videoyolo_amd/stream.py has no CPU path: the pipelined detect loop needs the net on its GPU (there is no fallback to
    fall back to), and says so instead of failing somewhere inside torch."""
    import videoyolo_amd as vy
    from videoyolo_amd import stream
    net = vy.yolo3_darknet53(voc_classes, pretrained_base=False)
    with pytest.raises(RuntimeError, match="reset_ctx"):
        stream.HostFedDetector(net, 4, (60, 80), 96)


def test_detect_heads_needs_a_device_and_three_heads():
    """net.detect_heads (vy_net_detect_heads: the detection tail alone on caller-supplied prediction tensors) refuses a net
    whose parameters are not on a device — there is no CPU fallback — before it looks at the tensors."""
    import numpy as np
    import pytest
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["a", "b", "c"], pretrained_base=False)
    heads = [np.zeros((1, 24, 2, 2), np.float32), np.zeros((1, 24, 4, 4), np.float32), np.zeros((1, 24, 8, 8), np.float32)]
    with pytest.raises(RuntimeError, match="not on a device"):
        net.detect_heads(heads, 64)
