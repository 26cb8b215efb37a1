"""-m gpu: the host-fed, pipelined detect loop (videoyolo_amd/stream.py; the reference's detect_yolo3.py:209-233):
uint8 frames in host memory -> copy stream -> resize + normalise -> net -> [gather] -> copy stream -> host.

  * one process: every batch of a stream comes back in order and equal, bit for bit, to the synchronous path
    (transforms.YOLO3VideoInferenceTransform + net(x) + .cpu()) — buffers are reused across batches, nothing leaks between
    slots, the results do not depend on how far the host runs ahead
  * two ranks sharing the GPU over gloo (RCCL needs two devices: the driver's scaling run covers it): frames scattered with
    even_split=False sizes (5 -> 3 + 2), rows gathered; what rank 0 holds equals the single-process result of the whole
    clip batch, bit for bit (frames are independent: SURVEY 8e) — also with an even split
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(ncls, size=96, mode="exact"):
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["c%d" % i for i in range(ncls)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233, obj_bias=-2.0)
    net.collect_params().reset_ctx("cuda:0")
    net.set_nms(0.45, 400, 100)
    net.set_conv_mode(mode)
    return net


def _sync_path(net, clip, size):
    from videoyolo_amd import transforms
    x = transforms.YOLO3VideoInferenceTransform(size, size)(clip)
    return [t.cpu().numpy() for t in net(x)]


@pytest.mark.parametrize("src_hw,size,batch,n", [((60, 80), 96, 3, 5), ((96, 96), 96, 2, 4), ((144, 256), 128, 4, 3)])
def test_pipelined_stream_equals_the_synchronous_path(src_hw, size, batch, n):
    from videoyolo_amd import stream
    rng = np.random.default_rng(3)
    clips = [rng.integers(0, 256, (batch,) + src_hw + (3,), dtype=np.uint8) for _ in range(n)]
    net = _net(20)
    want = [_sync_path(net, c, size) for c in clips]
    det = stream.HostFedDetector(net, batch, src_hw, size, depth=2)
    got = [[np.array(a) for a in out] for out in det.run(iter(clips))]     # copies: the pinned views are reused
    assert len(got) == n
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert a.shape == b.shape and np.array_equal(a, b)
    # frames that already lie in pinned host memory (a decoder's buffers) are copied in without the staging copy: same rows
    import torch
    pinned = [torch.from_numpy(c).pin_memory() for c in clips]
    got_p = [[np.array(a) for a in out] for out in det.run(iter(pinned))]
    for g, w in zip(got_p, want):
        assert all(np.array_equal(a, b) for a, b in zip(g, w))
    assert any((w[0] >= 0).any() for w in want), "the fixture keeps no detection: the comparison would be vacuous"
    assert not np.array_equal(want[0][1], want[1][1])
    # the slot protocol: a third batch cannot be submitted while two are uncollected
    a = det.submit(clips[0])
    b = det.submit(clips[1])
    with pytest.raises(RuntimeError):
        det.submit(clips[2])
    r1, r0 = [np.array(t) for t in det.result(b)], [np.array(t) for t in det.result(a)]
    assert np.array_equal(r0[2], want[0][2]) and np.array_equal(r1[2], want[1][2])
    with pytest.raises(ValueError):
        det.submit(clips[0][:, :-1])
    with pytest.raises(TypeError):
        det.submit(clips[0].astype(np.float32))


def test_full_size_stream_of_configs3():
    """BASELINE configs[3]'s shape through the pipelined loop: 30 classes, 608 x 608, clip batches of 32 frames of 360 x 640
    uint8 video (bicubic enlargement on the GPU), four batches over two slots — every batch bit-equal to the synchronous
    path, results in order."""
    from videoyolo_amd import stream
    rng = np.random.default_rng(9)
    pool = [rng.integers(0, 256, (32, 360, 640, 3), dtype=np.uint8) for _ in range(2)]
    clips = [pool[0], pool[1], pool[1][::-1].copy(), pool[0]]
    net = _net(30)
    want = [_sync_path(net, c, 608) for c in clips[:3]]
    det = stream.HostFedDetector(net, 32, (360, 640), 608, depth=2)
    got = [[np.array(a) for a in out] for out in det.run(iter(clips))]
    for i, g in enumerate(got):
        w = want[i] if i < 3 else want[0]
        assert all(np.array_equal(a, b) for a, b in zip(g, w)), "batch %d" % i
    assert np.array_equal(got[2][0], want[1][0][::-1])      # frames are independent: the reversed clip gives the reversed rows
    assert (want[0][0] >= 0).sum() > 32


@pytest.mark.parametrize("global_batch", [5, 4])
def test_two_ranks_scatter_and_gather_equal_one_rank(global_batch, tmp_path):
    from videoyolo_amd import launch
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stream_worker
    rc = launch.spawn_ranks(2, [os.path.join(ROOT, "tests", "stream_worker.py"), str(tmp_path), "--backend", "gloo", "--share-gpu",
                                "--global-batch", str(global_batch), "--batches", "3", "--classes", "30"], timeout=900)
    assert rc == 0, "a rank failed (exit code %d)" % rc
    got = dict(np.load(str(tmp_path / "gathered.npz")))
    net = _net(30)
    clips = stream_worker.clip_batches(3, global_batch, 60, 80)
    kept = 0
    for i, c in enumerate(clips):
        ids, scores, bboxes = _sync_path(net, c, 96)
        assert got["ids%d" % i].shape == (global_batch, 100, 1)
        assert np.array_equal(got["ids%d" % i], ids) and np.array_equal(got["scores%d" % i], scores)
        assert np.array_equal(got["bboxes%d" % i], bboxes)
        kept += int((ids >= 0).sum())
    assert kept > 0
