"""-m "not gpu": host logic of the training input contract (prefetch targets) and of the multi-process
helpers, the latter under a real 2-rank gloo process group."""
import os
import socket
import subprocess
import sys

import numpy as np

from oracle import targets_oracle as TOr
from videoyolo_amd import parallel, targets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prefetch_targets_match_the_reference_loop():
    rng = np.random.default_rng(0)
    for size, C in [(64, 3), (416, 20), (608, 30)]:
        gt, ids = TOr.synthetic_gt(3, size, C, m=6, seed=size, pad_to=9)
        gt[1, 2] = gt[1, 1]            # two boxes landing in the same (cell, anchor): last one wins
        ids[1, 2] = (ids[1, 1] + 1) % C
        mix = rng.uniform(0.3, 1.0, (3, 9, 1)).astype(np.float32)
        want = TOr.prefetch_targets(C, size, size, gt, ids, mix)
        got = targets.YOLOV3PrefetchTargetGenerator(C)(size, size, gt, ids, mix)
        for g, w in zip(got, want):
            assert g.shape == w.shape
            np.testing.assert_allclose(g, w, rtol=0, atol=1e-6)
        assert got[0].shape[1] == targets.num_anchors(size, size)
    assert targets.num_anchors(416, 416) == 10647 and targets.num_anchors(608, 608) == 22743


def test_cell_index_follows_float64_promotion():
    """yolo_target.py:115-119 under NumPy 1.x: the cell is int() of a FLOAT64 quotient.  Centres on
    stride multiples where fp32 and float64 truncate differently must land in the float64 cell, in the
    oracle and in the host generator alike (the device kernel: tests/test_gpu_targets.py)."""
    from conftest import check_float64_cell_case, float64_cell_case
    size, gt, ids, expected = float64_cell_case()
    # the case is only a test if fp32 arithmetic really disagrees on some cell
    f32 = [int(np.float32(240.0) / np.float32(416) * np.float32(26)), int(np.float32(120.0) / np.float32(416) * np.float32(52))]
    assert f32 == [15, 15] and [expected[0][2], expected[1][3]] == [14, 14]
    check_float64_cell_case(TOr.prefetch_targets(4, size, size, gt, ids), expected)
    check_float64_cell_case(targets.YOLOV3PrefetchTargetGenerator(4)(size, size, gt, ids), expected)


def test_split_sizes_even_split_false():
    assert parallel.split_sizes(64, 8) == [8] * 8
    assert parallel.split_sizes(10, 4) == [3, 3, 2, 2]         # gluon split_data(even_split=False)
    x = np.arange(10)
    parts = [parallel.scatter_frames(x, r, 4) for r in range(4)]
    assert np.array_equal(np.concatenate(parts), x) and [len(p) for p in parts] == [3, 3, 2, 2]


_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from videoyolo_amd import parallel
parallel.init_process_group("gloo")
r, w = parallel.rank(), parallel.world_size()
assert w == 2
# gradient all-reduce: sum over ranks of a flat buffer
g = torch.full((1000,), float(r + 1))
parallel.allreduce_(g)
assert torch.all(g == 3.0)
# SyncBN statistics: [2][C] doubles summed over ranks
s = torch.tensor([1.0 + r, 10.0 * (r + 1)], dtype=torch.float64)
parallel.allreduce_(s)
assert s.tolist() == [3.0, 30.0]
# frame scatter + host gather of detections
frames = np.arange(7 * 3).reshape(7, 3)
mine = parallel.scatter_frames(frames)
assert len(mine) == (4 if r == 0 else 3)
ids = torch.full((len(mine), 5, 1), float(r)); sc = torch.ones(len(mine), 5, 1); bb = torch.zeros(len(mine), 5, 4)
gi, gs, gb = parallel.gather_detections(ids, sc, bb)
assert gi.shape == (7, 5, 1) and gb.shape == (7, 5, 4)
assert gi[:4].eq(0).all() and gi[4:].eq(1).all()
# ... the same with the batch size known (no host-side size exchange: the pipelined loop's form), uneven and even slices
gi2, gs2, gb2 = parallel.gather_detections(ids, sc, bb, total=7)
assert torch.equal(gi2, gi) and torch.equal(gs2, gs) and torch.equal(gb2, gb)
e_ids = torch.arange(3 * 5, dtype=torch.float32).reshape(3, 5, 1) + 100 * r
ei, es, eb = parallel.gather_detections(e_ids, sc[:3] * (r + 1), bb[:3] + r, total=6)
assert ei.shape == (6, 5, 1) and torch.equal(ei[:3], e_ids - 100 * r) and torch.equal(ei[3:], e_ids - 100 * r + 100)
assert es[:3].eq(1).all() and es[3:].eq(2).all() and eb[:3].eq(0).all() and eb[3:].eq(1).all()
try:
    parallel.gather_detections(e_ids, sc[:3], bb[:3], total=8)   # every rank would hold 4 of 8: refused before any collective
    raise SystemExit("a slice that does not match `total` was accepted")
except ValueError:
    pass
torch.distributed.barrier()
print("rank", r, "ok")
"""


def test_two_rank_gloo_collectives(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o
