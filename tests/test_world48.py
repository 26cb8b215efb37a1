"""-m "not gpu": the host side of the multi-GPU path at WORLD SIZE 4 AND 8 (gloo on CPU, real processes started by
videoyolo_amd/launch.py — the launcher `python bench.py --gpus N` uses).  tests/world_worker.py holds the checks: uneven frame
scatter + result gather in rank order (all-gather and per-rank-broadcast paths, with and without the total known), any_rank,
the gloo side group and its all-ranks fallback when one rank cannot create it, describe_group (what an N-rank bench line says
it ran on), preflight (passes; fails with the same one-line reason on every rank), the gradient bucket table (covers every
trainable tensor once, identical on all ranks, bucket-wise all-reduce == whole-buffer all-reduce)."""
import os

import pytest

from videoyolo_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [4, 8])
def test_host_side_of_the_multi_gpu_path(world, capfd):
    rc = launch.spawn_ranks(world, [os.path.join(HERE, "world_worker.py")], timeout=600)
    out = capfd.readouterr()
    assert rc == 0, out.err[-3000:]
    assert "rank 0 of %d ok" % world in out.out
