"""-m "not gpu": the rank launcher (videoyolo_amd/launch.py) that `python bench.py --gpus N` uses, and the
replica broadcast helper, under real 2-rank gloo groups on CPU."""
import os
import subprocess
import sys

from videoyolo_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_OK = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from videoyolo_amd import parallel
parallel.init_process_group("gloo")
r, w = parallel.rank(), parallel.world_size()
assert w == 2 and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == r
# what Trainer.__init__ does with the flat parameter buffer: rank 0's values everywhere
buf = torch.full((1000,), float(r + 1))
parallel.broadcast_(buf, 0)
assert torch.all(buf == 1.0)
class FakeNet:
    _dev_params = torch.arange(64, dtype=torch.float32).mul(r + 1).view(torch.uint8)
    _mom = None
    _replicas_synced = False
net = FakeNet()
assert parallel.sync_replicas(net) and net._replicas_synced
assert torch.equal(net._dev_params.view(torch.float32), torch.arange(64, dtype=torch.float32))
# the recorded forward's resync: only rank 1 wrote parameters since the broadcast — BOTH ranks must enter the
# broadcast (all-reduce MAX of the dirty flags first), and rank 0's values win again
FakeNet._device = None
assert parallel.make_host_group() is not None and parallel.host_group() is not None
assert not parallel.sync_replicas_if_any_dirty(net)            # nobody dirty: no broadcast anywhere
if r == 1:
    net._dev_params.view(torch.float32).mul_(3.0)
    net._replicas_synced = False
assert parallel.any_rank(r == 1) and not parallel.any_rank(False)
assert parallel.sync_replicas_if_any_dirty(net) and net._replicas_synced
assert torch.equal(net._dev_params.view(torch.float32), torch.arange(64, dtype=torch.float32))
torch.distributed.barrier()
print("rank %%d of %%d ok" %% (r, w))
"""

_FAIL = r"""
import os, sys, time
if os.environ["RANK"] == "1":
    sys.exit(7)
time.sleep(600)   # rank 0 would hang in a collective: the parent must terminate it
"""


def test_needs_spawn_only_without_a_rank_environment(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert launch.needs_spawn(2) and not launch.needs_spawn(1)
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert not launch.needs_spawn(2)          # already a rank (torch.distributed.run or our own parent)


def test_spawn_ranks_runs_n_ranks_and_returns_zero(tmp_path, capfd):
    script = tmp_path / "ok.py"
    script.write_text(_OK % ROOT)
    assert launch.spawn_ranks(2, [str(script)], timeout=300) == 0
    assert "rank 0 of 2 ok" in capfd.readouterr().out      # rank 0's stdout is the parent's


def test_spawn_ranks_propagates_a_failing_rank_and_stops_the_others(tmp_path):
    script = tmp_path / "fail.py"
    script.write_text(_FAIL)
    import time
    t0 = time.time()
    assert launch.spawn_ranks(2, [str(script)], timeout=300) == 7
    assert time.time() - t0 < 60


def test_bench_parent_never_initialises_a_gpu():
    """`python bench.py --gpus 2` on a box without a GPU: the parent must get as far as starting two ranks
    (which then fail on torch.cuda) and return their non-zero code — not die on a GPU call itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["VY_LAUNCH_TRACE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert err.count("[launch] rank") == 2, err[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0
