"""-m gpu: the collectives of the training path executed by RCCL (backend "nccl") on the test box's one GPU, before
the driver's multi-GPU run does it for the first time.  One rank with VY_FORCE_COLLECTIVES=1 (videoyolo_amd.parallel):
every broadcast / all-reduce goes through the backend; over one rank each is the identity, so the step must equal the
same step without a process group bit for bit.  Reference behaviour this stands in for: train_yolov3.py:350-354
(SyncBatchNorm(num_devices)), :527-530 (Trainer, kvstore), :634 (trainer.step).  Two-rank semantics (the sums really
adding up) are covered over gloo in tests/test_gpu_multirank.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_training_collectives_run_through_rccl():
    env = dict(os.environ, VY_DIST_TIMEOUT_S="120")
    env.pop("VY_FORCE_COLLECTIVES", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert r["host_group_backend"] == "gloo"                    # the gloo side group beside the NCCL default group
    assert r["any_rank"] == [False, True]
    # six SyncBatchNorm layers (stem + five stride-2 convs): [2][C] doubles forward, then backward in reverse order
    assert r["sync_calls"] == [64, 128, 256, 512, 1024, 2048, 2048, 1024, 512, 256, 128, 64], r["sync_calls"]
    assert len(r["buckets"]) == 4                                # heads, stage 2, 1, 0 on the side stream
    assert r["losses_equal"] and r["grads_equal"] and r["params_equal"] and r["plain_equal"]
    assert r["f64_allreduce_identity"]
    assert r["failure_surfaces_as"] and "VyError" in r["failure_surfaces_as"], r["failure_surfaces_as"]
    assert r["group_marked_failed"]


def test_bench_runs_its_training_legs_over_rccl_on_one_gpu():
    """`bench.py --gpus 1 --backend nccl --force-collectives`: the driver's N > 1 line shape (also_train416 with the
    all-reduce timed alone, also_syncbn608) produced on one GPU through RCCL."""
    env = dict(os.environ, MASTER_PORT="29551")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "1", "--backend", "nccl",
                        "--force-collectives", "--size", "96", "--batch", "2", "--steps", "2", "--warmup", "1",
                        "--train-size", "96", "--train-batch", "2", "--train-steps", "2", "--syncbn-size", "96",
                        "--syncbn-batch", "2", "--cpu-frames", "0", "--no-pmc", "--no-latency", "--no-split-leg",
                        "--src-hw", "60x80", "--vid-src-hw", "48x64", "--vid-size", "96", "--vid-batch", "2",
                        "--multiscale-sizes", "64,96", "--multiscale-interval", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    # the line proves what it ran on: RCCL, its version, the device's PCI address, the pre-flight's 246.5 MB all-reduce
    col = r["collective"]
    assert col["backend"] == "nccl" and col["world"] == 1 and col["rccl_version"] and col["distinct_devices"] == 1
    d0 = col["devices"][0]
    assert d0["rank"] == 0 and d0["cus"] > 0 and d0["hbm_bytes"] > 100e9 and d0["streamk_enabled"] in (True, False)
    assert d0["pci_bus_id"] or d0["uuid"], d0
    assert col["preflight"]["ok"] and col["preflight"]["backend"] == "nccl" and col["preflight"]["allreduce_ms"] > 0
    assert r["also_train_multiscale"]["backend"] == "nccl"
    leg = r["also_train416"]
    assert leg["backend"] == "nccl" and leg["allreduce"].startswith("bucketed")
    assert leg["allreduce_alone_ms"] > 0 and leg["allreduce_bytes"] > 240e6
    assert "allreduce_exposed_ms" in leg and "allreduce_overlap_fraction" in leg
    assert r["also_syncbn608"]["batchnorm"].startswith("SyncBatchNorm(num_devices=1)")
    # the result gather of the host-fed legs went through RCCL too (an all-gather over one rank)
    assert "all-gather of the rows over nccl" in r["also_vid96"]["pipeline"] and r["also_vid96"]["frames_per_s"] > 0
    assert "all-gather of the rows over nccl" in r["also_hostfed96"]["pipeline"]
