"""-m gpu: the N > 1 product path under a real process group — 2 ranks sharing the one GPU of the test box
over gloo (RCCL needs 2 devices; the driver's scaling run covers it).  Ranks are started the way
`python bench.py --gpus N` starts them (videoyolo_amd/launch.py): plain child processes of a parent that
never touches the GPU.

  * Trainer broadcasts rank 0's parameters (ranks initialise with different seeds)
  * GradBucketOverlap: gradient buffer after the bucketed, overlapped all-reduce == one all-reduce after
    backward, bit for bit
  * data-parallel step (per-device BatchNorm) and SyncBatchNorm step against the oracle's data-parallel mode:
    losses 1e-4, gradients 2e-3 of each tensor's max, running statistics per device
  * after Trainer.step all ranks hold identical weights
  * bench.py --gpus 2 (both modes) prints a line with n_gpus == 2
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import frames

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2
# (classes, size, frames per rank): a small case, and BASELINE configs[4]'s shape — SyncBN training at 608 x 608,
# VOC classes — at 2 ranks x 2 frames
CASES = {"small": (3, 64, 2), "configs4_608": (20, 608, 2)}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    return CASES[request.param]


@pytest.fixture(scope="module")
def ranks(case, tmp_path_factory):
    from videoyolo_amd import launch
    C, S, PB = case
    out = tmp_path_factory.mktemp("dp")
    rc = launch.spawn_ranks(WORLD, [os.path.join(ROOT, "tests", "dp_worker.py"), str(out), "--backend", "gloo",
                                    "--share-gpu", "--size", str(S), "--per-rank", str(PB), "--classes", str(C)]
                            + (["--multiscale"] if S <= 96 else []),
                            timeout=1500)
    assert rc == 0, "a rank failed (exit code %d)" % rc
    return [dict(np.load(str(out / ("rank%d.npz" % r)))) for r in range(WORLD)]


@pytest.fixture(scope="module")
def problem(case):
    from videoyolo_amd import init
    from oracle import targets_oracle as T
    from oracle import yolo3_oracle as O
    C, S, PB = case
    params = init.synthetic_params(O.param_shapes(C), seed=17)      # rank 0's draw
    B = PB * WORLD
    x = frames(B, S, seed=8)
    gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=3, seed=4, pad_to=4)
    tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
    return params, x, gt_boxes, tg, (C, S, PB)


def test_parameters_are_broadcast_from_rank0(ranks, problem):
    params = problem[0]
    name = "stages.1.3.body.0.0.weight"
    assert np.array_equal(ranks[0]["bcast_before"], params[name])
    assert not np.array_equal(ranks[1]["bcast_before"], params[name])     # rank 1 drew other weights
    for r in ranks:
        assert np.array_equal(r["bcast_after"], params[name])


def test_bucketed_overlap_equals_plain_allreduce(ranks, problem):
    trainable = sum(v.size for k, v in problem[0].items() if "running" not in k)
    for r in ranks:
        assert bool(r["A_overlap_equals_plain"])
        assert float(r["A_grad_absmax"]) > 0
        b = r["A_buckets"]
        assert b.shape == (4, 2)                                          # heads, stages.2, stages.1, stages.0
        assert (np.diff(b[:, 0]) < 0).all()                               # deep -> shallow: descending offsets
        assert b[:, 1].sum() >= trainable                                 # the buckets cover every trainable tensor


def _check_against_oracle(ranks, problem, phase, sync_bn):
    from oracle import yolo3_train_oracle as TO
    params, x, gt_boxes, tg, (C, S, PB) = problem
    slices = [slice(r * PB, (r + 1) * PB) for r in range(WORLD)]
    orc = TO.OracleYolo3Train(C, dict(params), device_slices=slices, sync_bn=sync_bn)
    ref_losses = np.stack(orc.forward_train(x, gt_boxes, *tg))          # (4, B)
    ref_grads = orc.backward()
    for r, sl in enumerate(slices):
        np.testing.assert_allclose(ranks[r][phase + "_losses"], ref_losses[:, sl], rtol=1e-4, atol=1e-4)
        for key in ranks[r]:
            if key.startswith(phase + "_grad/"):
                name = key.split("/", 1)[1]
                want = ref_grads[name]
                err = np.abs(ranks[r][key] - want).max() / (np.abs(want).max() + 1e-6)
                assert err < 2e-3, (name, err)
            if key.startswith(phase + "_running/"):
                name = key.split("/", 1)[1]
                np.testing.assert_allclose(ranks[r][key], orc.new_running_dev[r][name], rtol=1e-4, atol=1e-5)
    return orc, ref_grads


def test_data_parallel_step_matches_the_oracle(ranks, problem):
    if problem[4][1] > 96:
        pytest.skip("full-size case: the SyncBN step below is the configs[4] check (one oracle run)")
    _check_against_oracle(ranks, problem, "A", sync_bn=False)
    # all-reduced gradients are the same bits on every rank
    for key in ranks[0]:
        if key.startswith("A_grad/"):
            assert np.array_equal(ranks[0][key], ranks[1][key]), key
    # per-device BatchNorm: the ranks' running statistics differ
    assert not np.array_equal(ranks[0]["A_running/stages.0.1.1.running_var"], ranks[1]["A_running/stages.0.1.1.running_var"])


def test_syncbn_step_matches_the_oracle(ranks, problem):
    from oracle import yolo3_train_oracle as TO
    params, (C, S, PB) = problem[0], problem[4]
    orc, ref_grads = _check_against_oracle(ranks, problem, "B", sync_bn=True)
    for r in ranks:
        calls = r["B_sync_calls"]
        # one exchange of [2][C] doubles per SyncBatchNorm layer and direction: stem 32 .. stages.2.0 1024
        assert len(calls) == 12 and sorted(set(calls.tolist())) == [64, 128, 256, 512, 1024, 2048]
        assert len(r["B_sync_calls_step2"]) == 12     # per-step log: it does not grow with the number of steps
    # synchronised layers share their running statistics, per-device layers do not
    assert np.array_equal(ranks[0]["B_running/stages.0.1.1.running_var"], ranks[1]["B_running/stages.0.1.1.running_var"])
    assert np.array_equal(ranks[0]["B_running/stages.1.0.1.running_mean"], ranks[1]["B_running/stages.1.0.1.running_mean"])
    assert not np.array_equal(ranks[0]["B_running/stages.0.2.body.0.1.running_mean"],
                              ranks[1]["B_running/stages.0.2.body.0.1.running_mean"])
    # Trainer.update(global batch): every rank applies the same update to the same weights
    B = PB * WORLD
    p_ref = {k: v.copy() for k, v in params.items()}
    TO.sgd_step(p_ref, ref_grads, {}, 1e-3, 0.9, 5e-4, B)
    for key in ranks[0]:
        if key.startswith("B_param/"):
            name = key.split("/", 1)[1]
            assert np.array_equal(ranks[0][key], ranks[1][key]), name
            atol = 2e-6 + 2e-3 * 1e-3 * float(np.abs(ref_grads[name]).max()) / B
            np.testing.assert_allclose(ranks[0][key], p_ref[name], rtol=0, atol=atol)
            assert np.abs(ranks[0][key] - params[name]).max() > 0
    assert np.array_equal(ranks[0]["B_param2/stages.0.0.0.weight"], ranks[1]["B_param2/stages.0.0.0.weight"])
    assert np.isfinite(ranks[0]["B_losses2"]).all()


def test_multiscale_step_under_the_hook_and_the_overlap(ranks, problem):
    """A recorded step at S, then one at S + 32 on the same net (SyncBatchNorm from the constructor argument, bucketed
    overlap on): the re-planned (larger) workspace is what the statistics callback and the buckets see — the second
    step's losses and gradients against the oracle's SyncBN mode at the new size, 12 exchanges, 4 buckets."""
    from oracle import targets_oracle as T
    from oracle import yolo3_train_oracle as TO
    params, _, _, _, (C, S, PB) = problem
    if "C_losses" not in ranks[0]:
        pytest.skip("full-size case runs without the multi-scale phase")
    B, S2 = PB * WORLD, S + 32
    x2 = frames(B, S2, seed=9)
    gt2, ids2 = T.synthetic_gt(B, S2, C, m=3, seed=5, pad_to=4)
    tg2 = T.prefetch_targets(C, S2, S2, gt2, ids2)
    slices = [slice(r * PB, (r + 1) * PB) for r in range(WORLD)]
    orc = TO.OracleYolo3Train(C, dict(params), device_slices=slices, sync_bn=True)
    ref_losses = np.stack(orc.forward_train(x2, gt2, *tg2))
    ref_grads = orc.backward()
    for r, sl in enumerate(slices):
        assert bool(ranks[r]["C_replanned"])
        np.testing.assert_allclose(ranks[r]["C_losses"], ref_losses[:, sl], rtol=1e-4, atol=1e-4)
        assert len(ranks[r]["C_sync_calls"]) == 12
        assert ranks[r]["C_buckets"].shape == (4, 2)
        for key in ranks[r]:
            if key.startswith("C_grad/"):
                name = key.split("/", 1)[1]
                err = np.abs(ranks[r][key] - ref_grads[name]).max() / (np.abs(ref_grads[name]).max() + 1e-6)
                assert err < 2e-3, (name, err)
                assert np.array_equal(ranks[0][key], ranks[r][key]), name


@pytest.mark.parametrize("extra", [[], ["--mode", "train"], ["--mode", "train", "--syncbn"]])
def test_bench_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2` without a rank environment: the parent spawns the ranks itself.  In the default
    mode the line also carries the training legs the driver's scaling run needs: also_train416 (configs[2]) with the
    forward / backward / exposed-all-reduce split, the all-reduce timed alone and the overlap fraction, and — N > 1 —
    also_syncbn608 (configs[4]) built through norm_layer=SyncBatchNorm."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
           "--size", "96", "--batch", "2", "--steps", "2", "--warmup", "1", "--cpu-frames", "0",
           "--train-size", "64", "--train-batch", "2", "--syncbn-size", "96", "--syncbn-batch", "2",
           "--train-steps", "2", "--no-split-leg", "--src-hw", "60x80", "--vid-src-hw", "48x64", "--vid-size", "96", "--vid-batch", "3",
           "--multiscale-sizes", "64,96", "--multiscale-interval", "2"] + extra   # (the split legs at 2 ranks: test_bench_under_torch_distributed_run)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout.decode()   # ONE line on stdout: gloo's "[Gloo] Rank 0 is connected ..." goes to stderr
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["value"] > 0
    assert r["scaling"] == "weak" and r["steps"] == 2
    # the line says what it ran on: two ranks of a gloo group that SHARE one device — reported, not inferred
    col = r["collective"]
    assert col["backend"] == "gloo" and col["world"] == 2 and col["ranks_in_order"] and col["distinct_devices"] == 1
    assert [d["rank"] for d in col["devices"]] == [0, 1] and col["devices"][0]["cus"] > 0 and col["ipc_mode"] == "dmabuf"
    assert col["preflight"]["ok"] and col["preflight"]["allreduce_bytes"] > 240e6 and col["preflight"]["world"] == 2
    if not extra:
        assert all(d["streamk_enabled"] in (True, False) for d in col["devices"])
        ms = r["also_train_multiscale"]
        assert ms["n_gpus"] == 2 and sorted(ms["sizes_in_order"]) == [64, 96] and ms["steps"] == 4 and ms["frames_per_s"] > 0
        assert set(ms["per_size"]) == {"64", "96"} and all(v["replan_ms"] > 0 for v in ms["per_size"].values())
    if not extra:
        t = r["also_train416"]
        for k in ("frames_per_s", "ms_per_step", "forward_ms", "backward_ms", "allreduce_exposed_ms", "sgd_ms",
                  "allreduce_alone_ms", "allreduce_overlap_fraction", "allreduce_busbw_GBps", "traffic"):
            assert k in t, k
        assert t["n_gpus"] == 2 and t["global_batch"] == 4 and t["size"] == 64 and t["frames_per_s"] > 0
        assert t["allreduce_bytes"] >= 61_000_000 * 4 and 0.0 <= t["allreduce_overlap_fraction"] <= 1.0
        # the host-fed, pipelined legs: frames scattered, rows gathered (configs[3] shape: 30 classes) — every rank entered
        hf, vid = r["also_hostfed96"], r["also_vid96"]
        assert hf["n_gpus"] == 2 and hf["global_batch"] == 4 and hf["frames_per_s"] > 0 and hf["source_frames"].startswith("60x80")
        assert vid["n_gpus"] == 2 and vid["classes"] == 30 and vid["global_batch"] == 6 and vid["frames_per_s"] > 0
        assert "all-gather" in vid["pipeline"] and vid["copy_in_bytes"] == 3 * 48 * 64 * 3
        sb = r["also_syncbn608"]
        assert sb["n_gpus"] == 2 and sb["size"] == 96 and sb["frames_per_s"] > 0
        assert sb["batchnorm"].startswith("SyncBatchNorm(num_devices=2)")
        assert "forward_ms" in sb and "allreduce_exposed_ms" in sb
    else:
        assert r["step_split"]["forward_ms"] > 0 and "allreduce_exposed_ms" in r["step_split"]   # (roofline: 416 / 608 only)


def test_bench_under_torch_distributed_run():
    """The driver's own command for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` — bench.py finds RANK / LOCAL_RANK / WORLD_SIZE in the environment,
    does not spawn, and rank 0 prints the one line (all legs of the default mode, the split ones included)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
           "--size", "96", "--batch", "2", "--steps", "2", "--warmup", "1", "--cpu-frames", "0",
           "--train-size", "64", "--train-batch", "2", "--syncbn-size", "96", "--syncbn-batch", "2", "--train-steps", "2",
           "--src-hw", "60x80", "--vid-src-hw", "48x64", "--vid-size", "96", "--vid-batch", "2",
           "--multiscale-sizes", "64,96", "--multiscale-interval", "2"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout.decode()   # exactly the one line of the contract
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["value"] > 0 and r["scaling"] == "weak"
    for leg in ("also_infer96_split", "also_hostfed96", "also_vid96", "also_train416", "also_train416_split", "also_syncbn608"):
        assert leg in r and r[leg]["n_gpus"] == 2, leg
    assert r["also_train416_split"]["frames_per_s"] > 0 and r["also_infer96_split"]["frames_per_s"] > 0


def test_default_bench_line_has_the_training_leg_on_one_gpu():
    """N = 1, default mode (small shapes): also_train416 present with a measured `traffic` when rocprofv3 exists."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "96", "--batch", "2", "--steps", "2", "--warmup", "1",
           "--cpu-frames", "1", "--train-size", "64", "--train-batch", "2", "--train-steps", "2",
           "--src-hw", "60x80", "--vid-src-hw", "48x64", "--vid-size", "96", "--vid-batch", "2",
           "--multiscale-sizes", "64,96", "--multiscale-interval", "2"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    t = r["also_train416"]
    assert t["n_gpus"] == 1 and t["allreduce"].startswith("none") and "also_syncbn608" not in r and "collective" not in r
    assert r["also_train_multiscale"]["steps"] == 4 and r["also_train_multiscale"]["replan_share"] < 1.0
    assert t["forward_ms"] > 0 and t["backward_ms"] > 0 and t["allreduce_exposed_ms"] < 0.5
    if os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert r["roofline"]["traffic"] and t["traffic"] and t["traffic"] > 0, (r["roofline"].get("traffic_note"), t.get("traffic_note"))
    assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline_torch"]["kind"] == "independent"
    assert r["cpu_baseline_torch"]["value"] > 0
    hf = r["also_hostfed96"]
    assert hf["n_gpus"] == 1 and hf["frames_per_s"] > 0 and hf["vs_resident"] > 0 and "exposed_ms_per_step" in hf
    assert r["also_vid96"]["classes"] == 30 and "all-gather" not in r["also_vid96"]["pipeline"]
