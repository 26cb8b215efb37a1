"""Rank program of tests/test_gpu_multirank.py (not a test module): the data-parallel PRODUCT path —
Trainer's parameter broadcast, GradBucketOverlap, SyncBatchNormHook, Trainer.step with world > 1 — under a
real process group.  Started through videoyolo_amd.launch.spawn_ranks; all ranks share cuda:0 over gloo
when there is one GPU (--share-gpu), or own a GPU each over RCCL (--backend nccl).

    python tests/dp_worker.py OUTDIR [--backend gloo] [--share-gpu] [--size 64] [--per-rank 2] [--classes 3]

Every rank writes OUTDIR/rank<r>.npz; the parent test compares the ranks with each other and with the
oracle's data-parallel mode.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WATCH = ["stages.0.0.0.weight", "stages.0.1.1.gamma", "stages.0.2.body.0.1.beta", "stages.1.0.0.weight",
         "stages.2.0.1.gamma", "yolo_blocks.0.body.0.0.weight", "yolo_blocks.2.tip.1.beta",
         "transitions.1.0.weight", "yolo_outputs.1.prediction.bias", "yolo_outputs.2.prediction.weight"]
RUNNING = ["stages.0.0.1.running_mean", "stages.0.1.1.running_var", "stages.0.2.body.0.1.running_mean",
           "stages.1.0.1.running_mean", "yolo_blocks.1.body.2.1.running_var"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--per-rank", type=int, default=2)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--multiscale", action="store_true", help="phase C: a second recorded step at size + 32")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import videoyolo_amd as vy
    from videoyolo_amd import _lib, autograd, init, parallel
    from conftest import frames
    from oracle import targets_oracle as T  # inputs only (synthetic gt + targets); the parent does the checking

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.share_gpu else int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    kw = {"device_id": dev} if args.backend == "nccl" else {}
    dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    C, S, pb = args.classes, args.size, args.per_rank
    B = pb * world
    x = frames(B, S, seed=8)
    gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=3, seed=4, pad_to=4)
    tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
    sl = slice(rank * pb, (rank + 1) * pb)
    mine = [x[sl], gt_boxes[sl]] + [t[sl] for t in tg]
    out = {}

    def make_net(**kw):
        net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False, **kw)
        # every rank draws DIFFERENT weights (as an unseeded net.initialize() would): rank 0's must win
        net.initialize(init="synthetic", seed=17 + 100 * rank)
        net.collect_params().reset_ctx(dev)
        return net

    def fwd_bwd(net):
        with autograd.record():
            losses = net(*mine)
            autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
        return np.stack([l.cpu().numpy() for l in losses])

    # ---- A. parameter broadcast + bucketed overlapped all-reduce == plain all-reduce, per-device BatchNorm
    net = make_net()
    before = net.collect_params()["stages.1.3.body.0.0.weight"].data()
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    out["bcast_before"] = before
    out["bcast_after"] = net.collect_params()["stages.1.3.body.0.0.weight"].data()
    trainer.enable_overlap()
    out["A_losses"] = fwd_bwd(net)
    trainer.allreduce_grads()
    torch.cuda.synchronize()
    g_overlap = net._grads.clone()
    for n in RUNNING:                                   # after ONE recorded forward
        out["A_running/" + n] = net.collect_params()[n].data()
    out["A_buckets"] = np.array(trainer._overlap.last_launched, np.int64)
    # the host-side mirror of the library's bucket bookkeeping (parallel.grad_bucket_table: what the world-4 / world-8 CPU
    # tests check for coverage) is what the library actually reported during this backward
    from videoyolo_amd import parallel as _par
    assert [tuple(b) for b in trainer._overlap.last_launched] == [(o, c) for _, o, c in _par.grad_bucket_table(net)], \
        (trainer._overlap.last_launched, _par.grad_bucket_table(net))
    trainer.disable_overlap()
    fwd_bwd(net)
    trainer.allreduce_grads()
    torch.cuda.synchronize()
    out["A_overlap_equals_plain"] = np.array(torch.equal(g_overlap, net._grads))
    out["A_grad_absmax"] = np.array(float(g_overlap.abs().max()))
    for n in WATCH:
        out["A_grad/" + n] = net.grad(n)
    del net, trainer, g_overlap

    # ---- B. SyncBatchNorm + one full Trainer.step.  The net is built ONLY through the constructor argument, the way
    # train_yolov3.py:350-354 does it: no explicit hook — the first recorded forward installs the exchange
    net = make_net(norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": world})
    trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
    trainer.enable_overlap()
    assert net._sync_hook is None
    out["B_losses"] = fwd_bwd(net)
    hook = net._sync_hook
    assert hook is not None, "norm_layer=SyncBatchNorm did not install the statistics exchange"
    trainer.allreduce_grads()
    for n in WATCH:
        out["B_grad/" + n] = net.grad(n)
    for n in RUNNING:
        out["B_running/" + n] = net.collect_params()[n].data()
    out["B_sync_calls"] = np.array(hook.calls, np.int64)
    trainer.update(B)
    for n in WATCH:
        out["B_param/" + n] = net.collect_params()[n].data()
    # second step through the one-call form
    l2 = None
    with autograd.record():
        losses = net(*mine)
        autograd.backward(losses)
    trainer.step(B)
    out["B_losses2"] = np.stack([l.cpu().numpy() for l in losses])
    out["B_param2/stages.0.0.0.weight"] = net.collect_params()["stages.0.0.0.weight"].data()
    out["B_sync_calls_step2"] = np.array(hook.calls, np.int64)     # the log covers one step, it does not grow
    torch.cuda.synchronize()
    del net, trainer, hook

    # ---- C. multi-scale under the hook and the bucket overlap (train_yolov3.py:258-271 trains at 320..608): a
    # recorded step at S, then one at S + 32 — the workspace is re-planned (and re-allocated: it grows) in between;
    # the statistics callback and the gradient buckets must follow it.  No update in between, so the second step is
    # comparable with the oracle on the original parameters.
    if args.multiscale:
        S2 = S + 32
        x2 = frames(B, S2, seed=9)
        gt2, ids2 = T.synthetic_gt(B, S2, C, m=3, seed=5, pad_to=4)
        tg2 = T.prefetch_targets(C, S2, S2, gt2, ids2)
        mine2 = [x2[sl], gt2[sl]] + [t[sl] for t in tg2]
        net = make_net(norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": world})
        trainer = vy.Trainer(net.collect_params(), 'sgd', {'learning_rate': 1e-3, 'wd': 5e-4, 'momentum': 0.9})
        trainer.enable_overlap()
        fwd_bwd(net)
        trainer.allreduce_grads()
        ws1 = (net._ws.data_ptr(), net._ws.numel())
        with autograd.record():
            losses = net(*mine2)
            autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
        trainer.allreduce_grads()
        out["C_losses"] = np.stack([l.cpu().numpy() for l in losses])
        out["C_replanned"] = np.array(net._ws.numel() > ws1[1])
        out["C_sync_calls"] = np.array(net._sync_hook.calls, np.int64)
        out["C_buckets"] = np.array(trainer._overlap.last_launched, np.int64)
        for n in WATCH:
            out["C_grad/" + n] = net.grad(n)
        torch.cuda.synchronize()
    np.savez(os.path.join(args.outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "done")


if __name__ == "__main__":
    main()
