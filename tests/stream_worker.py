"""Rank program of tests/test_gpu_stream.py (not a test module): the host-fed pipelined detect loop
(videoyolo_amd/stream.py) under a real process group — frames scattered, rows gathered.  All ranks share cuda:0 over gloo
on a one-GPU box (--share-gpu), or own a GPU each over RCCL.

    python tests/stream_worker.py OUTDIR [--backend gloo] [--share-gpu] [--global-batch 5] [--batches 3] [--classes 30]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clip_batches(n, gb, h, w):
    import numpy as np
    rng = np.random.default_rng(5)
    return [rng.integers(0, 256, (gb, h, w, 3), dtype=np.uint8) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--global-batch", type=int, default=5)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--classes", type=int, default=30)
    ap.add_argument("--size", type=int, default=96)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import videoyolo_amd as vy
    from videoyolo_amd import stream
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.share_gpu else int(os.environ["LOCAL_RANK"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    kw = {"device_id": dev} if args.backend == "nccl" else {}
    dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
    net = vy.yolo3_darknet53(["c%d" % i for i in range(args.classes)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233, obj_bias=-2.0)   # every rank: the same weights
    net.collect_params().reset_ctx(dev)
    net.set_nms(0.45, 400, 100)
    det = stream.HostFedDetector(net, args.global_batch, (60, 80), args.size, depth=2, gather=True)
    # (copies: a result is a view of its slot's pinned buffer, valid until the slot is submitted again)
    outs = [None if o is None else [np.array(a) for a in o] for o in det.run(clip_batches(args.batches, args.global_batch, 60, 80))]
    if rank == 0:
        np.savez(os.path.join(args.outdir, "gathered.npz"),
                 **{"%s%d" % (n, i): o[j] for i, o in enumerate(outs) for j, n in enumerate(("ids", "scores", "bboxes"))})
    else:
        assert all(o is None for o in outs)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
