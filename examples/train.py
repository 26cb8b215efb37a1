"""The reference's training loop (train_yolov3.py:494-640) on this package with synthetic VOC-shaped batches; with
--gpus N the script starts one process per GPU itself (RCCL), like `python train_yolov3.py --gpus 0,1,..`.

    python examples/train.py [--gpus 1] [--batch 16] [--size 416] [--steps 5] [--syncbn] [--conv-mode split_bf16x3_train]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videoyolo_amd import launch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="per GPU")
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--syncbn", action="store_true")
    ap.add_argument("--conv-mode", default="exact", choices=["exact", "split_bf16x3_train"],
                    help="split_bf16x3_train: opt-in split-fp32 convs and weight gradients on the bf16 matrix core (DESIGN.md 4.7, 4.8)")
    args = ap.parse_args()
    if launch.needs_spawn(args.gpus):            # this process only starts the ranks; it never touches a GPU
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import autograd, parallel, targets

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        parallel.init_process_group()
    classes = ["c%d" % i for i in range(20)]
    kw = dict(norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": world}) if args.syncbn else {}
    net = vy.yolo3_darknet53(classes, pretrained_base=False, **kw)     # train_yolov3.py:350-360
    net.initialize()                                                   # unseeded, like the reference: Trainer() broadcasts rank 0's
    net.collect_params().reset_ctx(dev)
    net.set_conv_mode(args.conv_mode)                                  # no counterpart in the reference (mxnet picks its conv algorithm)
    for p in net.collect_params(".*beta|.*gamma|.*bias").values():     # --no_wd, :495-497
        p.wd_mult = 0.0
    sched = vy.LRSequential([vy.LRScheduler("linear", base_lr=0, target_lr=1e-3, nepochs=0, iters_per_epoch=100, niters=2),
                             vy.LRScheduler("step", base_lr=1e-3, nepochs=10, iters_per_epoch=100, step_epoch=[6, 8],
                                            step_factor=0.1)])
    trainer = vy.Trainer(net.collect_params(), "sgd", {"wd": 5e-4, "momentum": 0.9, "lr_scheduler": sched})   # :517-530
    trainer.enable_overlap()
    gen = targets.YOLOV3PrefetchTargetGenerator(len(classes))
    g = torch.Generator().manual_seed(rank)
    for step in range(args.steps):
        x = torch.randn((args.batch, 3, args.size, args.size), generator=g).to(dev)
        gt_boxes, gt_ids = targets.synthetic_gt(args.batch, args.size, len(classes), m=8, seed=1000 * rank + step)
        fixed = gen(args.size, args.size, gt_boxes, gt_ids, device=dev)                 # yolo_target.py:13-148 on the GPU
        with autograd.record():
            obj, ctr, scl, cls = net(x, torch.as_tensor(gt_boxes).to(dev), *fixed)     # :625
            autograd.backward([obj + ctr + scl + cls])                                  # :626,631
        trainer.step(args.batch * world)                                                # :634
        if rank == 0:
            print("step %d  lr %.2e  obj %.3f  center %.3f  scale %.3f  cls %.3f" % (
                step, trainer.learning_rate, obj.mean().item(), ctr.mean().item(), scl.mean().item(), cls.mean().item()))
    if rank == 0:
        net.save_parameters("/tmp/yolo3_darknet53_example.params")                      # :293
        print("saved /tmp/yolo3_darknet53_example.params")


if __name__ == "__main__":
    main()
