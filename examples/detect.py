"""The reference's inference loop (detect_yolo3.py:199-265, 327-330) on this package, with synthetic frames:
decoded uint8 frames of any size -> resize + to_tensor + normalise on the GPU -> net(x) -> clip / filter /
normalise -> prediction lines, then VOC mAP against made-up ground truth.

    python examples/detect.py [--size 608] [--batch 4] [--frames 8] [--conv-mode split_bf16x3]
    python examples/detect.py --gpus 2 --batch 8       # starts its own ranks: a clip batch is scattered over the GPUs
                                                       # (split_and_load, detect_yolo3.py:211-213), the rows gathered to rank 0

The loop is PIPELINED (videoyolo_amd/stream.py): batch i + 1 is copied to the GPU and batch i - 1's rows are copied back
while batch i computes — three HIP streams, two buffer slots, no host synchronisation except when a finished batch is
asked for.  `--sync` runs the reference's shape instead (transform, net(x), .cpu(), one after the other; one GPU).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import launch, metrics, parallel, stream, transforms  # noqa: E402

SRC_HW = (360, 640)  # "decoded video frames"


def clip_batches(n_frames, batch, seed=0):
    """the same clip on every rank (a shared decoder / file): batches of up to `batch` frames"""
    rng = np.random.default_rng(seed)
    for start in range(0, n_frames, batch):
        n = min(batch, n_frames - start)
        yield start, rng.integers(0, 256, (n,) + SRC_HW + (3,), dtype=np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=4, help="frames per clip batch (over all GPUs)")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: all ranks on cuda:0 (use with --backend gloo)")
    ap.add_argument("--sync", action="store_true", help="the reference's synchronous loop instead of the pipelined one")
    ap.add_argument("--conv-mode", default="exact", choices=["exact", "split_bf16x3"],
                    help="exact: the parity path (default); split_bf16x3: the opt-in bf16 x 3 arithmetic (DESIGN.md 4)")
    args = ap.parse_args()
    if launch.needs_spawn(args.gpus):                            # started the reference's way: one process per GPU from here
        sys.exit(launch.spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if args.share_gpu:
            os.environ["LOCAL_RANK"] = "0"
        parallel.init_process_group(args.backend)
    rank = parallel.rank()
    classes = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
               "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
    net = vy.yolo3_darknet53(classes, pretrained_base=False)   # detect_yolo3.py:873
    net.initialize(init="synthetic", seed=233, obj_bias=-4.0)   # no checkpoint offline: synthetic weights (the same on every rank)
    net.collect_params().reset_ctx("cuda:%d" % local)           # :199
    net.set_nms(nms_thresh=0.45, nms_topk=400)                  # :200
    net.set_conv_mode(args.conv_mode)                           # no counterpart in the reference (mxnet picks its conv algorithm)
    metric = metrics.VOCMApMetric(iou_thresh=0.5, class_names=classes)
    rng = np.random.default_rng(1)
    lines = []

    def consume(start, n, ids, scores, bboxes):
        rows = transforms.postprocess(ids, scores, bboxes, args.size)    # :226, :256-265
        for i, r in enumerate(rows):
            lines.extend(transforms.prediction_lines("frame_%05d.jpg" % (start + i), r))      # :327-330
        gt_boxes = rng.uniform(0, args.size - 64, (n, 3, 2))
        gt_boxes = np.concatenate([gt_boxes, gt_boxes + rng.uniform(16, 64, (n, 3, 2))], -1)
        gt_ids = rng.integers(0, len(classes), (n, 3, 1)).astype(np.float64)
        metric.update(np.clip(bboxes, 0, args.size), ids, scores, gt_boxes, gt_ids)

    batches = list(clip_batches(args.frames, args.batch))
    if args.sync:
        if world > 1:
            sys.exit("--sync is the single-GPU reference loop")
        tf = transforms.YOLO3VideoInferenceTransform(args.size, args.size)
        for start, frames in batches:
            x = tf(frames, device="cuda:%d" % local)             # transforms.py:316-350, one kernel
            ids, scores, bboxes = [t.cpu().numpy() for t in net(x)]   # detect_yolo3.py:222, :233
            consume(start, len(frames), ids, scores, bboxes)
    else:
        # one detector per clip-batch size (the last batch of a clip may be short): buffers are allocated once each
        dets, pending = {}, []

        def drain_one():
            s0, n0, d, slot = pending.pop(0)
            out = d.result(slot)
            if out is not None:                                   # rank 0 holds the gathered rows of the whole clip batch
                consume(s0, n0, *[np.array(a) for a in out])      # (copies: the views belong to the slot)

        for start, frames in batches:
            n = len(frames)
            if n < world:
                sys.exit("a clip batch of %d frames cannot be scattered over %d GPUs" % (n, world))
            if n not in dets:
                while pending:
                    drain_one()
                dets[n] = stream.HostFedDetector(net, n, SRC_HW, args.size, depth=2)
            if len(pending) == dets[n].depth:
                drain_one()
            pending.append((start, n, dets[n], dets[n].submit(frames)))
        while pending:
            drain_one()
    if rank == 0:
        names, values = metric.get()
        print("%d frames, %d prediction lines; first: %s" % (args.frames, len(lines), lines[0].strip() if lines else "-"))
        print("%s = %.4f (random weights against random boxes: a plumbing check, not a score)" % (names[-1], values[-1]))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
