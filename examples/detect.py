"""The reference's inference loop (detect_yolo3.py:199-265, 327-330) on this package, with synthetic frames:
decoded uint8 frames of any size -> resize + to_tensor + normalise on the GPU -> net(x) -> clip / filter /
normalise -> prediction lines, then VOC mAP against made-up ground truth.

    python examples/detect.py [--size 608] [--batch 4] [--frames 8] [--conv-mode split_bf16x3]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import metrics, transforms  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--conv-mode", default="exact", choices=["exact", "split_bf16x3"],
                    help="exact: the parity path (default); split_bf16x3: the opt-in bf16 x 3 arithmetic (DESIGN.md 4.7)")
    args = ap.parse_args()
    classes = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
               "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
    net = vy.yolo3_darknet53(classes, pretrained_base=False)   # detect_yolo3.py:873
    net.initialize(init="synthetic", seed=233, obj_bias=-4.0)   # no checkpoint offline: synthetic weights
    net.collect_params().reset_ctx("cuda:0")                    # :199
    net.set_nms(nms_thresh=0.45, nms_topk=400)                  # :200
    net.set_conv_mode(args.conv_mode)                           # no counterpart in the reference (mxnet picks its conv algorithm)
    tf = transforms.YOLO3VideoInferenceTransform(args.size, args.size)
    metric = metrics.VOCMApMetric(iou_thresh=0.5, class_names=classes)
    rng = np.random.default_rng(0)
    lines = []
    for start in range(0, args.frames, args.batch):
        n = min(args.batch, args.frames - start)
        frames = rng.integers(0, 256, (n, 360, 640, 3), dtype=np.uint8)  # "decoded video frames"
        x = tf(frames)                                           # transforms.py:316-350, one kernel
        ids, scores, bboxes = net(x)                             # detect_yolo3.py:222
        rows = transforms.postprocess(ids, scores, bboxes, args.size)    # :226, :256-265
        for i, r in enumerate(rows):
            lines += transforms.prediction_lines("frame_%05d.jpg" % (start + i), r)      # :327-330
        gt_boxes = rng.uniform(0, args.size - 64, (n, 3, 2))
        gt_boxes = np.concatenate([gt_boxes, gt_boxes + rng.uniform(16, 64, (n, 3, 2))], -1)
        gt_ids = rng.integers(0, len(classes), (n, 3, 1)).astype(np.float64)
        metric.update(bboxes.clip(0, args.size), ids, scores, gt_boxes, gt_ids)
    names, values = metric.get()
    print("%d frames, %d prediction lines; first: %s" % (args.frames, len(lines), lines[0].strip() if lines else "-"))
    print("%s = %.4f (random weights against random boxes: a plumbing check, not a score)" % (names[-1], values[-1]))


if __name__ == "__main__":
    main()
