"""Where do the NMS kept rows of the split-fp32 conv mode differ from the oracle's, and why?
usage: python tools/split_keep_diff.py [--size 416] [--batch 1] [--obj-bias 0]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import videoyolo_amd as vy
from videoyolo_amd import init
from oracle import yolo3_oracle as O
from conftest import frames, VOC_CLASSES

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=416)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--obj-bias", type=float, default=0.0)
a = ap.parse_args()
params = init.synthetic_params(O.param_shapes(20), seed=233, obj_bias=a.obj_bias)
x = frames(a.batch, a.size)
net = vy.yolo3_darknet53(VOC_CLASSES, pretrained_base=False)
net.set_parameters(params)
net.collect_params().reset_ctx("cuda:0")
net.set_conv_mode("split_bf16x3")
net.set_nms(0.45, 400, 100)
ids, scores, bboxes, keep = [t.cpu().numpy() for t in net(x, return_index=True)]
orc = O.OracleYolo3(20, params)
r_ids, r_scores, r_bboxes, r_keep = orc(x)
det = orc.detections(x)  # (B, N*C, 6) in the reference's row order
for b in range(a.batch):
    bad = np.nonzero(keep[b] != r_keep[b])[0]
    print("image %d: %d of %d kept rows differ" % (b, len(bad), keep.shape[1]))
    for j in bad[:12]:
        g, w = int(keep[b, j]), int(r_keep[b, j])
        print("  slot %3d: split row %7d score %.9f | oracle row %7d score %.9f | oracle's scores of the two rows: %.9f %.9f (gap %.2e)"
              % (j, g, scores[b, j, 0], w, r_scores[b, j, 0], det[b, g, 1] if g >= 0 else -1, det[b, w, 1] if w >= 0 else -1,
                 abs(det[b, g, 1] - det[b, w, 1]) if g >= 0 and w >= 0 else float("nan")))
    print("  same SET of rows: %s" % (set(keep[b].tolist()) == set(r_keep[b].tolist())))
