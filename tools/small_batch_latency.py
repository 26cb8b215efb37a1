"""Latency of one inference call (frames resident in HBM -> detection rows) at small batches, 608x608, 20 classes.
usage: python tools/small_batch_latency.py [--size 608] [--batches 1,2,4,8,16]   (VY_CONV_SK=0 for plain launches)"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=608)
ap.add_argument("--batches", default="1,2,4,8,16")
ap.add_argument("--conv-mode", default="exact", choices=["exact", "split_bf16x3"])
ap.add_argument("--graph", action="store_true", help="hybridize: replay a captured HIP graph")
a = ap.parse_args()
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
net.set_nms(0.45, 400, 100)
net.set_conv_mode(a.conv_mode)
if a.graph:
    net.hybridize()
GF = {608: 139.76, 416: 65.43}.get(a.size)  # forward GFLOP per frame
for b in [int(t) for t in a.batches.split(",")]:
    x = torch.randn((b, 3, a.size, a.size), device="cuda:0")
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    names = [name for name, _, _, _ in net.profile(x)]
    sk = sum(1 for n_ in names if n_.endswith("sk"))
    sp = sum(1 for n_ in names if "|split" in n_ or "|wino" in n_)
    spk = sum(1 for n_ in names if "|split" in n_ and "k" in n_.split("|split")[1])
    print("batch %2d: %7.3f ms  %7.1f frames/s  %5.1f TFLOP/s  (%d stream-K launches, %d split-fp32 launches of which %d k-split)"
          % (b, ms, b / ms * 1e3, (GF * b / ms) if GF else 0.0, sk, sp, spk))
