"""How much of the inference step is the chip's power management?  Times the 608x608 batch-64 forward twice with the
same kernels and launch sequence: once on the synthetic normal frames / weights of bench.py, once with every weight,
BatchNorm shift and input pixel zero (all matrix operands zero: the arithmetic is the same, the switching power is not).
    python tools/zero_data_bench.py [--steps 20]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import videoyolo_amd as vy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=608)
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx(dev)
net.set_nms(0.45, 400, 100)


def run(x, label):
    """conv launches only (HIP events around every launch, median of the passes): with all-equal scores the detection
    tail degenerates, which is not what is being measured"""
    for _ in range(5):
        net(x)
    passes = [net.profile(x) for _ in range(max(3, args.steps // 4))]
    conv = sorted(sum(ms for name, ms, _, _ in p if "|" in name) for p in passes)[len(passes) // 2]
    gf = sum(fl for name, _, fl, _ in passes[0] if "|" in name) / 1e9
    print("%-28s conv launches %.2f ms per step  %.1f TFLOP/s" % (label, conv, gf / conv))
    return conv


x = torch.randn((args.batch, 3, args.size, args.size), generator=torch.Generator().manual_seed(233)).to(dev)
a = run(x, "normal frames and weights")
for p in net.collect_params().values():
    v = np.asarray(p.data())
    if p.name.endswith("running_var"):
        continue
    p.set_data(np.zeros_like(v))
b = run(torch.zeros_like(x), "all operands zero")
c = run(x, "normal frames, zero weights")
print("zero-operand step is %.1f %% shorter" % (100 * (1 - b / a)))
