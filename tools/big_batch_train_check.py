"""Training with activation planes of 4 GiB and more (round 4: the weight-gradient pixel table holds offsets relative to
each split's first pixel; round 3 returned VY_ERR_UNSUPPORTED here).  608x608, batch 92: the stage-0 planes are
92 x 610 x 610 x 32 x 4 B = 4.38 GB = 4.08 GiB.  The batch is ONE frame repeated, so every batch statistic equals the single-frame
statistic, every frame's forward / backward is the single-frame one, and the weight gradients (sums over the batch) must be
92 x the single-frame gradients up to the fp32 summation order — an independent check that every pixel of a > 4 GiB plane is
addressed correctly.
usage: python tools/big_batch_train_check.py [--batch 88] [--size 608]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import videoyolo_amd as vy
from videoyolo_amd import autograd, targets

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=92)
ap.add_argument("--size", type=int, default=608)
a = ap.parse_args()
C = 20
classes = ["c%d" % i for i in range(C)]


def grads(batch):
    net = vy.yolo3_darknet53(classes, pretrained_base=False)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    x1 = torch.randn((1, 3, a.size, a.size), generator=g)
    x = x1.repeat(batch, 1, 1, 1).cuda()
    gt1, gid1 = targets.synthetic_gt(1, a.size, C, m=8, seed=1)
    gt, gid = np.repeat(gt1, batch, 0), np.repeat(gid1, batch, 0)
    tg = targets.YOLOV3PrefetchTargetGenerator(C)(a.size, a.size, gt, gid, device="cuda:0")
    with autograd.record():
        losses = net(x, gt, *tg)
    autograd.backward(losses)
    torch.cuda.synchronize()
    out = {p.name: net._grads[p.offset:p.offset + p.size].double().cpu().numpy() for p in net.collect_params().values()
           if p.trainable}
    ls = [l.double().cpu().numpy() for l in losses]
    ws = net._ws.numel() / 2 ** 30
    del net
    torch.cuda.empty_cache()
    return out, ls, ws


g1, l1, _ = grads(1)
gB, lB, ws = grads(a.batch)
plane = a.batch * (a.size + 2) ** 2 * 32 * 4 / 2 ** 30
print("batch %d at %dx%d: workspace %.1f GiB, largest plane %.2f GiB" % (a.batch, a.size, a.size, ws, plane))
ok = all(np.allclose(lb, l1[i][0], rtol=1e-5, atol=1e-6) for i, lb in enumerate(lB))
print("per-frame losses equal the single-frame losses: %s" % ok)
worst = 0.0
for k in g1:
    scale = np.abs(g1[k]).max() * a.batch + 1e-30
    worst = max(worst, float(np.abs(gB[k] - a.batch * g1[k]).max() / scale))
print("max |grad(B) - B x grad(1)| / (B x max|grad(1)|) over %d tensors: %.3e" % (len(g1), worst))
ok &= worst < 2e-3
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
