"""Large-batch sanity of conv mode split_bf16x3 (Winograd cells included): 608x608, batch 160 (planes > 4 GiB) against the
same frames as batches of 64 — the heads must be bit-identical (the kernels' per-output arithmetic does not depend on the
tiling of the batch as long as the launch stays on the same kernel), and a 100-step soak of the batch-64 step.
usage: python tools/big_batch_split_check.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import videoyolo_amd as vy

net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
net.set_conv_mode("split_bf16x3")
g = torch.Generator().manual_seed(5)
x = torch.randn((160, 3, 608, 608), generator=g).cuda()
names = [r[0] for r in net.profile(x)]
print("batch 160: %d Winograd launches, %d split launches" % (sum("|wino" in n for n in names), sum("|split" in n for n in names)))
out = net(x)
big = [net.read_head(i).clone() for i in range(3)]
ok = True
for lo in (0, 64, 96):
    net(x[lo:lo + 64].contiguous())
    for i in range(3):
        ok &= bool(torch.equal(net.read_head(i), big[i][lo:lo + 64]))
print("heads of batch 160 identical to the same frames in batches of 64:", ok)
xs = x[:64].contiguous()
net(xs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    net(xs)
torch.cuda.synchronize()
print("soak: 100 steps of batch 64: %.1f frames/s; memory %.1f GiB" % (6400 / (time.perf_counter() - t0), torch.cuda.max_memory_allocated() / 2**30))
sys.exit(0 if ok else 1)
