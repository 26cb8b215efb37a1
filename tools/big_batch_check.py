"""Large-batch sanity (planes > 4 GiB: 64-bit base + 32-bit tile-relative offsets): batch 160 at 608x608,
frames of the big batch vs the same frames alone; plus a 300-step soak of the default bench shape.
usage: python tools/big_batch_check.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import videoyolo_amd as vy

net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
B = 160
x = torch.randn((B, 3, 608, 608), device="cuda:0")
out = [t.clone() for t in net(x, return_index=True)]
print("workspace GiB: %.1f" % (net._ws.numel() / 2 ** 30))
ok = True
for i in (0, 77, B - 1):
    one = net(x[i:i + 1], return_index=True)
    ok &= all(torch.equal(a[i:i + 1], b) for a, b in zip(out, one))
print("batch %d frames identical to single-frame runs: %s" % (B, ok))
x = x[:64].contiguous()
net(x)
torch.cuda.synchronize()
ts = []
for k in range(6):
    t = time.perf_counter()
    for _ in range(50):
        net(x)
    torch.cuda.synchronize()
    ts.append(64 * 50 / (time.perf_counter() - t))
print("soak 300 steps, fps per 50 steps:", " ".join("%.1f" % v for v in ts), " mem GiB %.1f" % (torch.cuda.memory_allocated() / 2 ** 30))
sys.exit(0 if ok else 1)
