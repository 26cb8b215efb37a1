"""Per-launch table of one inference pass (HIP events around every launch, median of 3 passes).
usage: python tools/layer_profile.py [--size 608] [--batch 64] [--out profiles/xxx.txt]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import videoyolo_amd as vy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=608)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--classes", type=int, default=20)
ap.add_argument("--out", default=None)
ap.add_argument("--conv-mode", default="exact", choices=["exact", "split_bf16x3"])
a = ap.parse_args()
net = vy.yolo3_darknet53(["c%d" % i for i in range(a.classes)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
net.set_conv_mode(a.conv_mode)
x = torch.randn((a.batch, 3, a.size, a.size), device="cuda:0")
for _ in range(2):
    net(x)
passes = [net.profile(x) for _ in range(3)]
pinfo = {p.name: p for p in net.collect_params().values()}
lines = ["%-34s %-18s %9s %9s %8s %8s" % ("launch", "w(O,I,k,k)", "ms", "GFLOP", "TFLOP/s", "GB/s")]
tot_ms = tot_fl = 0
for j, (name, _, fl, by) in enumerate(passes[0]):
    ms = sorted(p[j][1] for p in passes)[1]
    w = pinfo.get(name.split("|")[0] + ".0.weight") or pinfo.get(name.split("|")[0] + ".weight")
    shp = "x".join(map(str, w.shape)) if w is not None else "-"
    lines.append("%-34s %-18s %9.4f %9.2f %8.1f %8.0f" % (name, shp, ms, fl / 1e9, fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0))
    tot_ms += ms; tot_fl += fl
lines.append("%-34s %-18s %9.4f %9.2f %8.1f" % ("TOTAL", "", tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))
lines.append("frames/s (sum of launches): %.1f" % (a.batch / tot_ms * 1e3))
txt = "\n".join(lines)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")
