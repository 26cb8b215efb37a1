"""Where does the detection stage's time go at batch 1?  decode_nms (hist -> select -> collect -> sort_nms launches) timed
by the library's own per-launch events under different set_nms settings and objectness biases.
usage: python tools/nms_latency.py [--size 608] [--batch 1]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import videoyolo_amd as vy

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=608)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--trace", action="store_true", help="library built with -DVY_NMS_TRACE: phase stamps of sort_nms_kernel")
a = ap.parse_args()
net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
x = torch.randn((a.batch, 3, a.size, a.size), device="cuda:0")
for nms in [(0.45, 400, 100), (0.45, 100, 100), (0.45, 200, 100), (0.45, 1000, 100), (0.0, 400, 100), (0.45, 400, 10)]:
    net.set_nms(nms_thresh=nms[0], nms_topk=nms[1], post_nms=nms[2])
    for _ in range(3):
        net(x)
    ms = sorted(dict((n, t) for n, t, _, _ in net.profile(x))["decode_nms"] for _ in range(5))[2]
    ids, scores, boxes = net(x)
    print("nms_thresh %.2f topk %4d post_nms %3d : decode_nms %.1f us   kept rows %d" % (nms + (ms * 1e3, int((ids[0] >= 0).sum()))))
    if a.trace:
        import ctypes
        from videoyolo_amd import _lib
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        assert ctypes.CDLL(_lib.LIB_PATH).vy_debug_nms_trace(buf) == 0
        names = ["keys", "bitonic sort", "gather", "mask", "greedy", "scan", "rows", ]
        print("    sort_nms phases (us): " + "  ".join("%s %.1f" % (n, (buf[i + 1] - buf[i]) / 100.0) for i, n in enumerate(names)))
