"""PROBE: how far do the heads move when long-K launches are summed as S independent chains (VY_CONV_KSPLIT=1) instead of one?
One 608 x 608 frame, synthetic weights: |k-split - single chain| on the three head tensors, and whether the kept NMS rows change."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, R)
    import numpy as np
    import torch
    import videoyolo_amd as vy
    net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx("cuda:0")
    x = torch.from_numpy(np.random.default_rng(233).standard_normal((1, 3, 608, 608)).astype(np.float32))
    out = net(x, return_index=True)
    np.savez(sys.argv[1], keep=out[3].cpu().numpy(), **{"h%d" % i: net.read_head(i).cpu().numpy() for i in range(3)})
    sys.exit(0)
import numpy as np
res = {}
for ks in ("0", "1"):
    f = "/tmp/ksplit_num_%s.npz" % ks
    subprocess.run([sys.executable, os.path.abspath(__file__), f], env=dict(os.environ, VY_CONV_KSPLIT=ks), check=True)
    res[ks] = dict(np.load(f))
for i in range(3):
    d = np.abs(res["0"]["h%d" % i] - res["1"]["h%d" % i])
    print("head %d: max |k-split - single chain| = %.3e (mean %.3e, max |head| %.2f)" % (i, d.max(), d.mean(), np.abs(res["0"]["h%d" % i]).max()))
print("kept NMS rows identical:", bool(np.array_equal(res["0"]["keep"], res["1"]["keep"])),
      " differing slots:", int((res["0"]["keep"] != res["1"]["keep"]).sum()))
