"""Subprocess body of test_gpu_fullsize.py::test_stream_k_launches_equal_plain_launches: one seeded network, one seeded
batch; prints which inference launches ran as stream-K and SHA-256 digests of (a) the three raw heads + the detections
and (b) the four losses + all gradients of one recorded training step.  VY_CONV_SK is read once per process by the
library, so the two switch positions need two processes.
With a third argument "graph" the net is hybridized BEFORE its first forward — the capture is then the first time the
library sees these launches (its once-per-instance occupancy queries run inside the capture) — and only the inference
digest is produced, from two replays.
"infer" as the third argument stops after the inference digest (tools/archive/sk_sweep.py).
usage: python sk_digest_worker.py SIZE BATCH [graph|infer]"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import autograd, targets  # noqa: E402
from conftest import frames  # noqa: E402

S, B, C = int(sys.argv[1]), int(sys.argv[2]), 20
net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
net.set_nms(0.45, 400, 100)
x = torch.as_tensor(frames(B, S, seed=3)).cuda()

graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
if graph:
    net.hybridize()
    first = [t.clone() for t in net(x, return_index=True)]   # capture + first replay
    assert len(net._graphs) == 1
    again = net(x, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip(first, again)), "graph replays differ"
labels = [name for name, _, _, _ in net.profile(x)]
h = hashlib.sha256()
for t in net(x, return_index=True):
    h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
for i in range(3):
    h.update(np.ascontiguousarray(net.read_head(i).cpu().numpy()).tobytes())
out = {"sk_launches": [n for n in labels if n.endswith("sk")], "conv_launches": sum("|" in n for n in labels),
       "infer": h.hexdigest()}

if graph or (len(sys.argv) > 3 and sys.argv[3] == "infer"):
    print("DIGEST " + json.dumps(out))
    sys.exit(0)
gt, gid = targets.synthetic_gt(B, S, C, m=8, seed=1)
tg = targets.YOLOV3PrefetchTargetGenerator(C)(S, S, gt, gid, device="cuda:0")
with autograd.record():
    losses = net(x, gt, *tg)
autograd.backward(losses)
h = hashlib.sha256()
for l in losses:
    h.update(np.ascontiguousarray(l.cpu().numpy()).tobytes())
h.update(np.ascontiguousarray(net._grads.cpu().numpy()).tobytes())
out["train"] = h.hexdigest()
print("DIGEST " + json.dumps(out))
