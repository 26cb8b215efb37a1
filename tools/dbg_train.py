import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import videoyolo_amd as vy
from videoyolo_amd import autograd
from oracle import yolo3_train_oracle as TO
from test_gpu_train_parity import _setup, _net
C, B, S = 4, 2, 64
params, x, gt_boxes, tg = _setup(C, B, S)
orc = TO.OracleYolo3Train(C, dict(params))
das = {}
orig = orc._cell_bwd
def cb(t, da, grads):
    das[t["pre"]] = da
    return orig(t, da, grads)
orc._cell_bwd = cb
orc.forward_train(x, gt_boxes, *tg); ref = orc.backward()
net = _net(C, params)
with autograd.record():
    losses = net(x, gt_boxes, *tg); autograd.backward([sum(losses)])
outs = {t["pre"]: t["out"] for t in orc.tape if t.get("kind") == "cell"}
nbad = 0
for pre, want in outs.items():
    if ".body.1" in pre and pre.startswith("stages"): continue
    got = net.read_activation(pre).cpu().numpy()
    if got.shape != want.shape: want = want.repeat(2, axis=-1).repeat(2, axis=-2)
    if not np.array_equal(got, want):
        nbad += 1; print("fwd mismatch", pre, np.abs(got - want).max(), ((got > 0) != (want > 0)).sum())
print("forward-train activations differing:", nbad)
order = [t["pre"] for t in orc.tape if t.get("kind") == "cell"][::-1]
for pre in order:
    want = das[pre]
    got = net.read_grad_activation(pre).cpu().numpy()
    if got.shape != want.shape:
        B_, c, H2, W2 = got.shape
        got = got.reshape(B_, c, H2 // 2, 2, W2 // 2, 2).sum(axis=(3, 5))
    e = np.abs(got - want).max() / (np.abs(want).max() + 1e-9)
    gw = net.grad(pre + ".0.weight"); ew = np.abs(gw - ref[pre + ".0.weight"]).max() / (np.abs(ref[pre + ".0.weight"]).max() + 1e-9)
    gb = net.grad(pre + ".1.beta"); eb = np.abs(gb - ref[pre + ".1.beta"]).max() / (np.abs(ref[pre + ".1.beta"]).max() + 1e-9)
    if max(e, ew, eb) > 1e-3: print("%-26s da err %.2e   dW err %.2e  dbeta err %.2e  shape %s" % (pre, e, ew, eb, want.shape))
