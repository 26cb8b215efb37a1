import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["VY_SPLIT_ALWAYS"] = "1"
import numpy as np
import videoyolo_amd as vy
from videoyolo_amd import autograd, init
from oracle import targets_oracle as T
from oracle import yolo3_oracle as O
from conftest import frames
C, B, S = 4, int(os.environ.get("DBG_B", "2")), int(os.environ.get("DBG_S", "64"))
params = init.synthetic_params(O.param_shapes(C), seed=11)
x = frames(B, S, seed=5)
gt_boxes, gt_ids = T.synthetic_gt(B, S, C, m=3, seed=2, pad_to=5)
tg = T.prefetch_targets(C, S, S, gt_boxes, gt_ids)
def run(mode, xin=None):
    x = xin if xin is not None else globals()["x"]
    net = vy.yolo3_darknet53(["c%d" % i for i in range(C)], pretrained_base=False)
    net.set_parameters(params); net.collect_params().reset_ctx("cuda:0"); net.set_conv_mode(mode)
    with autograd.record():
        l = net(x, gt_boxes, *tg)
        autograd.backward([l[0] + l[1] + l[2] + l[3]])
    names = [p.name[:-9] for p in net.collect_params().values() if p.name.endswith(".0.weight")]
    acts = {}
    for n in names:
        try:
            acts[n] = net.read_activation(n).cpu().numpy()
        except Exception as e:
            acts[n] = None
    out = {}
    for n in names:
        try:
            out[n] = net.read_grad_activation(n).cpu().numpy()
        except Exception as e:
            out[n] = None
    grads = {p.name: net.grad(p.name) for p in net.collect_params().values() if p.trainable}
    return out, grads, acts
ea, eg, eac = run("exact")
if os.environ.get("DBG_PERTURB"):
    # intrinsic sensitivity: the EXACT path on an input perturbed by 1 ulp-ish (relative 2e-7) against itself
    pa, pg, pac = run("exact", (x * np.float32(1.0 + 2.0 ** -22)).astype(np.float32))
    worst = max((np.abs(ea[n] - pa[n]).max() / (np.abs(ea[n]).max() + 1e-12), n) for n in ea if ea[n] is not None)
    worst_a = max((np.abs(eac[n] - pac[n]).max() / (np.abs(eac[n]).max() + 1e-12), n) for n in ea if eac[n] is not None)
    print("EXACT vs EXACT on x * (1 + 2^-22): worst grad-act rel diff %.2e at %s; worst act rel diff %.2e" % (worst[0], worst[1], worst_a[0]))
    l2 = max((np.linalg.norm(eg[k] - pg[k]) / (np.linalg.norm(eg[k]) + 1e-30), k) for k in eg)
    mx = max((np.abs(eg[k] - pg[k]).max() / (np.abs(eg[k]).max() + 1e-30), k) for k in eg)
    print("   parameter gradients: worst rel L2 %.2e (%s), worst max-norm %.2e (%s)" % (l2[0], l2[1], mx[0], mx[1]))
sa, sg, sac = run("split_bf16x3")
l2 = max((np.linalg.norm(eg[k] - sg[k]) / (np.linalg.norm(eg[k]) + 1e-30), k) for k in eg)
mx = max((np.abs(eg[k] - sg[k]).max() / (np.abs(eg[k]).max() + 1e-30), k) for k in eg)
print("SPLIT vs EXACT parameter gradients: worst rel L2 %.2e (%s), worst max-norm %.2e (%s)" % (l2[0], l2[1], mx[0], mx[1]))
for n in ea:
    if ea[n] is None: continue
    d = np.abs(ea[n] - sa[n]).max() / (np.abs(ea[n]).max() + 1e-12)
    da = np.abs(eac[n] - sac[n]).max() / (np.abs(eac[n]).max() + 1e-12) if eac[n] is not None else -1
    print("%-28s grad-act rel diff %.2e   act rel diff %.2e   shape %s" % (n, d, da, ea[n].shape))
