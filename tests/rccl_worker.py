"""Subprocess body of tests/test_gpu_rccl.py: ONE rank, backend "nccl" (= RCCL), ``device_id=`` initialisation,
VY_FORCE_COLLECTIVES=1 — every collective of the training path runs through RCCL on the one GPU of the test box:
Trainer() broadcast, the gloo side group beside the NCCL default group, the float64 [2][C] SyncBatchNorm all-reduce from
the library's ctypes callback (6 layers forward + 6 backward), the 4-bucket gradient all-reduce on the side stream
(async_op), and the plain one-shot all-reduce.  An all-reduce over one rank is the identity, so every result must equal
the same step computed with no process group at all — which the worker computes first, in the same process."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import autograd, parallel, targets  # noqa: E402
from conftest import frames  # noqa: E402

S, B, C = 96, 2, 20
classes = ["c%d" % i for i in range(C)]
x = torch.as_tensor(frames(B, S, seed=3)).cuda()
gt, gid = targets.synthetic_gt(B, S, C, m=4, seed=1)
tg = targets.YOLOV3PrefetchTargetGenerator(C)(S, S, gt, gid, device="cuda:0")


def one_step(sync, overlap):
    kw = dict(norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": 1}) if sync else {}
    net = vy.yolo3_darknet53(classes, pretrained_base=False, **kw)
    net.initialize(init="synthetic", seed=233)
    net.collect_params().reset_ctx("cuda:0")
    tr = vy.Trainer(net.collect_params(), "sgd", {"learning_rate": 1e-3, "wd": 5e-4, "momentum": 0.9})
    if overlap:
        tr.enable_overlap()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with autograd.record():
            losses = net(x, gt, *tg)
            autograd.backward([losses[0] + losses[1] + losses[2] + losses[3]])
    tr.step(B)
    torch.cuda.synchronize()
    out = {"losses": [l.cpu().numpy() for l in losses], "grads": net._grads.cpu().numpy().copy(),
           "params": net._dev_params.view(torch.float32).cpu().numpy().copy(),
           "sync_calls": list(net._sync_hook.calls) if net._sync_hook is not None else None,
           "buckets": list(tr._overlap.last_launched) if tr._overlap is not None else None}
    return out


ref = one_step(sync=False, overlap=False)            # no process group yet: nothing is a collective
assert not parallel.collectives_active()

os.environ["VY_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
parallel.init_process_group("nccl")                   # device_id=cuda:0, finite timeout
import torch.distributed as dist  # noqa: E402
assert dist.get_backend() == "nccl" and parallel.collectives_active()

res = {}
a = one_step(sync=True, overlap=True)                # SyncBN callback + bucketed overlap + Trainer broadcast, over RCCL
hg = parallel.host_group()
res["host_group_backend"] = dist.get_backend(hg) if hg else None
res["any_rank"] = [parallel.any_rank(False, "cuda:0"), parallel.any_rank(True, "cuda:0")]
res["sync_calls"] = a["sync_calls"]
res["buckets"] = a["buckets"]
res["losses_equal"] = all(np.array_equal(u, v) for u, v in zip(a["losses"], ref["losses"]))
res["grads_equal"] = bool(np.array_equal(a["grads"], ref["grads"]))
res["params_equal"] = bool(np.array_equal(a["params"], ref["params"]))
b = one_step(sync=False, overlap=False)              # one flat 246.5 MB all-reduce after backward
res["plain_equal"] = bool(np.array_equal(b["grads"], ref["grads"]) and np.array_equal(b["params"], ref["params"]))
t = torch.arange(8, dtype=torch.float64, device="cuda:0")
dist.all_reduce(t)                                    # float64 through RCCL, as the SyncBN exchange uses it
res["f64_allreduce_identity"] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64)))

# the failure path: a collective that raises inside the library callback must fail THIS call loudly (VyError), mark the
# group as failed, and not hang
net = vy.yolo3_darknet53(classes, pretrained_base=False, norm_layer=vy.SyncBatchNorm, norm_kwargs={"num_devices": 1})
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
real = dist.all_reduce


def broken(tensor, *args, **kw):
    if tensor.dtype == torch.float64:   # the SyncBatchNorm statistics exchange, i.e. inside the library's callback
        raise RuntimeError("injected collective failure")
    return real(tensor, *args, **kw)


err = None
try:
    dist.all_reduce = broken
    with autograd.record():
        net(x, gt, *tg)
except Exception as e:
    err = "%s: %s" % (type(e).__name__, e)
finally:
    dist.all_reduce = real
res["failure_surfaces_as"] = err
res["group_marked_failed"] = parallel.failed() is not None
print("RESULT " + json.dumps(res))
sys.stdout.flush()
os._exit(0)   # the group was aborted on purpose: skip the orderly teardown
