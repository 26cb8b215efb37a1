"""Subprocess body of tests/test_gpu_streamk.py::test_stale_flags_are_cleared_after_an_error_return.  Started with
VY_CONV_SK_SLOTS=13, which makes EVERY conv launch of more than 13 tiles a stream-K launch on 13 blocks (the switch is
read once per process).  Sequence: a forward (reference) -> every hand-off flag of the workspace is raised by hand, as
an aborted launch sequence could leave them -> an entry point of the handle returns an error -> the next forward must
zero the flags first and reproduce the reference bit for bit."""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import videoyolo_amd as vy  # noqa: E402
from videoyolo_amd import _lib  # noqa: E402
from conftest import frames  # noqa: E402

net = vy.yolo3_darknet53(["c%d" % i for i in range(20)], pretrained_base=False)
net.initialize(init="synthetic", seed=233)
net.collect_params().reset_ctx("cuda:0")
x = torch.as_tensor(frames(2, 96, seed=3)).cuda()


def digest():
    h = hashlib.sha256()
    for t in net(x, return_index=True):
        h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    for i in range(3):
        h.update(np.ascontiguousarray(net.read_head(i).cpu().numpy()).tobytes())
    return h.hexdigest()


ref = digest()
labels = [n for n, _, _, _ in net.profile(x)]
en, off, nfl = ctypes.c_int32(), ctypes.c_size_t(), ctypes.c_int32()
_lib.check(net._lib.vy_net_streamk_state(net._h, ctypes.byref(en), ctypes.byref(off), ctypes.byref(nfl)))
flags = net._ws[off.value:off.value + 4 * nfl.value].view(torch.int32)
clean_before = bool((flags == 0).all().item())
flags.fill_(1)                      # what an aborted sequence could leave behind (and worse)
torch.cuda.synchronize()
rc = net._lib.vy_net_forward_infer(net._h, None, None, None, None, None, None)   # an error return on this handle
after = digest()
torch.cuda.synchronize()
print("RESULT " + json.dumps({
    "enabled": en.value, "sk_launches": sum(n.endswith("sk") for n in labels), "error_rc": rc,
    "flags_clean_before": clean_before, "flags_clean_after": bool((flags == 0).all().item()), "same": ref == after}))
