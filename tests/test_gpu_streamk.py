"""-m gpu: robustness of the chain-preserving stream-K conv launches (csrc/conv_igemm.hip, sk_schedule.h) — the parts
that are not arithmetic: the bind-time topology check and the recovery of the hand-off flags after an error.
(Bit-exactness of stream-K launches: tests/test_gpu_parity.py::test_stream_k_hand_off_stays_bit_exact and
tests/test_gpu_fullsize.py::test_stream_k_launches_equal_plain_launches.)"""
import ctypes
import json
import os
import subprocess
import sys

import pytest

from conftest import frames

pytestmark = pytest.mark.gpu


def test_topology_probe_enables_stream_k_on_the_mi355x(voc_classes, synth20):
    """vy_net_bind_workspace runs the placement probe (512 one-wave blocks record their XCC id): on an MI355X in SPX
    mode blocks L and L + 8 share an XCD and stream-K is enabled; anywhere else the net falls back to plain launches
    (and this test says which it was)."""
    import torch
    import videoyolo_amd as vy
    from videoyolo_amd import _lib
    net = vy.yolo3_darknet53(voc_classes, pretrained_base=False)
    net.set_parameters(synth20)
    net.collect_params().reset_ctx("cuda:0")
    net(frames(1, 64))
    en, off, nfl = ctypes.c_int32(-1), ctypes.c_size_t(), ctypes.c_int32()
    _lib.check(net._lib.vy_net_streamk_state(net._h, ctypes.byref(en), ctypes.byref(off), ctypes.byref(nfl)))
    props = torch.cuda.get_device_properties(0)
    assert nfl.value == 2048 and off.value % 256 == 0
    assert en.value == 1, "stream-K disabled: %d CUs, placement not the SPX round-robin" % props.multi_processor_count
    flags = net._ws[off.value:off.value + 4 * nfl.value].view(torch.int32)
    assert bool((flags == 0).all().item())      # the probe used this region as scratch and left it zeroed


def test_stale_flags_are_cleared_after_an_error_return():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, VY_CONV_SK="1", VY_CONV_SK_SLOTS="13")
    p = subprocess.run([sys.executable, os.path.join(here, "sk_recover_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert r["enabled"] == 1 and r["sk_launches"] >= 20, r       # the forwards really consumed hand-off flags
    assert r["error_rc"] != 0 and r["flags_clean_before"]
    assert r["same"], "a forward after an error return consumed stale stream-K flags"
    assert r["flags_clean_after"]
