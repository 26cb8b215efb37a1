"""-m "not gpu": libvyolo.so loads and exports every symbol include/vyolo.h declares; the host-only
entry points (graph construction, parameter table, planning, argument checking) behave.  No
compute entry point is called here (there is no GPU in this container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from videoyolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "vyolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vy_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libvyolo.so does not export %s" % n
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))
    assert lib.vy_version().decode().endswith("gfx950")


def test_product_does_not_touch_the_oracle():
    """The product path must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "videoyolo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.replace("the oracle", "").replace("CPU checker", ""), (dp, f)


@pytest.mark.parametrize("ncls,total", [(20, 61626049), (30, 61679899)])
def test_param_table_matches_reference_structure(ncls, total):
    from oracle import yolo3_oracle as O
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.vy_net_create(ncls, ctypes.byref(h)))
    n = lib.vy_net_num_params(h)
    table, train, end = {}, 0, 0
    for i in range(n):
        info = _lib.ParamInfo()
        _lib.check(lib.vy_net_param_info(h, i, ctypes.byref(info)))
        shape = tuple(info.shape[j] for j in range(info.ndim))
        table[info.name.decode()] = shape
        assert info.size == int(np.prod(shape))
        assert info.offset >= end and info.offset % 64 == 0   # disjoint, 256-B aligned
        end = info.offset + info.size
        if info.trainable:
            train += info.size
        assert info.backbone == int(info.name.decode().startswith("stages."))
    assert table == dict(O.param_shapes(ncls))
    assert train == total
    assert lib.vy_net_param_bytes(h) >= end * 4
    lib.vy_net_destroy(h)


def test_planning_and_argument_errors():
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.vy_net_create(20, ctypes.byref(h)))
    fixed = (32 << 20) + 8192      # stream-K scratch (kernels.h VY_SK_PARTIAL_BYTES + flags): in every plan
    # one plane per cell (parity taps / training): the conv-output activations, 38.6 M floats / frame at 416
    #  + borders + scratch
    _lib.check(lib.vy_net_set_keep_activations(h, 1))
    k1 = lib.vy_net_workspace_bytes(h, 1, 416, 416) - fixed
    assert 38.6e6 * 4 < k1 < 1.25 * 38.6e6 * 4 + 4e6
    # the default inference plan recycles planes by liveness: little more than half of that
    _lib.check(lib.vy_net_set_keep_activations(h, 0))
    b1 = lib.vy_net_workspace_bytes(h, 1, 416, 416)
    b2 = lib.vy_net_workspace_bytes(h, 2, 416, 416)
    b608 = lib.vy_net_workspace_bytes(h, 1, 608, 608)
    assert 0.45 * k1 < b1 - fixed < 0.6 * k1
    assert b1 < b2 < 2 * b1 + 1e6 and b608 - fixed > 2 * (b1 - fixed)
    # inference plans any size in [32, 4096] (ceil-sized feature maps, cropped upsample); training multiples of 32
    b400 = lib.vy_net_workspace_bytes(h, 1, 400, 416)
    assert lib.vy_net_workspace_bytes(h, 1, 384, 416) < b400 < b1
    assert lib.vy_net_workspace_bytes(h, 1, 16, 416) == 0 and lib.vy_net_workspace_bytes(h, 1, 416, 4100) == 0
    assert "[32, 4096]" in lib.vy_last_error().decode()
    assert lib.vy_net_train_workspace_bytes(h, 1, 400, 416) == 0
    assert "multiples of 32" in lib.vy_last_error().decode()
    assert lib.vy_net_train_workspace_bytes(h, 1, 416, 416) > b1
    assert lib.vy_net_workspace_bytes(h, 0, 416, 416) == 0
    rc = lib.vy_net_bind_workspace(h, None, 0, 1, 416, 416, None)
    assert rc == -1
    # forward before binding anything: a state error, not a crash
    dummy = ctypes.c_void_p(16)
    rc = lib.vy_net_forward_infer(h, dummy, dummy, dummy, dummy, None, None)
    assert rc == -2 and "not bound" in lib.vy_last_error().decode()
    info = _lib.ParamInfo()
    assert lib.vy_net_param_info(h, 10 ** 6, ctypes.byref(info)) == -1
    lib.vy_net_destroy(h)
    bad = ctypes.c_void_p()
    assert lib.vy_net_create(0, ctypes.byref(bad)) == -1
