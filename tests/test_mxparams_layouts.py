"""-m "not gpu": videoyolo_amd/mxparams.py against files packed BYTE BY BYTE here, not by its own writer.

The layouts are the module's recollection of mxnet's NDArray::Save / NDArray::LegacyLoad (src/ndarray/ndarray.cc) and of
MXNDArraySave's list container [UPSTREAM-RECALLED, UNVERIFIED: tests/golden/RUNBOOK.md captures a real file on a machine with
mxnet; tests/test_mxnet_goldens.py::test_mxparams_reads_a_real_mxnet_file reads it].  What this file adds to the writer /
reader round trip of tests/test_model_host.py: the three array headers the reader claims to know (V2 with a storage type and
int64 dims, V1 with uint32 dims, the pre-magic legacy form whose first word is ndim), every dtype flag, an is_none entry
(ndim 0: nothing after the shape), the `arg:` / `aux:` name prefixes of `save_params` files, and the error paths.
"""
import struct

import numpy as np
import pytest

from videoyolo_amd import mxparams

LIST_MAGIC, V1, V2, V3 = 0x112, 0xF993FAC8, 0xF993FAC9, 0xF993FACA
FLAG = {np.float32: 0, np.float64: 1, np.float16: 2, np.uint8: 3, np.int32: 4, np.int8: 5, np.int64: 6}


def _arr_v2(a, magic=V2):
    b = struct.pack("<Ii", magic, 0) + struct.pack("<I", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape)
    return b + struct.pack("<ii", 1, 0) + struct.pack("<i", FLAG[a.dtype.type]) + a.tobytes()


def _arr_v1(a):
    b = struct.pack("<I", V1) + struct.pack("<I", a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape)
    return b + struct.pack("<ii", 2, 3) + struct.pack("<i", FLAG[a.dtype.type]) + a.tobytes()     # saved from gpu(3): ignored


def _arr_legacy(a):
    b = struct.pack("<I", a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape)
    return b + struct.pack("<ii", 1, 0) + struct.pack("<i", FLAG[a.dtype.type]) + a.tobytes()


def _file(blobs, names):
    out = struct.pack("<QQQ", LIST_MAGIC, 0, len(blobs)) + b"".join(blobs) + struct.pack("<Q", len(names))
    for n in names:
        out += struct.pack("<Q", len(n.encode())) + n.encode()
    return out


def _rng_array(rng, dt, shape):
    if np.issubdtype(dt, np.floating):
        return rng.standard_normal(shape).astype(dt)
    return rng.integers(0, 100, shape).astype(dt)


@pytest.mark.parametrize("pack", [_arr_v2, lambda a: _arr_v2(a, V3), _arr_v1, _arr_legacy], ids=["v2", "v3", "v1", "legacy"])
def test_every_header_form_and_dtype(tmp_path, pack):
    rng = np.random.default_rng(5)
    arrays = {"stages.0.0.0.weight": _rng_array(rng, np.float32, (4, 3, 3, 3)), "f64": _rng_array(rng, np.float64, (5,)),
              "f16": _rng_array(rng, np.float16, (2, 7)), "u8": _rng_array(rng, np.uint8, (9,)), "i32": _rng_array(rng, np.int32, (3, 1)),
              "i8": _rng_array(rng, np.int8, (6,)), "i64": _rng_array(rng, np.int64, (2, 2))}
    f = tmp_path / "a.params"
    f.write_bytes(_file([pack(a) for a in arrays.values()], list(arrays)))
    got = mxparams.load(str(f))
    assert list(got) == list(arrays)
    for k, a in arrays.items():
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k


def test_save_params_prefixes_and_none_entries(tmp_path):
    w = np.arange(6, dtype=np.float32).reshape(2, 3)
    none = struct.pack("<Ii", V2, 0) + struct.pack("<I", 0)          # is_none: magic, storage type, ndim 0 — nothing else
    f = tmp_path / "b.params"
    f.write_bytes(_file([_arr_v2(w), none, _arr_v2(w + 1)], ["arg:conv0_weight", "aux:placeholder", "aux:bn0_moving_mean"]))
    got = mxparams.load(str(f))
    assert list(got) == ["conv0_weight", "placeholder", "bn0_moving_mean"]
    assert np.array_equal(got["conv0_weight"], w) and np.array_equal(got["bn0_moving_mean"], w + 1) and got["placeholder"].shape == ()


def test_writer_output_is_the_v2_layout_byte_for_byte(tmp_path):
    arrays = {"a.weight": np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2), "a.bias": np.ones(2, np.float32)}
    f = tmp_path / "c.params"
    mxparams.save(str(f), arrays)
    assert f.read_bytes() == _file([_arr_v2(a) for a in arrays.values()], list(arrays))


def test_errors(tmp_path):
    f = tmp_path / "d.params"
    f.write_bytes(struct.pack("<QQQ", 0x113, 0, 0))
    with pytest.raises(ValueError, match="not an mxnet NDArray list"):
        mxparams.load(str(f))
    sparse = struct.pack("<Ii", V2, 1)                               # row_sparse storage: refused, not mis-read
    f.write_bytes(_file([sparse], ["w"]))
    with pytest.raises(NotImplementedError):
        mxparams.load(str(f))
    f.write_bytes(_file([_arr_v2(np.zeros(3, np.float32))], ["a", "b"]))
    with pytest.raises(ValueError, match="2 names for 1 arrays"):
        mxparams.load(str(f))
