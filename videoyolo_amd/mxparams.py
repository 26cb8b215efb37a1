"""Reader/writer for mxnet's NDArray-dict file (``.params``) — SURVEY.md §8f row 4.

The reference saves/loads checkpoints with gluon ``save_parameters`` / ``load_parameters``
(train_yolov3.py:293-303,323-327, detect_yolo3.py:890), i.e. ``mx.nd.save`` of
{structural name: NDArray}.  [UPSTREAM-RECALLED, UNVERIFIED]: the container layout below is restated
from memory of mxnet's ``NDArray::Save`` (src/ndarray/ndarray.cc); no mxnet build and no sample file
exist offline, so it has only been round-tripped against itself (tests/test_model_host.py).  Treat a
real GluonCV file that fails to parse as a bug report against this module, not against the file.

    uint64  0x112 (list magic)        uint64  0 (reserved)
    uint64  n_arrays
      per array:  uint32 0xF993FAC9 (V2 magic; 0xF993FAC8 = V1, no storage type)
                  int32  storage type (0 = dense)            [V2 only]
                  uint32 ndim ; int64 dims[ndim]             (V1: uint32 dims)
                  int32  dev_type ; int32 dev_id
                  int32  dtype flag (0 = float32, 1 = float64, 2 = float16, 3 = uint8, 4 = int32, 6 = int64)
                  raw little-endian data
    uint64  n_names ;  per name: uint64 length ; bytes
"""
import struct

import numpy as np

_LIST_MAGIC = 0x112
_V1, _V2, _V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


def load(path):
    """Returns {name: numpy array}.  Names saved by gluon's save_parameters are the structural names
    (``stages.0.0.0.weight`` ...); older ``save_params`` files carry ``arg:``/``aux:`` prefixes, stripped here."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from("<" + fmt, buf, off)
        off += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    if take("Q") != _LIST_MAGIC:
        raise ValueError("%s: not an mxnet NDArray list file" % path)
    take("Q")
    arrays = []
    for _ in range(take("Q")):
        magic = take("I")
        if magic in (_V2, _V3):
            if take("i") != 0:
                raise NotImplementedError("sparse NDArray in a parameter file")
            ndim = take("I")
            shape = [take("q") for _ in range(ndim)]
        elif magic == _V1:
            ndim = take("I")
            shape = [take("I") for _ in range(ndim)]
        else:  # legacy: the word just read is ndim
            ndim = magic
            shape = [take("I") for _ in range(ndim)]
        if ndim == 0:
            arrays.append(np.zeros((), np.float32))
            continue
        take("ii")
        dt = np.dtype(_DTYPES[take("i")])
        n = int(np.prod(shape))
        arrays.append(np.frombuffer(buf, dt, n, off).reshape(shape).copy())
        off += n * dt.itemsize
    names = []
    for _ in range(take("Q")):
        ln = take("Q")
        names.append(buf[off:off + ln].decode())
        off += ln
    if len(names) != len(arrays):
        raise ValueError("%s: %d names for %d arrays" % (path, len(names), len(arrays)))
    return {(n.split(":", 1)[1] if n[:4] in ("arg:", "aux:") else n): a for n, a in zip(names, arrays)}


def save(path, arrays):
    """Writes {name: array} in the V2 layout above."""
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", _LIST_MAGIC, 0, len(arrays)))
        for a in arrays.values():
            a = np.ascontiguousarray(a)
            f.write(struct.pack("<IiI", _V2, 0, a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))
            f.write(a.tobytes())
        f.write(struct.pack("<Q", len(arrays)))
        for n in arrays:
            b = n.encode()
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
