"""`.params` checkpoint reader / writer (net.save_parameters / load_parameters,
train_yolov3.py:289-329, detect_yolo3.py:890).

File layout = MXNet's NDArray-list serialisation as recalled in SURVEY.md Appendix A.5
([UPSTREAM-UNVERIFIED]: no sample file exists in the reference tree and MXNet is not installable
here, so the reader has only been round-tripped against this writer):

    u64 0x112 | u64 0 | u64 n_arrays
    n_arrays x { u32 0xF993FAC9 | i32 stype(0) | u32 ndim | i64 dims[ndim] | i32 dev_type(1) | i32 dev_id(0)
                 | i32 dtype_flag (0=f32) | raw little-endian data }
    u64 n_names | n_names x { u64 len | bytes }

Keys are Gluon's structural names (`stages.0.0.0.weight`, `yolo_outputs.0.prediction.bias`, ...).
"""
import struct
from collections import OrderedDict

import numpy as np

_LIST_MAGIC = 0x112
_ND_MAGIC_V2 = 0xF993FAC9
_ND_MAGIC_V3 = 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


def save_params(path, arrays):
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", _LIST_MAGIC, 0, len(arrays)))
        for a in arrays.values():
            a = np.ascontiguousarray(a)
            f.write(struct.pack("<IiI", _ND_MAGIC_V2, 0, a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
        f.write(struct.pack("<Q", len(arrays)))
        for k in arrays.keys():
            b = k.encode("utf-8")
            f.write(struct.pack("<Q", len(b)))
            f.write(b)


def load_params(path):
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def rd(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, buf, off)
        off += struct.calcsize(fmt)
        return v

    magic, _, n = rd("<QQQ")
    if magic != _LIST_MAGIC:
        raise ValueError("%s: not an NDArray list file (magic %#x)" % (path, magic))
    arrs = []
    for _ in range(n):
        (m,) = rd("<I")
        if m not in (_ND_MAGIC_V2, _ND_MAGIC_V3):
            raise ValueError("%s: unsupported NDArray magic %#x" % (path, m))
        (stype,) = rd("<i")
        if stype != 0:
            raise ValueError("%s: sparse storage type %d not supported" % (path, stype))
        (ndim,) = rd("<I")
        shape = rd("<%dq" % ndim) if ndim else ()
        _dev_type, _dev_id, flag = rd("<iii")
        dt = np.dtype(_DTYPES[flag]).newbyteorder("<")
        cnt = int(np.prod(shape)) if ndim else 1
        a = np.frombuffer(buf, dtype=dt, count=cnt, offset=off).reshape(shape)
        off += cnt * dt.itemsize
        arrs.append(a)
    (nn,) = rd("<Q")
    names = []
    for _ in range(nn):
        (ln,) = rd("<Q")
        names.append(buf[off:off + ln].decode("utf-8"))
        off += ln
    if nn != n:
        raise ValueError("%s: %d arrays but %d names" % (path, n, nn))
    out = OrderedDict()
    for k, a in zip(names, arrs):
        for pre in ("arg:", "aux:"):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = np.array(a)
    return out
