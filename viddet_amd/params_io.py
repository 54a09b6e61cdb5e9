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
    """Strict reader of the layout of SURVEY.md A.5 ([UPSTREAM-UNVERIFIED]: no real MXNet file exists offline to validate it
    against).  Anything the layout does not allow - wrong magics, a non-zero reserved word, sparse storage, an unknown dtype
    flag, implausible ranks / dimensions, a truncated or over-long file, a name count that differs from the array count -
    raises ValueError naming the offset, never returns a partly decoded dict."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0

    def rd(fmt, what):
        nonlocal off
        need = struct.calcsize(fmt)
        if off + need > len(buf):
            raise ValueError("%s: truncated at byte %d while reading %s" % (path, off, what))
        v = struct.unpack_from(fmt, buf, off)
        off += need
        return v

    magic, reserved, n = rd("<QQQ", "the list header")
    if magic != _LIST_MAGIC:
        raise ValueError("%s: not an NDArray list file (magic %#x)" % (path, magic))
    if reserved != 0:
        raise ValueError("%s: reserved header word is %#x, expected 0" % (path, reserved))
    if n > (1 << 20):
        raise ValueError("%s: implausible array count %d" % (path, n))
    arrs = []
    for i in range(n):
        (m,) = rd("<I", "array %d's magic" % i)
        if m not in (_ND_MAGIC_V2, _ND_MAGIC_V3):
            raise ValueError("%s: array %d at byte %d: unsupported NDArray magic %#x" % (path, i, off - 4, m))
        (stype,) = rd("<i", "array %d's storage type" % i)
        if stype != 0:
            raise ValueError("%s: array %d: sparse storage type %d not supported" % (path, i, stype))
        (ndim,) = rd("<I", "array %d's rank" % i)
        if ndim > 8:
            raise ValueError("%s: array %d: implausible rank %d" % (path, i, ndim))
        shape = rd("<%dq" % ndim, "array %d's shape" % i) if ndim else ()
        if any(d < 0 for d in shape):
            raise ValueError("%s: array %d: negative dimension in %r" % (path, i, shape))
        _dev_type, _dev_id, flag = rd("<iii", "array %d's context / dtype" % i)
        if flag not in _DTYPES:
            raise ValueError("%s: array %d: unknown dtype flag %d" % (path, i, flag))
        dt = np.dtype(_DTYPES[flag]).newbyteorder("<")
        cnt = int(np.prod(shape)) if ndim else 1
        if off + cnt * dt.itemsize > len(buf):
            raise ValueError("%s: array %d %r: data runs past the end of the file" % (path, i, shape))
        a = np.frombuffer(buf, dtype=dt, count=cnt, offset=off).reshape(shape)
        off += cnt * dt.itemsize
        arrs.append(a)
    (nn,) = rd("<Q", "the name count")
    if nn != n:
        raise ValueError("%s: %d arrays but %d names" % (path, n, nn))
    names = []
    for i in range(nn):
        (ln,) = rd("<Q", "name %d's length" % i)
        if ln > 4096 or off + ln > len(buf):
            raise ValueError("%s: name %d: implausible length %d at byte %d" % (path, i, ln, off))
        names.append(buf[off:off + ln].decode("utf-8"))
        off += ln
    if off != len(buf):
        raise ValueError("%s: %d trailing bytes after the last name" % (path, len(buf) - off))
    if len(set(names)) != len(names):
        raise ValueError("%s: duplicate parameter names" % path)
    out = OrderedDict()
    for k, a in zip(names, arrs):
        for pre in ("arg:", "aux:"):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = np.array(a)
    return out
