"""Immutable codec tables shipped with the package (VPK packs, see tools/vpk.py)."""
import os
import struct
import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_DT = [np.float32, np.float64, np.int32, np.int64, np.uint8, np.int8, np.int16, np.uint16, np.uint32]
_cache = {}


def read_vpk(path):
    d = open(path, "rb").read()
    if d[:4] != b"VPK1":
        raise ValueError(f"{path}: not a VPK1 file")
    (count,) = struct.unpack_from("<I", d, 4)
    pos, res = 8, {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", d, pos); pos += 2
        name = d[pos:pos + nl].decode(); pos += nl
        code, ndim = struct.unpack_from("<BB", d, pos); pos += 2
        shape = struct.unpack_from(f"<{ndim}I", d, pos) if ndim else (); pos += 4 * ndim
        (nbytes,) = struct.unpack_from("<Q", d, pos); pos += 8
        pos += (-pos) % 8
        dt = np.dtype(_DT[code])
        res[name] = np.frombuffer(d, dtype=dt, count=nbytes // dt.itemsize, offset=pos).reshape(shape)
        pos += nbytes
    return res


def pack(name):
    if name not in _cache:
        _cache[name] = read_vpk(os.path.join(_DATA, name))
    return _cache[name]


def window_table(n):
    """Rising half-window (n/2 floats) of an n-sample block — the reference's vwin tables
    (lib/window.c:29-2122) taken verbatim."""
    return pack("common.vpk")[f"window/{n}"]
