"""VPK1 — a flat container of named numeric arrays.

The batched encoder's immutable tables (windows, aoTuV tuning rows, codebooks, the
per-class "mode pack" SURVEY.md §7 describes) travel in this format so that the C
library (csrc/vpk.h), the oracle (oracle/vpk.h is the same reader, compiled separately)
and Python all read one file.

Layout (little endian):
    char[4]  "VPK1"
    u32      count
    repeat count times:
        u16  name_len, char[name_len] name (no NUL)
        u8   dtype   (0=f32 1=f64 2=i32 3=i64 4=u8 5=i8 6=i16 7=u16 8=u32)
        u8   ndim    (<= 4)
        u32  shape[ndim]
        u64  nbytes
        pad to 8-byte file offset
        u8   data[nbytes]
"""
import struct
import numpy as np

_DT = [np.float32, np.float64, np.int32, np.int64, np.uint8, np.int8, np.int16, np.uint16, np.uint32]
_CODE = {np.dtype(t): i for i, t in enumerate(_DT)}


def write_vpk(path, arrays):
    """arrays: dict name -> array-like (ordered)."""
    out = bytearray()
    out += b"VPK1" + struct.pack("<I", len(arrays))
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        if a.dtype not in _CODE:
            raise TypeError(f"{name}: unsupported dtype {a.dtype}")
        if a.ndim > 4:
            raise ValueError(f"{name}: ndim {a.ndim} > 4")
        nb = name.encode()
        out += struct.pack("<H", len(nb)) + nb
        out += struct.pack("<BB", _CODE[a.dtype], a.ndim)
        out += struct.pack(f"<{a.ndim}I", *a.shape) if a.ndim else b""
        out += struct.pack("<Q", a.nbytes)
        out += b"\0" * ((-len(out)) % 8)
        out += a.tobytes()
    with open(path, "wb") as f:
        f.write(out)


def read_vpk(path):
    d = open(path, "rb").read()
    if d[:4] != b"VPK1":
        raise ValueError("not a VPK1 file")
    (count,) = struct.unpack_from("<I", d, 4)
    pos = 8
    res = {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", d, pos); pos += 2
        name = d[pos:pos + nl].decode(); pos += nl
        code, ndim = struct.unpack_from("<BB", d, pos); pos += 2
        shape = struct.unpack_from(f"<{ndim}I", d, pos) if ndim else (); pos += 4 * ndim
        (nbytes,) = struct.unpack_from("<Q", d, pos); pos += 8
        pos += (-pos) % 8
        res[name] = np.frombuffer(d, dtype=_DT[code], count=nbytes // np.dtype(_DT[code]).itemsize,
                                  offset=pos).reshape(shape)
        pos += nbytes
    return res
