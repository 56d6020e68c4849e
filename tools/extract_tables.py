#!/usr/bin/env python3
"""Extract the reference's immutable DATA tables into VPK packs.

Runs only in the authoring container (needs /root/reference as TEXT; nothing from the
reference is compiled, imported or executed).  Outputs are committed under
vorbis_aotuv_lancer_amd/data/ because the GPU box has no /root/reference.

    python tools/extract_tables.py windows     -> data/windows.vpk   (lib/window.c:29-2122)

Why the windows cannot be recomputed: the literals in lib/window.c were printed with 10
decimals, so e.g. vwin2048[0] = 0.0000009241F is NOT round(sin(pi/2 sin^2(...))) — the
table values themselves are the codec's windows and must be taken verbatim.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cinit  # noqa: E402
from vpk import write_vpk  # noqa: E402

REF = os.environ.get("VORBIS_REFERENCE", "/root/reference")
DATA = os.path.join(HERE, "..", "vorbis_aotuv_lancer_amd", "data")


def f32_table(values):
    return np.array([cinit.to_f32(v) for v in values], dtype=np.float32)


def extract_windows():
    d = cinit.parse_file(os.path.join(REF, "lib", "window.c"))
    arrays = {}
    for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        decl = d[f"vwin{n}"]
        t = f32_table(decl.value)
        assert t.shape == (n // 2,), (n, t.shape)
        arrays[f"window/{n}"] = t
    os.makedirs(DATA, exist_ok=True)
    write_vpk(os.path.join(DATA, "windows.vpk"), arrays)
    print("wrote windows.vpk:", {k: v.shape for k, v in arrays.items()})


STEPS = {"windows": extract_windows}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(STEPS)
    for s in todo:
        STEPS[s]()
