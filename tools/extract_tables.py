#!/usr/bin/env python3
"""Extract the reference's immutable DATA tables into VPK packs.

Runs only in the authoring container (needs /root/reference as TEXT; nothing from the
reference is compiled, imported or executed).  Outputs are committed under
vorbis_aotuv_lancer_amd/data/ because the GPU box has no /root/reference.

    python tools/extract_tables.py common   -> data/common.vpk  (windows lib/window.c:29-2122, psy/floor tables)

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


def extract_common():
    """Static tables of the psychoacoustic model, envelope detector and floor renderer:
    lib/masking.h:24-54 (ATH), :63-797 (tonemasks); lib/psy.c:41-111 (stereo thresholds, M3
    band tables, companders, aoTuV presets), :4517-4582 (FLOOR1_fromdB_LOOKUP);
    lib/envelope.c:39-40 (band edges); lib/window.c (windows)."""
    arrays = {}
    w = cinit.parse_file(os.path.join(REF, "lib", "window.c"))
    for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
        arrays[f"window/{n}"] = f32_table(w[f"vwin{n}"].value)
    m = cinit.parse_file(os.path.join(REF, "lib", "masking.h"))
    arrays["ATH"] = f32_table(m["ATH"].value)
    assert arrays["ATH"].shape == (88,)
    tm = np.array([[f32_table(c) for c in band] for band in m["tonemasks"].value], dtype=np.float32)
    assert tm.shape == (17, 6, 56), tm.shape
    arrays["tonemasks"] = tm
    p = cinit.parse_file(os.path.join(REF, "lib", "psy.c"))
    arrays["stereo_threshholds"] = np.array([float(x) for x in p["stereo_threshholds"].value], np.float64)
    arrays["stereo_threshholds_X"] = np.array([float(x) for x in p["stereo_threshholds_X"].value], np.float64)
    for k in ("m3n32", "m3n44", "m3n48", "m3n32x2", "m3n44x2", "m3n48x2", "freq_bfn128", "freq_bfn256",
              "stn_compand"):
        arrays[k] = np.array(p[k].value, np.int32)
    arrays["ntfix_offset"] = f32_table(p["ntfix_offset"].value)
    pre = p["set_aotuv_psy"].value
    arrays["aotuv_preset/ints"] = np.array([[r[0], r[2], r[3]] for r in pre], np.int32)  # tonecomp_endp, min_nn_lp, tonefix_end
    arrays["aotuv_preset/tonecomp_thres"] = f32_table([r[1] for r in pre])
    arrays["FLOOR1_fromdB_LOOKUP"] = f32_table(p["FLOOR1_fromdB_LOOKUP"].value)
    assert arrays["FLOOR1_fromdB_LOOKUP"].shape == (256,)
    e = cinit.parse_file(os.path.join(REF, "lib", "envelope.c"))
    arrays["envelope/band_begin"] = np.array(e["band_begin"].value, np.int32)
    arrays["envelope/band_end"] = np.array(e["band_end"].value, np.int32)
    os.makedirs(DATA, exist_ok=True)
    write_vpk(os.path.join(DATA, "common.vpk"), arrays)
    print("wrote common.vpk:", {k: v.shape for k, v in arrays.items()})


STEPS = {"common": extract_common}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(STEPS)
    for s in todo:
        STEPS[s]()
