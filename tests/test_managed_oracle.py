"""Managed-bitrate mode of the oracle (lib/mapping0.c:1097-1181 + :1204 loop over the 15
packetblobs, lib/bitrate.c:28-252, mode packs of vorbis_encode_init).

PARITY UNPINNED: the reference's dumps under tests/golden/ are VBR only, so nothing recorded from the
reference pins this mode.  What is checked here is what the domain offers without one: the rate the
reservoir logic steers to, the hard floor / ceiling when min / max are set, blob sizes growing with the
blob index, and that the VBR path (pinned by the dumps) is untouched by the managed-mode code."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import orc
from tests.signals import synth_signal


def run(oracle, bitrate, sig):
    st = orc.Stream(orc.Setup(oracle, 2, 44100, bitrate=bitrate))
    oracle.lib.orc_stream_set_capture(st.v, 0)
    oracle.lib.orc_stream_bitrate_state.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    res, af = (C.c_int64 * 2)(), C.c_double()
    out = []
    for at in range(0, sig.shape[1], 1024):
        st.write(sig[:, at:at + 1024])
        for b in st.blocks():
            oracle.lib.orc_stream_bitrate_state(st.v, res, C.byref(af))
            out.append((b, int(res[0]), int(res[1]), af.value))
    st.finish()
    for b in st.blocks():
        out.append((b, 0, 0, 0.))
    st.close()
    return out


def test_average_bitrate_is_steered_to_nominal(oracle):
    secs = 12
    sig = synth_signal(2, 44100, 44100 * secs, seed=5)
    out = run(oracle, 128000, sig)
    total = sum(len(b["packet"]) for b, *_ in out)
    assert abs(total * 8 / secs - 128000) < 0.03 * 128000
    for b, *_ in out:
        assert 0 <= b["choice"] < 15
        sizes = b["blob_bytes"]
        assert all(s > 0 for s in sizes)
        assert sizes[14] >= sizes[7] >= sizes[0]                     # lower noise curve = more bits
        assert b["packet"][:sizes[b["choice"]]] == b["blobs"][b["choice"]]
        assert len(b["packet"]) == sizes[b["choice"]]                # no min/max: nothing truncated or padded
    # the blob in the middle is not the VBR packet of the same setting: managed mode runs set_m3p differently
    # (lib/psy.c:4165-4173) and uses the managed books (lib/vorbisenc.c:513-530); only the framing agrees
    assert {b["W"] for b, *_ in out} == {0, 1}


def test_min_and_max_hold_through_silence(oracle):
    secs = 12
    sig = synth_signal(2, 44100, 44100 * secs, seed=5)
    sig[:, 44100 * 5:44100 * 8] = 0                                  # digital silence: the floor has to prop packets up
    out = run(oracle, (144000, 128000, 112000), sig)
    reservoir_bits = 2 * 128000
    padded = 0
    for b, avg_res, minmax_res, _ in out[:-3]:
        assert 0 <= minmax_res <= reservoir_bits                     # lib/bitrate.c:146-163: never under / over
        padded += len(b["packet"]) > b["blob_bytes"][b["choice"]]
        if len(b["packet"]) > b["blob_bytes"][b["choice"]]:
            assert set(b["packet"][b["blob_bytes"][b["choice"]]:]) == {0}
    assert padded > 0
    total = sum(len(b["packet"]) for b, *_ in out)
    assert 112000 * 0.98 <= total * 8 / secs <= 144000 * 1.02
