"""The reference's OWN test input on the device: test/util.c:29-45 gen_windowed_sine (2048 samples: a Hann-windowed
sine of period 32 and peak 0.95, then 1024 zeros), written in one piece to every channel and followed at once by
the end of the stream (test/write_read.c:85-99) — start-of-stream extrapolation, block switching and the
end-of-stream extrapolation of an undrained buffer within seven blocks.  For every (channels, rate) of the
reference's matrix (test/test.c:38-57: 1..8 channels x {44100, 48000, 32000, 22050, 16000, 96000} Hz) that has a
shipped mode pack, at the pack's quality: block sequence and packets against the oracle, through the batched front
end and through the reference's own entry points (include/vorbis_compat.h)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest
import torch

from tests import compat, orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gen_windowed_sine(n=2048, maximum=0.95):
    """test/util.c:29-45, evaluated in double and stored as float like the C code"""
    data = np.zeros(n, np.float32)
    half = n // 2
    k = np.arange(half, dtype=np.float64)
    x = np.sin(2.0 * k * np.pi * 1.0 / 32.0 + 0.4).astype(np.float32)          # data[k] = sin(...)  (float store)
    w = maximum * (0.5 - 0.5 * np.cos(2.0 * np.pi * k / (half - 1)))             # double
    data[:half] = (x.astype(np.float64) * w).astype(np.float32)                 # data[k] *= ...      (float store)
    return data


def classes():
    """(channels, rate, quality) of the shipped VBR packs that lie in the reference's test matrix"""
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "vorbis_aotuv_lancer_amd", "data", "mode_*ch_*_q*.vpk"))):
        m = re.match(r"mode_(\d+)ch_(\d+)_q(-?[\d.]+)\.vpk", os.path.basename(f))
        ch, rate, q = int(m.group(1)), int(m.group(2)), float(m.group(3))
        if 1 <= ch <= 8 and rate in (44100, 48000, 32000, 22050, 16000, 96000):
            out.append((ch, rate, q))
    return out


def oracle_blocks(oracle, ch, rate, q, pcm):
    st = orc.Stream(orc.Setup(oracle, ch, rate, q))
    oracle.lib.orc_stream_set_capture(st.v, 0)
    st.write(pcm)
    st.finish()
    want = [((b["lW"], b["W"], b["nW"], b["eos"], b["granulepos"], b["sequence"]), b["packet"]) for b in st.blocks()]
    st.close()
    return want


def test_matrix_is_not_empty():
    cl = classes()
    assert len(cl) >= 20 and {c[0] for c in cl} == {1, 2, 3, 4, 5, 6, 7, 8} and 96000 in {c[1] for c in cl}


@pytest.mark.parametrize("ch,rate,q", classes())
def test_windowed_sine_then_eos(oracle, cuda, ch, rate, q):
    import vorbis_aotuv_lancer_amd as v
    sine = gen_windowed_sine()
    pcm = np.repeat(sine[None, :], ch, axis=0)                # the same data on every channel (write_read.c:88-92)
    want = oracle_blocks(oracle, ch, rate, q, pcm)
    assert want and want[-1][0][3] == 1 and want[-1][0][4] == 2048      # e_o_s, last granule = samples written

    # batched front end: S = 3 streams with the same input, end declared before the first round
    S = 3
    enc = v.Encoder(v.Setup(ch, rate, q), S)
    fe = v.FrontEnd(enc)
    fe.write(torch.from_numpy(np.repeat(pcm[None], S, axis=0)).to(cuda))
    fe.finish()
    got = [[] for _ in range(S)]
    while True:
        info, packets, nbytes = fe.encode_round()
        if len(info) == 0:
            break
        packets, nbytes = packets.cpu().numpy(), nbytes.cpu().numpy()
        for k, pi in enumerate(info):
            got[int(pi["stream"])].append(((int(pi["lW"]), int(pi["W"]), int(pi["nW"]), int(pi["eos"]), int(pi["granulepos"]),
                                            int(pi["packetno"])), bytes(packets[k, :nbytes[k]])))
    for s in range(S):
        assert got[s] == want, f"stream {s}"
    fe.close()

    # the reference's own call sequence (test/write_read.c:85-115)
    dll = compat.bind(C.CDLL(v.COMPAT_LIB_PATH))
    st = compat.Stream(dll, ch, rate, q)
    assert st.write(pcm) == 0
    assert st.finish() == 0
    assert st.drain() == want
    st.close()
