"""examples/encode_raw.cpp — the reference's encoder_example.c written against the C ABI alone (no Python,
no torch): raw PCM file in, .ogg out.  Its packets must be the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from tests import orc
from tests.signals import synth_signal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_writes_the_oracles_packets(oracle, cuda, tmp_path):
    from tests.test_stream_wrapper import parse_pages, packets_of, unpack_headers
    exe = os.path.join(ROOT, "examples", "encode_raw")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    ch, rate, q = 2, 44100, 0.5
    nsamp = 40 * 1024 + 321                      # the last write is a short one
    sig = synth_signal(ch, rate, nsamp, seed=31)
    raw, ogg = tmp_path / "in.f32", tmp_path / "out.ogg"
    np.ascontiguousarray(sig.T).tofile(raw)      # interleaved
    data = os.path.join(ROOT, "vorbis_aotuv_lancer_amd", "data")
    subprocess.run([exe, str(ch), str(rate), str(q), str(raw), str(ogg), data], check=True, timeout=300)
    pages = parse_pages(open(ogg, "rb").read())
    pk = packets_of(pages)
    unpack_headers(*pk[:3])
    assert pages[0]["flags"] & 2 and pages[-1]["flags"] & 4 and pages[-1]["granule"] == nsamp

    st = orc.Stream(orc.Setup(oracle, ch, rate, q))
    oracle.lib.orc_stream_set_capture(st.v, 0)
    want = []
    for at in range(0, nsamp, 1024):
        st.write(sig[:, at:at + 1024])
        want.extend(b["packet"] for b in st.blocks())
    st.finish()
    want.extend(b["packet"] for b in st.blocks())
    st.close()
    assert pk[3:] == want
