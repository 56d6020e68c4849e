"""THE parity pin of the oracle: its packets must be byte-identical to the reference's scalar-C
build on the survey probe signal.  The goldens under tests/golden/ are
the packet dumps that the survey stage recorded from that build (md5s quoted in SURVEY.md):
    2ch 44.1 kHz q0.5 20 s  -> 989 packets, md5 0b15c75f94cb66bb39a5adaefcf26fbd
    6ch 48 kHz  q0.8 10 s  -> 567 packets, md5 4e93ce6323cdea862cb86cc145c6072c
    2ch 44.1 kHz q0.5 120 s -> 5866 packets (stored as per-packet length + crc32)
They exercise long/short/transition blocks, impulse and padding short blocks, coupled res-2
stereo and the uncoupled 5.1 map with its res-1 LFE submap."""
import hashlib
import os
import struct
import zlib

import numpy as np
import pytest

from tests import orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def split(d):
    out, p = [], 0
    while p < len(d):
        (n,) = struct.unpack_from("<i", d, p)
        p += 4
        out.append(d[p:p + n])
        p += n
    return out


@pytest.mark.parametrize("ch,rate,q,secs,golden,md5", [
    (2, 44100, 0.5, 20, "ref_scalar_2ch_44100_q05_20s.pkt", "0b15c75f94cb66bb39a5adaefcf26fbd"),
    (6, 48000, 0.8, 10, "ref_scalar_6ch_48000_q08_10s.pkt", "4e93ce6323cdea862cb86cc145c6072c"),
])
def test_oracle_packets_match_reference_dump(oracle, tmp_path, ch, rate, q, secs, golden, md5):
    ref = open(os.path.join(G, golden), "rb").read()
    assert hashlib.md5(ref).hexdigest() == md5          # the fixture is the dump SURVEY.md quotes
    out = str(tmp_path / "o.pkt")
    n, _ = orc.Setup(oracle, ch, rate, q).encode_probe(secs, out)
    got = open(out, "rb").read()
    pg, pr = split(got), split(ref)
    assert n == len(pr) == len(pg)
    bad = [i for i, (a, b) in enumerate(zip(pg, pr)) if a != b]
    assert not bad, f"first differing packet {bad[0]} of {len(pr)}"
    assert hashlib.md5(got).hexdigest() == md5


def test_oracle_packets_120s_len_crc(oracle, tmp_path):
    ref = np.load(os.path.join(G, "ref_scalar_2ch_44100_q05_120s.lencrc.npy"))
    out = str(tmp_path / "o.pkt")
    n, _ = orc.Setup(oracle, 2, 44100, 0.5).encode_probe(120, out)
    pk = split(open(out, "rb").read())
    assert n == len(pk) == ref.shape[0] == 5866
    got = np.array([(len(p), zlib.crc32(p)) for p in pk], dtype=np.uint32)
    assert np.array_equal(got, ref)
