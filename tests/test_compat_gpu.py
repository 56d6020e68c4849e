"""The reference's own entry points on the device (include/vorbis_compat.h): a program with the call sequence of
the reference's examples/encoder_example.c:179-236 reproduces the reference build's packet dump, and many
streams driven through vorbis_analysis_buffer/_wrote/_blockout + vorbis_analysis + vorbis_bitrate_* give the
oracle's packets while the device runs one batched round per block generation."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests import compat, orc
from tests.signals import synth_signal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def probe_pcm(oracle, ch, rate, secs):
    n = ((rate * secs + 1023) // 1024) * 1024
    out = np.empty((ch, n), np.float32)
    oracle.lib.orc_probe_signal.argtypes = [C.c_int, C.c_long, C.c_long, C.c_void_p]
    oracle.lib.orc_probe_signal(ch, rate, n, out.ctypes.data)
    return out


def test_example_program_reproduces_reference_dump(oracle, cuda, tmp_path):
    """examples/encoder_compat.c (plain C, gcc, the reference's call sequence) on the survey's probe signal:
    989 packets, md5 0b15c75f... = the reference's scalar build."""
    exe = os.path.join(ROOT, "examples", "encoder_compat")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "encoder_compat"])
    pcm = probe_pcm(oracle, 2, 44100, 20)
    raw, dump, ogg = tmp_path / "probe.f32", tmp_path / "probe.pkt", tmp_path / "probe.ogg"
    np.ascontiguousarray(pcm.T).tofile(raw)
    with open(raw, "rb") as fin, open(ogg, "wb") as fout:
        subprocess.run([exe, "2", "44100", "0.5", "--f32", "--no-eos", "--dump", str(dump)], stdin=fin, stdout=fout,
                       check=True, timeout=600)
    got = open(dump, "rb").read()
    ref = open(os.path.join(G, "ref_scalar_2ch_44100_q05_20s.pkt"), "rb").read()
    assert hashlib.md5(ref).hexdigest() == "0b15c75f94cb66bb39a5adaefcf26fbd"
    assert hashlib.md5(got).hexdigest() == "0b15c75f94cb66bb39a5adaefcf26fbd"
    assert got == ref
    # and the .ogg it wrote is a stream: headers first, its audio packets are the dump's
    from tests.test_stream_wrapper import parse_pages, packets_of, unpack_headers
    from tests.test_frontend_gpu import split_dump
    pk = packets_of(parse_pages(open(ogg, "rb").read()))
    unpack_headers(*pk[:3])
    assert pk[3:] == split_dump(ref)


def oracle_streams(oracle, ch, rate, q, sigs, bitrate=None, chunk=1024):
    setup = orc.Setup(oracle, ch, rate, q, bitrate=bitrate)
    want = []
    for sig in sigs:
        st = orc.Stream(setup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        seq = []
        for at in range(0, sig.shape[1], chunk):
            st.write(sig[:, at:at + chunk])
            seq.extend(st.blocks())
        st.finish()
        seq.extend(st.blocks())
        st.close()
        want.append([((b["lW"], b["W"], b["nW"], b["eos"], b["granulepos"], b["sequence"]), b["packet"]) for b in seq])
    return want


@pytest.fixture()
def dll():
    import vorbis_aotuv_lancer_amd as v
    d = compat.bind(C.CDLL(v.COMPAT_LIB_PATH))
    one = C.c_int(1)
    d.vorbis_mi355x_ctl(2, C.byref(one))     # VORBIS_MI355X_CARVE_AHEAD back to its default
    zero = C.c_int(0)
    d.vorbis_mi355x_ctl(5, C.byref(zero))    # VORBIS_MI355X_DEFER_BLOCKS off
    return d


def rounds(dll):
    n = C.c_longlong()
    dll.vorbis_mi355x_ctl(4, C.byref(n))
    return n.value


@pytest.mark.parametrize("defer,device_rounds", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_many_streams_through_the_reference_api(oracle, cuda, dll, defer, device_rounds, monkeypatch):
    """(device_rounds = 1: VORBIS_MI355X_DEVICE_ROUNDS, the pool's rounds built on the device and replayed as HIP graphs.)
    24 stereo q5 streams, each with its own vorbis_info / vorbis_dsp_state / vorbis_block, fed and drained in
    the application's order; streams of different length (end of stream at different times).  defer = 1:
    VORBIS_MI355X_DEFER_BLOCKS, the throughput mode (a block may be handed out later than the reference would; the
    packets and their order per stream are the same)."""
    ch, rate, q, NS = 2, 44100, 0.5, 24
    monkeypatch.setenv("VORBIS_MI355X_DEVICE_ROUNDS", str(device_rounds))      # (read when a pool is made)
    pool = C.c_int(32)
    dll.vorbis_mi355x_ctl(1, C.byref(pool))
    dflag = C.c_int(defer)
    dll.vorbis_mi355x_ctl(5, C.byref(dflag))
    lens = [(18 + 3 * (s % 5)) * 1024 + (0 if s % 2 else 333) for s in range(NS)]
    sigs = [synth_signal(ch, rate, lens[s], seed=900 + s, level=1.0 if s % 3 else 0.05) for s in range(NS)]
    want = oracle_streams(oracle, ch, rate, q, sigs)
    r0 = rounds(dll)
    ss = [compat.Stream(dll, ch, rate, q) for _ in range(NS)]
    got = [[] for _ in range(NS)]
    done = [False] * NS
    at = 0
    while not all(done):
        for s in range(NS):                       # every stream gets its next chunk (or its end) ...
            if done[s]:
                continue
            if at >= lens[s]:
                assert ss[s].finish() == 0
            else:
                assert ss[s].write(sigs[s][:, at:at + 1024]) == 0
        for s in range(NS):                       # ... then every stream is drained
            if done[s]:
                continue
            got[s].extend(ss[s].drain())
            if at >= lens[s]:
                done[s] = True
        at += 1024
    used = rounds(dll) - r0
    for s in range(NS):
        assert len(got[s]) == len(want[s]), (s, len(got[s]), len(want[s]))
        for k, (g, w) in enumerate(zip(got[s], want[s])):
            assert g[0] == w[0], (s, k, g[0], w[0])
            assert g[1] == w[1], f"stream {s} packet {k} differs from the oracle"
        assert got[s][-1][0][3] == 1                                   # e_o_s on the last packet
        assert ss[s].blockout() == 0                                   # stream over: nothing more (lib/block.c:566)
    nblocks = sum(len(g) for g in got)
    # one batched round per block generation, not one per block
    assert used < nblocks / 4, (used, nblocks)
    # where the host's time went (VORBIS_MI355X_TIMES): rounds were run, and the writes of this test (1024 samples, one
    # vorbis_analysis_wrote per vorbis_analysis_buffer) went through the pinned arena, not through staging copies
    times = (C.c_double * 8)()
    assert dll.vorbis_mi355x_ctl(6, times) == 0
    assert times[3] > 0 and all(t >= 0 for t in times)
    for st in ss:
        st.close()


def test_entry_point_bookkeeping(oracle, cuda, dll):
    """return codes of lib/analysis.c:50-60 and lib/bitrate.c:88-96, :229-252"""
    ch, rate = 2, 44100
    sig = synth_signal(ch, rate, 8 * 1024, seed=77)
    st = compat.Stream(dll, ch, rate, 0.5)
    op = compat.OggPacket()
    assert dll.vorbis_bitrate_flushpacket(st.vd, op) == 0              # nothing parked yet
    for at in range(0, sig.shape[1], 1024):
        st.write(sig[:, at:at + 1024])
    assert st.blockout() == 1
    assert dll.vorbis_bitrate_addblock(st.vb) == compat.OV_EINVAL      # not analysed yet (no reference counterpart: UB there)
    assert dll.vorbis_analysis(st.vb, op) == 0                         # VBR: op is filled directly (analysis.c:55-60)
    first = bytes(op.packet[:op.bytes])
    assert op.packetno == 3 and op.b_o_s == 0 and st.vb.sequence == 3
    assert st.vb.opb.endbyte == op.bytes and st.vb.pcmend in (256, 2048)
    assert dll.vorbis_bitrate_addblock(st.vb) == 0
    assert dll.vorbis_bitrate_addblock(st.vb) == -1                    # submitted without being claimed (bitrate.c:92)
    assert dll.vorbis_bitrate_flushpacket(st.vd, op) == 1
    assert bytes(op.packet[:op.bytes]) == first
    assert dll.vorbis_bitrate_flushpacket(st.vd, op) == 0
    assert st.finish() == 0
    assert st.finish() == compat.OV_EINVAL                             # after the end
    assert st.write(sig[:, :16]) == compat.OV_EINVAL
    st.close()
    # managed-bitrate stream: vorbis_analysis with op is refused (analysis.c:50-53), the bitrate interface works
    sm = compat.Stream(dll, ch, rate, bitrate=128000)
    for at in range(0, sig.shape[1], 1024):
        sm.write(sig[:, at:at + 1024])
    assert sm.blockout() == 1
    assert dll.vorbis_analysis(sm.vb, op) == compat.OV_EINVAL
    assert dll.vorbis_analysis(sm.vb, None) == 0 and dll.vorbis_bitrate_addblock(sm.vb) == 0
    assert dll.vorbis_bitrate_flushpacket(sm.vd, op) == 1 and op.bytes > 0
    sm.close()


@pytest.mark.parametrize("bitrate", [None, 128000])
def test_three_step_setup_encodes_the_same(oracle, cuda, dll, bitrate):
    """a stream set up the way oggenc does it (vorbis_encode_setup_vbr / _managed, vorbis_encode_ctl,
    vorbis_encode_setup_init) against the oracle"""
    ch, rate, q = 2, 44100, (0.5 if bitrate is None else None)
    sig = synth_signal(ch, rate, 12 * 1024, seed=31)
    want = oracle_streams(oracle, ch, rate, q, [sig], bitrate=bitrate)[0]
    st = compat.Stream(dll, ch, rate, q, bitrate=bitrate, three_step=True)
    got = []
    for at in range(0, sig.shape[1], 1024):
        assert st.write(sig[:, at:at + 1024]) == 0
        got.extend(st.drain())
    assert st.finish() == 0
    got.extend(st.drain())
    st.close()
    assert [g[1] for g in got] == [w[1] for w in want] and [g[0] for g in got] == [w[0] for w in want]


def test_managed_streams_and_slot_reuse(oracle, cuda, dll):
    """vorbis_encode_init classes through the same calls; a finished stream's slot serves the next stream"""
    ch, rate, NS = 2, 44100, 3
    pool = C.c_int(4)
    dll.vorbis_mi355x_ctl(1, C.byref(pool))
    for gen in range(2):
        sigs = [synth_signal(ch, rate, 14 * 1024, seed=40 + 10 * gen + s) for s in range(NS)]
        want = oracle_streams(oracle, ch, rate, None, sigs, bitrate=(144000, 128000, 112000))
        ss = [compat.Stream(dll, ch, rate, bitrate=(144000, 128000, 112000)) for _ in range(NS)]
        got = [[] for _ in range(NS)]
        for at in range(0, 14 * 1024, 1024):
            for s in range(NS):
                ss[s].write(sigs[s][:, at:at + 1024])
            for s in range(NS):
                got[s].extend(ss[s].drain())
        for s in range(NS):
            ss[s].finish()
            got[s].extend(ss[s].drain())
            assert got[s] == want[s], f"generation {gen} stream {s}"
        for st in ss:
            st.close()


def test_undrained_end_of_stream_without_carve_ahead(oracle, cuda, dll):
    """VORBIS_MI355X_CARVE_AHEAD = 0: a round holds the asking stream only, so an application that declares the
    end while blocks are still un-asked-for gets the reference's result (the end-of-stream extrapolation is fitted
    to the undrained buffer, lib/block.c:497-537), with a second stream of the pool moving at its own pace."""
    ch, rate, q = 2, 44100, 0.5
    zero = C.c_int(0)
    dll.vorbis_mi355x_ctl(2, C.byref(zero))
    sig = synth_signal(ch, rate, 6 * 1024, seed=5)
    other = synth_signal(ch, rate, 9 * 1024, seed=6)
    setup = orc.Setup(oracle, ch, rate, q)
    o = orc.Stream(setup)
    oracle.lib.orc_stream_set_capture(o.v, 0)
    for at in range(0, sig.shape[1], 1024):
        o.write(sig[:, at:at + 1024])
    gen = o.blocks()
    want = [next(gen)["packet"]]                  # one block asked for, the rest left in the buffer
    o.finish()
    want += [b["packet"] for b in o.blocks()]
    o.close()
    st, st2 = compat.Stream(dll, ch, rate, q), compat.Stream(dll, ch, rate, q)
    for at in range(0, other.shape[1], 1024):
        st2.write(other[:, at:at + 1024])
    for at in range(0, sig.shape[1], 1024):
        st.write(sig[:, at:at + 1024])
    assert st.blockout() == 1
    got = [p for _, p in st.packets_of_block()]
    assert st.finish() == 0
    got += [p for _, p in st.drain()]
    assert got == want
    want2 = oracle_streams(oracle, ch, rate, q, [other])[0]
    got2 = st2.drain()
    st2.finish()
    assert got2 + st2.drain() == want2
    st.close()
    st2.close()
    one = C.c_int(1)
    dll.vorbis_mi355x_ctl(2, C.byref(one))
