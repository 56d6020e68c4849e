"""Stream front end on the device (vbm_frontend_*: PCM in, packets out) against

  * the reference's own packet dumps (tests/golden/*.pkt, recorded from the reference's scalar build
    on the survey probe signal): start-of-stream LPC extrapolation, envelope search, block switching;
    the dumps end without the end-of-stream flush, which is therefore pinned by the oracle only, and
  * the oracle, stream by stream, for many concurrent streams with different block sequences:
    (lW, W, nW, block type, granulepos, packetno, e_o_s) and packet bytes of every block."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import orc
from tests.signals import synth_signal

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def split_dump(d):
    out, at = [], 0
    while at < len(d):
        n = int.from_bytes(d[at:at + 4], "little")
        out.append(d[at + 4:at + 4 + n])
        at += 4 + n
    return out


def collect(sink, info, packets, nbytes):
    packets = packets.cpu().numpy()
    nbytes = nbytes.cpu().numpy()
    for k, pi in enumerate(info):
        assert nbytes[k] >= 0, "packet buffer overflow"
        sink[int(pi["stream"])].append(((int(pi["lW"]), int(pi["W"]), int(pi["nW"]), int(pi["block_mode"]),
                                         int(pi["eos"]), int(pi["granulepos"]), int(pi["packetno"])),
                                        bytes(packets[k, :nbytes[k]])))


def drain(fe, sink, max_rounds=None, multi=False):
    """rounds until no stream has a block (or, with max_rounds, until that many rounds ran and no buffer
    is more than half full); sink[stream] collects (info fields, packet bytes).  multi: through
    vbm_frontend_encode_rounds (several rounds per call, joined at the end)"""
    if multi == "lazy":
        # lazy joins: the outputs of a call are complete once the next call has been enqueued (or after join)
        held = getattr(fe, "_held", None)
        while True:
            out = fe.encode_rounds(min_rounds=max_rounds or 64, max_rounds=4 if max_rounds is None else 16, lazy=True)
            if held is not None:
                collect(sink, held[0].copy(), held[1], held[2])
            held = (out[0], out[1], out[2])
            if not out[3] or max_rounds is not None:
                break
        fe._held = held
        return
    if multi:
        while True:
            info, packets, nbytes, counts = fe.encode_rounds(min_rounds=max_rounds or 64, max_rounds=4 if max_rounds is None else 16)
            collect(sink, info, packets, nbytes)
            if not counts or max_rounds is not None:
                return
    rounds = 0
    while True:
        if max_rounds is not None and rounds >= max_rounds and fe.max_buffered + 1024 <= fe.capacity // 2:
            return
        rounds += 1
        info, packets, nbytes = fe.encode_round()
        if len(info) == 0:
            return
        packets = packets.cpu().numpy()
        nbytes = nbytes.cpu().numpy()
        for k, pi in enumerate(info):
            assert nbytes[k] >= 0, "packet buffer overflow"
            sink[int(pi["stream"])].append(((int(pi["lW"]), int(pi["W"]), int(pi["nW"]), int(pi["block_mode"]),
                                             int(pi["eos"]), int(pi["granulepos"]), int(pi["packetno"])),
                                            bytes(packets[k, :nbytes[k]])))


def probe_pcm(oracle, ch, rate, secs):
    total = rate * secs
    n = ((total + 1023) // 1024) * 1024          # the probe driver writes whole 1024-sample chunks
    out = np.empty((ch, n), np.float32)
    oracle.lib.orc_probe_signal.argtypes = [C.c_int, C.c_long, C.c_long, C.c_void_p]
    oracle.lib.orc_probe_signal(ch, rate, n, out.ctypes.data)
    return out


@pytest.mark.parametrize("ch,rate,q,secs,golden", [
    (2, 44100, 0.5, 20, "ref_scalar_2ch_44100_q05_20s.pkt"),
    (6, 48000, 0.8, 10, "ref_scalar_6ch_48000_q08_10s.pkt"),
])
def test_frontend_reproduces_reference_packet_dump(oracle, cuda, ch, rate, q, secs, golden):
    import vorbis_aotuv_lancer_amd as v
    ref = split_dump(open(os.path.join(G, golden), "rb").read())
    pcm = probe_pcm(oracle, ch, rate, secs)
    enc = v.Encoder(v.Setup(ch, rate, q), 1)
    fe = v.FrontEnd(enc)
    got = [[]]
    dev_pcm = torch.from_numpy(pcm).to(cuda)
    for at in range(0, pcm.shape[1], 1024):
        fe.write(dev_pcm[None, :, at:at + 1024].contiguous())
        drain(fe, got)
    # the survey's probe driver stops after the last chunk without vorbis_analysis_wrote(vd, 0), so the
    # dump holds exactly the packets produced so far
    pk = [p for _, p in got[0]]
    assert len(pk) == len(ref), (len(pk), len(ref))
    bad = [i for i in range(len(ref)) if pk[i] != ref[i]]
    assert not bad, f"{len(bad)} packets differ from the reference dump, first at {bad[0]}"
    assert [m[6] for m, _ in got[0]] == list(range(3, 3 + len(pk)))                # packetno
    # end of stream (not in the reference dump: pinned by the oracle in the test below)
    fe.finish()
    drain(fe, got)
    assert len(got[0]) > len(ref)
    assert got[0][-1][0][4] == 1 and all(m[4] == 0 for m, _ in got[0][:-1])      # e_o_s on the last packet only


def frontend_vs_oracle(oracle, cuda, ch, rate, q, NS, seconds, need_modes=(0, 1, 2, 3), max_rounds=None, bitrate=None,
                       sigs=None, multi=False):
    import vorbis_aotuv_lancer_amd as v
    nsamp = int(seconds * rate) // 1024 * 1024
    if sigs is None:
        sigs = [synth_signal(ch, rate, nsamp, seed=500 + s, level=1.0 if s % 3 else 0.05) for s in range(NS)]
    # oracle: same write pattern (1024 at a time, drain after every write, then end of stream)
    osetup = orc.Setup(oracle, ch, rate, q, bitrate=bitrate)
    want = []
    for s in range(NS):
        st = orc.Stream(osetup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        seq = []
        for at in range(0, nsamp, 1024):
            st.write(sigs[s][:, at:at + 1024])
            seq.extend(st.blocks())
        st.finish()
        seq.extend(st.blocks())
        st.close()
        want.append([((b["lW"], b["W"], b["nW"], b["block_mode"], b["eos"], b["granulepos"], b["sequence"]), b["packet"])
                     for b in seq])
    enc = v.Encoder(v.Setup(ch, rate, q, bitrate=bitrate), NS)
    fe = v.FrontEnd(enc)
    got = [[] for _ in range(NS)]
    allp = torch.from_numpy(np.stack(sigs)).to(cuda)
    for at in range(0, nsamp, 1024):
        fe.write(allp[:, :, at:at + 1024].contiguous())
        drain(fe, got, max_rounds, multi)
    def flush_held():
        if getattr(fe, "_held", None) is not None:
            fe.join()
            collect(got, fe._held[0].copy(), fe._held[1], fe._held[2])
            fe._held = None
    # vorbis_analysis_wrote(vd, 0) fits its end-of-stream LPC to what the buffer holds at that moment
    # (lib/block.c:531-541), so, like the reference application loop, drain before finishing
    flush_held()
    drain(fe, got)
    fe.finish()
    drain(fe, got)
    modes = set()
    for s in range(NS):
        assert [m for m, _ in got[s]] == [m for m, _ in want[s]], f"stream {s}: block sequence differs"
        bad = [i for i in range(len(want[s])) if got[s][i][1] != want[s][i][1]]
        assert not bad, f"stream {s}: packet {bad[0]} differs"
        modes |= {m[3] for m, _ in got[s]}
    assert modes >= set(need_modes), modes


def test_frontend_many_streams_match_oracle(oracle, cuda):
    frontend_vs_oracle(oracle, cuda, 2, 44100, 0.5, NS=70, seconds=1.6)


@pytest.mark.parametrize("max_rounds", [None, 2])
def test_frontend_multi_round_calls_match_oracle(oracle, cuda, max_rounds):
    """vbm_frontend_encode_rounds: rounds with deferred joins (a round beside the long-block batch of the round
    before, streams changing block type from round to round) — drained completely, and two rounds per write"""
    frontend_vs_oracle(oracle, cuda, 2, 44100, 0.5, NS=200, seconds=1.6, multi=True, max_rounds=max_rounds)


@pytest.mark.parametrize("bitrate", [None, (144000, 128000, 112000)])
def test_frontend_lazy_joins_at_scale(oracle, cuda, monkeypatch, bitrate):
    """vbm_frontend_encode_rounds_lazy with a big batch (>= 1024 long blocks per write): its back half stays
    pending across calls (four workspaces in rotation), outputs are read one call late; 1100 streams carrying
    5 distinct signals, each compared with the oracle.  With a managed-bitrate setup the back half of a batch
    moves carried state too (the reservoirs, k_bitrate_choose): a stream's next block — in whatever batch of
    whatever round it lands — has to wait for it, or the blob choice goes wrong."""
    import vorbis_aotuv_lancer_amd as v
    monkeypatch.setenv("VBM_WORKSPACES", "4")
    K, S, ch, rate, q = 5, 1100, 2, 44100, (0.5 if bitrate is None else None)
    nsamp = (30 if bitrate is None else 22) * 1024
    base = [synth_signal(ch, rate, nsamp, seed=640 + k, level=1.0 if k % 2 else 0.05) for k in range(K)]
    sigs = [base[s % K] for s in range(S)]
    osetup = orc.Setup(oracle, ch, rate, q, bitrate=bitrate)
    want = []
    for k in range(K):
        st = orc.Stream(osetup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        seq = []
        for at in range(0, nsamp, 1024):
            st.write(base[k][:, at:at + 1024])
            seq.extend(b["packet"] for b in st.blocks())
        st.close()
        want.append(seq)
    enc = v.Encoder(v.Setup(ch, rate, q, bitrate=bitrate), S)
    fe = v.FrontEnd(enc)
    got = [[] for _ in range(S)]
    allp = torch.from_numpy(np.stack(sigs)).to(cuda)
    for at in range(0, nsamp, 1024):
        fe.write(allp[:, :, at:at + 1024].contiguous())
        drain(fe, got, 2, "lazy")
    fe.join()
    collect(got, fe._held[0].copy(), fe._held[1], fe._held[2])
    fe._held = None
    drain(fe, got)
    for s in range(S):
        assert [p for _, p in got[s]] == want[s % K], f"stream {s}"


def test_frontend_multi_round_managed(oracle, cuda):
    frontend_vs_oracle(oracle, cuda, 2, 44100, None, NS=5, seconds=2.0, bitrate=128000, multi=True)


def test_frontend_output_does_not_depend_on_round_policy(oracle, cuda):
    """Two rounds per write instead of draining: streams inside runs of short blocks fall behind their
    input and catch up later; blocks and packets stay those of the oracle (which drains every time)."""
    frontend_vs_oracle(oracle, cuda, 2, 44100, 0.5, NS=40, seconds=2.2, max_rounds=2)


# other mode families through the same kernels.  No reference dump exists for these
# classes (parity with the reference unpinned); the oracle is the general restatement that the three
# dumps pin for stereo q5 / q1 and 5.1 q8.
@pytest.mark.parametrize("ch,rate,q", [
    (2, 44100, 0.3),     # point-stereo coupling limits / lowpass of a lower quality
    (2, 44100, 0.9),     # no lowpass, near-lossless coupling
    (2, 44100, 1.0),     # top of the quality range (base_setting 11.999)
    (2, 44100, 0.0),     # bottom of the 256/2048 range
    (2, 48000, 0.5),     # ve_setup_48_stereo
    (2, 32000, 0.5),     # ve_setup_32_stereo
    (1, 44100, 0.5),     # uncoupled: residue type 1 on the main channel
    (6, 48000, 0.3),     # coupled 5.1: several coupling steps sharing channels (serial couple path)
    (2, 22050, 0.5),     # ve_setup_22_stereo: 512/1024 blocks
    (2, 16000, 0.5),     # ve_setup_16_stereo: 512/1024 blocks
    (1, 11025, 0.5),     # ve_setup_11_uncoupled: one block size (512), one mode
    (1, 8000, 0.5),      # ve_setup_8_uncoupled
    (2, 44100, -0.1),    # q < 0: 512/4096 blocks
    # round 3: the rest of the reference's test matrix (test/test.c:36-45: 1..8 channels, up to 96 kHz)
    (3, 44100, 0.5),     # ve_setup_44_uncoupled with 3 .. 8 channels: one submap, residue type 1, a vector per channel
    (4, 44100, 0.5),
    (5, 44100, 0.5),
    (7, 44100, 0.5),
    (8, 44100, 0.5),
    (8, 48000, 0.5),     # ve_setup_48_uncoupled
    (2, 96000, 0.5),     # ve_setup_X_stereo (lib/modes/setup_X.h, lib/vorbisenc.c:185-188)
    # the quality ladder (round 3, late): interpolated settings between the table rows of the templates
    (2, 44100, 0.7), (2, 48000, 0.2), (2, 48000, 0.9), (1, 44100, 0.2), (1, 44100, 0.9), (2, 32000, 0.2), (2, 22050, 0.8),
    (6, 48000, 0.5), (6, 44100, 0.5), (6, 48000, 0.1),
])
def test_frontend_other_mode_classes_match_oracle(oracle, cuda, ch, rate, q):
    # the 11 kHz and 8 kHz setups have a single block size: only block types 0 and 1 exist
    need = (0, 1) if rate < 16000 else (0, 1, 2, 3)
    frontend_vs_oracle(oracle, cuda, ch, rate, q, NS=6, seconds=1.7 if rate >= 16000 else 4.0, need_modes=need)


def test_complete_ogg_stream(oracle, cuda):
    """PCM -> device front end -> packets -> header packets + Ogg pages (N3): the file's pages parse back to
    the three headers plus exactly the packets the oracle produces, granule positions end at the sample
    count, e_o_s sits on the last page."""
    import vorbis_aotuv_lancer_amd as v
    from tests.test_stream_wrapper import parse_pages, packets_of, unpack_headers
    ch, rate, q = 2, 44100, 0.5
    nsamp = 50 * 1024
    sig = synth_signal(ch, rate, nsamp, seed=77)
    setup = v.Setup(ch, rate, q)
    enc = v.Encoder(setup, 1)
    fe = v.FrontEnd(enc)
    got = [[]]
    dev = torch.from_numpy(sig).to(cuda)
    for at in range(0, nsamp, 1024):
        fe.write(dev[None, :, at:at + 1024].contiguous())
        drain(fe, got)
    fe.finish()
    drain(fe, got)
    blob = v.write_ogg(setup, [p for _, p in got[0]], [(m[5], bool(m[4])) for m, _ in got[0]], serialno=7,
                       comments=["ENCODER=mi355x"])
    pages = parse_pages(blob)
    pk = packets_of(pages)
    hdr = v.header_packets(setup, ["ENCODER=mi355x"])
    assert pk[:3] == hdr and pk[3:] == [p for _, p in got[0]]
    unpack_headers(*pk[:3])
    assert pages[0]["flags"] & 2 and pages[-1]["flags"] & 4 and pages[-1]["granule"] == nsamp
    # the audio packets start on a fresh page after the headers (examples/encoder_example.c:150-157)
    header_pages = [pg for pg in pages if pg["granule"] == 0]
    assert sum(1 for pg in header_pages for lv in pg["lacing"] if lv < 255) == 3
    # and they are the oracle's
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


def test_frontend_reproduces_reference_120s_len_crc(oracle, cuda):
    """Two minutes of the probe signal from raw PCM: 5866 packets whose (length, crc32) the survey recorded
    from the reference's scalar build (tests/golden/ref_scalar_2ch_44100_q05_120s.lencrc.npy)."""
    import zlib
    import vorbis_aotuv_lancer_amd as v
    ref = np.load(os.path.join(G, "ref_scalar_2ch_44100_q05_120s.lencrc.npy"))
    pcm = probe_pcm(oracle, 2, 44100, 120)
    enc = v.Encoder(v.Setup(2, 44100, 0.5), 1)
    fe = v.FrontEnd(enc)
    got = [[]]
    dev_pcm = torch.from_numpy(pcm).to(cuda)
    for at in range(0, pcm.shape[1], 1024):
        fe.write(dev_pcm[None, :, at:at + 1024].contiguous())
        drain(fe, got)
    pk = [p for _, p in got[0]]
    assert len(pk) == ref.shape[0] == 5866
    mine = np.array([(len(p), zlib.crc32(p)) for p in pk], dtype=np.uint32)
    assert np.array_equal(mine, ref)


@pytest.mark.parametrize("layout", ["plain", "device arena", "pinned host arena"])
def test_streams_out_of_lock_step(oracle, cuda, layout):
    """Slots that start late, run at their own pace, end and are reused (vbm_frontend_write_streams /
    _restart_streams): every logical stream still equals the oracle run on its own.  The arena layouts go through
    vbm_frontend_write_streams_strided: every slot has its own [channels][2048] region, indexed by slot, in device memory
    or in pinned host memory that the append kernel reads over the bus (what the drop-in shim's vorbis_analysis_buffer
    hands out)."""
    import vorbis_aotuv_lancer_amd as v
    ch, rate, q = 2, 44100, 0.5
    osetup = orc.Setup(oracle, ch, rate, q)

    def oracle_stream(sig):
        st = orc.Stream(osetup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        seq = []
        for at in range(0, sig.shape[1], 1024):
            st.write(sig[:, at:at + 1024])
            seq.extend(st.blocks())
        st.finish()
        seq.extend(st.blocks())
        st.close()
        return [((b["lW"], b["W"], b["nW"], b["block_mode"], b["eos"], b["granulepos"], b["sequence"]), b["packet"]) for b in seq]

    sig = {name: synth_signal(ch, rate, n * 1024, seed=seed) for name, n, seed in
           [("a0", 30, 11), ("a1", 45, 12), ("b2", 25, 13), ("b3", 25, 14), ("c0", 12, 15)]}
    enc = v.Encoder(v.Setup(ch, rate, q), 4)
    fe = v.FrontEnd(enc)
    got = [[] for _ in range(4)]
    logical = {}
    A = 2048
    arena = None
    if layout == "device arena":
        arena = torch.zeros((4, ch, A), dtype=torch.float32, device=cuda)
    elif layout == "pinned host arena":
        arena = torch.zeros((4, ch, A), dtype=torch.float32).pin_memory()

    def feed(slots_and_names, chunk):
        ids, pcs = [], []
        for slot, name in slots_and_names:
            if chunk[name] * 1024 < sig[name].shape[1]:
                ids.append(slot)
                pcs.append(sig[name][:, chunk[name] * 1024:(chunk[name] + 1) * 1024])
                chunk[name] += 1
        if ids and arena is None:
            fe.write_streams(ids, torch.from_numpy(np.stack(pcs)).to(cuda).contiguous())
        elif ids:
            for slot, pc in zip(ids, pcs):
                arena[slot, :, :1024] = torch.from_numpy(np.ascontiguousarray(pc)).to(arena.device)
            torch.cuda.synchronize()
            fe.write_streams_strided(ids, arena, 1024, ch * A, A, by_slot=True)      # (returns when the samples are taken)
        drain(fe, got)

    chunk = {k: 0 for k in sig}
    for _ in range(20):                                    # slots 0, 1 run; slots 2, 3 have never been written
        feed([(0, "a0"), (1, "a1")], chunk)
    assert not got[2] and not got[3]
    fe.restart_streams([2, 3])
    for _ in range(10):                                    # a0 ends after 30 chunks
        feed([(0, "a0"), (1, "a1"), (2, "b2"), (3, "b3")], chunk)
    fe.finish([0])
    drain(fe, got)
    logical["a0"], got[0] = got[0], []
    fe.restart_streams([0])                                # slot 0 is reused
    for _ in range(15):
        feed([(0, "c0"), (1, "a1"), (2, "b2"), (3, "b3")], chunk)
    fe.finish([0, 1, 2, 3])
    drain(fe, got)
    logical["c0"], logical["a1"], logical["b2"], logical["b3"] = got
    for name in sig:
        assert chunk[name] * 1024 == sig[name].shape[1]
        assert logical[name] == oracle_stream(sig[name]), f"stream {name} differs from the oracle"


# ---- rounds built on the device (vbm_frontend_encode_rounds_device): no host in the loop ------------------------
def collect_device(sink, fe, out):
    """outputs of one encode_rounds_device call (device tensors) -> sink[stream]"""
    import vorbis_aotuv_lancer_amd as v
    info, packets, nbytes, counts = out
    nb = nbytes.cpu().numpy()
    live = np.flatnonzero(nb != -2)
    if len(live) == 0:
        return 0
    rec = info.cpu().numpy().view(np.dtype(v.PacketInfo))[:, 0]
    pk = packets[torch.from_numpy(live).to(packets.device)].cpu().numpy()
    for j, k in enumerate(live):
        pi = rec[k]
        assert nb[k] >= 0, "packet buffer overflow"
        assert pi["stream"] >= 0
        sink[int(pi["stream"])].append(((int(pi["lW"]), int(pi["W"]), int(pi["nW"]), int(pi["block_mode"]), int(pi["eos"]),
                                         int(pi["granulepos"]), int(pi["packetno"])), bytes(pk[j, :nb[k]])))
    # the counts are the live lanes per block type, and every region is filled from its start
    c = counts.cpu().numpy()
    assert int(c.sum()) == len(live)
    return len(live)


@pytest.mark.parametrize("ch,rate,q,NS,lazy", [(2, 44100, 0.5, 70, False), (2, 44100, 0.5, 70, True), (2, 44100, 0.5, 1100, False),
                                               (2, 44100, 0.5, 1100, True), (6, 48000, 0.8, 9, False),
                                               (1, 8000, 0.5, 6, False), (2, 44100, -0.1, 5, False),
                                               (2, 44100, 0.5, 1100, 2)])
def test_device_built_rounds(oracle, cuda, monkeypatch, ch, rate, q, NS, lazy):
    run_device_built_rounds(oracle, cuda, monkeypatch, ch, rate, q, NS, lazy, min(NS, 7))


def test_device_built_rounds_all_streams_switch_together(oracle, cuda, monkeypatch):
    """1100 streams with the SAME signal: every burst makes all of them ask for short blocks in the same round, the
    short types' lane regions (320 lanes) overflow, and most streams are put off — across calls too — and catch up
    later.  Packets must not depend on when a stream's blocks run."""
    run_device_built_rounds(oracle, cuda, monkeypatch, 2, 44100, 0.5, 1100, False, 1)


def run_device_built_rounds(oracle, cuda, monkeypatch, ch, rate, q, NS, lazy, K):
    """Two device-built rounds per 1024-sample write, outputs read one call late when lazy (lazy = 2: the feeding
    stream is never tied to the outputs; a consumer stream joins before it reads); then the host-built rounds drain
    what is left and end the streams.  Per stream, in order: block flags, granule positions, packet numbers and
    packet bytes of the oracle."""
    import vorbis_aotuv_lancer_amd as v
    monkeypatch.setenv("VBM_WORKSPACES", "4")
    nsamp = 26 * 1024
    base = [synth_signal(ch, rate, nsamp, seed=730 + k, level=1.0 if k % 3 else 0.05) for k in range(K)]
    if K == 1:      # a tone with three noise bursts: every burst switches every stream to short blocks at once
        rng = np.random.default_rng(5)
        t = np.arange(nsamp) / rate
        x = np.stack([0.3 * np.sin(2 * np.pi * 440 * t + c) for c in range(ch)]).astype(np.float32)
        for at in (6000, 13500, 20500):
            x[:, at:at + 200] += (0.6 * rng.standard_normal((ch, 200))).astype(np.float32)
        base = [x]
    osetup = orc.Setup(oracle, ch, rate, q)
    want = []
    for k in range(K):
        st = orc.Stream(osetup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        seq = []
        for at in range(0, nsamp, 1024):
            st.write(base[k][:, at:at + 1024])
            seq.extend(st.blocks())
        st.finish()
        seq.extend(st.blocks())
        st.close()
        want.append([((b["lW"], b["W"], b["nW"], b["block_mode"], b["eos"], b["granulepos"], b["sequence"]), b["packet"])
                     for b in seq])
    setup = v.Setup(ch, rate, q)
    lanes = v.lib.vbm_device_round_lanes(setup._h, NS)
    enc = v.Encoder(setup, NS, max_batch=lanes)
    fe = v.FrontEnd(enc)
    assert fe.device_lanes == lanes
    got = [[] for _ in range(NS)]
    allp = torch.from_numpy(np.stack([base[s % K] for s in range(NS)])).to(cuda)
    held = None
    total = 0
    for at in range(0, nsamp, 1024):
        fe.write(allp[:, :, at:at + 1024].contiguous())
        # (two rounds per write keep up with 256/2048 switching; 512-sample blocks at one size come 4 per write)
        out = fe.encode_rounds_device(nrounds=2 if setup.blocksizes[0] != setup.blocksizes[1] else 5, lazy=lazy)
        if lazy == 2:
            consumer = getattr(test_device_built_rounds, "_consumer", None) or torch.cuda.Stream(device=cuda)
            test_device_built_rounds._consumer = consumer
            fe.join(consumer)
            with torch.cuda.stream(consumer):
                total += collect_device(got, fe, out)
        elif lazy:
            if held is not None:
                total += collect_device(got, fe, held)
            held = out
        else:
            total += collect_device(got, fe, out)
    if lazy and lazy != 2:
        fe.join()
        total += collect_device(got, fe, held)
    modes, samples = fe.device_stats()
    assert sum(modes) == total
    assert fe.refused_writes == 0
    if K == 1:      # the case is only worth its name if short blocks came and a region overflowed
        assert modes[0] + modes[1] > 2 * NS, modes
    drain(fe, got)                      # host-built rounds take over: the mirrors are fetched from the device
    fe.finish()
    drain(fe, got)
    bad = [(s, k) for s in range(NS) for k, (g, w) in enumerate(zip(got[s], want[s % K])) if g != w]
    assert not bad, (len(bad), bad[:12], [got[s][k][0] for s, k in bad[:6]])
    for s in range(NS):
        assert len(got[s]) == len(want[s % K]), (s, len(got[s]), len(want[s % K]))
    fe.close()
