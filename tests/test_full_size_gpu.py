"""BASELINE configs[2] size on the device: 16384 stereo q5 streams through the front end and the per-block
path, and configs[4]: 8192 streams of 48 kHz 5.1 q8 (long + short blocks).  Size-independent properties: K distinct
signals are dealt round-robin over the streams, so every stream must produce exactly the packets of the first stream
that carries its signal (any lane / tile / batch-position / scheduling dependence would break this;
tests/gpuutil.py reports where), and those K are compared with the oracle."""
import numpy as np
import pytest
import torch

from tests import orc
from tests.gpuutil import TwinLedger
from tests.signals import burst_signal, synth_signal

pytestmark = pytest.mark.gpu


def oracle_packets(oracle, osetup, sig, nchunks, finish=False):
    st = orc.Stream(osetup)
    oracle.lib.orc_stream_set_capture(st.v, 0)
    want = []
    for c in range(nchunks):
        st.write(sig[:, c * 1024:(c + 1) * 1024])
        want.extend(b["packet"] for b in st.blocks())
    if finish:
        st.finish()
        want.extend(b["packet"] for b in st.blocks())
    st.close()
    return want


@pytest.mark.parametrize("S,K,ch,rate,q,nchunks,multi", [
    (16384, 16, 2, 44100, 0.5, 12, False),     # BASELINE configs[2]
    (16384, 16, 2, 44100, 0.5, 12, True),      # the same through vbm_frontend_encode_rounds (deferred joins, big batch held)
    (8192, 8, 6, 48000, 0.8, 9, False),        # BASELINE configs[4]: 5.1, block switching
])
def test_full_size_from_pcm(oracle, cuda, S, K, ch, rate, q, nchunks, multi):
    import vorbis_aotuv_lancer_amd as v
    sigs = [synth_signal(ch, rate, nchunks * 1024, seed=900 + k, level=1.0 if k % 5 else 0.02) for k in range(K)]
    base = torch.from_numpy(np.stack(sigs)).to(cuda)                       # [K, ch, n]
    setup = v.Setup(ch, rate, q)
    enc = v.Encoder(setup, S)
    fe = v.FrontEnd(enc)
    led = TwinLedger(S, K, cuda)
    rno = 0
    for c in range(nchunks):
        chunk = base[:, :, c * 1024:(c + 1) * 1024].repeat(S // K, 1, 1).contiguous()   # stream s carries signal s % K
        fe.write(chunk)
        while True:
            if multi:
                info_all, packets_all, nbytes_all, counts = fe.encode_rounds(min_rounds=64, max_rounds=4)
                if not counts:
                    break
                at = 0
                for cnt in counts:
                    led.add_host_round(info_all[at:at + cnt].copy(), packets_all[at:at + cnt], nbytes_all[at:at + cnt],
                                       f"write {c} round {rno}")
                    at += cnt
                    rno += 1
            else:
                info, packets, nbytes = fe.encode_round()
                if len(info) == 0:
                    break
                led.add_host_round(info, packets, nbytes, f"write {c} round {rno}")
                rno += 1
    led.finish("all rounds")
    assert led.nblocks >= S * (nchunks - 3)
    if ch == 6:
        assert led.nblocks > S * nchunks            # short blocks occurred
    osetup = orc.Setup(oracle, ch, rate, q)
    for k in range(K):
        assert led.lead_packets(k) == oracle_packets(oracle, osetup, sigs[k], nchunks), f"signal {k}: packets differ from the oracle"
    fe.close()
    enc.close()


@pytest.mark.parametrize("bitrate", [None, (144000, 128000, 112000)])
def test_full_size_benchmarked_path(oracle, cuda, monkeypatch, bitrate):
    """(Managed-bitrate case: `bench.py --bitrate`'s path — all fifteen packetblobs of a block per launch, the bitrate
    manager's choice ordered behind the batches before it, two workspaces, 28 writes.)
    The path bench.py times, at its size: 16384 stereo q5 streams, rounds built on the device replayed as HIP graphs,
    four workspaces, 2, 1, 1, 1 rounds per 1024-sample write (bench.py's default pattern), the feeding stream never tied to the outputs (lazy = 2, a
    consumer stream joins; the first six writes run three rounds each, as the bench's warm-up does, to clear the start of
    the streams where all of them deliver short blocks at once), 64 writes of signals with a noise burst every 40000
    samples (a fifth of the blocks are short ones).  Every stream against its twin (sequence check: a stream may be put off to a later round than its
    twin), the 16 distinct signals against the oracle, end of stream included."""
    import vorbis_aotuv_lancer_amd as v
    monkeypatch.setenv("VBM_WORKSPACES", "2" if bitrate else "4")
    S, K, ch, rate, q, nchunks = 16384, 16, 2, 44100, (None if bitrate else 0.5), (28 if bitrate else 64)
    sigs = [burst_signal(ch, rate, nchunks * 1024, seed=400 + k, period=(12000 if bitrate else 40000), level=1.0 if k % 5 else 0.05)
            for k in range(K)]
    base = torch.from_numpy(np.stack(sigs)).to(cuda)
    setup = v.Setup(ch, rate, bitrate=bitrate) if bitrate else v.Setup(ch, rate, q)
    lanes = v.lib.vbm_device_round_lanes(setup._h, S)
    enc = v.Encoder(setup, S, max_batch=lanes)
    fe = v.FrontEnd(enc)
    led = TwinLedger(S, K, cuda)
    consumer = torch.cuda.Stream(device=cuda)
    pattern = (2, 1, 1, 1)
    for c in range(nchunks):
        chunk = base[:, :, c * 1024:(c + 1) * 1024].repeat(S // K, 1, 1).contiguous()
        fe.write(chunk)
        info, packets, nbytes, counts = fe.encode_rounds_device(nrounds=3 if c < 6 else pattern[c % 4], lazy=2)
        fe.join(consumer)
        with torch.cuda.stream(consumer):
            led.add_device_rounds(info, packets, nbytes, f"write {c}")
        consumer.synchronize()          # (the ledger's temporaries go back to torch's allocator on the consumer stream)
    modes, samples = fe.device_stats()
    assert fe.refused_writes == 0
    assert sum(modes) == led.nblocks
    assert modes[0] + modes[1] >= 0.10 * sum(modes), modes        # block switching happened
    # host-built rounds drain what is left and end the streams
    torch.cuda.synchronize()
    while True:
        info, packets, nbytes = fe.encode_round()
        if len(info) == 0:
            break
        led.add_host_round(info, packets, nbytes, "drain")
    fe.finish()
    while True:
        info, packets, nbytes = fe.encode_round()
        if len(info) == 0:
            break
        led.add_host_round(info, packets, nbytes, "end of stream")
    led.finish("benchmarked path")
    osetup = orc.Setup(oracle, ch, rate, q, bitrate=bitrate)
    for k in range(K):
        assert led.lead_packets(k) == oracle_packets(oracle, osetup, sigs[k], nchunks, finish=True), \
            f"signal {k}: packets differ from the oracle"
    fe.close()
    enc.close()


def test_16384_streams_two_stream_form_is_deterministic(cuda):
    """the pipelined per-block path at full size: two runs (one- and two-stream form) give identical packets"""
    import vorbis_aotuv_lancer_amd as v
    S, ch = 16384, 2
    g = torch.Generator(device=cuda).manual_seed(1)
    setup = v.Setup(ch, 44100, 0.5)
    ids = np.arange(S, dtype=np.int32)
    fl = np.full(S, 3, np.uint8)
    blocks = [(0.4 * (torch.rand((S, ch, 2048), generator=g, device=cuda) - 0.5)).contiguous() for _ in range(4)]
    res = []
    for two in (False, True):
        enc = v.Encoder(setup, S)
        back = torch.cuda.Stream(device=cuda) if two else None
        outs = [(torch.empty((S, enc.max_packet_bytes), dtype=torch.uint8, device=cuda),
                 torch.empty((S,), dtype=torch.int32, device=cuda)) for _ in range(4)]
        for k in range(4):
            enc.analysis_batch(3, ids, fl, blocks[k], back_stream=back, out=outs[k])
        torch.cuda.synchronize()
        res.append(outs)
        enc.close()
    for k in range(4):
        assert bool((res[0][k][1] == res[1][k][1]).all()) and bool((res[0][k][1] > 0).all())
        assert bool((res[0][k][0] == res[1][k][0]).all())
