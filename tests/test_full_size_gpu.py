"""BASELINE configs[2] size on the device: 16384 stereo q5 streams through the front end and the per-block
path, and configs[4]: 8192 streams of 48 kHz 5.1 q8 (long + short blocks).  Size-independent properties: 16 distinct signals are dealt round-robin over the 16384 streams, so
every stream must produce exactly the packets of the first stream that carries its signal (any lane / tile /
batch-position dependence would break this), and those 16 are compared with the oracle."""
import numpy as np
import pytest
import torch

from tests import orc
from tests.signals import synth_signal

pytestmark = pytest.mark.gpu


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
    first = [[] for _ in range(K)]       # packets of streams 0..K-1
    nblocks = 0
    for c in range(nchunks):
        chunk = base[:, :, c * 1024:(c + 1) * 1024].repeat(S // K, 1, 1).contiguous()   # stream s carries signal s % K
        fe.write(chunk)
        pending = []
        while True:
            if multi:
                if not pending:
                    info_all, packets_all, nbytes_all, counts = fe.encode_rounds(min_rounds=64, max_rounds=4)
                    at = 0
                    for cnt in counts:
                        pending.append((info_all[at:at + cnt].copy(), packets_all[at:at + cnt], nbytes_all[at:at + cnt]))
                        at += cnt
                    if not pending:
                        break
                info, packets, nbytes = pending.pop(0)
            else:
                info, packets, nbytes = fe.encode_round()
                if len(info) == 0:
                    break
            nblocks += len(info)
            # compare on the device: every packet against the packet of stream (s % K) of the same round
            stream = torch.from_numpy(np.ascontiguousarray(info["stream"])).to(cuda).long()
            pos_of = torch.full((S,), -1, dtype=torch.long, device=cuda)
            pos_of[stream] = torch.arange(len(info), device=cuda)
            ref_pos = pos_of[stream % K]
            assert bool((ref_pos >= 0).all()), "a stream produced a block in a round in which its twin did not"
            assert bool((nbytes == nbytes[ref_pos]).all())
            assert bool((nbytes >= 0).all()), "packet buffer overflow"
            # bytes beyond a packet's length are zero (the packet words are cleared first), so whole rows compare
            assert bool((packets == packets[ref_pos]).all()), "identical input, different packets"
            for k in range(K):
                p = int(pos_of[k])
                if p >= 0:
                    first[k].append(bytes(packets[p, :int(nbytes[p])].cpu().numpy()))
    assert nblocks >= S * (nchunks - 3)
    if ch == 6:
        assert nblocks > S * nchunks            # short blocks occurred
    osetup = orc.Setup(oracle, ch, rate, q)
    for k in range(K):
        st = orc.Stream(osetup)
        oracle.lib.orc_stream_set_capture(st.v, 0)
        want = []
        for c in range(nchunks):
            st.write(sigs[k][:, c * 1024:(c + 1) * 1024])
            want.extend(b["packet"] for b in st.blocks())
        st.close()
        assert first[k] == want, f"signal {k}: packets differ from the oracle"


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
