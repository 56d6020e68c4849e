"""Full device pipeline vs the oracle: stage by stage and packet by packet.

NS independent streams are encoded by the oracle on the CPU (block sequence, raw block PCM,
every captured stage vector, packets).  The same blocks are then pushed through
vbm_analysis_batch in lock-step (k-th block of every stream per step, grouped by block type),
and every intermediate the C ABI exposes is compared bit-for-bit."""
import numpy as np
import pytest
import torch

from tests import orc
from tests.signals import synth_signal

pytestmark = pytest.mark.gpu

STAGES_F = ["mdct_raw", "logfft", "logmdct", "noise", "tone", "logmask", "mdct"]


def oracle_blocks(oracle, ch, rate, q, seconds, seed):
    setup = orc.Setup(oracle, ch, rate, q)
    st = orc.Stream(setup)
    sig = synth_signal(ch, rate, int(seconds * rate), seed=seed, level=1.0 if seed % 3 else 0.05)
    out = []
    for i in range(0, sig.shape[1], 1024):
        st.write(sig[:, i:i + 1024])
        out.extend(st.blocks())
    st.close()
    return out


def run_case(oracle, cuda, ch, rate, q, nstreams, seconds, check_stages=True, res1_channels=(), sub_batches=1,
             two_streams=False):
    import vorbis_aotuv_lancer_amd as v
    streams = [oracle_blocks(oracle, ch, rate, q, seconds, seed=100 + s) for s in range(nstreams)]
    nsteps = min(len(b) for b in streams)
    assert nsteps > 20
    setup = v.Setup(ch, rate, q)
    enc = v.Encoder(setup, nstreams)
    enc.set_sub_batches(sub_batches)
    assert enc.sub_batches == sub_batches
    seen_modes = set()
    mismatches = []
    pending = []
    back = torch.cuda.Stream(device=cuda) if two_streams else None
    for k in range(nsteps):
        by_mode = {}
        for s in range(nstreams):
            by_mode.setdefault(streams[s][k]["block_mode"], []).append(s)
        for mode, ids in sorted(by_mode.items()):
            seen_modes.add(mode)
            blks = [streams[s][k] for s in ids]
            pcm = torch.from_numpy(np.stack([b["pcm"] for b in blks])).to(cuda)
            wflags = [b["lW"] | (b["nW"] << 1) for b in blks]
            if two_streams:   # vbm_analysis_batch2: back half on its own stream; no host sync between the calls
                outs = (torch.empty((len(ids), enc.max_packet_bytes), dtype=torch.uint8, device=cuda),
                        torch.empty((len(ids),), dtype=torch.int32, device=cuda))
                pending.append((k, mode, blks, outs))
                enc.analysis_batch(mode, ids, wflags, pcm, back_stream=back, out=outs)
                continue
            packets, nbytes = enc.analysis_batch(mode, ids, wflags, pcm)
            nbytes = nbytes.cpu().numpy()
            packets = packets.cpu().numpy()
            if check_stages:
                for name in STAGES_F:
                    got = enc.fetch(name).cpu().numpy().view(np.uint32)
                    ref = np.concatenate([b[name] for b in blks]).view(np.uint32)
                    if not np.array_equal(got, ref):
                        bad = np.argwhere(got != ref)[0]
                        mismatches.append((k, mode, name, tuple(bad)))
                got = enc.fetch("post_valid").cpu().numpy()
                ref = np.concatenate([b["post_valid"] for b in blks])
                if not np.array_equal(got, ref):
                    mismatches.append((k, mode, "post_valid", ()))
                got = enc.fetch("residue").cpu().numpy()
                ref = np.concatenate([b["residue"] for b in blks])
                if res1_channels:
                    # res-1 encodes in place (lib/res0.c:372-375 on in[j] itself): after the packet
                    # kernel those channels hold the VQ remainder, not the quantised residue
                    keep = np.array([c not in res1_channels for c in range(ch)] * len(blks))
                    got, ref = got[keep], ref[keep]
                if not np.array_equal(got, ref):
                    mismatches.append((k, mode, "residue", tuple(np.argwhere(got != ref)[0])))
                got = enc.fetch("nonzero").cpu().numpy()
                ref = np.concatenate([b["nonzero"] for b in blks])
                if not np.array_equal(got, ref):
                    mismatches.append((k, mode, "nonzero", ()))
            for i, b in enumerate(blks):
                pk = bytes(packets[i, :max(nbytes[i], 0)])
                if nbytes[i] != len(b["packet"]) or pk != b["packet"]:
                    mismatches.append((k, mode, "packet", (ids[i], int(nbytes[i]), len(b["packet"]))))
            assert not mismatches, mismatches[:8]
    if two_streams:
        torch.cuda.synchronize()
        for k, mode, blks, (packets, nbytes) in pending:
            packets, nbytes = packets.cpu().numpy(), nbytes.cpu().numpy()
            for i, b in enumerate(blks):
                pk = bytes(packets[i, :max(nbytes[i], 0)])
                if nbytes[i] != len(b["packet"]) or pk != b["packet"]:
                    mismatches.append((k, mode, "packet", (i, int(nbytes[i]), len(b["packet"]))))
        assert not mismatches, mismatches[:8]
    enc.close()
    setup.close()
    return seen_modes, nsteps


def test_stereo_q5_stage_and_packet_parity(oracle, cuda):
    modes, nsteps = run_case(oracle, cuda, 2, 44100, 0.5, nstreams=24, seconds=4.0)
    assert modes == {0, 1, 2, 3}     # impulse, padding, transition and long blocks all occurred


def test_noise_mask_ring_form(oracle, cuda, monkeypatch):
    """VBM_NOISE_RING=1: the noise mask's sums in a 512-row ring with a scan wavefront beside the solves (opt-in: measured
    slower) — same stage outputs and packets"""
    monkeypatch.setenv("VBM_NOISE_RING", "1")
    modes, nsteps = run_case(oracle, cuda, 2, 44100, 0.5, nstreams=24, seconds=2.0)
    assert 3 in modes


def test_fused_packet_assembly(oracle, cuda, monkeypatch):
    """VBM_PACK_FUSED=1: one wavefront per stream-block builds the packet with the codewords in LDS (opt-in: 0.2 GB of
    traffic instead of 0.9, slower in the pipeline) — same packets, all block types"""
    monkeypatch.setenv("VBM_PACK_FUSED", "1")
    modes, nsteps = run_case(oracle, cuda, 2, 44100, 0.5, nstreams=24, seconds=2.0)
    assert 3 in modes


def test_stereo_q1_packet_parity(oracle, cuda):
    # q0.1: live noise normalisation (normal_start 16/128) and the 128x4 short floor
    run_case(oracle, cuda, 2, 44100, 0.1, nstreams=8, seconds=3.0)


def test_surround_51_q8_packet_parity(oracle, cuda):
    # 6 channels, two submaps: 5-channel res-2 (uncoupled) + LFE res-1
    run_case(oracle, cuda, 6, 48000, 0.8, nstreams=6, seconds=3.0, res1_channels=(5,))


def test_stereo_q5_many_streams_in_sub_batches(oracle, cuda):
    """150 streams = 3 tiles of 64 stream-blocks, encoded as 2 slices on separate internal HIP streams
    (vbm_encoder_set_sub_batches): packets must still equal the oracle's for every stream."""
    run_case(oracle, cuda, 2, 44100, 0.5, nstreams=150, seconds=0.7, check_stages=False, sub_batches=2)


def test_stereo_q5_two_stream_form(oracle, cuda):
    """vbm_analysis_batch2: back half (floor, couple/quantise, packets) of every call on a second stream,
    overlapping the front half of the next call, no host synchronisation in between; all block types."""
    modes, _ = run_case(oracle, cuda, 2, 44100, 0.5, nstreams=20, seconds=3.0, check_stages=False, two_streams=True)
    assert modes == {0, 1, 2, 3}


def test_two_stream_form_with_sub_batches(oracle, cuda):
    """two-stream form + sub-batches: the back half of a call runs as two slices (caller's back stream and an
    internal one beside it), joined before the workspace is handed back"""
    run_case(oracle, cuda, 2, 44100, 0.5, nstreams=150, seconds=0.7, check_stages=False, sub_batches=2, two_streams=True)
