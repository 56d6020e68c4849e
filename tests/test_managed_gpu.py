"""Managed bitrate on the device against the oracle: the 15 packetblobs of every block,
the bitrate manager's choice, and the packets handed out — block by block through vbm_analysis_batch and
from raw PCM through the stream front end.

PARITY UNPINNED by the reference (its dumps are VBR only): the oracle's managed mode is checked by the
properties in tests/test_managed_oracle.py, and this file holds the device path to the oracle bit for bit."""
import numpy as np
import pytest
import torch

from tests import orc
from tests.signals import synth_signal
from tests.test_frontend_gpu import frontend_vs_oracle

pytestmark = pytest.mark.gpu


def oracle_blocks(oracle, bitrate, seconds, seed, silence=None):
    st = orc.Stream(orc.Setup(oracle, 2, 44100, bitrate=bitrate))
    sig = synth_signal(2, 44100, int(seconds * 44100), seed=seed, level=1.0 if seed % 3 else 0.05)
    if silence:
        sig[:, silence[0]:silence[1]] = 0
    out = []
    for i in range(0, sig.shape[1], 1024):
        st.write(sig[:, i:i + 1024])
        out.extend(st.blocks())
    st.close()
    return out


@pytest.mark.parametrize("bitrate,silence", [(128000, None), ((144000, 128000, 112000), (44100, 3 * 44100))])
def test_blobs_choice_and_packets_block_by_block(oracle, cuda, bitrate, silence):
    import vorbis_aotuv_lancer_amd as v
    NS = 6
    streams = [oracle_blocks(oracle, bitrate, 3.5, 300 + s, silence if s % 2 else None) for s in range(NS)]
    nsteps = min(len(b) for b in streams)
    setup = v.Setup(2, 44100, bitrate=bitrate)
    enc = v.Encoder(setup, NS)
    modes, padded, choices = set(), 0, set()
    for k in range(nsteps):
        by_mode = {}
        for s in range(NS):
            by_mode.setdefault(streams[s][k]["block_mode"], []).append(s)
        for mode, ids in sorted(by_mode.items()):
            modes.add(mode)
            blks = [streams[s][k] for s in ids]
            pcm = torch.from_numpy(np.stack([b["pcm"] for b in blks])).to(cuda)
            wflags = [b["lW"] | (b["nW"] << 1) for b in blks]
            packets, nbytes = enc.analysis_batch(mode, ids, wflags, pcm)
            packets, nbytes = packets.cpu().numpy(), nbytes.cpu().numpy()
            choice = enc.fetch("choice").cpu().numpy()
            got = enc.fetch("mdct").cpu().numpy().view(np.uint32)          # after M1 (offset_select 1 only)
            ref = np.concatenate([b["mdct"] for b in blks]).view(np.uint32)
            assert np.array_equal(got, ref), (k, mode, "mdct")
            for kb in range(15):
                bp, bn = enc.fetch_blob(kb)
                bp, bn = bp.cpu().numpy(), bn.cpu().numpy()
                for i, b in enumerate(blks):
                    assert bn[i] == b["blob_bytes"][kb], (k, mode, ids[i], "blob size", kb, int(bn[i]), b["blob_bytes"][kb])
                    assert bytes(bp[i, :bn[i]]) == b["blobs"][kb], (k, mode, ids[i], "blob bytes", kb)
            for i, b in enumerate(blks):
                assert choice[i] == b["choice"], (k, mode, ids[i], "choice", int(choice[i]), b["choice"])
                assert nbytes[i] == len(b["packet"]) and bytes(packets[i, :nbytes[i]]) == b["packet"], (k, mode, ids[i])
                padded += len(b["packet"]) > b["blob_bytes"][b["choice"]]
                choices.add(b["choice"])
    assert modes == {0, 1, 2, 3}
    assert len(choices) >= 3
    if silence:
        assert padded > 0            # the floor was hit: zero bytes appended (lib/bitrate.c:179-188)
    enc.close()
    setup.close()


def test_managed_from_pcm_matches_oracle(oracle, cuda):
    frontend_vs_oracle(oracle, cuda, 2, 44100, None, NS=5, seconds=2.5, bitrate=128000)


def test_managed_min_max_from_pcm_with_silence(oracle, cuda):
    nsamp = int(4.0 * 44100) // 1024 * 1024
    sigs = [synth_signal(2, 44100, nsamp, seed=700 + s) for s in range(4)]
    for s in (1, 3):
        sigs[s][:, 44100:3 * 44100] = 0
    frontend_vs_oracle(oracle, cuda, 2, 44100, None, NS=4, seconds=4.0, bitrate=(144000, 128000, 112000), sigs=sigs)


@pytest.mark.parametrize("ch,rate,bitrate,secs", [
    (1, 44100, 64000, 2.0),       # uncoupled, residue 1
    (6, 48000, 320000, 1.7),      # 5.1: several coupling steps (general couple kernel), two submaps
    (2, 22050, 56000, 3.0),       # low rate: 512/1024 blocks, no M3
])
def test_managed_other_classes_from_pcm(oracle, cuda, ch, rate, bitrate, secs):
    frontend_vs_oracle(oracle, cuda, ch, rate, None, NS=4, seconds=secs, bitrate=bitrate)


# Other managed-bitrate classes (vorbis_encode_init): mono (the lane-per-bin couple kernel without coupling walks the
# blobs), 22.05 kHz (512 / 1024 blocks), coupled 5.1 (the general couple kernel: one launch per blob between the wide
# launches), and the stereo 44.1 / 48 kHz bitrate ladder.  Oracle-only pin (the reference's dumps are VBR).
@pytest.mark.parametrize("ch,rate,bitrate", [
    (1, 44100, 64000), (2, 22050, 56000), (6, 48000, 320000),
    (2, 44100, 64000), (2, 44100, 96000), (2, 44100, 160000), (2, 44100, 192000), (2, 44100, 256000), (2, 48000, 128000),
])
def test_other_managed_classes_from_pcm(oracle, cuda, ch, rate, bitrate):
    frontend_vs_oracle(oracle, cuda, ch, rate, None, NS=4, seconds=1.7, bitrate=bitrate)


def test_loop_over_the_blobs_gives_the_same(oracle, cuda, monkeypatch):
    """VBM_MANAGED_WIDE=0: the round-2 form of the back half (one pass per packetblob) is kept for A/B; same packets"""
    monkeypatch.setenv("VBM_MANAGED_WIDE", "0")
    frontend_vs_oracle(oracle, cuda, 2, 44100, None, NS=4, seconds=1.7, bitrate=(144000, 128000, 112000))
