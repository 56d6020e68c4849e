"""HIP window + FFT + log-power spectrum vs the CPU oracle — bit-exact."""
import numpy as np
import pytest
import torch

from tests import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def signal(rng, shape):
    t = np.arange(shape[-1], dtype=np.float32)
    f = rng.uniform(0.001, 0.4, shape[:-1] + (1,)).astype(np.float32)
    x = 0.4 * np.sin(2 * np.pi * f * t) + 0.05 * rng.standard_normal(shape)
    return x.astype(np.float32)


@pytest.mark.parametrize("n,nblocks", [(1024, 1), (1024, 70), (512, 3), (512, 300), (4096, 1), (4096, 40)])
def test_window_fft_log_512_1024_4096(oracle, cuda, n, nblocks):
    """block sizes of the 22/16/11/8 kHz modes (1024 over 512, plain 512) and of q < 0 (4096 over 512)"""
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(n * 3 + nblocks)
    x = signal(rng, (nblocks, n))
    lk = v.MdctLookup(n, short_n=512)
    assert np.array_equal(bits(lk.fft_twiddles), bits(orc.fft_twiddles(oracle, n)))
    wl, ws = window_table(max(n, 1024)), window_table(512)
    if n >= 1024:
        flags = rng.integers(0, 4, nblocks).astype(np.uint8)
        w = np.stack([oracle.apply_window(x[i], wl if flags[i] & 1 else ws, wl if flags[i] & 2 else ws)
                      for i in range(nblocks)])
        tf = torch.from_numpy(flags).to(cuda)
    else:
        w = oracle.apply_window(x, ws, ws)
        tf = None
    ref_log, ref_amp = orc.fft_logpower(oracle, w)
    got_log, got_amp = v.window_fft_log(lk, torch.from_numpy(x).to(cuda), tf)
    assert np.array_equal(bits(got_log.cpu().numpy()), bits(ref_log))
    assert np.array_equal(bits(got_amp.cpu().numpy()), bits(ref_amp))


@pytest.mark.parametrize("n,nblocks", [(2048, 1), (2048, 5), (2048, 300), (256, 1), (256, 9), (256, 700)])
def test_window_fft_log_bit_exact(oracle, cuda, n, nblocks):
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(n * 7 + nblocks)
    x = signal(rng, (nblocks, n))
    x[0, :] *= 1e-4          # a quiet block: exercises very negative dB values
    lk = v.MdctLookup(n, short_n=256)
    assert np.array_equal(bits(lk.fft_twiddles), bits(orc.fft_twiddles(oracle, n)))
    wl, ws = window_table(2048), window_table(256)
    if n == 2048:
        flags = rng.integers(0, 4, nblocks).astype(np.uint8)
        w = np.stack([oracle.apply_window(x[i], wl if flags[i] & 1 else ws, wl if flags[i] & 2 else ws)
                      for i in range(nblocks)])
        tf = torch.from_numpy(flags).to(cuda)
    else:
        w = oracle.apply_window(x, ws, ws)
        tf = None
    ref_log, ref_amp = orc.fft_logpower(oracle, w)
    got_log, got_amp = v.window_fft_log(lk, torch.from_numpy(x).to(cuda), tf)
    assert np.array_equal(bits(got_log.cpu().numpy()), bits(ref_log))
    assert np.array_equal(bits(got_amp.cpu().numpy()), bits(ref_amp))


def test_fft_silence_and_full_scale(oracle, cuda):
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    x = np.zeros((3, 2048), np.float32)
    x[1] = 1.0
    x[2, ::2] = -1.0
    lk = v.MdctLookup(2048, short_n=256)
    wl = window_table(2048)
    ref_log, ref_amp = orc.fft_logpower(oracle, oracle.apply_window(x, wl, wl))
    got_log, got_amp = v.window_fft_log(lk, torch.from_numpy(x).to(cuda))
    assert np.array_equal(bits(got_log.cpu().numpy()), bits(ref_log))
    assert np.array_equal(bits(got_amp.cpu().numpy()), bits(ref_amp))
