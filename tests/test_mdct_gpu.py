"""HIP window+MDCT vs the CPU oracle — bit-exact (integer compare of the float bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def signal(rng, shape):
    x = rng.standard_normal(shape).astype(np.float32) * 0.3
    x[..., ::7] += 0.5
    return x


@pytest.mark.parametrize("n", [4096, 2048, 1024, 512, 256])
@pytest.mark.parametrize("nblocks", [1, 3, 8, 9, 67, 1000])
def test_mdct_forward_bit_exact(oracle, cuda, n, nblocks):
    import vorbis_aotuv_lancer_amd as v
    rng = np.random.default_rng(n + nblocks)
    x = signal(rng, (nblocks, n))
    lk = v.MdctLookup(n, with_window=False)
    assert np.array_equal(bits(lk.trig), bits(oracle.mdct_trig(n)))
    got = v.mdct_forward(lk, torch.from_numpy(x).to(cuda)).cpu().numpy()
    ref = oracle.mdct_forward(x)
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("nblocks", [1, 5, 64, 513])
def test_window_mdct_long_all_neighbour_types(oracle, cuda, nblocks):
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(nblocks)
    x = signal(rng, (nblocks, 2048))
    x[0, :100] = -0.0  # zero regions must come out as +0 (lib/window.c:2248 stores 0.f)
    flags = rng.integers(0, 4, nblocks).astype(np.uint8)
    lk = v.MdctLookup(2048, short_n=256)
    got = v.window_mdct(lk, torch.from_numpy(x).to(cuda), torch.from_numpy(flags).to(cuda)).cpu().numpy()
    wl, ws = window_table(2048), window_table(256)
    ref = np.empty((nblocks, 1024), np.float32)
    for i in range(nblocks):
        w = oracle.apply_window(x[i], wl if flags[i] & 1 else ws, wl if flags[i] & 2 else ws)
        ref[i] = oracle.mdct_forward(w)
    assert np.array_equal(bits(got), bits(ref))
    # NULL flags == all-long neighbours
    got2 = v.window_mdct(lk, torch.from_numpy(x).to(cuda)).cpu().numpy()
    ref2 = oracle.mdct_forward(oracle.apply_window(x, wl, wl))
    assert np.array_equal(bits(got2), bits(ref2))


@pytest.mark.parametrize("nblocks", [1, 5, 130])
def test_window_mdct_long_1024_over_512(oracle, cuda, nblocks):
    """the 512/1024 block pair of the 22 kHz and 16 kHz modes (lib/modes/setup_22.h, setup_16.h)"""
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(50 + nblocks)
    x = signal(rng, (nblocks, 1024))
    flags = rng.integers(0, 4, nblocks).astype(np.uint8)
    lk = v.MdctLookup(1024, short_n=512)
    got = v.window_mdct(lk, torch.from_numpy(x).to(cuda), torch.from_numpy(flags).to(cuda)).cpu().numpy()
    wl, ws = window_table(1024), window_table(512)
    ref = np.stack([oracle.mdct_forward(oracle.apply_window(x[i], wl if flags[i] & 1 else ws, wl if flags[i] & 2 else ws))
                    for i in range(nblocks)])
    assert np.array_equal(bits(got), bits(ref))
    xs = signal(rng, (nblocks, 512))
    lks = v.MdctLookup(512, short_n=512)
    gots = v.window_mdct(lks, torch.from_numpy(xs).to(cuda)).cpu().numpy()
    assert np.array_equal(bits(gots), bits(oracle.mdct_forward(oracle.apply_window(xs, ws, ws))))


@pytest.mark.parametrize("nblocks", [1, 6, 90])
def test_window_mdct_long_4096_over_512(oracle, cuda, nblocks):
    """the 512/4096 block pair of q < 0 at 44.1/48 kHz (lib/modes/setup_44.h:40-45)"""
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(70 + nblocks)
    x = signal(rng, (nblocks, 4096))
    flags = rng.integers(0, 4, nblocks).astype(np.uint8)
    lk = v.MdctLookup(4096, short_n=512)
    got = v.window_mdct(lk, torch.from_numpy(x).to(cuda), torch.from_numpy(flags).to(cuda)).cpu().numpy()
    wl, ws = window_table(4096), window_table(512)
    ref = np.stack([oracle.mdct_forward(oracle.apply_window(x[i], wl if flags[i] & 1 else ws, wl if flags[i] & 2 else ws))
                    for i in range(nblocks)])
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("nblocks", [1, 7, 8, 200])
def test_window_mdct_short(oracle, cuda, nblocks):
    import vorbis_aotuv_lancer_amd as v
    from vorbis_aotuv_lancer_amd.tables import window_table
    rng = np.random.default_rng(100 + nblocks)
    x = signal(rng, (nblocks, 256))
    lk = v.MdctLookup(256, short_n=256)
    got = v.window_mdct(lk, torch.from_numpy(x).to(cuda)).cpu().numpy()
    ws = window_table(256)
    ref = oracle.mdct_forward(oracle.apply_window(x, ws, ws))
    assert np.array_equal(bits(got), bits(ref))


def test_mdct_edge_values(oracle, cuda):
    """denormals, zeros, large magnitudes survive identically (no flush-to-zero)."""
    import vorbis_aotuv_lancer_amd as v
    x = np.zeros((4, 2048), np.float32)
    x[1] = 1e-41  # subnormal
    x[2, ::2] = 1e30  # large but finite everywhere in the network (inf-inf NaN payloads are
    x[3] = np.float32(-0.0)  # platform-defined and not part of the parity contract)
    lk = v.MdctLookup(2048, with_window=False)
    got = v.mdct_forward(lk, torch.from_numpy(x).to(cuda)).cpu().numpy()
    ref = oracle.mdct_forward(x)
    for row, name in enumerate(["zeros", "subnormal", "large", "negative zero"]):
        assert np.array_equal(bits(got[row]), bits(ref[row])), name


def test_mdct_full_size_properties(oracle, cuda):
    """BASELINE config 2 size: 4096 streams x 2 ch x 8 long blocks.  Exact linearity under
    power-of-two scaling, batch-position independence, and a random sample vs the oracle."""
    import vorbis_aotuv_lancer_amd as v
    nb = 4096 * 2 * 8
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.rand((nb, 2048), generator=g) - 0.5).to(cuda)
    lk = v.MdctLookup(2048, short_n=256)
    y = v.window_mdct(lk, x)
    y4 = v.window_mdct(lk, x * 4.0)
    assert torch.equal(y4, y * 4.0)
    perm = torch.randperm(nb, generator=g).to(cuda)
    yp = v.window_mdct(lk, x[perm].contiguous())
    assert torch.equal(yp, y[perm])
    from vorbis_aotuv_lancer_amd.tables import window_table
    wl = window_table(2048)
    idx = np.random.default_rng(3).integers(0, nb, 64)
    ref = oracle.mdct_forward(oracle.apply_window(x[idx].cpu().numpy(), wl, wl))
    assert np.array_equal(bits(y[idx].cpu().numpy()), bits(ref))
