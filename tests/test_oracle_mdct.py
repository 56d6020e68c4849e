"""Oracle MDCT / window against first principles (CPU only).

The reference ships no golden vectors for mdct_forward; the restatement is
pinned end-to-end by the packet goldens (tests/test_oracle_packets.py).  Here it is
checked against the transform's definition in float64 and for internal consistency."""
import numpy as np
import pytest


def mdct_definition(x):
    n = x.shape[-1]
    j = np.arange(n)[None, :]
    k = np.arange(n // 2)[:, None]
    basis = np.cos(2 * np.pi / n * (j + .5 + n / 4) * (k + .5))
    return (4.0 / n) * (basis @ x.astype(np.float64).T).T


@pytest.mark.parametrize("n", [64, 256, 512, 2048])
def test_oracle_mdct_matches_definition(oracle, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((4, n)).astype(np.float32)
    got = oracle.mdct_forward(x)
    ref = mdct_definition(x)
    # float32 butterflies: error grows ~ log2(n) * eps * |x|
    assert np.abs(got - ref).max() < 4e-7 * np.sqrt(n)


@pytest.mark.parametrize("n", [256, 2048])
def test_oracle_mdct_trig_table(oracle, n):
    T = oracle.mdct_trig(n)
    i = np.arange(n // 4)
    assert np.array_equal(T[0:n // 2:2], np.cos(np.pi / n * 4 * i).astype(np.float32))
    assert np.array_equal(T[n // 2 + 1:n:2], np.sin(np.pi / (2 * n) * (2 * i + 1)).astype(np.float32))


def test_oracle_mdct_linearity_and_zero(oracle):
    n = 2048
    rng = np.random.default_rng(7)
    x = rng.standard_normal(n).astype(np.float32)
    assert np.all(oracle.mdct_forward(np.zeros(n, np.float32)) == 0)
    # exact power-of-two scaling commutes with every float op in the network
    assert np.array_equal(oracle.mdct_forward(x * 4.0), oracle.mdct_forward(x) * 4.0)


def test_oracle_window_regions(oracle):
    from vorbis_aotuv_lancer_amd.tables import window_table
    wl, ws = window_table(2048), window_table(256)
    x = np.ones(2048, np.float32)
    # long block between a short and a long neighbour (lW=0, nW=1): lib/window.c:2145-2152
    y = oracle.apply_window(x, ws, wl)
    assert np.all(y[:448] == 0) and np.array_equal(y[448:576], ws) and np.all(y[576:1024] == 1)
    assert np.array_equal(y[1024:], wl[::-1])
    y = oracle.apply_window(x, wl, ws)
    assert np.all(y[1024:1472] == 1) and np.array_equal(y[1472:1600], ws[::-1]) and np.all(y[1600:] == 0)
