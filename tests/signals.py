"""Deterministic synthetic PCM for the parity tests (numpy only)."""
import numpy as np


def synth_signal(ch, rate, nsamples, seed=0, level=1.0):
    """Two sines + noise + a short loud noise burst roughly every 1.3 s (forces short blocks),
    shaped like the survey probe signal but seeded per stream."""
    rng = np.random.default_rng(seed)
    t = np.arange(nsamples, dtype=np.float64) / rate
    f1 = rng.uniform(110.0, 1760.0)
    f2 = rng.uniform(2000.0, 6000.0)
    third = rate // 3
    phase = int(rng.integers(0, third))
    pos = np.arange(nsamples) + phase
    burst = ((pos // third) % 4 == 3) & ((pos % third) < 200)
    out = np.empty((ch, nsamples), np.float32)
    for c in range(ch):
        x = 0.3 * np.sin(2 * np.pi * f1 * (c + 1) * t) + 0.2 * np.sin(2 * np.pi * f2 * t + c)
        x += 0.05 * rng.uniform(-1, 1, nsamples)
        x += np.where(burst, 0.6 * rng.uniform(-1, 1, nsamples), 0.0)
        out[c] = (level * x).astype(np.float32)
    return out


def burst_signal(ch, rate, nsamples, seed=0, period=6000, burst=200, level=1.0):
    """like synth_signal, with a noise burst every `period` samples from a seeded phase on (block switching within a
    few writes, at different moments for different seeds)"""
    rng = np.random.default_rng(seed)
    t = np.arange(nsamples, dtype=np.float64) / rate
    f1 = rng.uniform(110.0, 1760.0)
    f2 = rng.uniform(2000.0, 6000.0)
    phase = int(rng.integers(0, period))
    pos = np.arange(nsamples) + phase
    on = (pos % period) < burst
    on &= np.arange(nsamples) >= 2500          # (the start of the stream switches by itself)
    out = np.empty((ch, nsamples), np.float32)
    for c in range(ch):
        x = 0.3 * np.sin(2 * np.pi * f1 * (c + 1) * t) + 0.2 * np.sin(2 * np.pi * f2 * t + c)
        x += 0.05 * rng.uniform(-1, 1, nsamples)
        x += np.where(on, 0.6 * rng.uniform(-1, 1, nsamples), 0.0)
        out[c] = (level * x).astype(np.float32)
    return out
