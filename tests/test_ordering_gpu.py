"""Ordering of the internal HIP streams, made adversarial on purpose.

The encode path runs on several internal HIP streams (front end, one per block type, front / back half of the big
batch, HIP graphs per workspace); events order them.  A missing edge normally shows once in many runs.  Here every
marked point of the host code (csrc/vbm_internal.h, enum vbm_delay_point) gets, one point per case, a kernel that
spins for milliseconds in front of the work behind it (vbm_debug_set_delay): whatever is not ordered behind the
delayed work runs ahead of it, deterministically.  Packets may not change: every stream against its twin and the
distinct signals against the oracle, for the host-built rounds, the device-built (graph) rounds and the two-stream
per-block form.  A second family of cases fills every scratch buffer with 0xFF between calls
(vbm_debug_poison_*): nothing a batch reads may be left over from the batch that used the workspace before."""
import numpy as np
import pytest
import torch

from tests import orc
from tests.gpuutil import TwinLedger
from tests.signals import burst_signal

pytestmark = pytest.mark.gpu

CH, RATE, Q = 2, 44100, 0.5
S, K = 2048, 8                  # 256 twins per signal: batches of 256 .. 2048 blocks (small and "big" ones, >= 1024)
DELAY_US = 3000                 # longer than any kernel of a batch of this size

POINTS = {"job_big": 0, "job_small": 1, "job_state": 2, "job_back": 3, "job_out": 4, "fe_fork": 5, "fe_shift": 6,
          "dev_big_front": 7, "dev_big_back": 8, "dev_small_front": 9, "dev_small_back": 10, "dev_plan": 11,
          "batch_front": 12, "batch_back": 13, "fe_write": 14, "dev_out": 15}


@pytest.fixture
def delay():
    import vorbis_aotuv_lancer_amd as v

    def set_points(*names):
        mask = 0
        for n in names:
            mask |= 1 << POINTS[n]
        assert v.lib.vbm_debug_set_delay(mask, DELAY_US if mask else 0) == 0
    yield set_points
    v.lib.vbm_debug_set_delay(0, 0)


_cache = {}


def signals(nchunks, period):
    key = (nchunks, period)
    if key not in _cache:
        _cache[key] = [burst_signal(CH, RATE, nchunks * 1024, seed=70 + k, period=period, level=1.0 if k % 3 else 0.05)
                       for k in range(K)]
    return _cache[key]


MANAGED = (144000, 128000, 112000)     # vorbis_encode_init(max, nominal, min): the bitrate manager's reservoirs are carried state


def make_setup(v, bitrate):
    return v.Setup(CH, RATE, bitrate=bitrate) if bitrate else v.Setup(CH, RATE, Q)


def oracle_want(oracle, nchunks, period, bitrate=None):
    key = ("want", nchunks, period, bitrate)
    if key not in _cache:
        osetup = orc.Setup(oracle, CH, RATE, None if bitrate else Q, bitrate=bitrate)
        want = []
        for sig in signals(nchunks, period):
            st = orc.Stream(osetup)
            oracle.lib.orc_stream_set_capture(st.v, 0)
            seq = []
            for c in range(nchunks):
                st.write(sig[:, c * 1024:(c + 1) * 1024])
                seq.extend(b["packet"] for b in st.blocks())
            st.finish()
            seq.extend(b["packet"] for b in st.blocks())
            st.close()
            want.append(seq)
        _cache[key] = want
    return _cache[key]


def drain_host(fe, led, label):
    while True:
        info, packets, nbytes = fe.encode_round()
        if len(info) == 0:
            return
        led.add_host_round(info, packets, nbytes, label)


def run_host_rounds(cuda, nchunks, period, poison=False, multi=False, bitrate=None):
    import vorbis_aotuv_lancer_amd as v
    base = torch.from_numpy(np.stack(signals(nchunks, period))).to(cuda)
    enc = v.Encoder(make_setup(v, bitrate), S)
    fe = v.FrontEnd(enc)
    led = TwinLedger(S, K, cuda)
    for c in range(nchunks):
        if poison:
            enc.debug_poison(-1, 0xFF)
            fe.debug_poison(0xFF)
        fe.write(base[:, :, c * 1024:(c + 1) * 1024].repeat(S // K, 1, 1).contiguous())
        if multi:
            while True:
                info, packets, nbytes, counts = fe.encode_rounds(min_rounds=64, max_rounds=4)
                if not counts:
                    break
                led.add_host_round(info.copy(), packets, nbytes, f"write {c}")
        else:
            drain_host(fe, led, f"write {c}")
    fe.finish()
    drain_host(fe, led, "end of stream")
    led.finish("host-built rounds")
    fe.close()
    enc.close()
    return led


def run_device_rounds(cuda, monkeypatch, nchunks, period, poison=False, lazy=2, bitrate=None):
    import vorbis_aotuv_lancer_amd as v
    monkeypatch.setenv("VBM_WORKSPACES", "4")
    base = torch.from_numpy(np.stack(signals(nchunks, period))).to(cuda)
    setup = make_setup(v, bitrate)
    enc = v.Encoder(setup, S, max_batch=v.lib.vbm_device_round_lanes(setup._h, S))
    fe = v.FrontEnd(enc)
    led = TwinLedger(S, K, cuda)
    consumer = torch.cuda.Stream(device=cuda)
    for c in range(nchunks):
        if poison:
            fe.join()
            enc.debug_poison(-1, 0xFF)
            fe.debug_poison(0xFF)
        fe.write(base[:, :, c * 1024:(c + 1) * 1024].repeat(S // K, 1, 1).contiguous())
        info, packets, nbytes, counts = fe.encode_rounds_device(nrounds=2, lazy=lazy)
        fe.join(consumer)
        with torch.cuda.stream(consumer):
            led.add_device_rounds(info, packets, nbytes, f"write {c}")
        consumer.synchronize()
    fe.device_stats()
    assert fe.refused_writes == 0
    torch.cuda.synchronize()
    drain_host(fe, led, "drain")
    fe.finish()
    drain_host(fe, led, "end of stream")
    led.finish("device-built rounds")
    fe.close()
    enc.close()
    return led


def check_oracle(led, oracle, nchunks, period, bitrate=None):
    want = oracle_want(oracle, nchunks, period, bitrate)
    for k in range(K):
        assert led.lead_packets(k) == want[k], f"signal {k}: packets differ from the oracle"
    assert led.modes[0] + led.modes[1] > 0 and led.modes[2] > 0 and led.modes[3] > 0, led.modes   # all block types ran


@pytest.mark.parametrize("point", ["none", "job_big", "job_small", "job_state", "job_back", "job_out", "fe_fork", "fe_shift",
                                   "fe_write"])
def test_host_built_rounds_with_delays(oracle, cuda, delay, point):
    nchunks, period = 10, 6000
    if point != "none":
        delay(point)
    check_oracle(run_host_rounds(cuda, nchunks, period), oracle, nchunks, period)


def test_host_built_rounds_deferred_joins_with_delays(oracle, cuda, delay):
    """several rounds in flight (vbm_frontend_encode_rounds) with the small batches and every back half held up"""
    nchunks, period = 10, 6000
    delay("job_small", "job_back")
    check_oracle(run_host_rounds(cuda, nchunks, period, multi=True), oracle, nchunks, period)


@pytest.mark.parametrize("point", ["none", "dev_big_front", "dev_big_back", "dev_small_front", "dev_small_back", "dev_plan",
                                   "dev_out", "fe_shift", "fe_write"])
def test_device_built_rounds_with_delays(oracle, cuda, monkeypatch, delay, point):
    nchunks, period = 18, 12000
    if point != "none":
        delay(point)
    check_oracle(run_device_rounds(cuda, monkeypatch, nchunks, period), oracle, nchunks, period)


@pytest.mark.parametrize("point", ["none", "job_back", "job_small", "job_out", "job_state"])
def test_managed_host_built_rounds_with_delays(oracle, cuda, delay, point):
    """Managed bitrate: the front half and the fifteen packetblobs of a batch run beside the back halves of the batches
    before it; the bitrate manager's choice (reservoirs: state carried from block to block, lib/bitrate.c:98-226) is the
    step that waits for them.  A stream changes block type — and with it the internal HIP stream — at every burst; with
    the back half of a batch held up, a choice that did not wait would see reservoirs one block old.  (Tried and not kept: a
    case that switches the wait off and expects different packets — the reservoirs' sums do not depend on the order of the
    additions, so a mis-ordered choice only shows when it flips a choice, which a third of a second of audio does not do.)"""
    nchunks, period = 10, 6000
    if point != "none":
        delay(point)
    check_oracle(run_host_rounds(cuda, nchunks, period, bitrate=MANAGED), oracle, nchunks, period, MANAGED)


@pytest.mark.parametrize("point", ["none", "job_back", "job_big", "job_out"])
def test_managed_device_built_rounds_with_delays(oracle, cuda, monkeypatch, delay, point):
    nchunks, period = 12, 6000
    if point != "none":
        delay(point)
    check_oracle(run_device_rounds(cuda, monkeypatch, nchunks, period, bitrate=MANAGED), oracle, nchunks, period, MANAGED)


def test_host_built_rounds_poisoned_scratch(oracle, cuda):
    nchunks, period = 10, 6000
    check_oracle(run_host_rounds(cuda, nchunks, period, poison=True), oracle, nchunks, period)


def test_device_built_rounds_poisoned_scratch(oracle, cuda, monkeypatch):
    nchunks, period = 18, 12000
    check_oracle(run_device_rounds(cuda, monkeypatch, nchunks, period, poison=True), oracle, nchunks, period)


@pytest.mark.parametrize("point", ["batch_front", "batch_back"])
def test_two_stream_form_with_delays(cuda, delay, point):
    """vbm_analysis_batch2 (front half and back half on two streams, workspaces alternating): six calls, one half held
    up at every call, against the one-stream form without delays"""
    import vorbis_aotuv_lancer_amd as v
    g = torch.Generator(device=cuda).manual_seed(3)
    setup = v.Setup(CH, RATE, Q)
    ids = np.arange(S, dtype=np.int32)
    fl = np.full(S, 3, np.uint8)
    blocks = [(0.4 * (torch.rand((S, CH, 2048), generator=g, device=cuda) - 0.5)).contiguous() for _ in range(6)]
    res = []
    for two in (False, True):
        if two:
            delay(point)
        enc = v.Encoder(setup, S)
        back = torch.cuda.Stream(device=cuda) if two else None
        outs = [(torch.empty((S, enc.max_packet_bytes), dtype=torch.uint8, device=cuda),
                 torch.empty((S,), dtype=torch.int32, device=cuda)) for _ in range(6)]
        for k in range(6):
            enc.analysis_batch(3, ids, fl, blocks[k], back_stream=back, out=outs[k])
        torch.cuda.synchronize()
        res.append(outs)
        enc.close()
    for k in range(6):
        assert bool((res[0][k][1] == res[1][k][1]).all()) and bool((res[0][k][1] > 0).all())
        assert bool((res[0][k][0] == res[1][k][0]).all()), f"call {k}: packets differ with {point} held up"
