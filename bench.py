#!/usr/bin/env python3
"""Headline benchmark of the MI355X batched Vorbis (aoTuV) encode path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[2]: "1 GPU: full pipe incl. residue VQ on device, 16384 streams, q5 coupled
stereo" — 16384 concurrent 44.1 kHz stereo q5 streams per GPU, the largest single-GPU configuration (configs[1]
asks for "VQ on host", which this product does not do: nothing runs on the host).

`value` is measured FROM RAW PCM: one step = every stream receives 1024 new samples (23.2 ms of audio, resident
in HBM) through the device front end — PCM intake, envelope search, block switching, block carve-out
(vbm_frontend_*) — and the blocks that come out go through the whole per-block path (window, MDCT, FFT, psy,
floor fit/encode, couple/quantise, residue VQ, packet assembly).  The signal is SURVEY.md §8(d)'s: two sines +
noise + a 200-sample burst every ~1.33 s at a per-stream phase, so ~15 % of the blocks are short ones.
value = audio seconds WRITTEN inside the timed region / wall seconds = concurrent streams encodable at 1x realtime — or
the audio seconds of the blocks that came out if that is less (streams falling behind their input would otherwise
count as served); the ratio of the two is reported.  Before the W warm-up steps the streams are started up (PRIME
writes, the first ten with four rounds each, part of the set-up like loading a model: every stream begins with short
blocks, which would otherwise leave a backlog of rounds for the timed region to drain, and one burst period has to pass
before as many streams are catching up after a burst as in the steady state).  The per-block path alone (pre-cut long blocks, no front end: the
§8(a) measurement) is timed in the same run and reported as `per_block_path`.

Streams shard across ranks with no data-path collective: weak scaling, every rank encodes its own
16384 streams.  The control path (barrier, max-over-ranks time, work sum) is vorbis_aotuv_lancer_amd/shard.py over
a gloo group: no RCCL anywhere.  --dry-run exercises that control path without a device.
"""
import argparse
import json
import os
import sys
import time

# the block types of a blockout round run on internal HIP streams next to the caller's; the ROCm runtime folds
# HIP streams onto 4 hardware queues by default, which makes two of them share one
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# lazy joins (the back half of a write's long-block batch beside the next write) need more than two workspaces
os.environ.setdefault("VBM_WORKSPACES", "4")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAZY_JOIN = os.environ.get("VBM_BENCH_LAZY_JOIN", "1") != "0"   # from PCM: vbm_frontend_encode_rounds_lazy
STREAMS_PER_GPU = int(os.environ.get("VBM_BENCH_STREAMS", "16384"))
CHANNELS = 2
RATE = 44100
QUALITY = 0.5
N_LONG, N_SHORT = 2048, 256
HOP = N_LONG // 2
TWO_STREAMS = os.environ.get("VBM_BENCH_TWO_STREAMS", "1") != "0"   # per-block leg: vbm_analysis_batch2
MIN_ROUNDS = int(os.environ.get("VBM_BENCH_MIN_ROUNDS", "1"))   # from PCM: blockout rounds per write before the
                                                                 # buffers decide (more while one is past half full)
DISTINCT_STEPS = 8              # per-block leg: PCM for this many consecutive blocks per stream is kept in HBM
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# algorithmic HBM bytes per long channel-block and stage
n = N_LONG // 2
STAGE_BYTES = {
    "window_mdct": 4 * N_LONG + 4 * n,              # 12 288: read block, write spectrum
    "window_fft_log": 4 * N_LONG + 4 * n + 4,       # read block, write logfft + ampmax
    "transpose": 2 * (4 * n + 4 * n),               # mdct and logfft, read + write
    "prologue": 16,
    "noisemask": 4 * n * 2 + 4 * n * 3 + 4 * 32,    # mdct, lastmdct in; logmdct, noise, epeak, npeak out
    "tonemask": 4 * n + 4 * n,                      # logfft in, tone out
    "offset_and_mix": 4 * n * 4 + 4 * n * 3,        # noise, tone, logmdct, mdct in; logmask, mdct, lastmdct out
    "floor_fit": 4 * 905 * 2 + 4 * 29,              # first 905 bins of logmdct + logmask in, posts out
    "floor_encode": 4 * 29 * 2 + 4 * n,             # posts in, deviations + ilogmask out
    "couple_quantize": 4 * 928 * 3 + 4 * 928,       # mdct, epeak, ilogmask in, residue out (to lowpass)
    "pack": 4 * 912 + 240,                          # residue in, ~240 B packet out per channel
    "packet_out": 2 * 240,
}


# kernels behind every stage name (csrc/capi_encoder.cpp: kStageNames), as rocprofv3 lists them
STAGE_KERNELS = {
    "window_mdct": ["k_window_mdct<2048>"], "window_fft_log": ["k_window_fft_log<2048>"], "prologue": ["k_prologue"],
    "noisemask": ["k_noisemask<2,4>"], "tonemask": ["k_tonemask<8>"], "offset_and_mix": ["k_mix<1,false,true,false>", "k_block_state"],
    "floor_fit": ["k_floor_fit"], "floor_encode": ["k_floor_encode", "k_floor_render"],
    "couple_quantize": ["k_couple_m6stats", "k_couple_fast<1>"],
    "pack": ["k_zero_u128", "k_pack_head", "k_nonzero_propagate", "k_res_vq", "k_res_offsets", "k_res_emit"],
    "packet_out": ["k_from_tiled<int>"],
}


def stream_params(torch, dev, lo, hi):
    """per-stream signal parameters, a function of the GLOBAL stream index only (a stream sounds the same whichever
    rank encodes it)"""
    g = torch.Generator(device="cpu").manual_seed(0x9E3779B9)
    total = hi                                        # draw for [0, hi), keep [lo, hi): index-stable
    f1 = (110.0 + 1650.0 * torch.rand(total, generator=g))[lo:hi]
    f2 = (2000.0 + 4000.0 * torch.rand(total, generator=g))[lo:hi]
    # burst phase uniform over the whole 4/3-s period: any window of the run sees the same ~15 % short blocks
    phase = torch.randint(0, 4 * (RATE // 3), (total,), generator=g)[lo:hi]
    return (f1.view(-1, 1, 1).to(dev), f2.view(-1, 1, 1).to(dev), phase.view(-1, 1, 1).to(dev))


def synth_pcm(torch, dev, params, gen, first, count):
    """samples [first, first + count) of every stream of this rank: SURVEY.md §8(d) signal, [S][ch][count] float32"""
    f1, f2, phase = params
    S = f1.shape[0]
    pos = torch.arange(first, first + count, device=dev, dtype=torch.int64)
    t = pos.to(torch.float64) / RATE
    chan = torch.arange(1, CHANNELS + 1, device=dev, dtype=torch.float32).view(1, CHANNELS, 1)
    a1 = (2 * np.pi * f1.double() * chan.double() * t).remainder(2 * np.pi).float()
    a2 = (2 * np.pi * f2.double() * t).remainder(2 * np.pi).float() + chan
    x = 0.3 * torch.sin(a1) + 0.2 * torch.sin(a2)
    x += 0.05 * (2 * torch.rand((S, CHANNELS, count), generator=gen, device=dev) - 1)
    third = RATE // 3
    p = pos.view(1, 1, -1) + phase
    burst = ((p // third) % 4 == 3) & ((p % third) < 200)
    x += torch.where(burst, 0.6 * (2 * torch.rand((S, CHANNELS, count), generator=gen, device=dev) - 1),
                     torch.zeros((), device=dev))
    return x.contiguous()


def cpu_baseline(seconds_of_audio=600):
    """The oracle's full encoder (CPU restatement, bit-identical to the reference's scalar build) on
    the host cores of this box, one stereo q5 stream of the survey probe signal per thread
    (the reference itself cannot travel to the GPU box)."""
    import subprocess
    import threading
    so = os.path.join(ROOT, "oracle", "build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    from tests import orc
    o = orc.Oracle(so)
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(cores, int(os.environ.get("VBM_BENCH_CPU_THREADS", "16"))))
    res = [None] * threads

    def run(i):
        st = orc.Setup(o, CHANNELS, RATE, QUALITY)          # own setup + stream state per thread
        res[i] = st.encode_probe(seconds_of_audio)          # one C call; ctypes drops the GIL

    t0 = time.time()
    th = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.time() - t0
    per_core = [seconds_of_audio / r[1] for r in res]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": threads * seconds_of_audio / wall,
            "unit": f"x realtime on {threads} host cores (= streams at 1x; one stream per thread); oracle port, "
                    "not the reference build",
            "cores": threads, "kind": "port",
            "per_core": sum(per_core) / threads, "cpu_model": model, "host_cpus_available": cores,
            "restatement_vs_reference": {"scalar": 0.65, "lancer_sse2": 0.57,
                                         "note": "authoring container, 1 core: oracle 50-59x realtime vs the "
                                                 "survey's reference builds at 81x (scalar) / 93x (SSE2), BASELINE.md 2; "
                                                 "divide the figures here by these to read them as reference-equivalents"},
            "sample": f"{threads} x {seconds_of_audio} s of 44.1 kHz stereo q5 (survey probe signal, {res[0][0]} packets "
                      f"each, block switching + envelope search included), oracle/ scalar C, {wall:.1f} s wall"}


def compat_path():
    """The DROP-IN path (include/vorbis_compat.h): examples/compat_bench drives 16384 streams through the reference's own
    entry points — vorbis_analysis_buffer / _wrote / _blockout, vorbis_analysis, vorbis_bitrate_addblock / _flushpacket, the loop
    of the reference's examples/encoder_example.c:179-236 — from 4 host threads that own one device pool of 4096 streams
    each; host PCM in, packets back on the host (H2D / D2H inside).  Two delivery modes: the reference's (every block as
    soon as the stream has it) and VORBIS_MI355X_DEFER_BLOCKS (same packets, a lagging stream's extra blocks come later)."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "compat_bench")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "compat_bench"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
        out = {}
        threads, per, writes = 4, STREAMS_PER_GPU // 4, 32
        for name, defer in (("reference_delivery", "0"), ("deferred_delivery", "1")):
            env = dict(os.environ, VORBIS_MI355X_DEFER_BLOCKS=defer)
            r = subprocess.run([exe, str(threads), str(per), str(per), str(writes), "8"], capture_output=True, text=True,
                               timeout=120, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode or not line:
                return {"error": f"compat_bench failed (rc {r.returncode}): {r.stderr[-200:]}"}
            d = json.loads(line[-1])
            out[name] = {"value": d["value"], "wall_s": d["wall_s"], "device_rounds": d["device_rounds_total"],
                         "encoded_over_input": d["encoded_audio_s"] / d["input_audio_s"]}
        best = max(out, key=lambda k: out[k]["value"])
        return {"value": out[best]["value"], "mode": best, "unit": "x realtime (streams at 1x) through vorbis_analysis_* / "
                "vorbis_bitrate_*, host PCM in, host packets out", "streams": threads * per, "threads": threads,
                "pool_streams": per, "writes_timed": writes, **out}
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": str(e)}


def roof_of(stage, ms_per_launch, channel_blocks, traffic):
    alg = STAGE_BYTES[stage] * channel_blocks
    ach = alg / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
    return {"kernel": stage, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "traffic": traffic.get(stage),
            "algorithmic_bytes_per_launch": alg, "channel_blocks_per_launch": channel_blocks, "kernel_ms": ms_per_launch}


def load_traffic(ncb):
    """PMC-measured HBM bytes per launch and stage (tools/gpu_profile.sh -> profiles/pmc_traffic.json): a committed
    measurement of an earlier run, tagged with the commit it was taken at — never mixed in silently."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return {}, None
    tj = json.load(open(tpath))
    if tj.get("channel_blocks_per_step") != ncb:
        return {}, None
    src = {"file": "profiles/pmc_traffic.json", "measured_at": tj.get("commit"),
           "mode": "per launch of the step's full-size batch (PMC passes over the per-block leg: every launch there is one such batch)"}
    if tj.get("from_pcm"):
        src["from_pcm_total_bytes_per_write"] = tj["from_pcm"].get("total_bytes_per_write")
    if tj.get("valu_instructions_per_write_from_pcm"):
        src["valu_instructions_per_write_from_pcm"] = tj["valu_instructions_per_write_from_pcm"]
    return tj.get("hbm_bytes_per_launch", {}), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-pcm", action="store_true", help="(default) value = whole encoder from raw PCM")
    ap.add_argument("--per-block", action="store_true",
                    help="value = the per-block path alone on pre-cut long blocks (the §8(a) measurement; no front "
                         "end, no block switching) — for A/B work on the kernels")
    ap.add_argument("--only", choices=["pcm", "block", "solo"], default=None, help="run one leg only")
    ap.add_argument("--bitrate", type=int, default=0,
                    help="managed-bitrate setup of this nominal rate (vorbis_encode_init, SURVEY 8f N2: all 15 "
                         "packetblobs per block) instead of the q5 VBR setup of the headline metric")
    ap.add_argument("--dry-run", action="store_true",
                    help="no device: every rank 'encodes' its streams in VBM_BENCH_DRYRUN_MS milliseconds per step "
                         "(comma list, one entry per rank) — exercises sharding, barrier, max-over-ranks and the "
                         "aggregate on the gloo control group")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")

    # control group first, before anything touches the GPU: gloo over TCP on 127.0.0.1, no RCCL
    from vorbis_aotuv_lancer_amd import shard
    shard.init_control(rank, world)
    lo, hi = shard.stream_range(STREAMS_PER_GPU * world, rank, world)     # this rank's global stream ids

    if args.dry_run:
        ms = [float(x) for x in os.environ.get("VBM_BENCH_DRYRUN_MS", "10").split(",")]
        my_ms = ms[rank % len(ms)]
        shard.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(my_ms * 1e-3)
        shard.barrier()
        dt_local = time.perf_counter() - t0
        audio_local = (hi - lo) * HOP / RATE * args.steps
        dt = shard.max_over_ranks(dt_local)
        value = shard.aggregate_throughput(audio_local, dt_local)
        ranges = shard.gather_objects((lo, hi))
        audio_all = shard.sum_over_ranks(audio_local)          # collectives: every rank calls them
        if rank == 0:
            print(json.dumps({"metric": "realtime-stream-equivalents/node (44.1kHz stereo q5) + MDCT HBM GB/s",
                              "value": value, "unit": "x realtime (dry run: no device work)", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                              "data": "none (dry run)",
                              "config": {"workload": "dry run of the control path", "stream_ranges": ranges,
                                         "audio_s_all_ranks": audio_all,
                                         "control_backend": shard.backend()}}), flush=True)
        shard.finish()
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the encode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import vorbis_aotuv_lancer_amd as v

    setup = v.Setup(CHANNELS, RATE, bitrate=args.bitrate) if args.bitrate else v.Setup(CHANNELS, RATE, QUALITY)
    S = hi - lo
    enc = v.Encoder(setup, S, max_batch=max(S, v.lib.vbm_device_round_lanes(setup._h, S)))
    params = stream_params(torch, dev, lo, hi)
    ncb = S * CHANNELS
    traffic, traffic_src = load_traffic(ncb)
    # (the per-block leg first: the front end adds HIP streams; the solo leg last: its encoder's internal streams would
    # change which hardware queues the streams of the timed legs land on — measured: per-block step 3.7 ms instead of 2.8)
    legs = [args.only] if args.only else ["block", "pcm", "solo"]

    def barrier():
        shard.barrier()
        torch.cuda.synchronize()

    def timed(step, after=None):
        for k in range(args.warmup):
            step(k)
        if after:
            after()
        barrier()
        enc.profile_begin(args.steps)   # HIP events between the stage kernels, on the stream each is launched on
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k)
        if after:
            after()
        barrier()
        dt_local = time.perf_counter() - t0
        stage_ms, calls = enc.profile_end()
        return dt_local, stage_ms, calls, enc.profile_blocks

    results = {}

    # ---- leg "pcm": the whole encoder from raw PCM (the headline value) -----------------------------------------
    def leg_pcm():
        enc.reset()
        fe = v.FrontEnd(enc)
        gen = torch.Generator(device=dev).manual_seed(99 + lo)
        chunks = []
        stat = {"rounds": 0, "blocks": 0, "samples": 0, "modes": np.zeros(4, np.int64), "mean_bytes": 0.0}
        kept = []
        counting = [False]

        DEVICE_ROUNDS = os.environ.get("VBM_BENCH_DEVICE_ROUNDS", "1") != "0"
        CONSUMER = os.environ.get("VBM_BENCH_CONSUMER", "1") != "0"
        consumer = torch.cuda.Stream(device=dev) if CONSUMER else None
        # rounds per write, cycled: a stream needs ~1.2 blocks per 1024 samples at this signal's block-switching rate
        # (55 long + 10 short + 2 transition blocks per 57 writes), a stream inside a burst eight; streams that fell
        # behind catch up one extra block per round.  "2,1,1,1" (1.25 per write) keeps every stream's buffer bounded over
        # hundreds of writes now that the bursts of the streams are spread evenly over the burst period (320 writes:
        # encoded / input 0.9985, max_buffered_samples_at_end 16192, as after 96 writes; "2,1,1": 15168, 3 % slower); "1"
        # does not (the buffers grow until a write is refused, which aborts the run).
        PATTERN = [int(x) for x in os.environ.get("VBM_BENCH_ROUNDS", "2,1,1,1").split(",")]

        def step_pcm(k):
            fe.write(chunks[k])
            if DEVICE_ROUNDS:
                # rounds built on the device: the call only enqueues (no decision ever comes back to the host)
                nr = PATTERN[k % len(PATTERN)]
                if counting[0]:
                    stat["rounds"] += nr
                # lazy=2: the stream that feeds PCM never waits for packets; a consumer's stream is tied to the outputs of
                # every call instead (what a writer of Ogg pages would wait on).  VBM_BENCH_CONSUMER=0: the feeding
                # stream itself waits (lazy join: all but the newest long-block batch and the last round).
                if CONSUMER:
                    kept.append(fe.encode_rounds_device(nrounds=nr, lazy=2, device=dev))
                    fe.join(consumer)
                else:
                    kept.append(fe.encode_rounds_device(nrounds=nr, lazy=LAZY_JOIN, device=dev))
                del kept[:-6]
                return
            info, pk_, nb_, counts = fe.encode_rounds(min_rounds=MIN_ROUNDS, max_rounds=16, headroom=HOP, device=dev,
                                                      lazy=LAZY_JOIN)
            if counting[0] and len(info):
                stat["rounds"] += len(counts)
                stat["blocks"] += len(info)
                # what a block advances its stream by: the distance between block centres (lib/block.c:745-759)
                stat["samples"] += int((np.where(info["W"] != 0, N_LONG, N_SHORT) // 4
                                        + np.where(info["nW"] != 0, N_LONG, N_SHORT) // 4).sum())
                stat["modes"] += np.bincount(info["block_mode"], minlength=4)
            kept.append((info, pk_, nb_))     # lazy: the outputs of a call are complete after the next one
            del kept[:-3]

        base_stats = [None]

        def count_from_now():
            fe.join()
            counting[0] = True
            if DEVICE_ROUNDS:
                base_stats[0] = fe.device_stats()

        # Stream start-up (set-up, not warm-up): every stream begins with two or three short blocks and a transition
        # block, all streams at the same moment, which overflows the short types' lane regions for a dozen rounds.
        # PRIME writes with four rounds each clear that, so that warm-up and timed region see the steady state.
        # The rest of the PRIME writes run at the steady pattern until one whole burst period (57 writes) has passed: the
        # share of streams that are catching up after a burst has then reached its steady state (before that, fewer blocks
        # come out than go in).
        PRIME = int(os.environ.get("VBM_BENCH_PRIME", "64"))
        gen_prime = torch.Generator(device=dev).manual_seed(7 + lo)
        for k in range(PRIME):
            fe.write(synth_pcm(torch, dev, params, gen_prime, k * HOP, HOP))
            if DEVICE_ROUNDS:
                kept.append(fe.encode_rounds_device(nrounds=4 if k < 10 else PATTERN[k % len(PATTERN)],
                                                    lazy=2 if CONSUMER else LAZY_JOIN, device=dev))
                if CONSUMER:
                    fe.join(consumer)
                del kept[:-6]
            else:
                fe.encode_rounds(min_rounds=4, max_rounds=16, headroom=HOP, device=dev)
        fe.join()
        torch.cuda.synchronize()
        # warmup is not counted; the timed region starts with a join so that nothing of it is left pending
        PROF_STEPS = 8
        nsteps_total = args.warmup + args.steps + PROF_STEPS
        chunks[:] = [synth_pcm(torch, dev, params, gen, (PRIME + k) * HOP, HOP) for k in range(nsteps_total)]   # resident in HBM
        for k in range(args.warmup):
            step_pcm(k)
        count_from_now()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step_pcm(args.warmup + k)
        t_enq = time.perf_counter() - t0         # the host's share: every call above only enqueues
        fe.join()
        barrier()
        dt_local = time.perf_counter() - t0
        stat["host_enqueue_ms_per_step"] = 1e3 * t_enq / max(args.steps, 1)
        end_stats = fe.device_stats() if DEVICE_ROUNDS else None
        counting[0] = False
        # stage times: a few more steps with HIP events between the stage kernels (outside the timed region: the
        # device-built rounds run as HIP graphs, which have no room for events; with timing on they are launched
        # kernel by kernel)
        if os.environ.get("VBM_BENCH_NO_STAGE_PASS"):      # kernel traces of the timed region alone (tools/gpu_pcm_trace.sh)
            PROF_STEPS = 0
        blocks_prof, stage_ms, calls = 0, {}, 0
        if PROF_STEPS:
            enc.profile_begin(PROF_STEPS * 4)
            for k in range(PROF_STEPS):
                step_pcm(args.warmup + args.steps + k)
            fe.join()
            torch.cuda.synchronize()
            blocks_prof = int(v.lib.vbm_encoder_profile_blocks(enc._h))
            stage_ms, calls = enc.profile_end()
        if DEVICE_ROUNDS:
            modes1, samples1 = end_stats               # totals kept on the device: (blocks per type, samples advanced)
            modes0, samples0 = base_stats[0]
            stat["modes"] = np.array([a - b for a, b in zip(modes1, modes0)], np.int64)
            stat["blocks"] = int(stat["modes"].sum())
            stat["samples"] = samples1 - samples0
            nb_last = kept[-1][2]
            live = nb_last[nb_last >= 0]
            stat["mean_bytes"] = float(live.float().mean().item()) if len(live) else 0.0
            stat["max_buffered_end"] = fe.max_buffered
            stat["refused_writes"] = fe.refused_writes
            if fe.refused_writes:
                raise SystemExit(f"invalid run: {fe.refused_writes} writes were refused (stream buffers full): more rounds per write needed")
        else:
            nb_last = kept[-1][2]
            stat["mean_bytes"] = float(nb_last.float().mean().item()) if len(nb_last) else 0.0
        encoded_s, input_s = stat["samples"] / RATE, S * HOP / RATE * args.steps
        results["pcm"] = dict(dt_local=dt_local, stage_ms=stage_ms, calls=calls, blocks_prof=blocks_prof, stat=stat,
                              audio_local=min(encoded_s, input_s), encoded_s=encoded_s, input_s=input_s, prof_steps=PROF_STEPS)
        fe.close()
        del chunks, kept

    # ---- leg "block": the per-block path alone on pre-cut long blocks (§8a) ---------------------------------------
    def leg_block():
        enc.reset()
        gen = torch.Generator(device=dev).manual_seed(1234 + lo)
        x = synth_pcm(torch, dev, params, gen, 0, (DISTINCT_STEPS + 1) * HOP)
        blocks = [x[:, :, k * HOP:k * HOP + N_LONG].contiguous() for k in range(DISTINCT_STEPS)]
        del x
        ids = np.arange(S, dtype=np.int32)
        wflags = np.full(S, 3, dtype=np.uint8)   # lW = nW = long
        back_q = torch.cuda.Stream(device=dev)
        outs = [(torch.empty((S, enc.max_packet_bytes), dtype=torch.uint8, device=dev),
                 torch.empty((S,), dtype=torch.int32, device=dev)) for _ in range(2)]
        last = [None]

        def step_block(k):
            if TWO_STREAMS:   # back half of this step beside the front half of the next (vbm_analysis_batch2)
                last[0] = enc.analysis_batch(3, ids, wflags, blocks[k % DISTINCT_STEPS], back_stream=back_q, out=outs[k & 1])
            else:
                last[0] = enc.analysis_batch(3, ids, wflags, blocks[k % DISTINCT_STEPS])

        dt_local, stage_ms, calls, blocks_prof = timed(step_block)
        # The MDCT kernel on its own (same launch shape as in the pipeline: one long block of every channel), timed
        # with HIP events inside the library on this stream: the kernel's own HBM rate.
        import ctypes as C
        lk = v.MdctLookup(N_LONG, short_n=N_SHORT)
        xb = blocks[0].reshape(ncb, N_LONG)
        y = torch.empty((ncb, N_LONG // 2), device=dev)
        ms = C.c_float()
        st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        v.check(v.lib.vbm_window_mdct_time(lk._h, xb.data_ptr(), y.data_ptr(), None, ncb, 3, st_, C.byref(ms)))
        v.check(v.lib.vbm_window_mdct_time(lk._h, xb.data_ptr(), y.data_ptr(), None, ncb, 20, st_, C.byref(ms)))
        results["block"] = dict(dt_local=dt_local, stage_ms=stage_ms, calls=calls, blocks_prof=blocks_prof,
                                audio_local=S * HOP / RATE * args.steps, mdct_alone_ms=ms.value / 20,
                                mean_bytes=float(last[0][1].float().mean().item()))

    # ---- leg "solo": every stage of the per-block path ALONE (one HIP stream, the two mask branches one after the other:
    #      no kernel runs beside another), HIP events between the stages on that stream.  What `roofline` is made of;
    #      `rocprofv3 --kernel-trace --stats -- python bench.py --only solo` shows the same launches kernel by kernel
    #      (profiles/rNN/solo_kernel_stats.csv).
    def leg_solo():
        os.environ["VBM_OVERLAP_BRANCHES"] = "0"
        enc1 = v.Encoder(setup, S)
        del os.environ["VBM_OVERLAP_BRANCHES"]
        gen = torch.Generator(device=dev).manual_seed(1234 + lo)
        x = synth_pcm(torch, dev, params, gen, 0, (DISTINCT_STEPS + 1) * HOP)
        blocks = [x[:, :, k * HOP:k * HOP + N_LONG].contiguous() for k in range(DISTINCT_STEPS)]
        del x
        ids = np.arange(S, dtype=np.int32)
        wflags = np.full(S, 3, dtype=np.uint8)
        nsolo = int(os.environ.get("VBM_BENCH_SOLO_STEPS", "12"))
        for k in range(3):
            enc1.analysis_batch(3, ids, wflags, blocks[k % DISTINCT_STEPS])
        torch.cuda.synchronize()
        enc1.profile_begin(nsolo)
        for k in range(nsolo):
            enc1.analysis_batch(3, ids, wflags, blocks[(3 + k) % DISTINCT_STEPS])
        torch.cuda.synchronize()
        stage_ms, calls = enc1.profile_end()
        results["solo"] = dict(stage_ms={k: ms_ / max(calls, 1) for k, ms_ in stage_ms.items()}, calls=calls)
        enc1.close()

    for leg in legs:
        {"pcm": leg_pcm, "block": leg_block, "solo": leg_solo}[leg]()

    # ---- reduce over ranks (gloo): max of the time, sum of the work ------------------------------------------------
    for name, r in results.items():
        if name == "solo":
            continue
        r["dt"] = shard.max_over_ranks(r["dt_local"])
        r["value"] = shard.aggregate_throughput(r["audio_local"], r["dt_local"])

    if rank == 0:
        def stage_view(r):
            calls = max(r["calls"], 1)
            per_step = {k: ms_ / max(r.get("prof_steps", args.steps), 1) for k, ms_ in r["stage_ms"].items()}
            per_launch = {k: ms_ / calls for k, ms_ in r["stage_ms"].items()}
            cb = r["blocks_prof"] * CHANNELS // calls if r["blocks_prof"] else ncb   # channel-blocks per timed launch
            dominant = max(per_step, key=per_step.get) if per_step else None
            return per_step, per_launch, cb, dominant

        if set(results) == {"solo"}:        # (profiling runs: rocprofv3 over the solo launches only)
            print(json.dumps({"metric": "solo stage times of the per-block path", "stage_solo_ms": results["solo"]["stage_ms"],
                              "channel_blocks_per_launch": ncb, "launches_timed": results["solo"]["calls"]}), flush=True)
            shard.finish()
            return
        headline = "block" if (args.per_block or "pcm" not in results) else "pcm"
        R = results[headline]
        per_step, per_launch, cb, dominant = stage_view(R)
        if "solo" in results:
            # the dominant kernel = the stage that takes longest ALONE at the step's launch size (one launch per step);
            # its duration is the HIP-event time on its own stream with nothing beside it
            solo = results["solo"]["stage_ms"]
            dominant = max(solo, key=solo.get)
            roofline = roof_of(dominant, solo[dominant], ncb, traffic)
            roofline["kernel_ms_source"] = ("HIP events around the stage's launches on their stream, leg 'solo' of this run: "
                                            f"{results['solo']['calls']} launches of {ncb} long channel-blocks, one HIP stream, "
                                            "nothing beside them")
            roofline["kernels"] = STAGE_KERNELS.get(dominant, [])
            roofline["in_situ_ms_per_step"] = per_step.get(dominant)
            roofline["launches_per_step"] = 1
            if args.bitrate and dominant == "pack":
                # managed bitrate: the stage timer "pack" brackets the whole back half of a block — floor encode, couple /
                # quantise and packet assembly of all fifteen packetblobs (capi_encoder.cpp: managed_back) and the choice
                blobs = 15
                alg = blobs * (STAGE_BYTES["floor_encode"] + STAGE_BYTES["couple_quantize"] + STAGE_BYTES["pack"]) * ncb
                ach = alg / (solo[dominant] * 1e-3) / 1e9
                roofline.update({"kernel": "managed back half (floor encode + couple/quantise + pack, 15 packetblobs)",
                                 "algorithmic_bytes_per_launch": alg, "achieved": ach, "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                                 "kernels": STAGE_KERNELS["floor_encode"] + STAGE_KERNELS["couple_quantize"] + STAGE_KERNELS["pack"]
                                            + ["k_bitrate_choose", "k_blob_gather"],
                                 "note": "bound by instruction issue, not HBM: k_couple_fast (15 blobs in one launch) and k_res_vq "
                                         "are 9 of the step's 17.6 ms and run at 60-80 % of the vector issue rate (DESIGN.md 6a)"})
        elif dominant:
            roofline = roof_of(dominant, per_launch[dominant], cb, traffic)
        else:
            roofline = None
        if traffic_src and roofline:
            roofline["traffic_source"] = traffic_src
        line = {
            "metric": "realtime-stream-equivalents/node (44.1kHz stereo q5) + MDCT HBM GB/s",
            "value": R["value"],
            "unit": "x realtime (concurrent 44.1 kHz stereo q5 streams encodable at 1x)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": R["dt"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[2]: full pipe incl. residue VQ on device, {STREAMS_PER_GPU} streams/GPU x 44.1 kHz "
                            "coupled stereo q5, " +
                            ("from raw PCM: one 1024-sample write per stream per step through the device front end, "
                             "long + short blocks as the envelope detector decides; PCM resident in HBM, packets left "
                             "in HBM" if headline == "pcm" else
                             "per-block path alone: one pre-cut long block (2048) per stream per step"),
                "signal": "SURVEY 8(d): 0.3 sin(f1) + 0.2 sin(f2) + 0.05 noise, f1 in [110,1760] Hz, f2 in [2,6] kHz per "
                          "stream, + 0.6-amplitude 200-sample noise burst every 1.33 s at a per-stream phase",
                "streams_per_gpu": STREAMS_PER_GPU, "channels": CHANNELS,
                "from_pcm": headline == "pcm",
                "block_switching": ("in the timed region: PCM intake, envelope search, block switching and block carve-out "
                                    "run on the device (vbm_frontend_*)" if headline == "pcm" else
                                    "not in the timed region (pre-cut long blocks)"),
                "value_counts": "min(audio seconds written, audio seconds of the blocks that came out) inside the timed region, "
                                "all ranks / max-over-ranks wall",
                "pipeline_complete": True,
                "control_backend": shard.backend(),
                **({"managed_bitrate": args.bitrate,
                    "note": "NOT the headline setup: managed bitrate, 15 packetblobs per block (lib/mapping0.c:1204)"}
                   if args.bitrate else {}),
                "parallelism": f"stream-shard x{world} (no collective, control over gloo)",
            },
            "roofline": roofline,
            "stage_ms_per_step": per_step,
            "stage_ms_per_launch": per_launch,
        }
        if traffic_src and traffic_src.get("valu_instructions_per_write_from_pcm") and headline == "pcm":
            # What bounds the step is instruction issue, not HBM: vector instructions issued per write (PMC,
            # committed measurement) against what 1024 SIMDs issue per second at one instruction per four cycles
            n_valu = traffic_src["valu_instructions_per_write_from_pcm"]
            rate = 256 * 4 * 2.4e9 / 4
            line["issue_bound"] = {"valu_instructions_per_write": n_valu, "simd_issue_rate_per_s": rate,
                                   "bound_ms": n_valu / rate * 1e3, "frac_of_step": n_valu / rate * 1e3 / (R["dt"] / args.steps * 1e3),
                                   "hbm_bytes_per_write": traffic_src.get("from_pcm_total_bytes_per_write"),
                                   "hbm_frac_of_peak": (traffic_src.get("from_pcm_total_bytes_per_write") or 0) /
                                                       (R["dt"] / args.steps) / 1e9 / HBM_PEAK_GBPS,
                                   "source": "profiles/pmc_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE over "
                                             "`bench.py --only pcm`), measured at " + str(traffic_src.get("measured_at"))}
        if "solo" in results:
            line["stage_solo_ms"] = results["solo"]["stage_ms"]
            line["stage_solo_roofline_frac"] = {k: (STAGE_BYTES[k] * ncb / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms_ > 0 else None)
                                                for k, ms_ in results["solo"]["stage_ms"].items() if k in STAGE_BYTES}
        if "pcm" in results:
            st = results["pcm"]["stat"]
            line["config"].update({
                "blocks_encoded": st["blocks"], "rounds": st["rounds"], "rounds_per_write": st["rounds"] / args.steps,
                "host_enqueue_ms_per_step": st.get("host_enqueue_ms_per_step"),
                "rounds_pattern": os.environ.get("VBM_BENCH_ROUNDS", "2,1,1,1"),
                "outputs_joined_on": ("a consumer stream after every call (the feeding stream never waits for packets)"
                                      if os.environ.get("VBM_BENCH_CONSUMER", "1") != "0" else "the feeding stream (lazy join)"),
                "blocks_by_mode": {"impulse_short": int(st["modes"][0]), "padding_short": int(st["modes"][1]),
                                   "transition_long": int(st["modes"][2]), "long": int(st["modes"][3])},
                "short_block_fraction": float(st["modes"][:2].sum() / max(st["blocks"], 1)),
                "input_audio_s_per_rank": S * HOP / RATE * args.steps,
                "encoded_audio_s_rank0": results["pcm"]["encoded_s"],
                "encoded_over_input": results["pcm"]["encoded_s"] / results["pcm"]["input_s"],
                "stream_start_up": "VBM_BENCH_PRIME (64) writes before the warm-up, the first ten with four rounds each: set-up (one burst "
                                   "period of every stream has passed when the warm-up starts), not timed",
                "mean_packet_bytes_last_call": st["mean_bytes"],
                "rounds_built_on": "device (no host synchronisation inside the timed region)" if st.get("max_buffered_end") is not None
                                   else "host (one synchronisation per round)",
                "max_buffered_samples_at_end": st.get("max_buffered_end"),
            })
        if "block" in results:
            B = results["block"]
            bstep, blaunch, bcb, bdom = stage_view(B)
            line["per_block_path"] = {
                "ms_per_step": B["dt"] / args.steps * 1e3, "value": B["value"],
                "note": "per-block path alone (§8a): pre-cut long blocks, front end and block switching NOT in the "
                        "timed region; two HIP streams (vbm_analysis_batch2)" if TWO_STREAMS else "one HIP stream",
                "stage_ms_per_step": bstep, "dominant": roof_of(bdom, blaunch[bdom], bcb, traffic),
                "mean_packet_bytes": B["mean_bytes"],
            }
            r_mdct = roof_of("window_mdct", blaunch["window_mdct"], bcb, traffic)
            r_mdct["in_pipeline_ms"] = r_mdct["kernel_ms"]
            r_mdct["in_pipeline_frac"] = r_mdct["frac"]
            r_mdct["kernel_ms"] = B["mdct_alone_ms"]
            r_mdct["achieved"] = r_mdct["algorithmic_bytes_per_launch"] / (B["mdct_alone_ms"] * 1e-3) / 1e9
            r_mdct["frac"] = r_mdct["achieved"] / HBM_PEAK_GBPS
            r_mdct["note"] = "kernel alone, 20 launches of the step's blocks; in_pipeline_* = beside the previous step's back half"
            line["mdct_roofline"] = r_mdct
        if not args.no_cpu_baseline and world == 1:   # host baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.only and not args.bitrate and not os.environ.get("VBM_BENCH_NO_COMPAT"):
            # (outside the headline: what a user of the reference's own API gets; the bench's device objects are gone by now)
            enc.close()
            torch.cuda.synchronize()
            line["compat_path"] = compat_path()
        print(json.dumps(line), flush=True)
    shard.finish()


if __name__ == "__main__":
    main()
