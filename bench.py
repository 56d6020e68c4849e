#!/usr/bin/env python3
"""Headline benchmark of the MI355X batched Vorbis (aoTuV) encode path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[2]: "1 GPU: full pipe incl. residue VQ on device, 16384 streams,
q5 coupled stereo" — 16384 concurrent 44.1 kHz stereo q5 streams per GPU, long blocks (2048
samples), synthetic PCM already resident in HBM, the WHOLE per-block path on the device (window,
MDCT, FFT, psy, floor fit/encode, couple/quantise, residue VQ, packet assembly).  configs[1]
("MDCT+psy on device, VQ on host") is not what this implementation does: nothing runs on the
host, so the configuration that matches the product is configs[2]; it fits one GPU (~6 GB).
VBM_BENCH_STREAMS=4096 reproduces configs[1]'s stream count.  One "step" = vbm_analysis_batch over one long block of every stream (each step advances
every stream by 1024 samples = 23.2 ms of audio; consecutive steps feed consecutive overlapping
blocks so the carried aoTuV state evolves as in a real encode).  Packets stay on the device.

Streams shard across ranks with no data-path collective: weak scaling, every
rank encodes its own 16384 streams; value = audio seconds encoded by all ranks / max-over-ranks
wall time = number of streams that could be encoded at 1x realtime.
"""
import argparse
import json
import os
import sys
import time

if "--from-pcm" in sys.argv:
    # the four block types of a blockout round run on four internal HIP streams next to the caller's; the ROCm
    # runtime folds HIP streams onto 4 hardware queues by default, which makes two of them share one
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # lazy joins (the back half of a write's long-block batch beside the next write) need a third workspace
    os.environ.setdefault("VBM_WORKSPACES", "4")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAZY_JOIN = os.environ.get("VBM_BENCH_LAZY_JOIN", "1") != "0"   # --from-pcm: vbm_frontend_encode_rounds_lazy
kept = []
STREAMS_PER_GPU = int(os.environ.get("VBM_BENCH_STREAMS", "16384"))
CHANNELS = 2
RATE = 44100
QUALITY = 0.5
N_LONG = 2048
HOP = N_LONG // 2
SPLIT = int(os.environ.get("VBM_BENCH_SPLIT", "1"))   # sub-batches per step, one HIP stream each
TWO_STREAMS = os.environ.get("VBM_BENCH_TWO_STREAMS", "1") != "0"   # vbm_analysis_batch2: front / back half streams
MAX_ROUNDS = int(os.environ.get("VBM_BENCH_MAX_ROUNDS", "1"))   # --from-pcm: blockout rounds per write before the
                                                                 # buffers decide (more while one is past half full)
DISTINCT_STEPS = 8              # PCM for this many consecutive blocks per stream is kept in HBM
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


def synth_blocks(dev, seed):
    """[DISTINCT_STEPS][streams][ch][N] block-major PCM cut from one continuous signal per stream."""
    g = torch.Generator(device=dev).manual_seed(seed)
    S, C = STREAMS_PER_GPU, CHANNELS
    total = (DISTINCT_STEPS + 1) * HOP
    t = torch.arange(total, device=dev, dtype=torch.float32) / RATE
    f1 = 110.0 + 1650.0 * torch.rand((S, 1, 1), generator=g, device=dev)
    f2 = 2000.0 + 4000.0 * torch.rand((S, 1, 1), generator=g, device=dev)
    chan = torch.arange(1, C + 1, device=dev, dtype=torch.float32).view(1, C, 1)
    x = 0.3 * torch.sin(2 * np.pi * f1 * chan * t) + 0.2 * torch.sin(2 * np.pi * f2 * t + chan)
    x += 0.05 * (2 * torch.rand((S, C, total), generator=g, device=dev) - 1)
    blocks = [x[:, :, k * HOP:k * HOP + N_LONG].contiguous() for k in range(DISTINCT_STEPS)]
    return blocks


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
            "unit": f"x realtime on {threads} host cores (= streams at 1x; one stream per thread)",
            "cores": threads, "kind": "port",
            "per_core": sum(per_core) / threads, "cpu_model": model, "host_cpus_available": cores,
            "restatement_vs_reference": {"scalar": 0.65, "lancer_sse2": 0.57,
                                         "note": "authoring container, 1 core: oracle 50-59x realtime vs the "
                                                 "survey's reference builds at 81x (scalar) / 93x (SSE2), BASELINE.md 2; "
                                                 "divide the figures here by these to read them as reference-equivalents"},
            "sample": f"{threads} x {seconds_of_audio} s of 44.1 kHz stereo q5 (survey probe signal, {res[0][0]} packets "
                      f"each, block switching + envelope search included), oracle/ scalar C, {wall:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--from-pcm", action="store_true",
                    help="time the whole encoder from raw PCM (stream front end: envelope search + block "
                         "carve-out on the device, SURVEY 8f N1) instead of the per-block path alone")
    ap.add_argument("--bitrate", type=int, default=0,
                    help="managed-bitrate setup of this nominal rate (vorbis_encode_init, SURVEY 8f N2: all 15 "
                         "packetblobs per block) instead of the q5 VBR setup of the headline metric")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the encode path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import vorbis_aotuv_lancer_amd as v

    setup = v.Setup(CHANNELS, RATE, bitrate=args.bitrate) if args.bitrate else v.Setup(CHANNELS, RATE, QUALITY)
    blocks = synth_blocks(dev, seed=1234 + rank)      # resident in HBM before timing starts
    LONG = 3
    # The 16384 streams of a step are submitted as SPLIT sub-batches, each to its own encoder object
    # (its share of the stream state + workspace) on its own HIP stream, so the serial stages of one
    # sub-batch (few wavefronts) overlap with the wide stages of another.
    per = STREAMS_PER_GPU // SPLIT
    assert per * SPLIT == STREAMS_PER_GPU
    encs = [v.Encoder(setup, per) for _ in range(SPLIT)]
    enc = encs[0]
    queues = [torch.cuda.Stream(device=dev) for _ in range(SPLIT)] if SPLIT > 1 else [torch.cuda.current_stream()]
    ids = np.arange(per, dtype=np.int32)
    wflags = np.full(per, 3, dtype=np.uint8)   # lW = nW = long
    parts = [[blk[p * per:(p + 1) * per] for blk in blocks] for p in range(SPLIT)]   # contiguous views

    back_queues = [torch.cuda.Stream(device=dev, priority=int(os.environ.get("VBM_BENCH_BACK_PRIORITY", "0")))
                   for _ in range(SPLIT)]
    outs = [[(torch.empty((per, enc.max_packet_bytes), dtype=torch.uint8, device=dev),
              torch.empty((per,), dtype=torch.int32, device=dev)) for _ in range(2)] for _ in range(SPLIT)]
    fe = None
    if args.from_pcm:
        # one continuous signal per stream, cut into 1024-sample writes (23.2 ms of audio per step)
        assert SPLIT == 1
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        nchunks = args.steps + args.warmup
        t = torch.arange(nchunks * HOP, device=dev, dtype=torch.float32) / RATE
        f1 = 110.0 + 1650.0 * torch.rand((STREAMS_PER_GPU, 1, 1), generator=g, device=dev)
        f2 = 2000.0 + 4000.0 * torch.rand((STREAMS_PER_GPU, 1, 1), generator=g, device=dev)
        chan = torch.arange(1, CHANNELS + 1, device=dev, dtype=torch.float32).view(1, CHANNELS, 1)
        chunks = []
        for k in range(nchunks):
            tk = t[k * HOP:(k + 1) * HOP]
            x = 0.3 * torch.sin(2 * np.pi * f1 * chan * tk) + 0.2 * torch.sin(2 * np.pi * f2 * tk + chan)
            x += 0.05 * (2 * torch.rand((STREAMS_PER_GPU, CHANNELS, HOP), generator=g, device=dev) - 1)
            chunks.append(x.contiguous())
        fe = v.FrontEnd(enc)
        round_count = [0, 0]

    def step(k):
        if fe is not None:
            # Round policy: a stream inside a run of short blocks has up to 8 blocks per write, each in its
            # own round, and such rounds hold a handful of blocks.  Two rounds per write keep every stream
            # ahead of its input on average (a lagging stream gains one block per step); more only while
            # some buffer is past half of its capacity.
            trace = os.environ.get("VBM_BENCH_TRACE")
            if trace:
                torch.cuda.synchronize(); t_a = time.perf_counter()
            fe.write(chunks[k])
            out = None
            rounds = 0
            if not trace and os.environ.get("VBM_BENCH_SINGLE_ROUNDS", "0") != "1":
                # all rounds of the write in one call: a round runs beside the long-block batch of the round before
                # it (vbm_frontend_encode_rounds), everything joined at the end
                info, pk_, nb_, counts = fe.encode_rounds(min_rounds=MAX_ROUNDS, max_rounds=16, headroom=HOP, device=dev,
                                                          lazy=LAZY_JOIN)
                round_count[0] += len(counts)
                round_count[1] += len(info)
                if LAZY_JOIN:      # the outputs of a call are complete after the next one: keep them alive
                    kept.append((info, pk_, nb_))
                    del kept[:-3]
                return (pk_, nb_) if len(info) else None
            while True:
                info, pk_, nb_ = fe.encode_round(dev)
                if trace:
                    torch.cuda.synchronize(); t_b = time.perf_counter()
                    print(f"step {k} round {rounds}: {len(info)} blocks, modes "
                          f"{np.bincount(info['block_mode'], minlength=4).tolist() if len(info) else []}, "
                          f"{(t_b - t_a) * 1e3:.2f} ms, max_buffered {fe.max_buffered}", file=sys.stderr)
                    t_a = t_b
                if len(info) == 0:
                    break
                out = (pk_, nb_) if out is None else out
                rounds += 1
                round_count[0] += 1
                round_count[1] += len(info)
                if rounds >= MAX_ROUNDS and fe.max_buffered + HOP <= fe.capacity // 2:
                    break
            return out
        out = None
        for p in range(SPLIT):
            with torch.cuda.stream(queues[p]):
                if TWO_STREAMS:   # back half of this step beside the front half of the next (vbm_analysis_batch2)
                    out = encs[p].analysis_batch(LONG, ids, wflags, parts[p][k % DISTINCT_STEPS],
                                                 back_stream=back_queues[p], out=outs[p][k & 1])
                else:
                    out = encs[p].analysis_batch(LONG, ids, wflags, parts[p][k % DISTINCT_STEPS])
        return out

    def barrier():
        if fe is not None:
            fe.join()                # lazy joins: everything begun so far
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    barrier()
    for each in encs:
        each.profile_begin(args.steps)  # HIP events between the stage kernels, on the stream each is launched on
    t0 = time.perf_counter()
    for k in range(args.steps):
        res = step(args.warmup + k)
        if res is not None:
            pk, nb = res
    barrier()
    dt = time.perf_counter() - t0
    stage_ms, calls = enc.profile_end()
    for other in encs[1:]:
        other.profile_end()
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    audio_s_per_step = STREAMS_PER_GPU * HOP / RATE * world
    value = audio_s_per_step * args.steps / dt
    ncb = STREAMS_PER_GPU * CHANNELS
    # the three transform stages are launched once per call on the whole batch, the later stages once
    # per sub-batch (include/vorbis_mi355x.h, vbm_encoder_set_sub_batches)
    sub = enc.sub_batches
    front = ("window_mdct", "window_fft_log", "transpose")
    launches = {k: max(calls, 1) * SPLIT * (1 if k in front else sub) for k in stage_ms}
    units = {k: (ncb // SPLIT) // (1 if k in front else sub) for k in stage_ms}   # channel-blocks per launch
    per_launch_ms = {k: ms / launches[k] for k, ms in stage_ms.items()}
    per_step_ms = {k: ms / max(calls, 1) for k, ms in stage_ms.items()}       # summed over the launches of a step
    dominant = max(per_step_ms, key=per_step_ms.get)

    def roof(stage):
        ms = per_launch_ms[stage]
        ach = STAGE_BYTES[stage] * units[stage] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": stage, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                "algorithmic_bytes_per_launch": STAGE_BYTES[stage] * units[stage],
                "channel_blocks_per_launch": units[stage], "kernel_ms": ms}

    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("channel_blocks_per_step") == ncb and tj.get("sub_batches") == sub:
            traffic = tj.get("hbm_bytes_per_launch", {})

    # The MDCT kernel on its own (same launch shape as in the pipeline: one long block of every channel),
    # timed with HIP events inside the library on this stream: the kernel's own HBM rate.  Inside the
    # pipeline it shares the GPU with the previous step's back half (vbm_analysis_batch2), which is what
    # stage_ms_per_launch["window_mdct"] shows.
    mdct_alone_ms = None
    if not args.from_pcm:
        import ctypes as C
        lk = v.MdctLookup(N_LONG, short_n=256)
        x = blocks[0].reshape(ncb, N_LONG)
        y = torch.empty((ncb, N_LONG // 2), device=dev)
        ms = C.c_float()
        st_ = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        v.check(v.lib.vbm_window_mdct_time(lk._h, x.data_ptr(), y.data_ptr(), None, ncb, 3, st_, C.byref(ms)))
        v.check(v.lib.vbm_window_mdct_time(lk._h, x.data_ptr(), y.data_ptr(), None, ncb, 20, st_, C.byref(ms)))
        mdct_alone_ms = ms.value / 20

    if rank == 0:
        r_dom = roof(dominant)
        r_dom["traffic"] = traffic.get(dominant)
        r_mdct = roof("window_mdct")
        r_mdct["traffic"] = traffic.get("window_mdct")
        if mdct_alone_ms:
            r_mdct["in_pipeline_ms"] = r_mdct["kernel_ms"]
            r_mdct["in_pipeline_frac"] = r_mdct["frac"]
            r_mdct["kernel_ms"] = mdct_alone_ms
            r_mdct["achieved"] = r_mdct["algorithmic_bytes_per_launch"] / (mdct_alone_ms * 1e-3) / 1e9
            r_mdct["frac"] = r_mdct["achieved"] / HBM_PEAK_GBPS
            r_mdct["note"] = "kernel alone, 20 launches of the step's blocks; in_pipeline_* = beside the previous step's back half"
        mean_bytes = float(nb.float().mean().item())
        line = {
            "metric": "realtime-stream-equivalents/node (44.1kHz stereo q5) + MDCT HBM GB/s",
            "value": value,
            "unit": "x realtime (concurrent 44.1 kHz stereo q5 streams encodable at 1x)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[2]: full pipe incl. residue VQ on device, {STREAMS_PER_GPU} streams/GPU x 44.1 kHz "
                            "coupled stereo q5, long blocks (2048), one block per stream per step, PCM resident "
                            "in HBM, packets left in HBM",
                "realtime_factor_at_this_concurrency": (HOP / RATE) / (dt / args.steps),
                "streams_per_gpu": STREAMS_PER_GPU, "channels": CHANNELS, "blocksize": N_LONG,
                "channel_blocks_per_step": ncb * world,
                "stages": list(stage_ms.keys()),
                "pipeline_complete": True,
                "from_pcm": bool(args.from_pcm),
                **({"managed_bitrate": args.bitrate,
                    "note": "NOT the headline setup: managed bitrate, 15 packetblobs per block (lib/mapping0.c:1204)"}
                   if args.bitrate else {}),
                **({"blocks_encoded": round_count[1], "rounds": round_count[0], "max_rounds_per_write": MAX_ROUNDS}
                   if args.from_pcm else {}),
                "block_switching": ("in the timed region: PCM intake, envelope search and block carve-out run on the "
                                    "device (vbm_frontend_*); one 1024-sample write per stream per step"
                                    if args.from_pcm else
                                    "not in the timed region (long blocks only; --from-pcm times the stream front "
                                    "end as well)"),
                "mean_packet_bytes": mean_bytes,
                "parallelism": f"stream-shard x{world} (no collective); "
                               + ("front half (transforms, psychoacoustics) and back half (floor, couple/quantise, "
                                  "packets) of consecutive steps on two HIP streams (vbm_analysis_batch2)"
                                  if TWO_STREAMS and not args.from_pcm else "one HIP stream"),
            },
            "roofline": r_dom,
            "mdct_roofline": r_mdct,
            "stage_ms_per_launch": per_launch_ms,
            "stage_ms_per_step": per_step_ms,
            "sub_batches": sub,
        }
        if not args.no_cpu_baseline and world == 1:   # host baseline: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
