#!/usr/bin/env python3
"""Headline benchmark of the MI355X batched Vorbis encode path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 4096 concurrent 44.1 kHz stereo q5 streams per GPU,
long blocks (2048 samples), synthetic PCM already resident in HBM.  One "step" = one pass of
the implemented hot-path stages over one batch of BLOCKS_PER_STREAM consecutive long blocks
of every stream (each long block advances a stream by 1024 samples).  Streams shard across
ranks with no data-path collective, so scaling is weak: every rank encodes
its own 4096 streams; `value` = audio seconds encoded by all ranks / max-over-ranks wall time.

The JSON line carries which stages are inside the timed region (`config.stages`): until
the whole pipe of SURVEY.md §8a is on the device, `value` covers only those stages and
`config.pipeline_complete` is false.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 4096
CHANNELS = 2
RATE = 44100
N_LONG = 2048
BLOCKS_PER_STREAM = 16          # long blocks per stream per step (16 hops = 0.3715 s of audio)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MDCT_BYTES_PER_BLOCK = 6 * N_LONG  # read N*4 + write (N/2)*4 per (block, channel)


def synth_pcm(nblocks, device, seed):
    """Deterministic synthetic block-major PCM in [-1, 1): two sines + noise per block row."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(N_LONG, device=device, dtype=torch.float32) / RATE
    f1 = 110.0 + 1650.0 * torch.rand((nblocks, 1), generator=g, device=device)
    f2 = 2000.0 + 4000.0 * torch.rand((nblocks, 1), generator=g, device=device)
    x = 0.3 * torch.sin(2 * np.pi * f1 * t) + 0.2 * torch.sin(2 * np.pi * f2 * t)
    x += 0.05 * (2 * torch.rand((nblocks, N_LONG), generator=g, device=device) - 1)
    return x.contiguous()


def cpu_baseline(sample_blocks=4096):
    """Oracle (CPU restatement, 1 thread) on a bounded sample of the same workload."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    from tests import orc
    from vorbis_aotuv_lancer_amd.tables import window_table
    o = orc.Oracle(so)
    rng = np.random.default_rng(0)
    x = (rng.random((sample_blocks, N_LONG), dtype=np.float32) - 0.5)
    wl = window_table(N_LONG)
    t0 = time.perf_counter()
    o.mdct_forward(o.apply_window(x, wl, wl))
    dt = time.perf_counter() - t0
    audio_s = sample_blocks / CHANNELS * (N_LONG // 2) / RATE
    return {"value": audio_s / dt, "unit": "realtime-stream-equivalents (same stages)", "cores": 1,
            "kind": "port",
            "sample": f"{sample_blocks} long channel-blocks, window+mdct_forward only, oracle/ scalar C, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    nblocks = STREAMS_PER_GPU * CHANNELS * BLOCKS_PER_STREAM
    pcm = synth_pcm(nblocks, dev, seed=1234 + rank)   # resident in HBM before timing starts
    spec = torch.empty((nblocks, N_LONG // 2), device=dev, dtype=torch.float32)
    lookup = v.MdctLookup(N_LONG, short_n=256)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        v.check(v.lib.vbm_window_mdct_batch(lookup._h, pcm.data_ptr(), spec.data_ptr(), None, nblocks, stream),
                "vbm_window_mdct_batch")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()          # torch's current stream == the stream the kernel is launched on
        step()
        b.record()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    audio_s_per_step = STREAMS_PER_GPU * BLOCKS_PER_STREAM * (N_LONG // 2) / RATE * world
    value = audio_s_per_step * args.steps / dt
    achieved = MDCT_BYTES_PER_BLOCK * nblocks / (kernel_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "mdct_pmc_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("nblocks") == nblocks:
            traffic = tj.get("hbm_bytes_per_launch")

    if rank == 0:
        line = {
            "metric": "realtime-stream-equivalents/node (44.1kHz stereo q5) + MDCT HBM GB/s",
            "value": value,
            "unit": "x realtime (concurrent 44.1 kHz stereo streams encodable at 1x)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "configs[1]: 4096 streams/GPU x 44.1 kHz stereo q5, long blocks (2048), "
                            f"{BLOCKS_PER_STREAM} blocks/stream/step, PCM resident in HBM",
                "streams_per_gpu": STREAMS_PER_GPU, "channels": CHANNELS, "blocksize": N_LONG,
                "blocks_per_step": nblocks * world,
                "stages": ["window", "mdct_forward"],
                "pipeline_complete": False,
                "parallelism": f"stream-shard x{world} (no collective)",
            },
            "roofline": {"kernel": "k_window_mdct<2048>", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": MDCT_BYTES_PER_BLOCK * nblocks,
                         "kernel_ms": kernel_ms},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
