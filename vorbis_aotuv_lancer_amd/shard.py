"""Stream sharding across the GPUs of one node: streams are independent, so each rank owns a
contiguous range of stream ids and there is NO data-path collective and no RCCL traffic.  The only cross-rank
traffic is control — a barrier, a MAX-reduction of the step time, a SUM of the work — and it runs over a gloo
(TCP, CPU tensors) group that is created before anything touches the GPU.  bench.py is the user."""
import os

import torch
import torch.distributed as dist

_OWN_GROUP = False


def init_control(rank, world):
    """Create the control group (gloo) for `world` ranks; nothing to do for one rank.  Reads MASTER_ADDR /
    MASTER_PORT from the environment (torch.distributed.run sets them), defaulting to 127.0.0.1."""
    global _OWN_GROUP
    if world <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    # gloo announces its connections on stdout; bench.py's stdout carries exactly one JSON line, so the
    # announcement goes to stderr
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    _OWN_GROUP = True


def finish():
    global _OWN_GROUP
    if _OWN_GROUP and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    _OWN_GROUP = False


def _multi():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def backend():
    return dist.get_backend() if _multi() else "none (single rank)"


def barrier():
    if _multi():
        dist.barrier()


def stream_range(total_streams, rank, world):
    """Contiguous, balanced partition: rank r gets [lo, hi)."""
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def max_over_ranks(seconds, device=None):
    """Largest value of `seconds` over all ranks (whole-job time of an embarrassingly parallel step)."""
    if not _multi():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device=None):
    if not _multi():
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(audio_seconds_local, wall_seconds_local, device=None):
    """Whole-job audio-seconds per wall-second: sum of the work, max of the time."""
    return sum_over_ranks(audio_seconds_local, device) / max_over_ranks(wall_seconds_local, device)


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank"""
    if not _multi():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
