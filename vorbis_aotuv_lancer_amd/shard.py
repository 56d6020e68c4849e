"""Stream sharding across the GPUs of one node: streams are independent, so each
rank owns a contiguous range of stream ids and there is no data-path collective.  The only
cross-rank traffic is control: a barrier and a MAX-reduction of the step time."""
import torch
import torch.distributed as dist


def stream_range(total_streams, rank, world):
    """Contiguous, balanced partition: rank r gets [lo, hi)."""
    base, extra = divmod(total_streams, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def max_over_ranks(seconds, device=None):
    """Largest value of `seconds` over all ranks (whole-job time of an embarrassingly parallel step)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(audio_seconds_local, wall_seconds_local, device=None):
    """Whole-job audio-seconds per wall-second: sum of the work, max of the time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return audio_seconds_local / wall_seconds_local
    w = torch.tensor([audio_seconds_local], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return float(w.item()) / max_over_ranks(wall_seconds_local, device)
