"""N > 1 control path on CPU (gloo, world_size 2): stream sharding is a disjoint cover, the
step-time reduction takes the slowest rank, throughput aggregates work / max time."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vorbis_aotuv_lancer_amd.shard import stream_range, max_over_ranks, aggregate_throughput
    lo, hi = stream_range(total, rank, world)
    owned = torch.zeros(total, dtype=torch.int32)
    owned[lo:hi] = 1
    dist.all_reduce(owned)                      # every stream owned exactly once
    t = max_over_ranks(0.010 * (rank + 1))      # rank 1 is the slow one
    thr = aggregate_throughput(float(hi - lo), 0.010 * (rank + 1))
    if rank == 0:
        torch.save({"owned": owned, "t": t, "thr": thr, "lo": lo, "hi": hi}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_stream_sharding_and_time_reduction(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    total = 16385                                # odd on purpose: ragged split
    mp.spawn(_worker, args=(2, port, total, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    assert bool((r["owned"] == 1).all())
    assert (r["lo"], r["hi"]) == (0, 8193)
    assert abs(r["t"] - 0.020) < 1e-12
    assert abs(r["thr"] - total / 0.020) < 1e-6
