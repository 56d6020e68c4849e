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


def test_bench_control_path_dry_run(tmp_path):
    """bench.py's own rank arithmetic at world size 2 (gloo, no device): stream ranges are a disjoint cover with
    per-rank offsets, n_gpus = 2, the time is the slower rank's and value = sum of the work / max of the time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VBM_BENCH_DRYRUN_MS="5,20", VBM_BENCH_STREAMS="1000")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
                                       "--warmup", "0", "--dry-run"], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert outs[1].strip() == ""                               # rank 0 alone prints the line
    assert len(outs[0].strip().splitlines()) == 1              # ONE line on stdout, and it is JSON
    line = json.loads(outs[0].strip())
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 10
    assert line["config"]["control_backend"] == "gloo"
    assert line["config"]["stream_ranges"] == [[0, 1000], [1000, 2000]]
    audio = 2 * 1000 * 1024 / 44100 * 10
    assert abs(line["config"]["audio_s_all_ranks"] - audio) < 1e-9
    ms = line["ms_per_step"]
    assert 20.0 <= ms < 40.0                                   # the slow rank's 20 ms sleeps, not the mean (12.5)
    assert abs(line["value"] - audio / (ms * 10 / 1e3)) < 1e-6 * line["value"]
