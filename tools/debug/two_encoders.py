"""Experiment: the bench's from-PCM loop with the streams split over K independent encoder objects in ONE process
(K dependency chains in flight instead of one).  usage: two_encoders.py K [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "8"))
os.environ.setdefault("VBM_WORKSPACES", "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vorbis_aotuv_lancer_amd as v
import bench

K = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda", 0)
S = 16384 // K
setup = v.Setup(2, 44100, 0.5)
encs, fes, params, gens = [], [], [], []
for k in range(K):
    e = v.Encoder(setup, S, max_batch=max(S, v.lib.vbm_device_round_lanes(setup._h, S)))
    encs.append(e); fes.append(v.FrontEnd(e))
    params.append(bench.stream_params(torch, dev, k * S, (k + 1) * S))
    gens.append(torch.Generator(device=dev).manual_seed(7 + k))
consumer = torch.cuda.Stream(device=dev)
PAT = [2, 1, 1, 1]
kept = []
def step(i, pat):
    for k in range(K):
        fes[k].write(chunks[k][i])
        kept.append(fes[k].encode_rounds_device(nrounds=pat, lazy=2, device=dev)); fes[k].join(consumer)
    del kept[:-6 * K]
PRIME = 64
for i in range(PRIME):
    for k in range(K):
        fes[k].write(bench.synth_pcm(torch, dev, params[k], gens[k], i * 1024, 1024))
        kept.append(fes[k].encode_rounds_device(nrounds=4 if i < 10 else PAT[i % 4], lazy=2, device=dev)); fes[k].join(consumer)
    del kept[:-6 * K]
for f in fes: f.join()
torch.cuda.synchronize()
chunks = [[bench.synth_pcm(torch, dev, params[k], gens[k], (PRIME + i) * 1024, 1024) for i in range(steps + 8)] for k in range(K)]
for i in range(8): step(i, PAT[i % 4])
for f in fes: f.join()
base = [f.device_stats() for f in fes]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): step(8 + i, PAT[i % 4])
for f in fes: f.join()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
end = [f.device_stats() for f in fes]
enc_s = sum(e[1] - b[1] for e, b in zip(end, base)) / 44100
inp_s = 16384 * 1024 / 44100 * steps
print(f"K={K} ms/step {dt/steps*1e3:.3f} streams {min(enc_s, inp_s)/dt:.0f} enc/in {enc_s/inp_s:.4f} refused {[f.refused_writes for f in fes]}")
