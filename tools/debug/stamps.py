"""Device-side timeline of the from-PCM pipeline (VBM_DEBUG_STAMPS=1): where every step's front end, big batch halves
and small batches start and end, in ms from the first stamp of the window."""
import os, sys, ctypes as C
os.environ["VBM_DEBUG_STAMPS"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("VBM_WORKSPACES", "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vorbis_aotuv_lancer_amd as v
import bench
dev = torch.device("cuda:0")
S = 16384
setup = v.Setup(2, 44100, 0.5)
enc = v.Encoder(setup, S, max_batch=v.lib.vbm_device_round_lanes(setup._h, S))
fe = v.FrontEnd(enc)
params = bench.stream_params(torch, dev, 0, S)
gen = torch.Generator(device=dev).manual_seed(1)
N = 40
chunks = [bench.synth_pcm(torch, dev, params, gen, k * 1024, 1024) for k in range(N)]
torch.cuda.synchronize()
kept = []
consumer = torch.cuda.Stream(device=dev)
pat = [int(x) for x in os.environ.get("VBM_BENCH_ROUNDS", "2,1").split(",")]
for k in range(N):
    fe.write(chunks[k])
    kept.append(fe.encode_rounds_device(nrounds=3 if k < 8 else pat[k % len(pat)], lazy=2)); del kept[:-6]
    fe.join(consumer)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 8192))()
n = v.lib.vbm_debug_stamps_read(buf, 8192)
a = np.frombuffer(buf, dtype=np.uint64)[:2 * n].reshape(-1, 2)
names = {0: "write", 1: "encode", 2: "run r0", 3: "run r>0", 4: "fe end", 10: "BIG front start", 11: "BIG front end", 12: "BIG back start",
         13: "BIG back end", 20: "t0 start", 21: "t1 start", 22: "t2 start", 23: "t3 start", 30: "t0 end", 31: "t1 end", 32: "t2 end", 33: "t3 end"}
order = np.argsort(a[:, 1], kind="stable")
a = a[order]
writes = [i for i in range(len(a)) if a[i, 0] == 0]
i0 = writes[-7]
t0 = int(a[i0, 1])
for tag, t in a[i0:]:
    print(f"{(int(t) - t0) / 1e5:9.3f} ms  {names.get(int(tag), int(tag))}")
