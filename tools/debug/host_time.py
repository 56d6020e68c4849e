"""How long does the host spend inside write() / encode_rounds_device() per step (no synchronisation)?"""
import os, sys, time
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
chunks = [bench.synth_pcm(torch, dev, params, gen, k * 1024, 1024) for k in range(40)]
torch.cuda.synchronize()
if os.environ.get('USER_STREAM'):
    ust = torch.cuda.Stream(); torch.cuda.set_stream(ust)
tw = []; te = []
kept = []
for k in range(40):
    t0 = time.perf_counter(); fe.write(chunks[k]); t1 = time.perf_counter()
    kept.append(fe.encode_rounds_device(nrounds=2, lazy=True)); t2 = time.perf_counter()
    del kept[:-3]
    tw.append(t1 - t0); te.append(t2 - t1)
    if k == 19: torch.cuda.synchronize(); tstart = time.perf_counter()
torch.cuda.synchronize(); tend = time.perf_counter()
print("host ms per step: write %.3f  encode %.3f  (steps 20..39); wall per step %.3f" % (1e3*np.mean(tw[20:]), 1e3*np.mean(te[20:]), 1e3*(tend-tstart)/20))
