"""debug: device-built rounds at 1100 streams; where do wrong packets come from?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("VBM_WORKSPACES", "4")
import vorbis_aotuv_lancer_amd as v
from tests import orc
from tests.signals import synth_signal
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "oracle", "build", "liboracle.so")
oracle = orc.Oracle(so)
ch, rate, q, NS, K = 2, 44100, 0.5, int(os.environ.get("NS", "1100")), 7
nsamp = 26 * 1024
base = [synth_signal(ch, rate, nsamp, seed=730 + k, level=1.0 if k % 3 else 0.05) for k in range(K)]
osetup = orc.Setup(oracle, ch, rate, q)
want = []
for k in range(K):
    st = orc.Stream(osetup); oracle.lib.orc_stream_set_capture(st.v, 0)
    seq = []
    for at in range(0, nsamp, 1024):
        st.write(base[k][:, at:at + 1024]); seq.extend(st.blocks())
    st.close()
    want.append([b["packet"] for b in seq])
setup = v.Setup(ch, rate, q)
lanes = v.lib.vbm_device_round_lanes(setup._h, NS)
enc = v.Encoder(setup, NS, max_batch=lanes)
fe = v.FrontEnd(enc)
dev = torch.device("cuda:0")
allp = torch.from_numpy(np.stack([base[s % K] for s in range(NS)])).to(dev)
got = [[] for _ in range(NS)]
call = 0
for at in range(0, nsamp, 1024):
    fe.write(allp[:, :, at:at + 1024].contiguous())
    info, packets, nbytes, counts = fe.encode_rounds_device(nrounds=2, lazy=False)
    torch.cuda.synchronize()
    nb = nbytes.cpu().numpy()
    rec = info.cpu().numpy().view(np.dtype(v.PacketInfo))[:, 0]
    pk = packets.cpu().numpy()
    for k in np.flatnonzero(nb != -2):
        got[int(rec[k]["stream"])].append((call, int(k) // lanes, int(k) % lanes, int(rec[k]["block_mode"]), bytes(pk[k, :nb[k]])))
    if call < 12:
        print("call", call, "counts", counts.cpu().numpy().tolist(), "max_buffered", fe.max_buffered)
    call += 1
bad = 0
for s in range(NS):
    for k, g in enumerate(got[s]):
        if k < len(want[s % K]) and g[4] != want[s % K][k]:
            bad += 1
            if bad <= 25:
                print("BAD stream", s, "packet", k, "call", g[0], "round", g[1], "lane", g[2], "mode", g[3])
print("bad total", bad, "lanes", lanes)
