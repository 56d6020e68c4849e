"""Dump the floor posts of the first steps of the 5.1 q8 case (run once per VBM_FLOORFIT_COOP setting, then compare)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from tests import orc
from tests.test_pipeline_gpu import oracle_blocks
import vorbis_aotuv_lancer_amd as v
o = orc.Oracle(os.path.join("oracle", "build", "liboracle.so"))
ch, rate, q, ns = 6, 48000, 0.8, 3
streams = [oracle_blocks(o, ch, rate, q, 1.2, seed=100 + s) for s in range(ns)]
enc = v.Encoder(v.Setup(ch, rate, q), ns)
dev = torch.device("cuda:0")
out = {}
for k in range(6):
    by_mode = {}
    for s in range(ns):
        by_mode.setdefault(streams[s][k]["block_mode"], []).append(s)
    for mode, ids in sorted(by_mode.items()):
        blks = [streams[s][k] for s in ids]
        pcm = torch.from_numpy(np.stack([b["pcm"] for b in blks])).to(dev)
        wflags = [b["lW"] | (b["nW"] << 1) for b in blks]
        enc.analysis_batch(mode, ids, wflags, pcm)
        out[f"post_{k}_{mode}"] = enc.fetch("post").cpu().numpy()
        out[f"valid_{k}_{mode}"] = enc.fetch("post_valid").cpu().numpy()
np.savez(sys.argv[1], **out)
