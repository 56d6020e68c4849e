"""debug: front end vs oracle block sequence on the probe signal; prints the first divergence"""
import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import orc
from tests.test_frontend_gpu import probe_pcm, drain
import vorbis_aotuv_lancer_amd as v
o = orc.Oracle("oracle/build/liboracle.so")
ch, rate, q, secs = 2, 44100, 0.5, int(sys.argv[1]) if len(sys.argv) > 1 else 20
pcm = probe_pcm(o, ch, rate, secs)
st = orc.Stream(orc.Setup(o, ch, rate, q)); o.lib.orc_stream_set_capture(st.v, 0)
want = []
for at in range(0, pcm.shape[1], 1024):
    st.write(pcm[:, at:at + 1024]); want.extend(st.blocks())
st.finish(); want.extend(st.blocks())
enc = v.Encoder(v.Setup(ch, rate, q), 1); fe = v.FrontEnd(enc)
got = [[]]
dev = torch.from_numpy(pcm).cuda()
for at in range(0, pcm.shape[1], 1024):
    fe.write(dev[None, :, at:at + 1024].contiguous()); drain(fe, got)
fe.finish(); drain(fe, got)
g = got[0]
print("oracle blocks", len(want), "device blocks", len(g))
for i in range(min(len(want), len(g))):
    w = (want[i]["lW"], want[i]["W"], want[i]["nW"], want[i]["block_mode"], want[i]["eos"], want[i]["granulepos"], want[i]["sequence"])
    if w != g[i][0] or want[i]["packet"] != g[i][1]:
        print("first divergence at block", i)
        for k in range(max(0, i - 3), min(len(want), len(g), i + 4)):
            wk = (want[k]["lW"], want[k]["W"], want[k]["nW"], want[k]["block_mode"], want[k]["eos"], want[k]["granulepos"], want[k]["sequence"])
            print(k, "oracle", wk, "device", g[k][0], "packet equal" if want[k]["packet"] == g[k][1] else "PACKET DIFFERS")
        break
else:
    print("common prefix identical")
