"""debug: device front end with full drain vs 2 rounds per write; list differing packets"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.signals import synth_signal
from tests.test_frontend_gpu import drain
import vorbis_aotuv_lancer_amd as v
ch, rate, q, NS, seconds = 2, 44100, 0.5, 40, 2.2
nsamp = int(seconds * rate) // 1024 * 1024
sigs = [synth_signal(ch, rate, nsamp, seed=500 + s, level=1.0 if s % 3 else 0.05) for s in range(NS)]
allp = torch.from_numpy(np.stack(sigs)).cuda()
def run(max_rounds, log=None):
    enc = v.Encoder(v.Setup(ch, rate, q), NS); fe = v.FrontEnd(enc)
    got = [[] for _ in range(NS)]
    for at in range(0, nsamp, 1024):
        fe.write(allp[:, :, at:at + 1024].contiguous())
        before = [len(g) for g in got]
        drain(fe, got, max_rounds)
        if log is not None: log.append([len(g) - b for g, b in zip(got, before)])
    fe.finish(); drain(fe, got)
    return got
la, lb = [], []
A = run(None, la); B = run(2, lb)
for s in range(NS):
    assert len(A[s]) == len(B[s])
    for i in range(len(A[s])):
        if A[s][i] != B[s][i]:
            print("stream", s, "block", i, "of", len(A[s]), "info A", A[s][i][0], "info B", B[s][i][0], "len", len(A[s][i][1]), len(B[s][i][1]))
            # which write produced it in each policy
            for name, log in (("A", la), ("B", lb)):
                acc = 0
                for w, row in enumerate(log):
                    acc += row[s]
                    if acc > i: print("   policy", name, "emitted it after write", w, "(", row[s], "blocks that write )"); break
                else: print("   policy", name, "emitted it after finish()")
print("done")
