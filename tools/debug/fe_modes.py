"""debug: how many streams of the bench's stationary signal sit in which block mode per round"""
import sys, os, numpy as np, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vorbis_aotuv_lancer_amd as v
S, CH, RATE, HOP = 4096, 2, 44100, 1024
dev = torch.device("cuda:0")
enc = v.Encoder(v.Setup(CH, RATE, 0.5), S); fe = v.FrontEnd(enc)
g = torch.Generator(device=dev).manual_seed(99)
f1 = 110.0 + 1650.0 * torch.rand((S, 1, 1), generator=g, device=dev)
f2 = 2000.0 + 4000.0 * torch.rand((S, 1, 1), generator=g, device=dev)
chan = torch.arange(1, CH + 1, device=dev, dtype=torch.float32).view(1, CH, 1)
t = torch.arange(40 * HOP, device=dev, dtype=torch.float32) / RATE
for k in range(40):
    tk = t[k * HOP:(k + 1) * HOP]
    x = 0.3 * torch.sin(2 * np.pi * f1 * chan * tk) + 0.2 * torch.sin(2 * np.pi * f2 * tk + chan)
    x += 0.05 * (2 * torch.rand((S, CH, HOP), generator=g, device=dev) - 1)
    fe.write(x.contiguous())
    rounds = []
    while True:
        info, pk, nb = fe.encode_round(dev)
        if len(info) == 0: break
        rounds.append(dict(collections.Counter(int(m) for m in info["block_mode"])))
    if k >= 4: print(k, rounds)
