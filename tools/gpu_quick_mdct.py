"""ad-hoc: time the window+MDCT kernel at a few batch sizes (run on the GPU box)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vorbis_aotuv_lancer_amd as v

dev = torch.device("cuda:0")
for n, nb in ((2048, 131072), (2048, 16384), (256, 131072 * 8)):
    lk = v.MdctLookup(n, short_n=256)
    x = (torch.rand((nb, n), device=dev) - 0.5)
    y = torch.empty((nb, n // 2), device=dev)
    ms = C.c_float()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    v.check(v.lib.vbm_window_mdct_time(lk._h, x.data_ptr(), y.data_ptr(), None, nb, 3, st, C.byref(ms)))
    v.check(v.lib.vbm_window_mdct_time(lk._h, x.data_ptr(), y.data_ptr(), None, nb, 20, st, C.byref(ms)))
    per = ms.value / 20
    print(f"n={n} blocks={nb}: {per*1e3:.1f} us/launch, {nb*6*n/per/1e6:.1f} GB/s algorithmic "
          f"({nb*6*n/per/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
