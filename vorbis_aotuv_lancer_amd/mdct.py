"""Batched forward MDCT — host mirror of mdct_lookup / mdct_init / mdct_forward
(reference lib/mdct.h:55-80, lib/mdct.c:54-92, :1799-1869) and of the window + MDCT pair
mapping0_forward runs per channel (lib/mapping0.c:825-843)."""
import ctypes as C
import numpy as np
import torch

from ._lib import lib, check
from .tables import window_table


class MdctLookup:
    """Device-resident MDCT lookup for one block size (mdct_lookup, lib/mdct.h:55-73).

    n: 2048 (long) or 256 (short); short_n: the short size of the mode pair, needed by
    long blocks whose neighbour is short (lib/window.c:2143-2151)."""

    def __init__(self, n, short_n=256, with_window=True):
        self.n = int(n)
        self.short_n = int(short_n) if self.n != short_n else self.n
        self._h = C.c_void_p()
        wn = ws = None
        if with_window:
            wn = np.ascontiguousarray(window_table(self.n))
            ws = np.ascontiguousarray(window_table(self.short_n))
        check(lib.vbm_mdct_plan_create(C.byref(self._h), self.n, self.short_n,
                                       wn.ctypes.data if wn is not None else None,
                                       ws.ctypes.data if ws is not None else None),
              "vbm_mdct_plan_create")

    @property
    def trig(self):
        p = lib.vbm_mdct_plan_trig(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n + self.n // 4,)).copy()

    @property
    def fft_twiddles(self):
        p = lib.vbm_mdct_plan_fft_twiddles(self._h)
        return np.ctypeslib.as_array(p, shape=(self.n,)).copy()

    def close(self):
        if self._h:
            lib.vbm_mdct_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prep(lookup, x):
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("expected a contiguous float32 CUDA tensor")
    if x.shape[-1] != lookup.n:
        raise ValueError(f"last dimension must be the block size {lookup.n}")
    return x.numel() // lookup.n


def mdct_forward(lookup, x, out=None):
    """mdct_forward(init, in, out) for every row of x (…, n) -> (…, n/2)."""
    nb = _prep(lookup, x)
    if out is None:
        out = torch.empty(x.shape[:-1] + (lookup.n // 2,), dtype=torch.float32, device=x.device)
    check(lib.vbm_mdct_forward_batch(lookup._h, x.data_ptr(), out.data_ptr(), nb, _stream()),
          "vbm_mdct_forward_batch")
    return out


def window_mdct(lookup, pcm, wflags=None, out=None):
    """_vorbis_apply_window + mdct_forward for every row of pcm (…, n).
    wflags: optional uint8 tensor, one per block, bit0 = lW, bit1 = nW."""
    nb = _prep(lookup, pcm)
    if out is None:
        out = torch.empty(pcm.shape[:-1] + (lookup.n // 2,), dtype=torch.float32, device=pcm.device)
    fp = None
    if wflags is not None:
        if not (wflags.is_cuda and wflags.dtype == torch.uint8 and wflags.numel() == nb):
            raise ValueError("wflags must be a uint8 CUDA tensor with one entry per block")
        fp = wflags.data_ptr()
    check(lib.vbm_window_mdct_batch(lookup._h, pcm.data_ptr(), out.data_ptr(), fp, nb, _stream()),
          "vbm_window_mdct_batch")
    return out


def window_fft_log(lookup, pcm, wflags=None):
    """_vorbis_apply_window + drft_forward + log-power spectrum (lib/mapping0.c:825-888).
    Returns (logfft (…, n/2), local_ampmax (…,))."""
    nb = _prep(lookup, pcm)
    logfft = torch.empty(pcm.shape[:-1] + (lookup.n // 2,), dtype=torch.float32, device=pcm.device)
    amp = torch.empty(pcm.shape[:-1], dtype=torch.float32, device=pcm.device)
    fp = None
    if wflags is not None:
        if not (wflags.is_cuda and wflags.dtype == torch.uint8 and wflags.numel() == nb):
            raise ValueError("wflags must be a uint8 CUDA tensor with one entry per block")
        fp = wflags.data_ptr()
    check(lib.vbm_window_fft_log_batch(lookup._h, pcm.data_ptr(), logfft.data_ptr(), amp.data_ptr(), fp, nb,
                                       _stream()), "vbm_window_fft_log_batch")
    return logfft, amp
