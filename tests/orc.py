"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).  TEST INFRASTRUCTURE ONLY:
nothing under vorbis_aotuv_lancer_amd/ may import this module."""
import ctypes as C
import numpy as np


class _Mdct(C.Structure):
    _fields_ = [("n", C.c_int), ("log2n", C.c_int), ("trig", C.POINTER(C.c_float)),
                ("bitrev", C.POINTER(C.c_int)), ("scale", C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        self._mdct = {}

    # ---- MDCT ------------------------------------------------------------------
    def mdct_lookup(self, n):
        if n not in self._mdct:
            m = _Mdct()
            self.lib.orc_mdct_init(C.byref(m), n)
            self._mdct[n] = m
        return self._mdct[n]

    def mdct_trig(self, n):
        m = self.mdct_lookup(n)
        return np.ctypeslib.as_array(m.trig, shape=(n + n // 4,)).copy()

    def mdct_forward(self, x):
        """x: (..., n) float32 -> (..., n/2)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[-1]
        m = self.mdct_lookup(n)
        flat = x.reshape(-1, n)
        out = np.empty((flat.shape[0], n // 2), np.float32)
        for i in range(flat.shape[0]):
            self.lib.orc_mdct_forward(C.byref(m), _p(flat[i]), _p(out[i]))
        return out.reshape(x.shape[:-1] + (n // 2,))

    def apply_window(self, x, win_l, win_r):
        """in-place copy: x (..., n); win_l / win_r rising half windows (their length*2 = ln/rn)"""
        x = np.array(x, dtype=np.float32, copy=True)
        n = x.shape[-1]
        win_l = np.ascontiguousarray(win_l, np.float32)
        win_r = np.ascontiguousarray(win_r, np.float32)
        flat = x.reshape(-1, n)
        for i in range(flat.shape[0]):
            self.lib.orc_apply_window(_p(flat[i]), C.c_long(n), _p(win_l), C.c_long(2 * len(win_l)),
                                      _p(win_r), C.c_long(2 * len(win_r)))
        return x


# ---- full encoder ---------------------------------------------------------------------
import os as _os

_DATA = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "vorbis_aotuv_lancer_amd", "data")


class Setup:
    """orc_setup for one (channels, rate, quality) class, from the committed mode pack."""

    def __init__(self, oracle, ch, rate, q):
        self.o = oracle
        lib = oracle.lib
        lib.orc_setup_load.restype = C.c_void_p
        lib.orc_setup_load.argtypes = [C.c_char_p, C.c_char_p]
        lib.orc_encode_probe.restype = C.c_long
        lib.orc_encode_probe.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
        mode = _os.path.join(_DATA, f"mode_{ch}ch_{rate}_q{q:g}.vpk")
        self.h = lib.orc_setup_load(_os.path.join(_DATA, "common.vpk").encode(), mode.encode())
        if not self.h:
            raise RuntimeError(f"cannot load {mode}")
        self.ch, self.rate, self.q = ch, rate, q

    def encode_probe(self, secs, out_path=None):
        """SURVEY.md Appendix B signal + driver; returns (packets, seconds)."""
        t = C.c_double()
        n = self.o.lib.orc_encode_probe(self.h, secs, out_path.encode() if out_path else None, C.byref(t))
        return n, t.value


class _Drft(C.Structure):
    _fields_ = [("n", C.c_int), ("trigcache", C.POINTER(C.c_float)), ("splitcache", C.POINTER(C.c_int))]


def _drft(oracle, n):
    cache = oracle.__dict__.setdefault("_drft", {})
    if n not in cache:
        d = _Drft()
        oracle.lib.orc_drft_init(C.byref(d), n)
        cache[n] = d
    return cache[n]


def fft_twiddles(oracle, n):
    d = _drft(oracle, n)
    return np.ctypeslib.as_array(d.trigcache, shape=(3 * n,))[n:2 * n].copy()


def fft_logpower(oracle, windowed):
    """windowed: (..., n) float32 -> (logfft (..., n/2), local_ampmax (...))  [lib/mapping0.c:847-888]"""
    x = np.array(windowed, dtype=np.float32, copy=True)
    n = x.shape[-1]
    d = _drft(oracle, n)
    oracle.lib.orc_fft_logpower.restype = C.c_float
    flat = x.reshape(-1, n)
    amp = np.empty(flat.shape[0], np.float32)
    for i in range(flat.shape[0]):
        amp[i] = oracle.lib.orc_fft_logpower(C.byref(d), _p(flat[i]), n)
    return flat[:, :n // 2].reshape(x.shape[:-1] + (n // 2,)).copy(), amp.reshape(x.shape[:-1])
