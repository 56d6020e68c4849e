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
