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


def mode_pack_name(ch, rate, q=None, bitrate=None):
    if bitrate is None:
        return f"mode_{ch}ch_{rate}_q{q:g}.vpk"
    mx, nom, mn = bitrate if isinstance(bitrate, (tuple, list)) else (-1, bitrate, -1)
    return f"mode_{ch}ch_{rate}_b{nom}" + (f"_max{mx}" if mx > 0 else "") + (f"_min{mn}" if mn > 0 else "") + ".vpk"


class Setup:
    """orc_setup for one (channels, rate, quality) class, from the committed mode pack."""

    def __init__(self, oracle, ch, rate, q=None, bitrate=None):
        """bitrate = nominal or (max, nominal, min) bits/s selects a managed-bitrate pack
        (vorbis_encode_init) instead of the VBR one"""
        self.o = oracle
        lib = oracle.lib
        lib.orc_setup_load.restype = C.c_void_p
        lib.orc_setup_load.argtypes = [C.c_char_p, C.c_char_p]
        lib.orc_encode_probe.restype = C.c_long
        lib.orc_encode_probe.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
        mode = _os.path.join(_DATA, mode_pack_name(ch, rate, q, bitrate))
        self.managed = bitrate is not None
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


class Stream:
    """One oracle encoder stream with stage capture (orc_stream + orc_block)."""

    def __init__(self, setup):
        self.setup = setup
        lib = self.lib = setup.o.lib
        lib.orc_stream_new.restype = C.c_void_p
        lib.orc_stream_new.argtypes = [C.c_void_p]
        lib.orc_block_new.restype = C.c_void_p
        lib.orc_block_new.argtypes = [C.c_void_p]
        lib.orc_analysis_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.orc_analysis_wrote.argtypes = [C.c_void_p, C.c_int]
        lib.orc_analysis_blockout.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_analysis.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_block_packet.restype = C.POINTER(C.c_ubyte)
        lib.orc_block_packet.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        lib.orc_block_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.orc_block_info64.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.orc_block_cap.restype = C.c_void_p
        lib.orc_block_cap.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.orc_stream_set_capture.argtypes = [C.c_void_p, C.c_int]
        lib.orc_stream_free.argtypes = [C.c_void_p]
        lib.orc_block_free.argtypes = [C.c_void_p]
        self.v = lib.orc_stream_new(setup.h)
        self.vb = lib.orc_block_new(setup.h)
        lib.orc_stream_set_capture(self.v, 1)
        self.ch = setup.ch

    def write(self, pcm):
        """pcm: (ch, n) float32"""
        pcm = np.ascontiguousarray(pcm, np.float32)
        n = pcm.shape[1]
        bufs = (C.c_void_p * 8)()
        self.lib.orc_analysis_buffer(self.v, n, bufs)
        for c in range(self.ch):
            C.memmove(bufs[c], pcm[c].ctypes.data, n * 4)
        self.lib.orc_analysis_wrote(self.v, n)

    def finish(self):
        """vorbis_analysis_wrote(vd, 0): end of stream"""
        self.lib.orc_analysis_wrote(self.v, 0)

    def _arr(self, name, c, count, dtype):
        p = self.lib.orc_block_cap(self.vb, name.encode(), c)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    def blocks(self):
        """Drain ready blocks: yields dicts with the raw block PCM, the block flags, every
        captured stage vector and the packet."""
        info = (C.c_int * 8)()
        while self.lib.orc_analysis_blockout(self.v, self.vb) == 1:
            self.lib.orc_block_info(self.vb, info)
            N = info[4]
            pcm = np.stack([self._arr("pcm", c, N, np.float32) for c in range(self.ch)])
            self.lib.orc_analysis(self.v, self.vb)
            self.lib.orc_block_info(self.vb, info)
            n = N // 2
            nb = C.c_long()
            pk = self.lib.orc_block_packet(self.vb, C.byref(nb))
            d = {"lW": info[0], "W": info[1], "nW": info[2], "blocktype": info[3], "N": N,
                 "block_mode": info[5], "eos": info[6], "pcm": pcm, "packet": bytes(pk[:nb.value])}
            i64 = (C.c_int64 * 2)()
            self.lib.orc_block_info64(self.vb, i64)
            d["granulepos"], d["sequence"] = int(i64[0]), int(i64[1])
            if self.setup.managed:
                sizes = (C.c_int * 15)()
                self.lib.orc_block_choice.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
                d["choice"] = self.lib.orc_block_choice(self.vb, sizes)
                d["blob_bytes"] = list(sizes)
                self.lib.orc_block_blob.restype = C.POINTER(C.c_ubyte)
                self.lib.orc_block_blob.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_long)]
                d["blobs"] = []
                for k in range(15):
                    bp = self.lib.orc_block_blob(self.vb, k, C.byref(nb))
                    d["blobs"].append(bytes(bp[:min(nb.value, sizes[k])]))
            for name in ("mdct_raw", "mdct", "logfft", "logmdct", "noise", "tone", "logmask", "epeak"):
                d[name] = np.stack([self._arr(name, c, n, np.float32) for c in range(self.ch)])
            for name in ("ilogmask", "residue"):
                d[name] = np.stack([self._arr(name, c, n, np.int32) for c in range(self.ch)])
            d["post"] = np.stack([self._arr("post", c, 65, np.int32) for c in range(self.ch)])
            d["post_valid"] = np.array([self._arr("post_valid", c, 1, np.int32)[0] for c in range(self.ch)])
            d["nonzero"] = np.array([self._arr("nonzero", c, 1, np.int32)[0] for c in range(self.ch)])
            d["local_ampmax"] = np.array([self._arr("local_ampmax", c, 1, np.float32)[0] for c in range(self.ch)])
            d["global_ampmax"] = self._arr("global_ampmax", 0, 1, np.float32)[0]
            yield d

    def close(self):
        self.lib.orc_stream_free(self.v)
        self.lib.orc_block_free(self.vb)


