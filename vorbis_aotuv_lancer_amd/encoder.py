"""Batched encoder — host mirror of the reference's per-block encode calls
(vorbis_analysis / vorbis_bitrate_addblock / vorbis_bitrate_flushpacket, reference
lib/analysis.c:29, lib/bitrate.c:73, :229) over the C ABI of include/vorbis_mi355x.h."""
import ctypes as C
import os

import numpy as np
import torch

from ._lib import lib, check

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def mode_pack_path(channels, rate, quality=None, bitrate=None):
    """bitrate: nominal bits/s or (max, nominal, min) as given to vorbis_encode_init (-1 = unset)"""
    if bitrate is None:
        return os.path.join(_DATA, f"mode_{channels}ch_{rate}_q{quality:g}.vpk")
    mx, nom, mn = bitrate if isinstance(bitrate, (tuple, list)) else (-1, bitrate, -1)
    name = f"mode_{channels}ch_{rate}_b{nom}" + (f"_max{mx}" if mx > 0 else "") + (f"_min{mn}" if mn > 0 else "")
    return os.path.join(_DATA, name + ".vpk")


class Setup:
    """codec_setup_info + looks for one (channels, rate, quality) class (vorbis_encode_init_vbr +
    vorbis_analysis_init in the reference) or, with bitrate=, one managed-bitrate class
    (vorbis_encode_init), loaded from the shipped mode pack."""

    def __init__(self, channels, rate, quality=None, bitrate=None):
        self.channels, self.rate, self.quality, self.bitrate = channels, rate, quality, bitrate
        self._h = C.c_void_p()
        path = mode_pack_path(channels, rate, quality, bitrate)
        if not os.path.exists(path):
            raise FileNotFoundError(f"no mode pack for {channels} ch / {rate} Hz / "
                                    f"{'q%g' % quality if bitrate is None else 'bitrate %s' % (bitrate,)}: {path} "
                                    "(generate one with tools/make_modepack.py)")
        check(lib.vbm_setup_create(C.byref(self._h), os.path.join(_DATA, "common.vpk").encode(), path.encode()),
              "vbm_setup_create")
        info = self.table("info")
        self.blocksizes = (int(info[2]), int(info[3]))

    def table(self, name):
        data, count, kind = C.c_void_p(), C.c_long(), C.c_char()
        check(lib.vbm_setup_table(self._h, name.encode(), C.byref(data), C.byref(count), C.byref(kind)),
              f"vbm_setup_table({name})")
        dt = {b"f": np.float32, b"i": np.int32, b"u": np.uint32, b"b": np.int8, b"d": np.float64}[kind.value]
        buf = (C.c_char * (count.value * np.dtype(dt).itemsize)).from_address(data.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def close(self):
        if self._h:
            lib.vbm_setup_destroy(self._h)
            self._h = C.c_void_p()


class Encoder:
    """Device-resident state of `nstreams` encoder streams + batch workspace."""

    def __init__(self, setup, nstreams, max_batch=None):
        self.setup = setup
        self.nstreams = nstreams
        self.max_batch = max_batch or nstreams
        self._h = C.c_void_p()
        check(lib.vbm_encoder_create(C.byref(self._h), setup._h, nstreams, self.max_batch), "vbm_encoder_create")
        self.max_packet_bytes = lib.vbm_encoder_max_packet_bytes(self._h)

    def reset(self):
        check(lib.vbm_encoder_reset(self._h), "vbm_encoder_reset")

    def analysis_batch(self, block_mode, stream_ids, wflags, pcm, back_stream=None, out=None):
        """pcm: CUDA float32 tensor [nsb, channels, blocksize]; returns (packets uint8 [nsb, max_bytes],
        nbytes int32 [nsb]) on the device.  back_stream (torch.cuda.Stream): run the second half of the path
        (floor, couple/quantise, packets) there, so that it overlaps the first half of the next call
        (vbm_analysis_batch2); the outputs are then ready on back_stream and must be passed in as `out`
        (tensors the caller keeps alive)."""
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        fl = np.ascontiguousarray(wflags, dtype=np.uint8)
        nsb = len(ids)
        N = self.setup.blocksizes[block_mode >> 1]
        if not (pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous()
                and tuple(pcm.shape) == (nsb, self.setup.channels, N)):
            raise ValueError(f"pcm must be a contiguous CUDA float32 tensor of shape ({nsb}, {self.setup.channels}, {N})")
        if out is not None:
            packets, nbytes = out
        elif back_stream is not None:
            raise ValueError("back_stream needs caller-owned output tensors (out=...)")
        else:
            packets = torch.empty((nsb, self.max_packet_bytes), dtype=torch.uint8, device=pcm.device)
            nbytes = torch.empty((nsb,), dtype=torch.int32, device=pcm.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        sb = st if back_stream is None else C.c_void_p(back_stream.cuda_stream)
        check(lib.vbm_analysis_batch2(self._h, block_mode, nsb, ids.ctypes.data, fl.ctypes.data, pcm.data_ptr(),
                                      packets.data_ptr(), nbytes.data_ptr(), st, sb), "vbm_analysis_batch2")
        self._last = (nsb, pcm.device)
        return packets, nbytes

    def fetch(self, name):
        """Intermediate of the last batch as a CUDA tensor, block-major."""
        nsb, dev = self._last
        rows, kind = C.c_long(), C.c_char()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_encoder_fetch(self._h, name.encode(), None, C.byref(rows), C.byref(kind), st), "vbm_encoder_fetch")
        dt = torch.float32 if kind.value == b"f" else torch.int32
        per_sb = name in ("global_ampmax", "packet_bytes", "choice")
        count = nsb if per_sb else nsb * self.setup.channels
        shape = (count,) if rows.value == 1 else (count, rows.value)
        out = torch.empty(shape, dtype=dt, device=dev)
        check(lib.vbm_encoder_fetch(self._h, name.encode(), out.data_ptr(), C.byref(rows), C.byref(kind), st),
              "vbm_encoder_fetch")
        return out

    def fetch_blob(self, k):
        """Managed bitrate: packetblob k of the last batch before the bitrate manager chose
        (packets uint8 [nsb, max_bytes], nbytes int32 [nsb])."""
        nsb, dev = self._last
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        packets = torch.empty((nsb, self.max_packet_bytes), dtype=torch.uint8, device=dev)
        nbytes = torch.empty((nsb,), dtype=torch.int32, device=dev)
        check(lib.vbm_encoder_fetch_blob(self._h, k, packets.data_ptr(), nbytes.data_ptr(), st), "vbm_encoder_fetch_blob")
        return packets, nbytes

    def debug_poison(self, w=-1, byte=0xFF):
        """test instrumentation: fill the scratch arrays of workspace w (-1: all) with `byte` (device idle)"""
        check(lib.vbm_debug_poison_workspace(self._h, w, byte), "vbm_debug_poison_workspace")

    def set_sub_batches(self, n):
        """Slices of a batch that run on separate internal HIP streams after the transforms."""
        check(lib.vbm_encoder_set_sub_batches(self._h, n), "vbm_encoder_set_sub_batches")

    @property
    def sub_batches(self):
        return lib.vbm_encoder_sub_batches(self._h)

    def profile_begin(self, max_calls):
        check(lib.vbm_encoder_profile_begin(self._h, max_calls), "vbm_encoder_profile_begin")

    def profile_end(self):
        """-> ({stage name: total ms}, calls covered)"""
        n = lib.vbm_encoder_stage_count()
        self.profile_blocks = int(lib.vbm_encoder_profile_blocks(self._h))
        ms = (C.c_float * n)()
        calls = C.c_int()
        check(lib.vbm_encoder_profile_end(self._h, ms, C.byref(calls)), "vbm_encoder_profile_end")
        return {lib.vbm_encoder_stage_name(k).decode(): float(ms[k]) for k in range(n)}, calls.value

    def close(self):
        if self._h:
            lib.vbm_encoder_destroy(self._h)
            self._h = C.c_void_p()


class PacketInfo(C.Structure):
    """vbm_packet_info (include/vorbis_mi355x.h)"""
    _fields_ = [("stream", C.c_int), ("block_mode", C.c_int), ("lW", C.c_int), ("W", C.c_int), ("nW", C.c_int),
                ("eos", C.c_int), ("granulepos", C.c_longlong), ("packetno", C.c_longlong)]


class FrontEnd:
    """Stream front end of an Encoder: vorbis_analysis_buffer/_wrote/_blockout (+ envelope detector) for
    all its streams on the device, feeding vbm_analysis_batch (reference lib/block.c:405-812,
    lib/envelope.c; examples/encoder_example.c:190-235 is the loop this mirrors)."""

    def __init__(self, encoder):
        self.enc = encoder
        self._h = C.c_void_p()
        check(lib.vbm_frontend_create(C.byref(self._h), encoder._h), "vbm_frontend_create")
        self._info = (PacketInfo * encoder.nstreams)()
        # zero-copy record view of the info array: fields stream, block_mode, lW, W, nW, eos, granulepos, packetno
        self._info_view = np.ctypeslib.as_array(self._info)

    def reset(self):
        check(lib.vbm_frontend_reset(self._h), "vbm_frontend_reset")

    def write(self, pcm):
        """pcm: CUDA float32 [nstreams, channels, vals] — `vals` new samples for every stream."""
        S, ch = self.enc.nstreams, self.enc.setup.channels
        if not (pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous() and pcm.dim() == 3
                and pcm.shape[0] == S and pcm.shape[1] == ch):
            raise ValueError(f"pcm must be a contiguous CUDA float32 tensor of shape ({S}, {ch}, vals)")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_write(self._h, pcm.data_ptr(), int(pcm.shape[2]), st), "vbm_frontend_write")

    def write_streams(self, stream_ids, pcm):
        """pcm: CUDA float32 [len(stream_ids), channels, vals] — new samples for the listed streams only."""
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        ch = self.enc.setup.channels
        if not (pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous() and pcm.dim() == 3
                and pcm.shape[0] == len(ids) and pcm.shape[1] == ch):
            raise ValueError(f"pcm must be a contiguous CUDA float32 tensor of shape ({len(ids)}, {ch}, vals)")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_write_streams(self._h, ids.ctypes.data, len(ids), pcm.data_ptr(), int(pcm.shape[2]), st),
              "vbm_frontend_write_streams")

    def write_streams_strided(self, stream_ids, pcm, vals, stream_stride, ch_stride, by_slot=False):
        """vbm_frontend_write_streams_strided: channel c of stream_ids[k] at pcm + (stream_ids[k] if by_slot else k) *
        stream_stride + c * ch_stride floats; pcm: a CUDA float32 tensor (or pinned host memory the device can read)."""
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_write_streams_strided(self._h, ids.ctypes.data, len(ids), pcm.data_ptr(), int(vals),
                                                     int(stream_stride), int(ch_stride), int(bool(by_slot)), st),
              "vbm_frontend_write_streams_strided")

    def restart_streams(self, stream_ids):
        """a new stream starts in each listed slot"""
        ids = np.ascontiguousarray(stream_ids, dtype=np.int32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_restart_streams(self._h, ids.ctypes.data, len(ids), st), "vbm_frontend_restart_streams")

    def finish(self, stream_ids=None):
        """vorbis_analysis_wrote(vd, 0) for the listed streams (default: all)."""
        ids = np.arange(self.enc.nstreams, dtype=np.int32) if stream_ids is None else \
            np.ascontiguousarray(stream_ids, dtype=np.int32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_finish(self._h, ids.ctypes.data, len(ids), st), "vbm_frontend_finish")

    @property
    def max_buffered(self):
        return lib.vbm_frontend_max_buffered(self._h)

    @property
    def capacity(self):
        return lib.vbm_frontend_capacity(self._h)

    def encode_round(self, device=None):
        """One blockout round over all streams.  Returns (info records, packets uint8 [n, max_bytes], nbytes int32
        [n]) for the n blocks that came out (n may be 0: every stream needs more PCM).  The info records are a
        numpy structured view (fields of vbm_packet_info) that the next round overwrites."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        S = self.enc.nstreams
        packets = torch.empty((S, self.enc.max_packet_bytes), dtype=torch.uint8, device=dev)
        nbytes = torch.empty((S,), dtype=torch.int32, device=dev)
        n = C.c_int()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_encode_round(self._h, packets.data_ptr(), nbytes.data_ptr(), self._info, C.byref(n), st),
              "vbm_frontend_encode_round")
        k = n.value
        return self._info_view[:k], packets[:k], nbytes[:k]

    def join(self, stream=None):
        """vbm_frontend_join: `stream` (default: the current one) waits for everything begun so far (lazy calls included)"""
        st = stream if stream is not None else torch.cuda.current_stream()
        check(lib.vbm_frontend_join(self._h, C.c_void_p(st.cuda_stream)), "vbm_frontend_join")

    def encode_rounds(self, min_rounds=1, max_rounds=8, headroom=1024, device=None, lazy=False, cap_blocks=None):
        """Up to max_rounds blockout rounds in one call (vbm_frontend_encode_rounds: a round runs beside the
        long-block batch of the round before it; everything is joined at the end).  Returns (info records,
        packets uint8 [n, max_bytes], nbytes int32 [n], blocks per round) over all rounds, in round order.

        The outputs live in a ring of three buffer sets owned by this object (cap_blocks slots each, default
        nstreams * min(max_rounds, 4): rounds stop when fewer than nstreams slots are left): what a call returns
        stays valid until the third call after it.  That also covers lazy=True, where the device still writes a
        call's packets while the next call is being enqueued (vbm_frontend_encode_rounds_lazy): the buffers are
        never handed back to torch's allocator while work from a non-torch stream is pending on them."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        S = self.enc.nstreams
        cap = cap_blocks or S * min(max_rounds, 4)
        ring = getattr(self, "_ring", None)
        if ring is None or self._ring_cap < cap or self._ring_dev != dev:
            if ring is not None:
                self.join()
                torch.cuda.synchronize(dev)
            self._ring = [((PacketInfo * cap)(), torch.empty((cap, self.enc.max_packet_bytes), dtype=torch.uint8, device=dev),
                           torch.empty((cap,), dtype=torch.int32, device=dev)) for _ in range(3)]
            self._ring_views = [np.ctypeslib.as_array(r[0]) for r in self._ring]
            self._ring_cap, self._ring_dev, self._ring_at = cap, dev, 0
        at = self._ring_at
        self._ring_at = (at + 1) % 3
        info, packets, nbytes = self._ring[at]
        per_round = (C.c_int * max_rounds)()
        nr = C.c_int()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn = lib.vbm_frontend_encode_rounds_lazy if lazy else lib.vbm_frontend_encode_rounds
        check(fn(self._h, min_rounds, max_rounds, headroom, packets.data_ptr(), nbytes.data_ptr(),
                 info, cap, per_round, C.byref(nr), st), "vbm_frontend_encode_rounds")
        counts = [per_round[r] for r in range(nr.value)]
        k = sum(counts)
        return self._ring_views[at][:k], packets[:k], nbytes[:k], counts

    @property
    def device_lanes(self):
        """output slots of one device-built round (vbm_device_round_lanes): the Encoder needs max_batch >= this"""
        return lib.vbm_device_round_lanes(self.enc.setup._h, self.enc.nstreams)

    def encode_rounds_device(self, nrounds=2, lazy=False, device=None):
        """vbm_frontend_encode_rounds_device: `nrounds` rounds built and run on the device, nothing read back.
        Returns device tensors (info uint8 [nrounds * lanes, 40] = vbm_packet_info records, packets uint8
        [nrounds * lanes, max_bytes], nbytes int32 [nrounds * lanes] with -2 = empty lane, counts int32 [nrounds, 4]),
        complete on the current stream when the call's work has run (lazy: the call's long-block batch one call
        later; lazy=2: not tied to the current stream at all — a consumer calls join(stream) before it reads).  They
        live in a ring of three sets per round count owned by this object."""
        dev = device or torch.device("cuda", torch.cuda.current_device())
        lanes = self.device_lanes
        rings = getattr(self, "_drings", None)
        if rings is None:
            rings = self._drings = {}
            self._dring_dev = dev
        if (nrounds, dev) not in rings:      # one ring of three sets per round count (callers may vary it from call to call)
            n = nrounds * lanes
            rings[(nrounds, dev)] = [[(torch.zeros((n, C.sizeof(PacketInfo)), dtype=torch.uint8, device=dev),
                                       torch.empty((n, self.enc.max_packet_bytes), dtype=torch.uint8, device=dev),
                                       torch.full((n,), -2, dtype=torch.int32, device=dev),
                                       torch.zeros((nrounds, 4), dtype=torch.int32, device=dev)) for _ in range(3)], 0]
        ring = rings[(nrounds, dev)]
        at = ring[1]
        ring[1] = (at + 1) % 3
        info, packets, nbytes, counts = ring[0][at]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.vbm_frontend_encode_rounds_device(self._h, nrounds, packets.data_ptr(), nbytes.data_ptr(), info.data_ptr(),
                                                    counts.data_ptr(), int(lazy), st),
              "vbm_frontend_encode_rounds_device")
        return info, packets, nbytes, counts

    def debug_poison(self, byte=0xFF):
        """test instrumentation: fill the front end's scratch (block buffers, search spectra, round lists) with `byte`"""
        check(lib.vbm_debug_poison_frontend(self._h, byte), "vbm_debug_poison_frontend")

    def device_stats(self):
        """running totals of the device-built rounds: ([blocks of type 0..3], samples all streams advanced by)"""
        out = (C.c_ulonglong * 6)()
        check(lib.vbm_frontend_device_stats(self._h, out), "vbm_frontend_device_stats")
        self.refused_writes = int(out[5])      # must stay 0 (a stream's buffer was full: samples lost)
        return [int(out[k]) for k in range(4)], int(out[4])

    def close(self):
        if self._h:
            if getattr(self, "_drings", None) is not None:
                self.join()
                torch.cuda.synchronize(self._dring_dev)
                self._drings = None
            if getattr(self, "_ring", None) is not None:      # nothing may still be writing the ring
                self.join()
                torch.cuda.synchronize(self._ring_dev)
                self._ring = None
            lib.vbm_frontend_destroy(self._h)
            self._h = C.c_void_p()
