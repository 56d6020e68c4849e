"""ctypes binding of libvorbis_mi355x.so (C ABI: include/vorbis_mi355x.h)."""
import ctypes as C
import os

# torch first: it brings its own HIP runtime (torch/lib/libamdhip64.so).  Loaded after it, this library binds to
# that runtime and shares device, streams and allocations with torch; loaded before it, the process would hold two
# HIP runtimes and this library's calls would go to one that torch's streams and pointers mean nothing to.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvorbis_mi355x.so")
# the reference's own entry points (include/vorbis_compat.h: vorbis_analysis, vorbis_bitrate_*, ogg_stream_* ...) are a
# separate shim library over this one: nothing here loads it, a process gets those global names only by asking for them
COMPAT_LIB_PATH = os.path.join(_HERE, "libvorbis_mi355x_compat.so")


class VbmError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise VbmError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the encode path.")
    return C.CDLL(LIB_PATH)


lib = _load()

_c_float_p = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/vorbis_mi355x.h declares
SIGNATURES = {
    "vbm_version": (C.c_char_p, []),
    "vbm_device_count": (C.c_int, []),
    "vbm_last_error": (C.c_char_p, []),
    "vbm_mdct_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vbm_mdct_plan_destroy": (None, [C.c_void_p]),
    "vbm_mdct_plan_trig": (_c_float_p, [C.c_void_p]),
    "vbm_mdct_forward_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "vbm_window_mdct_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "vbm_window_fft_log_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                           C.c_void_p]),
    "vbm_mdct_plan_fft_twiddles": (_c_float_p, [C.c_void_p]),
    "vbm_setup_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p]),
    "vbm_setup_destroy": (None, [C.c_void_p]),
    "vbm_setup_table": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_long),
                                  C.POINTER(C.c_char)]),
    "vbm_encoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "vbm_encoder_destroy": (None, [C.c_void_p]),
    "vbm_encoder_reset": (C.c_int, [C.c_void_p]),
    "vbm_encoder_max_packet_bytes": (C.c_int, [C.c_void_p]),
    "vbm_analysis_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "vbm_analysis_batch2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbm_analysis_round": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "vbm_encoder_fetch": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_char),
                                    C.c_void_p]),
    "vbm_encoder_fetch_blob": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbm_encoder_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "vbm_encoder_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "vbm_encoder_profile_blocks": (C.c_longlong, [C.c_void_p]),
    "vbm_encoder_set_sub_batches": (C.c_int, [C.c_void_p, C.c_int]),
    "vbm_encoder_sub_batches": (C.c_int, [C.c_void_p]),
    "vbm_frontend_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "vbm_frontend_destroy": (None, [C.c_void_p]),
    "vbm_frontend_reset": (C.c_int, [C.c_void_p]),
    "vbm_frontend_write": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vbm_frontend_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vbm_frontend_write_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vbm_comment_packet": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p]),
    "vbm_frontend_round_types": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbm_frontend_write_streams_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_long,
                                                     C.c_int, C.c_void_p]),
    "vbm_frontend_restart_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vbm_frontend_max_buffered": (C.c_int, [C.c_void_p]),
    "vbm_frontend_capacity": (C.c_int, [C.c_void_p]),
    "vbm_frontend_encode_round": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "vbm_frontend_encode_round_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.POINTER(C.c_int), C.c_void_p]),
    "vbm_packets_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vbm_frontend_encode_rounds": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    "vbm_analysis_round_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "vbm_analysis_round_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vbm_analysis_round_join_lazy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vbm_frontend_encode_rounds_lazy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    "vbm_frontend_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vbm_device_round_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "vbm_frontend_encode_rounds_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_int, C.c_void_p]),
    "vbm_frontend_device_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "vbm_analysis_round_wait_workspace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vbm_header_packets": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]),
    "vbm_ogg_stream_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "vbm_ogg_stream_destroy": (None, [C.c_void_p]),
    "vbm_ogg_stream_packetin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_longlong]),
    "vbm_ogg_stream_pageout": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_long)]),
    "vbm_encoder_stage_count": (C.c_int, []),
    "vbm_encoder_stage_name": (C.c_char_p, [C.c_int]),
    "vbm_host_mdct_trig": (C.c_int, [C.c_int, C.c_void_p]),
    "vbm_host_fft_twiddles": (C.c_int, [C.c_int, C.c_void_p]),
    "vbm_host_book_lattice": (C.c_int, [C.c_long, C.c_long, C.c_long, C.c_int, C.POINTER(C.c_int)]),
    "vbm_debug_set_delay": (C.c_int, [C.c_uint, C.c_int]),
    "vbm_debug_poison_workspace": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "vbm_debug_poison_frontend": (C.c_int, [C.c_void_p, C.c_int]),
    "vbm_window_mdct_time": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_float)]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def check(rc, what=""):
    if rc != 0:
        msg = lib.vbm_last_error().decode()
        raise VbmError(f"{what} failed with code {rc}" + (f": {msg}" if msg else ""))
