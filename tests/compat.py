"""ctypes mirror of include/vorbis_compat.h (the reference's public structs, include/vorbis/codec.h:27-149) for the
tests: the calls below are the reference application's own call sequence, made through the C ABI."""
import ctypes as C

import numpy as np

OV_EINVAL, OV_EIMPL = -131, -130


class OggpackBuffer(C.Structure):
    _fields_ = [("endbyte", C.c_long), ("endbit", C.c_int), ("buffer", C.c_void_p), ("ptr", C.c_void_p),
                ("storage", C.c_long)]


class OggPacket(C.Structure):
    _fields_ = [("packet", C.POINTER(C.c_ubyte)), ("bytes", C.c_long), ("b_o_s", C.c_long), ("e_o_s", C.c_long),
                ("granulepos", C.c_int64), ("packetno", C.c_int64)]


class VorbisInfo(C.Structure):
    _fields_ = [("version", C.c_int), ("channels", C.c_int), ("rate", C.c_long), ("bitrate_upper", C.c_long),
                ("bitrate_nominal", C.c_long), ("bitrate_lower", C.c_long), ("bitrate_window", C.c_long),
                ("codec_setup", C.c_void_p)]


class VorbisDspState(C.Structure):
    _fields_ = [("analysisp", C.c_int), ("vi", C.POINTER(VorbisInfo)), ("pcm", C.c_void_p), ("pcmret", C.c_void_p),
                ("pcm_storage", C.c_int), ("pcm_current", C.c_int), ("pcm_returned", C.c_int),
                ("preextrapolate", C.c_int), ("eofflag", C.c_int), ("lW", C.c_long), ("W", C.c_long), ("nW", C.c_long),
                ("centerW", C.c_long), ("granulepos", C.c_int64), ("sequence", C.c_int64), ("glue_bits", C.c_int64),
                ("time_bits", C.c_int64), ("floor_bits", C.c_int64), ("res_bits", C.c_int64),
                ("backend_state", C.c_void_p)]


class VorbisBlock(C.Structure):
    _fields_ = [("pcm", C.c_void_p), ("opb", OggpackBuffer), ("lW", C.c_long), ("W", C.c_long), ("nW", C.c_long),
                ("pcmend", C.c_int), ("mode", C.c_int), ("eofflag", C.c_int), ("granulepos", C.c_int64),
                ("sequence", C.c_int64), ("vd", C.POINTER(VorbisDspState)), ("localstore", C.c_void_p),
                ("localtop", C.c_long), ("localalloc", C.c_long), ("totaluse", C.c_long), ("reap", C.c_void_p),
                ("glue_bits", C.c_long), ("time_bits", C.c_long), ("floor_bits", C.c_long), ("res_bits", C.c_long),
                ("internal", C.c_void_p)]


class VorbisComment(C.Structure):
    _fields_ = [("user_comments", C.POINTER(C.c_char_p)), ("comment_lengths", C.POINTER(C.c_int)),
                ("comments", C.c_int), ("vendor", C.c_char_p)]


class RateManage2(C.Structure):
    """struct ovectl_ratemanage2_arg (include/vorbis/vorbisenc.h)"""
    _fields_ = [("management_active", C.c_int), ("bitrate_limit_min_kbps", C.c_long), ("bitrate_limit_max_kbps", C.c_long),
                ("bitrate_limit_reservoir_bits", C.c_long), ("bitrate_limit_reservoir_bias", C.c_double),
                ("bitrate_average_kbps", C.c_long), ("bitrate_average_damping", C.c_double)]


def bind(dll):
    """argtypes / restypes of the entry points the tests call"""
    P = C.POINTER
    dll.vorbis_info_init.argtypes = [P(VorbisInfo)]
    dll.vorbis_info_init.restype = None
    dll.vorbis_info_clear.argtypes = [P(VorbisInfo)]
    dll.vorbis_info_clear.restype = None
    dll.vorbis_info_blocksize.argtypes = [P(VorbisInfo), C.c_int]
    dll.vorbis_encode_init_vbr.argtypes = [P(VorbisInfo), C.c_long, C.c_long, C.c_float]
    dll.vorbis_encode_init.argtypes = [P(VorbisInfo), C.c_long, C.c_long, C.c_long, C.c_long, C.c_long]
    dll.vorbis_comment_init.argtypes = [P(VorbisComment)]
    dll.vorbis_comment_init.restype = None
    dll.vorbis_comment_add_tag.argtypes = [P(VorbisComment), C.c_char_p, C.c_char_p]
    dll.vorbis_comment_add_tag.restype = None
    dll.vorbis_comment_query.argtypes = [P(VorbisComment), C.c_char_p, C.c_int]
    dll.vorbis_comment_query.restype = C.c_char_p
    dll.vorbis_comment_query_count.argtypes = [P(VorbisComment), C.c_char_p]
    dll.vorbis_comment_clear.argtypes = [P(VorbisComment)]
    dll.vorbis_comment_clear.restype = None
    dll.vorbis_analysis_init.argtypes = [P(VorbisDspState), P(VorbisInfo)]
    dll.vorbis_block_init.argtypes = [P(VorbisDspState), P(VorbisBlock)]
    dll.vorbis_block_clear.argtypes = [P(VorbisBlock)]
    dll.vorbis_dsp_clear.argtypes = [P(VorbisDspState)]
    dll.vorbis_dsp_clear.restype = None
    dll.vorbis_analysis_headerout.argtypes = [P(VorbisDspState), P(VorbisComment), P(OggPacket), P(OggPacket), P(OggPacket)]
    dll.vorbis_analysis_buffer.argtypes = [P(VorbisDspState), C.c_int]
    dll.vorbis_analysis_buffer.restype = P(P(C.c_float))
    dll.vorbis_analysis_wrote.argtypes = [P(VorbisDspState), C.c_int]
    dll.vorbis_analysis_blockout.argtypes = [P(VorbisDspState), P(VorbisBlock)]
    dll.vorbis_analysis.argtypes = [P(VorbisBlock), P(OggPacket)]
    dll.vorbis_bitrate_addblock.argtypes = [P(VorbisBlock)]
    dll.vorbis_bitrate_flushpacket.argtypes = [P(VorbisDspState), P(OggPacket)]
    dll.vorbis_mi355x_ctl.argtypes = [C.c_int, C.c_void_p]
    dll.vorbis_encode_setup_vbr.argtypes = [P(VorbisInfo), C.c_long, C.c_long, C.c_float]
    dll.vorbis_encode_setup_managed.argtypes = [P(VorbisInfo), C.c_long, C.c_long, C.c_long, C.c_long, C.c_long]
    dll.vorbis_encode_setup_init.argtypes = [P(VorbisInfo)]
    dll.vorbis_encode_ctl.argtypes = [P(VorbisInfo), C.c_int, C.c_void_p]
    dll.vorbis_commentheader_out.argtypes = [P(VorbisComment), P(OggPacket)]
    return dll


class Stream:
    """One reference-API encoder stream: vorbis_info + vorbis_dsp_state + vorbis_block, driven like
    examples/encoder_example.c drives them."""

    def __init__(self, dll, ch, rate, q=None, bitrate=None, three_step=False):
        """three_step: vorbis_encode_setup_vbr / _managed + vorbis_encode_ctl + vorbis_encode_setup_init, as oggenc does"""
        self.dll, self.ch = dll, ch
        self.vi, self.vd, self.vb = VorbisInfo(), VorbisDspState(), VorbisBlock()
        dll.vorbis_info_init(self.vi)
        if bitrate is None:
            rc = (dll.vorbis_encode_setup_vbr if three_step else dll.vorbis_encode_init_vbr)(self.vi, ch, rate, q)
        else:
            mx, nom, mn = bitrate if isinstance(bitrate, (tuple, list)) else (-1, bitrate, -1)
            rc = (dll.vorbis_encode_setup_managed if three_step else dll.vorbis_encode_init)(self.vi, ch, rate, mx, nom, mn)
        assert rc == 0, rc
        if three_step:
            if bitrate is None:
                assert dll.vorbis_encode_ctl(self.vi, 0x15, None) == 0       # OV_ECTL_RATEMANAGE2_SET, NULL: no management
            assert dll.vorbis_encode_setup_init(self.vi) == 0
        assert dll.vorbis_analysis_init(self.vd, self.vi) == 0
        assert dll.vorbis_block_init(self.vd, self.vb) == 0

    def write(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.float32)
        n = pcm.shape[1]
        buf = self.dll.vorbis_analysis_buffer(self.vd, n)
        for c in range(self.ch):
            C.memmove(buf[c], pcm[c].ctypes.data, n * 4)
        return self.dll.vorbis_analysis_wrote(self.vd, n)

    def finish(self):
        return self.dll.vorbis_analysis_wrote(self.vd, 0)

    def blockout(self):
        return self.dll.vorbis_analysis_blockout(self.vd, self.vb)

    def packets_of_block(self):
        """vorbis_analysis + vorbis_bitrate_addblock + flushpacket loop for the block blockout just returned"""
        out = []
        assert self.dll.vorbis_analysis(self.vb, None) == 0
        assert self.dll.vorbis_bitrate_addblock(self.vb) == 0
        op = OggPacket()
        while self.dll.vorbis_bitrate_flushpacket(self.vd, op):
            out.append(((int(self.vb.lW), int(self.vb.W), int(self.vb.nW), int(op.e_o_s), int(op.granulepos),
                         int(op.packetno)), bytes(op.packet[:op.bytes])))
        return out

    def drain(self):
        out = []
        while self.blockout() == 1:
            out.extend(self.packets_of_block())
        return out

    def close(self):
        self.dll.vorbis_block_clear(self.vb)
        self.dll.vorbis_dsp_clear(self.vd)
        self.dll.vorbis_info_clear(self.vi)


class VorbisBlockInternal(C.Structure):
    """leading members of the reference's vorbis_block_internal (lib/codec_internal.h:42-49)"""
    _fields_ = [("pcmdelay", C.c_void_p), ("ampmax", C.c_float), ("blocktype", C.c_int),
                ("packetblob", C.POINTER(OggpackBuffer) * 15)]


class VorbisFuncMapping(C.Structure):
    """vorbis_func_mapping (lib/backends.h:121-128)"""
    _fields_ = [("pack", C.c_void_p), ("unpack", C.c_void_p), ("free_info", C.c_void_p),
                ("forward", C.CFUNCTYPE(C.c_int, C.POINTER(VorbisBlock))), ("inverse", C.c_void_p)]
