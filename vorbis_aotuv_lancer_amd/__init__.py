"""vorbis_aotuv_lancer_amd — MI355X-native batched Vorbis (aoTuV) encode path.

Host-side mirror (Python) of the reference's per-block encode interface over the C ABI in
include/vorbis_mi355x.h.  PyTorch is used only as plumbing (device memory, streams,
torch.distributed); every transform runs in the hand-written gfx950 kernels of
libvorbis_mi355x.so.  There is no CPU fallback: importing works anywhere, but any compute
call raises if the HIP library or a GPU is missing.
"""
from ._lib import lib, LIB_PATH, COMPAT_LIB_PATH, VbmError, check  # noqa: F401
from .tables import window_table  # noqa: F401
from .mdct import MdctLookup, mdct_forward, window_mdct, window_fft_log  # noqa: F401

from .encoder import Setup, Encoder, FrontEnd, PacketInfo  # noqa: F401,E402
from .stream import header_packets, OggStream, write_ogg  # noqa: F401,E402

__all__ = ["Setup", "Encoder", "FrontEnd", "PacketInfo", "header_packets", "OggStream", "write_ogg", "lib", "LIB_PATH", "COMPAT_LIB_PATH", "VbmError", "check", "window_table",
           "MdctLookup", "mdct_forward", "window_mdct", "window_fft_log"]
