"""Stream wrapper (host only): the three Vorbis header packets of a Setup and Ogg page framing —
vorbis_analysis_headerout (reference lib/info.c:636-717) and libogg's ogg_stream_packetin/_pageout
(page format: reference doc/framing.html) over the C ABI of include/vorbis_mi355x.h."""
import ctypes as C

from ._lib import lib, check


def header_packets(setup, comments=(), vendor=None):
    """-> [identification, comment, setup] packets (bytes) = packets 0..2 of a stream"""
    arr = (C.c_char_p * max(len(comments), 1))(*[c.encode() if isinstance(c, str) else c for c in comments])
    lens = (C.c_long * 3)()
    v = vendor.encode() if isinstance(vendor, str) else vendor
    check(lib.vbm_header_packets(setup._h, v, arr, len(comments), None, 0, lens), "vbm_header_packets")
    total = sum(lens)
    buf = (C.c_ubyte * total)()
    check(lib.vbm_header_packets(setup._h, v, arr, len(comments), buf, total, lens), "vbm_header_packets")
    raw = bytes(buf)
    return [raw[:lens[0]], raw[lens[0]:lens[0] + lens[1]], raw[lens[0] + lens[1]:]]


class OggStream:
    """One logical Ogg bitstream (ogg_stream_state)."""

    def __init__(self, serialno):
        self._h = C.c_void_p()
        check(lib.vbm_ogg_stream_create(C.byref(self._h), serialno), "vbm_ogg_stream_create")

    def packetin(self, packet, granulepos, eos=False):
        check(lib.vbm_ogg_stream_packetin(self._h, packet, len(packet), 1 if eos else 0, granulepos),
              "vbm_ogg_stream_packetin")

    def pageout(self, flush=False):
        """-> bytes of the next complete page, or None when none is due"""
        page, n = C.c_void_p(), C.c_long()
        rc = lib.vbm_ogg_stream_pageout(self._h, 1 if flush else 0, C.byref(page), C.byref(n))
        if rc < 0:
            check(rc, "vbm_ogg_stream_pageout")
        return C.string_at(page, n.value) if rc == 1 else None

    def pages(self, flush=False):
        out = []
        while True:
            p = self.pageout(flush)
            if p is None:
                return out
            out.append(p)

    def close(self):
        if self._h:
            lib.vbm_ogg_stream_destroy(self._h)
            self._h = C.c_void_p()


def write_ogg(setup, packets, infos, serialno=1, comments=()):
    """Headers + the audio packets of ONE stream (in order, with their (granulepos, eos)) -> .ogg bytes,
    paged as the reference application does (examples/encoder_example.c:139-157, 211-233)."""
    os_ = OggStream(serialno)
    out = []
    for i, h in enumerate(header_packets(setup, comments)):
        os_.packetin(h, 0)
    out += os_.pages(flush=True)                     # audio data starts on a fresh page
    for pkt, (granulepos, eos) in zip(packets, infos):
        os_.packetin(pkt, granulepos, eos)
        out += os_.pages()
    out += os_.pages(flush=True)
    os_.close()
    return b"".join(out)
