"""The reference's internal plugin seam on the device: vorbis_func_mapping.forward(vorbis_block *) = mapping0_forward
(reference lib/backends.h:121-128, lib/mapping0.c:1500-1506, lib/registry.c:42-44; called from lib/analysis.c:45).
A libvorbis that keeps its own blockout hands every carved block — host PCM, window flags, block type in its
vorbis_block_internal — to include/vorbis_compat.h: mapping0_exportbundle_mi355x.forward.  Here the oracle plays that
libvorbis: its blockout carves the blocks, the device encodes them one by one through the bundle, and the packets must be
the oracle's own mapping0_forward output, byte for byte, including what lands in vbi->packetblob[PACKETBLOBS/2]."""
import ctypes as C

import numpy as np
import pytest

from tests import compat, orc
from tests.signals import burst_signal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ch,rate,q", [(2, 44100, 0.5), (6, 48000, 0.8), (1, 44100, 0.5)])
def test_mapping0_forward_bundle(oracle, cuda, ch, rate, q):
    import vorbis_aotuv_lancer_amd as v
    dll = compat.bind(C.CDLL(v.COMPAT_LIB_PATH))
    bundle = compat.VorbisFuncMapping.in_dll(dll, "mapping0_exportbundle_mi355x")
    assert not bundle.pack and not bundle.unpack and not bundle.inverse
    nstreams, nchunks = 3, 24
    sigs = [burst_signal(ch, rate, nchunks * 1024, seed=300 + k, period=7000) for k in range(nstreams)]
    osetup = orc.Setup(oracle, ch, rate, q)
    streams = [compat.Stream(dll, ch, rate, q) for _ in range(nstreams)]      # three device slots of one pool
    ostreams = []
    for k in range(nstreams):
        st = orc.Stream(osetup)
        ostreams.append(st)
    modes = set()
    nblocks = 0
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    blob = compat.OggpackBuffer()                 # the caller's packetblob[7]: grown by the callee with realloc
    for c in range(nchunks + 1):
        for k in range(nstreams):                 # the streams' blocks interleave: each keeps its own device state
            st = ostreams[k]
            if c < nchunks:
                st.write(sigs[k][:, c * 1024:(c + 1) * 1024])
            else:
                st.finish()
            for b in st.blocks():                 # the oracle's blockout carved b["pcm"]; its own packet is b["packet"]
                vb = compat.VorbisBlock()
                vbi = compat.VorbisBlockInternal()
                vbi.blocktype = b["blocktype"]
                vbi.packetblob[7] = C.pointer(blob)
                pcm = np.ascontiguousarray(b["pcm"], np.float32)
                rows = (C.c_void_p * ch)(*[pcm[i].ctypes.data for i in range(ch)])
                vb.pcm = C.cast(rows, C.c_void_p)
                vb.lW, vb.W, vb.nW, vb.pcmend = b["lW"], b["W"], b["nW"], b["N"]
                vb.vd = C.pointer(streams[k].vd)
                vb.internal = C.cast(C.pointer(vbi), C.c_void_p)
                assert bundle.forward(C.byref(vb)) == 0
                got = C.string_at(vb.opb.buffer, vb.opb.endbyte)
                assert got == b["packet"], f"stream {k} block {nblocks}: packet differs from mapping0_forward's"
                assert C.string_at(blob.buffer, blob.endbyte) == b["packet"] and blob.endbit == 0
                modes.add(b["block_mode"])
                nblocks += 1
    assert modes == {0, 1, 2, 3} or (ch == 1 and len(modes) >= 3), modes
    assert nblocks > nstreams * nchunks
    # malformed blocks are refused (OV_EINVAL), not encoded
    vb = compat.VorbisBlock()
    assert dll.vbm_mapping0_forward(C.byref(vb)) == compat.OV_EINVAL
    libc.free(blob.buffer)
    for st in ostreams:
        st.close()
    for s in streams:
        s.close()
