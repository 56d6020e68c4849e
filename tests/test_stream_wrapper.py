"""Stream wrapper (host only — runs without a GPU).

Header packets: the product packs them like the reference's ENCODE side (vorbis_analysis_headerout).  The
check reads them back with a restatement of the reference's DECODE side — _vorbis_unpack_info/_comment/_books
(lib/info.c:203-430), vorbis_staticbook_unpack (lib/codebook.c:277-395), floor1_unpack (lib/floor1.c:119-180),
res0_unpack (lib/res0.c:191-249), mapping0_unpack (lib/mapping0.c:95-160), including their validity checks —
and compares every field with the mode pack.  No reference header bytes exist to compare with
(parity with the reference unpinned); pack and unpack are different code in the reference, so the round
trip is an independent check.

Ogg pages: parsed back per the reference's doc/framing.html with an independent CRC (bitwise, no table)."""
import glob
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vpk  # noqa: E402

DATA = os.path.join(ROOT, "vorbis_aotuv_lancer_amd", "data")


class BitReader:
    """oggpack_read: LSb first"""

    def __init__(self, data):
        self.d, self.pos = data, 0

    def read(self, bits):
        v = 0
        for i in range(bits):
            byte = self.pos >> 3
            if byte >= len(self.d):
                raise EOFError
            v |= ((self.d[byte] >> (self.pos & 7)) & 1) << i
            self.pos += 1
        return v

    def bytes_left(self):
        return len(self.d) - ((self.pos + 7) >> 3)


def ilog(v):
    return int(v).bit_length()


def maptype1_quantvals(entries, dim):
    vals = max(int(np.floor(np.float32(entries) ** (np.float32(1.) / np.float32(dim)))), 1)
    while True:
        acc, acc1, ok = 1, 1, True
        for _ in range(dim):
            if entries // vals < acc:
                ok = False
                break
            acc *= vals
            acc1 *= vals + 1
        if ok and acc <= entries and acc1 > entries:
            return vals
        vals += -1 if (not ok or acc > entries) else 1


def unpack_book(r):
    assert r.read(24) == 0x564342
    dim, entries = r.read(16), r.read(24)
    assert ilog(dim) + ilog(entries) <= 24
    if r.read(1) == 0:
        unused = r.read(1)
        lengths = []
        for _ in range(entries):
            if unused:
                lengths.append(r.read(5) + 1 if r.read(1) else 0)
            else:
                lengths.append(r.read(5) + 1)
    else:
        length = r.read(5) + 1
        lengths = []
        while len(lengths) < entries:
            num = r.read(ilog(entries - len(lengths)))
            assert length <= 32 and num <= entries - len(lengths)
            lengths += [length] * num
            length += 1
    book = {"dim": dim, "entries": entries, "lengthlist": lengths, "maptype": r.read(4)}
    if book["maptype"] in (1, 2):
        book["q_min"], book["q_delta"] = r.read(32), r.read(32)
        book["q_quant"], book["q_sequencep"] = r.read(4) + 1, r.read(1)
        qv = maptype1_quantvals(entries, dim) if book["maptype"] == 1 else entries * dim
        book["quantlist"] = [r.read(book["q_quant"]) for _ in range(qv)]
    else:
        assert book["maptype"] == 0
    return book


def unpack_headers(h0, h1, h2):
    out = {}
    r = BitReader(h0)
    assert r.read(8) == 1 and bytes(r.read(8) for _ in range(6)) == b"vorbis"
    assert r.read(32) == 0
    out["channels"], out["rate"] = r.read(8), r.read(32)
    out["bitrates"] = [r.read(32), r.read(32), r.read(32)]
    out["blocksizes"] = [1 << r.read(4), 1 << r.read(4)]
    assert r.read(1) == 1 and out["rate"] >= 1 and out["channels"] >= 1 and 64 <= out["blocksizes"][0] <= out["blocksizes"][1] <= 8192
    assert len(h0) == 30

    r = BitReader(h1)
    assert r.read(8) == 3 and bytes(r.read(8) for _ in range(6)) == b"vorbis"
    out["vendor"] = bytes(r.read(8) for _ in range(r.read(32)))
    out["comments"] = [bytes(r.read(8) for _ in range(r.read(32))) for _ in range(r.read(32))]
    assert r.read(1) == 1

    r = BitReader(h2)
    assert r.read(8) == 5 and bytes(r.read(8) for _ in range(6)) == b"vorbis"
    out["books"] = [unpack_book(r) for _ in range(r.read(8) + 1)]
    nb = len(out["books"])
    for _ in range(r.read(6) + 1):          # time backend placeholders
        assert r.read(16) == 0
    floors = []
    for _ in range(r.read(6) + 1):
        assert r.read(16) == 1             # floor type 1
        f = {"partitions": r.read(5)}
        f["partitionclass"] = [r.read(4) for _ in range(f["partitions"])]
        maxclass = max(f["partitionclass"], default=-1)
        f["class_dim"], f["class_subs"], f["class_book"], f["class_subbook"] = [], [], [], []
        for _ in range(maxclass + 1):
            f["class_dim"].append(r.read(3) + 1)
            f["class_subs"].append(r.read(2))
            f["class_book"].append(r.read(8) if f["class_subs"][-1] else None)
            assert f["class_book"][-1] is None or f["class_book"][-1] < nb
            f["class_subbook"].append([r.read(8) - 1 for _ in range(1 << f["class_subs"][-1])])
            assert all(-1 <= x < nb for x in f["class_subbook"][-1])
        f["mult"] = r.read(2) + 1
        rangebits = r.read(4)
        count = sum(f["class_dim"][c] for c in f["partitionclass"])
        assert count <= 63
        f["postlist"] = [0, 1 << rangebits] + [r.read(rangebits) for _ in range(count)]
        assert len(set(f["postlist"])) == len(f["postlist"])   # no repeated posts (zero-length segments)
        floors.append(f)
    out["floors"] = floors
    residues = []
    for _ in range(r.read(6) + 1):
        res = {"type": r.read(16)}
        assert res["type"] in (0, 1, 2)
        res["begin"], res["end"], res["grouping"] = r.read(24), r.read(24), r.read(24) + 1
        res["partitions"], res["groupbook"] = r.read(6) + 1, r.read(8)
        res["secondstages"] = []
        for _ in range(res["partitions"]):
            cascade = r.read(3)
            if r.read(1):
                cascade |= r.read(5) << 3
            res["secondstages"].append(cascade)
        res["booklist"] = [r.read(8) for _ in range(sum(bin(c).count("1") for c in res["secondstages"]))]
        assert res["groupbook"] < nb and all(b < nb and out["books"][b]["maptype"] != 0 for b in res["booklist"])
        gb = out["books"][res["groupbook"]]
        assert gb["dim"] >= 1 and res["partitions"] ** gb["dim"] <= gb["entries"]
        residues.append(res)
    out["residues"] = residues
    maps = []
    ch = out["channels"]
    for _ in range(r.read(6) + 1):
        assert r.read(16) == 0
        m = {"submaps": r.read(4) + 1 if r.read(1) else 1}
        m["coupling"] = []
        if r.read(1):
            for _ in range(r.read(8) + 1):
                mag, ang = r.read(ilog(ch - 1)), r.read(ilog(ch - 1))
                assert mag != ang and mag < ch and ang < ch
                m["coupling"].append((mag, ang))
        assert r.read(2) == 0
        m["chmuxlist"] = [r.read(4) for _ in range(ch)] if m["submaps"] > 1 else [0] * ch
        assert all(x < m["submaps"] for x in m["chmuxlist"])
        m["floorsubmap"], m["residuesubmap"] = [], []
        for _ in range(m["submaps"]):
            r.read(8)
            m["floorsubmap"].append(r.read(8))
            m["residuesubmap"].append(r.read(8))
            assert m["floorsubmap"][-1] < len(floors) and m["residuesubmap"][-1] < len(residues)
        maps.append(m)
    out["maps"] = maps
    modes = []
    for _ in range(r.read(6) + 1):
        md = (r.read(1), r.read(16), r.read(16), r.read(8))
        assert md[1] == 0 and md[2] == 0 and md[3] < len(maps)
        modes.append(md)
    out["modes"] = modes
    assert r.read(1) == 1                      # framing bit
    assert r.bytes_left() == 0                 # nothing but padding bits after it
    return out


@pytest.mark.parametrize("pack", sorted(os.path.basename(p) for p in glob.glob(os.path.join(DATA, "mode_*.vpk"))))
def test_header_packets_read_back_to_the_mode_pack(pack):
    import vorbis_aotuv_lancer_amd as v
    d = vpk.read_vpk(os.path.join(DATA, pack))
    ch, rate, q = int(d["info/channels"][0]), int(d["info/rate"][0]), float(d["info/quality"][0])
    if int(d["info/managed"][0]):
        av, mn, mx, _ = [int(x) for x in d["bi/rates"]]
        setup = v.Setup(ch, rate, bitrate=(mx, av, mn))
    else:
        setup = v.Setup(ch, rate, q)
    comments = ["TITLE=parity", "ARTIST=" + "x" * 300]
    hdr = v.header_packets(setup, comments)
    u = unpack_headers(*hdr)
    assert (u["channels"], u["rate"]) == (ch, rate)
    assert u["blocksizes"] == [int(x) for x in d["info/blocksizes"]]
    assert u["bitrates"] == [int(x) & 0xffffffff for x in d["info/bitrates"]]
    assert u["vendor"] == b"AO; aoTuV [20110424] (based on libvorbis 1.3.7)"
    assert u["comments"] == [c.encode() for c in comments]
    modes_, maps_, floors_, residues_, books_, _ = [int(x) for x in d["info/counts"]]
    assert (len(u["modes"]), len(u["maps"]), len(u["floors"]), len(u["residues"]), len(u["books"])) == \
        (modes_, maps_, floors_, residues_, books_)
    for i, b in enumerate(u["books"]):
        head = [int(x) for x in d[f"book/{i}/head"]]
        assert (b["dim"], b["entries"], b["maptype"]) == (head[0], head[1], head[2])
        assert b["lengthlist"] == [int(x) for x in d[f"book/{i}/lengthlist"]]
        if b["maptype"]:
            assert (b["q_min"], b["q_delta"], b["q_quant"], b["q_sequencep"]) == \
                (head[3] & 0xffffffff, head[4] & 0xffffffff, head[5], head[6])
            assert b["quantlist"] == [abs(int(x)) for x in d[f"book/{i}/quantlist"]][:len(b["quantlist"])]
    for i, f in enumerate(u["floors"]):
        P = int(d[f"floor/{i}/partitions"][0])
        assert f["partitions"] == P and f["mult"] == int(d[f"floor/{i}/mult"][0])
        assert f["partitionclass"] == [int(x) for x in d[f"floor/{i}/partitionclass"][:P]]
        nc = len(f["class_dim"])
        assert f["class_dim"] == [int(x) for x in d[f"floor/{i}/class_dim"][:nc]]
        assert f["class_subs"] == [int(x) for x in d[f"floor/{i}/class_subs"][:nc]]
        for c in range(nc):
            if f["class_subs"][c]:
                assert f["class_book"][c] == int(d[f"floor/{i}/class_book"][c])
            assert f["class_subbook"][c] == [int(x) for x in d[f"floor/{i}/class_subbook"][c][:1 << f["class_subs"][c]]]
        want = [int(x) for x in d[f"floor/{i}/postlist"][:len(f["postlist"])]]
        want[1] = 1 << ilog(want[1] - 1)     # the range is sent as a bit count (lib/floor1.c:105, :164): 12 reads back as 16
        assert f["postlist"] == want
    for i, r in enumerate(u["residues"]):
        head = [int(x) for x in d[f"residue/{i}/head"]]
        assert [r["type"], r["begin"], r["end"], r["grouping"], r["partitions"], r["groupbook"]] == \
            [head[0], head[1], head[2], head[3], head[4], head[6]]
        assert r["secondstages"] == [int(x) for x in d[f"residue/{i}/secondstages"][:r["partitions"]]]
        assert r["booklist"] == [int(x) for x in d[f"residue/{i}/booklist"][:len(r["booklist"])]]
    for i, m in enumerate(u["maps"]):
        assert m["submaps"] == int(d[f"map/{i}/submaps"][0])
        steps = int(d[f"map/{i}/coupling_steps"][0])
        assert m["coupling"] == [(int(d[f"map/{i}/coupling_mag"][k]), int(d[f"map/{i}/coupling_ang"][k])) for k in range(steps)]
        assert m["chmuxlist"] == [int(x) for x in d[f"map/{i}/chmuxlist"][:ch]]
        assert m["floorsubmap"] == [int(x) for x in d[f"map/{i}/floorsubmap"][:m["submaps"]]]
        assert m["residuesubmap"] == [int(x) for x in d[f"map/{i}/residuesubmap"][:m["submaps"]]]
    for i, md in enumerate(u["modes"]):
        assert list(md) == [int(x) for x in d[f"mode/{i}"]]
    setup.close()


def crc_bitwise(data):
    """doc/framing.html:363-366: direct CRC-32, polynomial 0x04c11db7, initial value and final XOR 0"""
    r = 0
    for byte in data:
        r ^= byte << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04c11db7) & 0xffffffff if r & 0x80000000 else (r << 1) & 0xffffffff
    return r


def parse_pages(blob):
    pages, at = [], 0
    while at < len(blob):
        assert blob[at:at + 4] == b"OggS" and blob[at + 4] == 0
        flags = blob[at + 5]
        granule, serial, seq, crc = struct.unpack_from("<qIII", blob, at + 6)
        nseg = blob[at + 26]
        lacing = list(blob[at + 27:at + 27 + nseg])
        size = 27 + nseg + sum(lacing)
        page = bytearray(blob[at:at + size])
        page[22:26] = b"\0\0\0\0"
        assert crc_bitwise(page) == crc, "page CRC"
        pages.append({"flags": flags, "granule": granule, "serial": serial, "seq": seq, "lacing": lacing,
                      "body": blob[at + 27 + nseg:at + size]})
        at += size
    return pages


def packets_of(pages):
    out, cur, open_ = [], b"", False
    for pg in pages:
        assert bool(pg["flags"] & 1) == open_, "continued-packet flag"
        at = 0
        for lv in pg["lacing"]:
            cur += pg["body"][at:at + lv]
            at += lv
            if lv < 255:
                out.append(cur)
                cur = b""
        open_ = bool(pg["lacing"]) and pg["lacing"][-1] == 255
    assert not open_
    return out


def test_ogg_pages_round_trip():
    import vorbis_aotuv_lancer_amd as v
    rng = np.random.default_rng(5)
    sizes = [30, 74, 4225, 0, 1, 254, 255, 256, 509, 510, 511, 3000, 70000, 12] + [int(x) for x in rng.integers(1, 900, 400)]
    packets = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in sizes]
    os_ = v.OggStream(0x1234abcd)
    blob = b""
    gp = 0
    for i, p in enumerate(packets):
        gp += 1024 if i >= 3 else 0
        os_.packetin(p, gp if i >= 3 else 0, eos=(i == len(packets) - 1))
        if i == 2:
            blob += b"".join(os_.pages(flush=True))      # headers flushed: audio starts on a fresh page
        else:
            blob += b"".join(os_.pages())
    blob += b"".join(os_.pages(flush=True))
    pages = parse_pages(blob)
    assert packets_of(pages) == packets
    assert [pg["seq"] for pg in pages] == list(range(len(pages)))
    assert all(pg["serial"] == 0x1234abcd for pg in pages)
    assert pages[0]["flags"] & 2 and not any(pg["flags"] & 2 for pg in pages[1:])        # b_o_s on the first page only
    assert pages[-1]["flags"] & 4 and not any(pg["flags"] & 4 for pg in pages[:-1])      # e_o_s on the last page only
    assert pages[0]["lacing"] == [30]                                                    # first page = first packet alone
    assert all(len(pg["lacing"]) <= 255 for pg in pages)
    # granule position of a page = that of the last packet ENDING on it, -1 if none ends there
    gps, g = [], 0
    for i in range(len(packets)):
        g += 1024 if i >= 3 else 0
        gps.append(g if i >= 3 else 0)
    done = 0
    for pg in pages:
        ends = sum(1 for lv in pg["lacing"] if lv < 255)
        done += ends
        assert pg["granule"] == (gps[done - 1] if ends else -1)
    os_.close()


def test_header_bytes_equal_the_oracles(oracle):
    """The three header packets of every shipped setup, packed by the product from the mode pack
    (csrc/capi_stream.cpp) and by the oracle from its own setup structs (oracle/orc_headers.c, restating
    lib/info.c:500-617, lib/codebook.c:158-275, floor1_pack, res0_pack, mapping0_pack): byte for byte."""
    import ctypes as C
    import glob
    import re
    import vorbis_aotuv_lancer_amd as v
    from tests import orc
    lib = oracle.lib
    lib.orc_header_packets.restype = C.c_long
    lib.orc_header_packets.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_long,
                                       C.POINTER(C.c_long)]
    comments = ["ENCODER=test", "TITLE=header parity", ""]
    arr = (C.c_char_p * len(comments))(*[c.encode() for c in comments])
    packs = sorted(glob.glob(os.path.join(os.path.dirname(v.LIB_PATH), "data", "mode_*.vpk")))
    assert len(packs) >= 20
    for path in packs:
        m = re.match(r"mode_(\d+)ch_(\d+)_(q|b)(-?[\d.]+?)(?:_max(\d+))?(?:_min(\d+))?\.vpk", os.path.basename(path))
        ch, rate = int(m.group(1)), int(m.group(2))
        if m.group(3) == "q":
            kw = dict(q=float(m.group(4)))
            setup = v.Setup(ch, rate, float(m.group(4)))
        else:
            br = (int(m.group(5) or -1), int(m.group(4)), int(m.group(6) or -1))
            kw = dict(bitrate=br)
            setup = v.Setup(ch, rate, bitrate=br)
        mine = v.header_packets(setup, comments)
        osetup = orc.Setup(oracle, ch, rate, **kw)
        lens = (C.c_long * 3)()
        total = lib.orc_header_packets(osetup.h, None, arr, len(comments), None, 0, lens)
        assert total > 0
        buf = (C.c_ubyte * total)()
        assert lib.orc_header_packets(osetup.h, None, arr, len(comments), buf, total, lens) == total
        raw = bytes(buf)
        theirs = [raw[:lens[0]], raw[lens[0]:lens[0] + lens[1]], raw[lens[0] + lens[1]:]]
        for k in range(3):
            assert mine[k] == theirs[k], (os.path.basename(path), k, len(mine[k]), len(theirs[k]))
        setup.close()
