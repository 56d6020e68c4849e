// Stream wrapper, host only: the three Vorbis header packets and Ogg page framing,
// so that the packets of vbm_frontend_encode_round become a playable .ogg stream.
//
//   vbm_header_packets   vorbis_analysis_headerout (reference lib/info.c:636-717):
//                        _vorbis_pack_info :500-527, _vorbis_pack_comment :529-559, _vorbis_pack_books :561-617
//                        with vorbis_staticbook_pack (lib/codebook.c:158-273), floor1_pack (lib/floor1.c:77-113),
//                        res0_pack (lib/res0.c:161-188), mapping0_pack (lib/mapping0.c:51-92).
//                        Source of the setup: the mode pack the vbm_setup was created from.
//   vbm_ogg_stream_*     libogg's ogg_stream_packetin / _pageout / _flush.  libogg is an external dependency
//                        of the reference (cmake/FindOgg.cmake, not vendored, no version pin); the page format is
//                        restated from the reference's own doc/framing.html (capture pattern, header layout,
//                        lacing, CRC: "direct algorithm, initial val and final XOR = 0, generator polynomial
//                        0x04c11db7" :363-366).  When a page ends is not normative; the policy here is libogg
//                        1.3's: the first page holds the first packet alone, later pages close once they carry
//                        more than 4096 body bytes and at least four packets, at 255 segments, on flush, or at
//                        the end of the stream.
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "vorbis_mi355x.h"
#include "vpk.h"
#include "vbm_internal.h"

const char *vbm_setup_handle_mode_path(const vbm_setup_handle *h);

namespace {

// oggpack_write: LSb first (doc/02-bitpacking.tex)
struct BitPacker {
    std::vector<uint8_t> buf;
    uint64_t acc = 0;
    int nbits = 0;
    void write(unsigned long value, int bits)
    {
        if (bits <= 0) return;
        uint64_t v = value;
        if (bits < 64) v &= (bits == 64) ? ~0ull : ((1ull << bits) - 1ull);
        acc |= v << nbits;
        nbits += bits;
        while (nbits >= 8) {
            buf.push_back((uint8_t)(acc & 0xff));
            acc >>= 8;
            nbits -= 8;
        }
    }
    void string(const char *s, size_t n) { for (size_t i = 0; i < n; i++) write((unsigned char)s[i], 8); }
    std::vector<uint8_t> finish()
    {
        if (nbits > 0) buf.push_back((uint8_t)(acc & 0xff));
        acc = 0;
        nbits = 0;
        return buf;
    }
};

int ov_ilog(uint32_t v)
{
    int ret = 0;
    while (v) { ret++; v >>= 1; }
    return ret;
}

int icount(unsigned v)
{
    int ret = 0;
    while (v) { ret += v & 1; v >>= 1; }
    return ret;
}

// lib/sharedbook.c:174-212
long maptype1_quantvals(long entries, long dim)
{
    if (entries < 1) return 0;
    long vals = (long)floor(pow((float)entries, 1.f / dim));
    if (vals < 1) vals = 1;
    while (1) {
        long acc = 1, acc1 = 1;
        int i;
        for (i = 0; i < dim; i++) {
            if (entries / vals < acc) break;
            acc *= vals;
            if (LONG_MAX / (vals + 1) < acc1) acc1 = LONG_MAX;
            else acc1 *= vals + 1;
        }
        if (i >= dim && acc <= entries && acc1 > entries) return vals;
        if (i < dim || acc > entries) vals--;
        else vals++;
    }
}

struct PackFile {
    vpk_file f;
    bool ok;
    explicit PackFile(const char *path) { ok = vpk_open(&f, path) == 0; }
    ~PackFile() { if (ok) vpk_close(&f); }
    template <typename T> const T *get(const std::string &name, int dtype, size_t *n = nullptr) const
    {
        size_t nn = 0;
        const void *p = vpk_get(&f, name.c_str(), dtype, &nn);
        if (!p) throw std::string("mode pack entry missing or mistyped: ") + name;
        if (n) *n = nn;
        return (const T *)p;
    }
    const int *i32(const std::string &n, size_t *c = nullptr) const { return get<int>(n, VPK_I32, c); }
    const int64_t *i64(const std::string &n, size_t *c = nullptr) const { return get<int64_t>(n, VPK_I64, c); }
};

// lib/codebook.c:158-273; head = {dim, entries, maptype, q_min, q_delta, q_quant, q_sequencep, quantlist length}
void pack_book(BitPacker &o, const int64_t *head, const signed char *lengthlist, const int *quantlist)
{
    const long dim = (long)head[0], entries = (long)head[1], maptype = (long)head[2];
    long i, j;
    int ordered = 0;
    o.write(0x564342, 24);
    o.write(dim, 16);
    o.write(entries, 24);
    for (i = 1; i < entries; i++)
        if (lengthlist[i - 1] == 0 || lengthlist[i] < lengthlist[i - 1]) break;
    if (i == entries) ordered = 1;
    if (ordered) {
        long count = 0;
        o.write(1, 1);
        o.write(lengthlist[0] - 1, 5);
        for (i = 1; i < entries; i++) {
            char cur = lengthlist[i], last = lengthlist[i - 1];
            if (cur > last) {
                for (j = last; j < cur; j++) {
                    o.write(i - count, ov_ilog((uint32_t)(entries - count)));
                    count = i;
                }
            }
        }
        o.write(i - count, ov_ilog((uint32_t)(entries - count)));
    } else {
        o.write(0, 1);
        for (i = 0; i < entries; i++)
            if (lengthlist[i] == 0) break;
        if (i == entries) {
            o.write(0, 1);
            for (i = 0; i < entries; i++) o.write(lengthlist[i] - 1, 5);
        } else {
            o.write(1, 1);
            for (i = 0; i < entries; i++) {
                if (lengthlist[i] == 0) o.write(0, 1);
                else {
                    o.write(1, 1);
                    o.write(lengthlist[i] - 1, 5);
                }
            }
        }
    }
    o.write(maptype, 4);
    if (maptype == 1 || maptype == 2) {
        if (!head[7]) throw std::string("codebook with a value mapping but no quantlist");
        o.write((unsigned long)head[3], 32);
        o.write((unsigned long)head[4], 32);
        o.write(head[5] - 1, 4);
        o.write(head[6], 1);
        long quantvals = (maptype == 1) ? maptype1_quantvals(entries, dim) : entries * dim;
        if (quantvals > (long)head[7]) throw std::string("quantlist shorter than the codebook needs");
        for (i = 0; i < quantvals; i++) o.write(labs(quantlist[i]), (int)head[5]);
    } else if (maptype != 0) {
        throw std::string("unknown codebook map type");
    }
}

std::vector<uint8_t> pack_info(const PackFile &m)
{
    BitPacker o;
    const int ch = *m.i32("info/channels");
    const long rate = (long)*m.i64("info/rate");
    const int *bs = m.i32("info/blocksizes");
    const int64_t *br = m.i64("info/bitrates");   // upper, nominal, lower (lib/vorbisenc.c:876-884)
    o.write(0x01, 8);
    o.string("vorbis", 6);
    o.write(0x00, 32);
    o.write(ch, 8);
    o.write(rate, 32);
    o.write((unsigned long)br[0], 32);
    o.write((unsigned long)br[1], 32);
    o.write((unsigned long)br[2], 32);
    o.write(ov_ilog(bs[0] - 1), 4);
    o.write(ov_ilog(bs[1] - 1), 4);
    o.write(1, 1);
    return o.finish();
}

std::vector<uint8_t> pack_comment(const char *vendor, const char *const *comments, int ncomments)
{
    BitPacker o;
    const size_t vb = strlen(vendor);
    o.write(0x03, 8);
    o.string("vorbis", 6);
    o.write(vb, 32);
    o.string(vendor, vb);
    o.write(ncomments, 32);
    for (int i = 0; i < ncomments; i++) {
        if (comments[i]) {
            const size_t n = strlen(comments[i]);
            o.write(n, 32);
            o.string(comments[i], n);
        } else {
            o.write(0, 32);
        }
    }
    o.write(1, 1);
    return o.finish();
}

std::vector<uint8_t> pack_setup(const PackFile &m)
{
    BitPacker o;
    const int ch = *m.i32("info/channels");
    const int *counts = m.i32("info/counts");   // modes, maps, floors, residues, books, psys
    const int modes = counts[0], maps = counts[1], floors = counts[2], residues = counts[3], books = counts[4];
    o.write(0x05, 8);
    o.string("vorbis", 6);

    o.write(books - 1, 8);
    for (int i = 0; i < books; i++) {
        const std::string pre = "book/" + std::to_string(i) + "/";
        size_t nl = 0, nq = 0;
        const int64_t *head = m.i64(pre + "head");
        const signed char *ll = m.get<signed char>(pre + "lengthlist", VPK_I8, &nl);
        const int *ql = m.i32(pre + "quantlist", &nq);
        if ((long)nl != (long)head[1]) throw std::string("codebook lengthlist size mismatch");
        pack_book(o, head, ll, ql);
    }

    o.write(0, 6);    // times: hook placeholders
    o.write(0, 16);

    o.write(floors - 1, 6);
    for (int i = 0; i < floors; i++) {
        const std::string pre = "floor/" + std::to_string(i) + "/";
        o.write(1, 16);   // floor type 1
        // floor1_pack
        const int partitions = *m.i32(pre + "partitions");
        const int *pclass = m.i32(pre + "partitionclass"), *cdim = m.i32(pre + "class_dim"), *csubs = m.i32(pre + "class_subs");
        const int *cbook = m.i32(pre + "class_book"), *csub = m.i32(pre + "class_subbook"), *postlist = m.i32(pre + "postlist");
        const int mult = *m.i32(pre + "mult");
        int count = 0, maxclass = -1;
        const int maxposit = postlist[1];
        o.write(partitions, 5);
        for (int j = 0; j < partitions; j++) {
            o.write(pclass[j], 4);
            if (maxclass < pclass[j]) maxclass = pclass[j];
        }
        for (int j = 0; j < maxclass + 1; j++) {
            o.write(cdim[j] - 1, 3);
            o.write(csubs[j], 2);
            if (csubs[j]) o.write(cbook[j], 8);
            for (int k = 0; k < (1 << csubs[j]); k++) o.write(csub[j * 8 + k] + 1, 8);
        }
        o.write(mult - 1, 2);
        o.write(ov_ilog(maxposit - 1), 4);
        const int rangebits = ov_ilog(maxposit - 1);
        for (int j = 0, k = 0; j < partitions; j++) {
            count += cdim[pclass[j]];
            for (; k < count; k++) o.write(postlist[k + 2], rangebits);
        }
    }

    o.write(residues - 1, 6);
    for (int i = 0; i < residues; i++) {
        const std::string pre = "residue/" + std::to_string(i) + "/";
        const int *head = m.i32(pre + "head");   // type, begin, end, grouping, partitions, partvals, groupbook
        const int *second = m.i32(pre + "secondstages"), *booklist = m.i32(pre + "booklist");
        o.write(head[0], 16);
        // res0_pack
        int acc = 0;
        o.write(head[1], 24);
        o.write(head[2], 24);
        o.write(head[3] - 1, 24);
        o.write(head[4] - 1, 6);
        o.write(head[6], 8);
        for (int j = 0; j < head[4]; j++) {
            if (ov_ilog(second[j]) > 3) {
                o.write(second[j], 3);
                o.write(1, 1);
                o.write(second[j] >> 3, 5);
            } else {
                o.write(second[j], 4);
            }
            acc += icount(second[j]);
        }
        for (int j = 0; j < acc; j++) o.write(booklist[j], 8);
    }

    o.write(maps - 1, 6);
    for (int i = 0; i < maps; i++) {
        const std::string pre = "map/" + std::to_string(i) + "/";
        const int submaps = *m.i32(pre + "submaps"), steps = *m.i32(pre + "coupling_steps");
        const int *chmux = m.i32(pre + "chmuxlist"), *fsub = m.i32(pre + "floorsubmap"), *rsub = m.i32(pre + "residuesubmap");
        const int *mag = m.i32(pre + "coupling_mag"), *ang = m.i32(pre + "coupling_ang");
        o.write(0, 16);   // mapping type 0
        // mapping0_pack
        if (submaps > 1) {
            o.write(1, 1);
            o.write(submaps - 1, 4);
        } else {
            o.write(0, 1);
        }
        if (steps > 0) {
            o.write(1, 1);
            o.write(steps - 1, 8);
            for (int k = 0; k < steps; k++) {
                o.write(mag[k], ov_ilog(ch - 1));
                o.write(ang[k], ov_ilog(ch - 1));
            }
        } else {
            o.write(0, 1);
        }
        o.write(0, 2);
        if (submaps > 1)
            for (int k = 0; k < ch; k++) o.write(chmux[k], 4);
        for (int k = 0; k < submaps; k++) {
            o.write(0, 8);
            o.write(fsub[k], 8);
            o.write(rsub[k], 8);
        }
    }

    o.write(modes - 1, 6);
    for (int i = 0; i < modes; i++) {
        const int *md = m.i32("mode/" + std::to_string(i));   // blockflag, windowtype, transformtype, mapping
        o.write(md[0], 1);
        o.write(md[1], 16);
        o.write(md[2], 16);
        o.write(md[3], 8);
    }
    o.write(1, 1);
    return o.finish();
}

}  // namespace

extern "C" int vbm_header_packets(const vbm_setup_handle *setup, const char *vendor, const char *const *comments,
                                  int ncomments, uint8_t *buf, long cap, long *lens)
{
    if (!setup || !lens || ncomments < 0 || (ncomments && !comments)) return VBM_EINVAL;
    // the scalar build of the reference identifies itself with this string (lib/info.c:43)
    if (!vendor) vendor = "AO; aoTuV [20110424] (based on libvorbis 1.3.7)";
    try {
        PackFile m(vbm_setup_handle_mode_path(setup));
        if (!m.ok) {
            g_vbm_err = "cannot reopen the mode pack";
            return VBM_EFAULT;
        }
        std::vector<uint8_t> p0 = pack_info(m), p1 = pack_comment(vendor, comments, ncomments), p2 = pack_setup(m);
        lens[0] = (long)p0.size();
        lens[1] = (long)p1.size();
        lens[2] = (long)p2.size();
        const long total = lens[0] + lens[1] + lens[2];
        if (!buf) return VBM_OK;   // size query
        if (cap < total) return VBM_EINVAL;
        memcpy(buf, p0.data(), p0.size());
        memcpy(buf + p0.size(), p1.data(), p1.size());
        memcpy(buf + p0.size() + p1.size(), p2.data(), p2.size());
        return VBM_OK;
    } catch (const std::string &e) {
        g_vbm_err = e;
        return VBM_EFAULT;
    }
}

// the comment header alone (reference vorbis_commentheader_out, lib/info.c:600-617): size query with buf == NULL
extern "C" int vbm_comment_packet(const char *vendor, const char *const *comments, int ncomments, uint8_t *buf, long cap,
                                  long *len)
{
    if (!len || ncomments < 0 || (ncomments && !comments)) return VBM_EINVAL;
    if (!vendor) vendor = "AO; aoTuV [20110424] (based on libvorbis 1.3.7)";
    const std::vector<uint8_t> p1 = pack_comment(vendor, comments, ncomments);
    *len = (long)p1.size();
    if (!buf) return VBM_OK;
    if (cap < *len) return VBM_EINVAL;
    memcpy(buf, p1.data(), p1.size());
    return VBM_OK;
}

// ---- Ogg pages (doc/framing.html) -----------------------------------------------------------------
struct vbm_ogg_stream {
    int serialno;
    long pageno = 0;
    bool b_o_s_done = false;   // the first page has gone out
    bool e_o_s = false;
    bool prev_open = false;    // the last page ended inside a packet: the next one starts with its continuation
    std::vector<uint8_t> body;        // bytes of the queued segments
    std::vector<uint8_t> lacing;      // one value per queued segment
    std::vector<long long> granule;   // granulepos of the packet that ENDS at this segment (else -1)
    std::vector<uint8_t> page;        // last page handed out
    uint32_t crc_table[256];
};

extern "C" int vbm_ogg_stream_create(vbm_ogg_stream **out, int serialno)
{
    if (!out) return VBM_EINVAL;
    vbm_ogg_stream *os = new vbm_ogg_stream();
    os->serialno = serialno;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i << 24;
        for (int k = 0; k < 8; k++) r = (r & 0x80000000u) ? (r << 1) ^ 0x04c11db7u : (r << 1);
        os->crc_table[i] = r;
    }
    *out = os;
    return VBM_OK;
}

extern "C" void vbm_ogg_stream_destroy(vbm_ogg_stream *os) { delete os; }

extern "C" int vbm_ogg_stream_packetin(vbm_ogg_stream *os, const uint8_t *packet, long bytes, int e_o_s, long long granulepos)
{
    if (!os || bytes < 0 || (bytes && !packet) || os->e_o_s) return VBM_EINVAL;
    const long segs = bytes / 255 + 1;
    os->body.insert(os->body.end(), packet, packet + bytes);
    for (long i = 0; i < segs - 1; i++) {
        os->lacing.push_back(255);
        os->granule.push_back(-1);
    }
    os->lacing.push_back((uint8_t)(bytes % 255));
    os->granule.push_back(granulepos);
    if (e_o_s) os->e_o_s = true;
    return VBM_OK;
}

extern "C" int vbm_ogg_stream_pageout(vbm_ogg_stream *os, int flush, const uint8_t **page, long *bytes)
{
    if (!os || !page || !bytes) return VBM_EINVAL;
    *page = nullptr;
    *bytes = 0;
    const int maxvals = os->lacing.size() > 255 ? 255 : (int)os->lacing.size();
    if (maxvals == 0) return 0;
    int vals = 0;
    long long granule_pos = -1;
    bool force = flush != 0 || (os->e_o_s) || !os->b_o_s_done;
    if (!os->b_o_s_done) {
        // the first page holds the first packet alone
        granule_pos = 0;
        for (vals = 0; vals < maxvals; vals++)
            if (os->lacing[vals] < 255) { vals++; break; }
    } else {
        long acc = 0;
        int packets_done = 0, packet_just_done = 0;
        for (vals = 0; vals < maxvals; vals++) {
            if (acc > 4096 && packet_just_done >= 4) { force = true; break; }
            acc += os->lacing[vals];
            if (os->lacing[vals] < 255) {
                granule_pos = os->granule[vals];
                packet_just_done = ++packets_done;
            } else {
                packet_just_done = 0;
            }
        }
        if (vals == 255) force = true;
    }
    if (!force) return 0;

    long body_bytes = 0;
    for (int i = 0; i < vals; i++) body_bytes += os->lacing[i];
    std::vector<uint8_t> &pg = os->page;
    pg.assign(27 + vals + body_bytes, 0);
    memcpy(pg.data(), "OggS", 4);
    pg[4] = 0;   // stream structure version
    uint8_t flags = 0;
    if (os->prev_open) flags |= 0x01;   // continued packet
    if (!os->b_o_s_done) flags |= 0x02;
    if (os->e_o_s && vals == (int)os->lacing.size()) flags |= 0x04;
    pg[5] = flags;
    for (int i = 0; i < 8; i++) pg[6 + i] = (uint8_t)(((unsigned long long)granule_pos >> (8 * i)) & 0xff);
    for (int i = 0; i < 4; i++) pg[14 + i] = (uint8_t)(((uint32_t)os->serialno >> (8 * i)) & 0xff);
    for (int i = 0; i < 4; i++) pg[18 + i] = (uint8_t)(((uint32_t)os->pageno >> (8 * i)) & 0xff);
    pg[26] = (uint8_t)vals;
    for (int i = 0; i < vals; i++) pg[27 + i] = os->lacing[i];
    memcpy(pg.data() + 27 + vals, os->body.data(), body_bytes);
    uint32_t crc = 0;
    for (size_t i = 0; i < pg.size(); i++) crc = (crc << 8) ^ os->crc_table[((crc >> 24) & 0xff) ^ pg[i]];
    for (int i = 0; i < 4; i++) pg[22 + i] = (uint8_t)((crc >> (8 * i)) & 0xff);

    os->body.erase(os->body.begin(), os->body.begin() + body_bytes);
    os->lacing.erase(os->lacing.begin(), os->lacing.begin() + vals);
    os->granule.erase(os->granule.begin(), os->granule.begin() + vals);
    os->pageno++;
    os->b_o_s_done = true;
    os->prev_open = (pg[27 + vals - 1] == 255);
    *page = pg.data();
    *bytes = (long)pg.size();
    return 1;
}
