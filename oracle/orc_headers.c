/* ORACLE (test infrastructure) — the three Vorbis header packets, packed from the oracle's own setup structs.
 *
 * Restates the encode side of
 *   _vorbis_pack_info / _vorbis_pack_comment / _vorbis_pack_books   lib/info.c:500-617
 *   vorbis_staticbook_pack                                         lib/codebook.c:158-275
 *   floor1_pack                                                    lib/floor1.c:77-113
 *   res0_pack                                                      lib/res0.c:161-188
 *   mapping0_pack                                                  lib/mapping0.c:51-92
 *   vorbis_analysis_headerout                                      lib/info.c:638-717 (three packets, in order)
 * so that the product's header bytes (csrc/capi_stream.cpp, packed from the mode pack) are compared with a second,
 * independent packer working from different data structures.  The reference holds no header bytes to compare with
 * (parity with the reference unpinned); vendor string of the scalar build: lib/info.c:43.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static void put_string(orc_bits *o, const char *s, long bytes)
{
    while (bytes--) orc_bits_write(o, (unsigned char)*s++, 8);
}

static int icount(unsigned int v)
{
    int ret = 0;
    while (v) { ret += v & 1; v >>= 1; }
    return ret;
}

/* lib/info.c:500-527 */
static void pack_info(orc_bits *o, const orc_setup *s)
{
    orc_bits_write(o, 0x01, 8);
    put_string(o, "vorbis", 6);
    orc_bits_write(o, 0x00, 32);
    orc_bits_write(o, (unsigned long)s->channels, 8);
    orc_bits_write(o, (unsigned long)s->rate, 32);
    orc_bits_write(o, (unsigned long)s->bitrate_upper, 32);
    orc_bits_write(o, (unsigned long)s->bitrate_nominal, 32);
    orc_bits_write(o, (unsigned long)s->bitrate_lower, 32);
    orc_bits_write(o, (unsigned long)orc_ilog((uint32_t)(s->blocksizes[0] - 1)), 4);
    orc_bits_write(o, (unsigned long)orc_ilog((uint32_t)(s->blocksizes[1] - 1)), 4);
    orc_bits_write(o, 1, 1);
}

/* lib/info.c:529-556 */
static void pack_comment(orc_bits *o, const char *vendor, const char *const *comments, int ncomments)
{
    long bytes = (long)strlen(vendor);
    int i;
    orc_bits_write(o, 0x03, 8);
    put_string(o, "vorbis", 6);
    orc_bits_write(o, (unsigned long)bytes, 32);
    put_string(o, vendor, bytes);
    orc_bits_write(o, (unsigned long)ncomments, 32);
    for (i = 0; i < ncomments; i++) {
        if (comments[i]) {
            long n = (long)strlen(comments[i]);
            orc_bits_write(o, (unsigned long)n, 32);
            put_string(o, comments[i], n);
        } else {
            orc_bits_write(o, 0, 32);
        }
    }
    orc_bits_write(o, 1, 1);
}

/* lib/codebook.c:158-275 (_book_maptype1_quantvals: orc_book.c) */
long orc_maptype1_quantvals(long entries, long dim);

static int pack_book(orc_bits *o, const orc_book *c)
{
    long i, j;
    int ordered = 0;
    orc_bits_write(o, 0x564342, 24);
    orc_bits_write(o, (unsigned long)c->dim, 16);
    orc_bits_write(o, (unsigned long)c->entries, 24);

    for (i = 1; i < c->entries; i++)
        if (c->lengthlist[i - 1] == 0 || c->lengthlist[i] < c->lengthlist[i - 1]) break;
    if (i == c->entries) ordered = 1;

    if (ordered) {
        long count = 0;
        orc_bits_write(o, 1, 1);
        orc_bits_write(o, (unsigned long)(c->lengthlist[0] - 1), 5);
        for (i = 1; i < c->entries; i++) {
            char this_ = c->lengthlist[i], last = c->lengthlist[i - 1];
            if (this_ > last) {
                for (j = last; j < this_; j++) {
                    orc_bits_write(o, (unsigned long)(i - count), orc_ilog((uint32_t)(c->entries - count)));
                    count = i;
                }
            }
        }
        orc_bits_write(o, (unsigned long)(i - count), orc_ilog((uint32_t)(c->entries - count)));
    } else {
        orc_bits_write(o, 0, 1);
        for (i = 0; i < c->entries; i++)
            if (c->lengthlist[i] == 0) break;
        if (i == c->entries) {
            orc_bits_write(o, 0, 1);
            for (i = 0; i < c->entries; i++) orc_bits_write(o, (unsigned long)(c->lengthlist[i] - 1), 5);
        } else {
            orc_bits_write(o, 1, 1);
            for (i = 0; i < c->entries; i++) {
                if (c->lengthlist[i] == 0) {
                    orc_bits_write(o, 0, 1);
                } else {
                    orc_bits_write(o, 1, 1);
                    orc_bits_write(o, (unsigned long)(c->lengthlist[i] - 1), 5);
                }
            }
        }
    }

    orc_bits_write(o, (unsigned long)c->maptype, 4);
    switch (c->maptype) {
    case 0:
        break;
    case 1:
    case 2: {
        long quantvals;
        if (!c->quantlist) return -1;
        orc_bits_write(o, (unsigned long)(uint32_t)c->q_min, 32);
        orc_bits_write(o, (unsigned long)(uint32_t)c->q_delta, 32);
        orc_bits_write(o, (unsigned long)(c->q_quant - 1), 4);
        orc_bits_write(o, (unsigned long)c->q_sequencep, 1);
        quantvals = c->maptype == 1 ? orc_maptype1_quantvals(c->entries, c->dim) : (long)c->entries * c->dim;
        for (i = 0; i < quantvals; i++) orc_bits_write(o, (unsigned long)labs((long)c->quantlist[i]), c->q_quant);
        break;
    }
    default:
        return -1;
    }
    return 0;
}

/* lib/floor1.c:77-113 */
static void pack_floor1(orc_bits *o, const orc_floor *info)
{
    int j, k, count = 0, rangebits, maxposit = info->postlist[1], maxclass = -1;
    orc_bits_write(o, (unsigned long)info->partitions, 5);
    for (j = 0; j < info->partitions; j++) {
        orc_bits_write(o, (unsigned long)info->partitionclass[j], 4);
        if (maxclass < info->partitionclass[j]) maxclass = info->partitionclass[j];
    }
    for (j = 0; j < maxclass + 1; j++) {
        orc_bits_write(o, (unsigned long)(info->class_dim[j] - 1), 3);
        orc_bits_write(o, (unsigned long)info->class_subs[j], 2);
        if (info->class_subs[j]) orc_bits_write(o, (unsigned long)info->class_book[j], 8);
        for (k = 0; k < (1 << info->class_subs[j]); k++) orc_bits_write(o, (unsigned long)(info->class_subbook[j][k] + 1), 8);
    }
    orc_bits_write(o, (unsigned long)(info->mult - 1), 2);
    orc_bits_write(o, (unsigned long)orc_ilog((uint32_t)(maxposit - 1)), 4);
    rangebits = orc_ilog((uint32_t)(maxposit - 1));
    for (j = 0, k = 0; j < info->partitions; j++) {
        count += info->class_dim[info->partitionclass[j]];
        for (; k < count; k++) orc_bits_write(o, (unsigned long)info->postlist[k + 2], rangebits);
    }
}

/* lib/res0.c:161-188 */
static void pack_res(orc_bits *o, const orc_residue *info)
{
    int j, acc = 0;
    orc_bits_write(o, (unsigned long)info->begin, 24);
    orc_bits_write(o, (unsigned long)info->end, 24);
    orc_bits_write(o, (unsigned long)(info->grouping - 1), 24);
    orc_bits_write(o, (unsigned long)(info->partitions - 1), 6);
    orc_bits_write(o, (unsigned long)info->groupbook, 8);
    for (j = 0; j < info->partitions; j++) {
        if (orc_ilog((uint32_t)info->secondstages[j]) > 3) {
            orc_bits_write(o, (unsigned long)info->secondstages[j], 3);
            orc_bits_write(o, 1, 1);
            orc_bits_write(o, (unsigned long)(info->secondstages[j] >> 3), 5);
        } else {
            orc_bits_write(o, (unsigned long)info->secondstages[j], 4);
        }
        acc += icount((unsigned int)info->secondstages[j]);
    }
    for (j = 0; j < acc; j++) orc_bits_write(o, (unsigned long)info->booklist[j], 8);
}

/* lib/mapping0.c:51-92 */
static void pack_map(orc_bits *o, const orc_setup *s, const orc_map *info)
{
    int i;
    if (info->submaps > 1) {
        orc_bits_write(o, 1, 1);
        orc_bits_write(o, (unsigned long)(info->submaps - 1), 4);
    } else {
        orc_bits_write(o, 0, 1);
    }
    if (info->coupling_steps > 0) {
        orc_bits_write(o, 1, 1);
        orc_bits_write(o, (unsigned long)(info->coupling_steps - 1), 8);
        for (i = 0; i < info->coupling_steps; i++) {
            orc_bits_write(o, (unsigned long)info->coupling_mag[i], orc_ilog((uint32_t)(s->channels - 1)));
            orc_bits_write(o, (unsigned long)info->coupling_ang[i], orc_ilog((uint32_t)(s->channels - 1)));
        }
    } else {
        orc_bits_write(o, 0, 1);
    }
    orc_bits_write(o, 0, 2);
    if (info->submaps > 1)
        for (i = 0; i < s->channels; i++) orc_bits_write(o, (unsigned long)info->chmuxlist[i], 4);
    for (i = 0; i < info->submaps; i++) {
        orc_bits_write(o, 0, 8);
        orc_bits_write(o, (unsigned long)info->floorsubmap[i], 8);
        orc_bits_write(o, (unsigned long)info->residuesubmap[i], 8);
    }
}

/* lib/info.c:558-617 */
static int pack_books(orc_bits *o, const orc_setup *s)
{
    int i;
    orc_bits_write(o, 0x05, 8);
    put_string(o, "vorbis", 6);
    orc_bits_write(o, (unsigned long)(s->books - 1), 8);
    for (i = 0; i < s->books; i++)
        if (pack_book(o, &s->book[i])) return -1;
    orc_bits_write(o, 0, 6);      /* times: placeholders */
    orc_bits_write(o, 0, 16);
    orc_bits_write(o, (unsigned long)(s->floors - 1), 6);
    for (i = 0; i < s->floors; i++) {
        orc_bits_write(o, 1, 16);                               /* floor type 1 */
        pack_floor1(o, &s->floor[i]);
    }
    orc_bits_write(o, (unsigned long)(s->residues - 1), 6);
    for (i = 0; i < s->residues; i++) {
        orc_bits_write(o, (unsigned long)s->residue[i].type, 16);
        pack_res(o, &s->residue[i]);
    }
    orc_bits_write(o, (unsigned long)(s->maps - 1), 6);
    for (i = 0; i < s->maps; i++) {
        orc_bits_write(o, 0, 16);                               /* mapping type 0 */
        pack_map(o, s, &s->map[i]);
    }
    orc_bits_write(o, (unsigned long)(s->modes - 1), 6);
    for (i = 0; i < s->modes; i++) {
        orc_bits_write(o, (unsigned long)s->mode_blockflag[i], 1);
        orc_bits_write(o, 0, 16);                               /* windowtype */
        orc_bits_write(o, 0, 16);                               /* transformtype */
        orc_bits_write(o, (unsigned long)s->mode_mapping[i], 8);
    }
    orc_bits_write(o, 1, 1);
    return 0;
}

long orc_header_packets(const orc_setup *s, const char *vendor, const char *const *comments, int ncomments,
                        unsigned char *buf, long cap, long *lens)
{
    orc_bits o[3];
    long total = 0, at = 0;
    int i, bad;
    if (!vendor) vendor = "AO; aoTuV [20110424] (based on libvorbis 1.3.7)";   /* lib/info.c:43 */
    for (i = 0; i < 3; i++) orc_bits_init(&o[i]);
    pack_info(&o[0], s);
    pack_comment(&o[1], vendor, comments, ncomments);
    bad = pack_books(&o[2], s);
    for (i = 0; i < 3; i++) {
        lens[i] = orc_bits_bytes(&o[i]);
        total += lens[i];
    }
    if (!bad && buf && cap >= total)
        for (i = 0; i < 3; i++) {
            memcpy(buf + at, o[i].buf, (size_t)lens[i]);
            at += lens[i];
        }
    for (i = 0; i < 3; i++) orc_bits_clear(&o[i]);
    return bad ? -1 : total;
}
