/* ORACLE — test infrastructure only.
 *
 * LSb-first bit packer (the subset of libogg's oggpack_* that the encoder uses; semantics
 * from the reference's doc/02-bitpacking.tex and from its own fast writer
 * lib/codebook.c:80-110) and the encode side of the codebooks:
 *   ov_ilog                   lib/sharedbook.c:35-39
 *   _float32_unpack           lib/sharedbook.c:66-80
 *   _make_words               lib/sharedbook.c:85-169
 *   _book_maptype1_quantvals  lib/sharedbook.c:174-207
 *   vorbis_book_init_encode   lib/sharedbook.c:303-317
 *   vorbis_book_encode        lib/codebook.c:402-410
 *   local_book_besterror      lib/res0.c:316-378
 */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

/* ---- bits ------------------------------------------------------------------------ */
void orc_bits_init(orc_bits *b)
{
    b->storage = 256;
    b->buf = (unsigned char *)calloc(1, (size_t)b->storage);
    b->endbyte = 0;
    b->endbit = 0;
}

void orc_bits_reset(orc_bits *b)
{
    b->endbyte = 0;
    b->endbit = 0;
    b->buf[0] = 0;
}

void orc_bits_clear(orc_bits *b)
{
    free(b->buf);
    memset(b, 0, sizeof(*b));
}

void orc_bits_write(orc_bits *b, unsigned long value, int bits)
{
    int i;
    if (bits <= 0) return;
    if (b->endbyte + 8 >= b->storage) {
        long ns = b->storage + 256;
        b->buf = (unsigned char *)realloc(b->buf, (size_t)ns);
        memset(b->buf + b->storage, 0, (size_t)(ns - b->storage));
        b->storage = ns;
    }
    if (bits < 32) value &= (1UL << bits) - 1;
    else value &= 0xffffffffUL;
    /* bit-serial on purpose: unambiguous, and speed is irrelevant for the checker */
    for (i = 0; i < bits; i++) {
        if (b->endbit == 0) b->buf[b->endbyte] = 0;
        b->buf[b->endbyte] |= (unsigned char)(((value >> i) & 1UL) << b->endbit);
        if (++b->endbit == 8) {
            b->endbit = 0;
            b->endbyte++;
        }
    }
}

long orc_bits_bytes(const orc_bits *b) { return b->endbyte + (b->endbit + 7) / 8; }

/* libogg oggpack_writetrunc: cut the buffer back to `bits` bits, clearing the bits above in the last byte */
void orc_bits_writetrunc(orc_bits *b, long bits)
{
    long bytes = bits >> 3;
    if (!b->buf) return;
    bits -= bytes * 8;
    b->endbyte = bytes;
    b->endbit = (int)bits;
    b->buf[bytes] &= (unsigned char)((1u << bits) - 1);
}

/* ---- books ------------------------------------------------------------------------- */
int orc_ilog(uint32_t v)
{
    int ret;
    for (ret = 0; v; ret++) v >>= 1;
    return ret;
}

/* lib/sharedbook.c:46-48, 66-80 */
static float float32_unpack(long val)
{
    double mant = val & 0x1fffff;
    int sign = (int)(val & 0x80000000);
    long e = (val & 0x7fe00000L) >> 21;
    if (sign) mant = -mant;
    e = e - (21 - 1) - 768;
    if (e > 63) e = 63;
    if (e < -63) e = -63;
    return (float)ldexp(mant, (int)e);
}

/* canonical Huffman assignment, lowest codeword first, then bit-reversed for the LSb packer */
static uint32_t *make_words(const signed char *l, long n)
{
    long i, j, count = 0;
    uint32_t marker[33];
    uint32_t *r = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(*r));
    memset(marker, 0, sizeof(marker));
    for (i = 0; i < n; i++) {
        long length = l[i];
        if (length > 0) {
            uint32_t entry = marker[length];
            if (length < 32 && (entry >> length)) { free(r); return NULL; }
            r[count++] = entry;
            for (j = length; j > 0; j--) {
                if (marker[j] & 1) {
                    if (j == 1) marker[1]++;
                    else marker[j] = marker[j - 1] << 1;
                    break;
                }
                marker[j]++;
            }
            for (j = length + 1; j < 33; j++) {
                if ((marker[j] >> 1) == entry) {
                    entry = marker[j];
                    marker[j] = marker[j - 1] << 1;
                } else
                    break;
            }
        } else
            count++; /* sparsecount == 0 on the encode side */
    }
    for (i = 0, count = 0; i < n; i++) {
        uint32_t temp = 0;
        for (j = 0; j < l[i]; j++) {
            temp <<= 1;
            temp |= (r[count] >> j) & 1;
        }
        r[count++] = temp;
    }
    return r;
}

long orc_maptype1_quantvals(long entries, long dim)
{
    long vals;
    if (entries < 1) return 0;
    vals = (long)floor(pow((float)entries, 1.f / dim));
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

void orc_book_init_encode(orc_book *c)
{
    c->codelist = make_words(c->lengthlist, c->entries);
    c->quantvals = (int)orc_maptype1_quantvals(c->entries, c->dim);
    c->minval = (int)rint(float32_unpack(c->q_min));
    c->delta = (int)rint(float32_unpack(c->q_delta));
}

int orc_book_encode(const orc_book *b, int a, orc_bits *opb)
{
    if (a < 0 || a >= b->entries) return 0;
    orc_bits_write(opb, b->codelist[a], b->lengthlist[a]);
    return b->lengthlist[a];
}

/* Nearest lattice entry with the reference's tie rule (lowest entry index wins) and the
 * in-place subtraction of the chosen point. */
int orc_book_besterror(const orc_book *book, int *a)
{
    int dim = book->dim;
    int i, j, o;
    int minval = book->minval, del = book->delta, qv = book->quantvals;
    int ze = qv >> 1;
    int index = 0;
    int p[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    if (del != 1) {
        for (i = 0, o = dim; i < dim; i++) {
            int v = (a[--o] - minval + (del >> 1)) / del;
            int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
            index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
            p[o] = v * del + minval;
        }
    } else {
        for (i = 0, o = dim; i < dim; i++) {
            int v = a[--o] - minval;
            int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
            index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
            p[o] = v * del + minval;
        }
    }

    if (book->lengthlist[index] <= 0) {
        int best = -1;
        int e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int maxval = book->minval + book->delta * (book->quantvals - 1);
        for (i = 0; i < book->entries; i++) {
            if (book->lengthlist[i] > 0) {
                int dist = 0;
                for (j = 0; j < dim; j++) {
                    int val = e[j] - a[j];
                    dist += val * val;
                }
                if (best == -1 || dist < best) {
                    memcpy(p, e, sizeof(p));
                    best = dist;
                    index = i;
                }
            }
            /* lattice odometer: 0,+d,-d,+2d,-2d,... per component, component 0 fastest
               (the reference also steps it after the last entry, reading one past e[]; that
               value is never used, so the step is skipped here) */
            if (i + 1 >= book->entries) break;
            j = 0;
            while (e[j] >= maxval) e[j++] = 0;
            if (e[j] >= 0) e[j] += book->delta;
            e[j] = -e[j];
        }
    }

    if (index > -1)
        for (i = 0; i < dim; i++) *a++ -= p[i];
    return index;
}

/* test hook: {quantvals, minval, delta} of a maptype-1 book header (lib/sharedbook.c:303-317) and the unpacked
 * floats themselves, for the reference's self-test books (lib/sharedbook.c:474-560) */
void orc_book_lattice(long q_min, long q_delta, long entries, long dim, int *out, float *unpacked)
{
    out[0] = (int)orc_maptype1_quantvals(entries, dim);
    out[1] = (int)rint(float32_unpack(q_min));
    out[2] = (int)rint(float32_unpack(q_delta));
    unpacked[0] = float32_unpack(q_min);
    unpacked[1] = float32_unpack(q_delta);
}

