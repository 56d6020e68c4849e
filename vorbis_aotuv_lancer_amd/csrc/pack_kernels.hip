// Packet assembly for gfx950: header bits, floor-1 entropy coding, residue classification,
// cascaded lattice-VQ encode and the aoTuV block-state update, one lane per stream-block.
//
// mapping0_forward loop C (reference lib/mapping0.c:1204-1313) for the VBR blob, as launches:
//   k_pack_head     packet type / mode / window bits :1211-1218; floor1_encode's bit emission
//                   (lib/floor1.c:856-942); nonzero[] propagation over the coupling steps
//                   (lib/psy.c:5133-5140).  Serial, short.
//   k_block_state   the aoTuV block-state update of lib/mapping0.c:1297-1305.  It only depends on the block
//                   type and is launched right after offset_and_mix: the psychoacoustics of the stream's
//                   next block read it, the rest of this block's path does not.
//   per residue submap (lib/mapping0.c:1273-1295):
//   k_res_vq        res*_class (_01class lib/res0.c:406-468, _2class :473-526) and the res2
//                   interleave (:781-787) of its partition slice, then
//                   the cascade of _01forward (:528-640): _encodepart :384-404 ->
//                   local_book_besterror :316-378.  A partition's stages only touch that
//                   partition's samples, so partitions are sliced over blockIdx.y; every codeword
//                   goes to a scratch slot (code | length << 32) and the partition's bit count per
//                   stage to lenT.
//   k_res_offsets   walks _01forward's emission order (stage, partition word, [phrase codeword],
//                   partition, vector) over the bit counts: integer prefix sum -> bit offset of
//                   every (stage, partition, vector) run; emits the phrase codewords.  Serial, short.
//   k_res_emit      ORs every run into the packet at its offset (sliced over partitions).
// Bits are appended LSb first (libogg oggpack semantics).  Packet buffers are 32-bit-word-major
// tiles packetT[word][64 lanes], zeroed before k_pack_head; runs written by different wavefronts
// meet inside words, hence atomicOr.  packet_bytes[sb] = oggpack_bytes(), or -1 on overflow.
// The nearest-codeword search walks the compact list of used entries (ascending entry order,
// so the reference's lowest-index tie rule holds) instead of stepping the lattice odometer
// through unused entries.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "batch.h"
#include "kernels.h"

namespace {

// append-only writer of the head kernel (the lane owns its packet exclusively at that point)
struct BitW {
    uint32_t *base;    // &packet words [0][lane], row stride 64
    int nwords, maxwords;
    uint64_t acc;
    int nbits;
};

__device__ __forceinline__ void bw_write(BitW &w, uint32_t value, int bits)
{
    if (bits <= 0) return;
    if (bits < 32) value &= (1u << bits) - 1u;
    w.acc |= (uint64_t)value << w.nbits;
    w.nbits += bits;
    if (w.nbits >= 32) {
        if (w.nwords < w.maxwords) w.base[(size_t)w.nwords * 64] = (uint32_t)w.acc;
        w.acc >>= 32;
        w.nbits -= 32;
        w.nwords++;
    }
}

// returns the bit position after the last bit written
__device__ __forceinline__ int bw_finish(BitW &w)
{
    if (w.nbits > 0 && w.nwords < w.maxwords) w.base[(size_t)w.nwords * 64] = (uint32_t)w.acc;
    return w.nwords * 32 + w.nbits;
}

// OR `bits` low bits of `value` into the packet at bit position `pos`
__device__ __forceinline__ void or_bits(uint32_t *words, int maxwords, int pos, uint32_t value, int bits)
{
    if (bits <= 0) return;
    if (bits < 32) value &= (1u << bits) - 1u;
    const uint64_t v = (uint64_t)value << (pos & 31);
    const int wi = pos >> 5;
    if (wi < maxwords) atomicOr(&words[(size_t)wi * 64], (uint32_t)v);
    if ((v >> 32) && wi + 1 < maxwords) atomicOr(&words[(size_t)(wi + 1) * 64], (uint32_t)(v >> 32));
}

__device__ __forceinline__ int ilog(uint32_t v)
{
    return v ? 32 - __clz(v) : 0;
}

__device__ __forceinline__ int book_encode(const vbm_book *bk, int a, BitW &w)
{
    if (a < 0 || a >= bk->entries) return 0;
    bw_write(w, bk->codelist[a], bk->lengthlist[a]);
    return bk->lengthlist[a];
}

// the fields of a book the VQ search reads per vector, fetched once per (partition, stage): the book
// is addressed per lane, and the compiler cannot hoist its loads over the stores of the search loop
struct book_regs {
    int dim, minval, delta, quantvals, used, entries;
    const signed char *lengthlist;
    const uint32_t *codelist;
    const int *used_point, *used_index;
    const short *used_pack;
    const int *used_norm;
};
__device__ __forceinline__ book_regs load_book(const vbm_book *book)
{
    book_regs r;
    r.dim = book->dim; r.minval = book->minval; r.delta = book->delta; r.quantvals = book->quantvals;
    r.used = book->used; r.entries = book->entries;
    r.lengthlist = book->lengthlist; r.codelist = book->codelist;
    r.used_point = book->used_point; r.used_index = book->used_index;
    r.used_pack = book->used_pack; r.used_norm = book->used_norm;
    return r;
}

// num / den as C computes it (truncation toward zero) for den > 0.  Below 2^23 the correctly rounded float
// quotient cannot reach the next integer (it is at least 1/den away, the rounding error is below
// |num| / den * 2^-24), so its truncation is the integer quotient: ~10 instructions instead of ~40.
__device__ __forceinline__ int div_trunc(int num, int den)
{
    if (abs(num) < (1 << 23)) return (int)((float)num / (float)den);
    return num / den;
}

// local_book_besterror (lib/res0.c:316-378); a[] is the vector (dim <= 8) in registers
__device__ __forceinline__ int besterror(const book_regs *book, int *a)
{
    const int dim = book->dim;
    int i, j, o;
    const int minval = book->minval, del = book->delta, qv = book->quantvals;
    const int ze = (qv >> 1);
    int index = 0;
    int p[VBM_MAX_BOOK_DIM] = {0, 0, 0, 0, 0, 0, 0, 0};

    if (del != 1) {
        for (i = 0, o = dim; i < dim; i++) {
            int v = div_trunc(a[--o] - minval + (del >> 1), del);
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
        // exhaustive search over the entries that have a codeword, first minimum wins (lib/res0.c:343-370).
        // |pt - a|^2 = |pt|^2 - 2 pt.a + |a|^2: the last term is common, so entries are compared by
        // |pt|^2 - 2 pt.a (same order, same ties); pt.a as packed 16-bit dot products, one 16-byte
        // load per entry.  Only the winner's index is tracked; its point is fetched afterwards.
        const int used = book->used;
        int bi = 0;
        bool small = book->used_pack != nullptr;
        for (j = 0; j < dim; j++) small = small && (a[j] >= -32768 && a[j] <= 32767);
        if (small) {
            typedef short short2v __attribute__((ext_vector_type(2)));
            uint32_t pa[4] = {0u, 0u, 0u, 0u};
            for (j = 0; j < dim; j++) pa[j >> 1] |= ((uint32_t)a[j] & 0xffffu) << ((j & 1) * 16);
            const uint4 *pk = reinterpret_cast<const uint4 *>(book->used_pack);
            const int *__restrict__ nrm = book->used_norm;
            const int words = (dim + 1) >> 1;
            int best = 0;
            for (i = 0; i < used; i++) {
                const uint4 v = pk[i];
                int dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.x), __builtin_bit_cast(short2v, pa[0]), 0, false);
                if (words > 1) dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.y), __builtin_bit_cast(short2v, pa[1]), dot, false);
                if (words > 2) {
                    dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.z), __builtin_bit_cast(short2v, pa[2]), dot, false);
                    dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.w), __builtin_bit_cast(short2v, pa[3]), dot, false);
                }
                const int score = nrm[i] - 2 * dot;
                if (i == 0 || score < best) { best = score; bi = i; }
            }
        } else {
            int best = -1;
            const int *pt = book->used_point;
            for (i = 0; i < used; i++, pt += dim) {
                int dist = 0;
                for (j = 0; j < dim; j++) {
                    int val = pt[j] - a[j];
                    dist += val * val;
                }
                if (best == -1 || dist < best) { best = dist; bi = i; }
            }
        }
        if (used > 0) {
            const int *pt = book->used_point + (size_t)bi * dim;
            for (j = 0; j < dim; j++) p[j] = pt[j];
            index = book->used_index[bi];
        }
    }

    if (index > -1)
        for (i = 0; i < dim; i++) a[i] -= p[i];
    return index;
}

// local_book_besterror split in two for the fused kernel.  vq_lattice: the lattice step (lib/res0.c:316-341) — quantise every
// component, index of the lattice point, the point itself in p[]; returns true when that entry has no codeword, i.e. the
// exhaustive search of :343-370 has to find the nearest entry that has one.  That search is rare per vector and long (it
// visits every used entry): run by the lane that needs it, it makes the whole wavefront wait for `used` steps whenever ANY
// of its 64 vectors needs it.  vq_search_wave instead serves one such vector with all 64 lanes: lane l scores the used
// entries l, l + 64, ... and a wave reduction picks the minimum, the lowest entry on ties (the source keeps the first).
__device__ __forceinline__ bool vq_lattice(const book_regs *book, const int *a, int &index, int *p)
{
    const int dim = book->dim;
    const int minval = book->minval, del = book->delta, qv = book->quantvals;
    const int ze = (qv >> 1);
    index = 0;
    int i, o;
    if (del != 1) {
        for (i = 0, o = dim; i < dim; i++) {
            int v = div_trunc(a[--o] - minval + (del >> 1), del);
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
    return book->lengthlist[index] <= 0;
}

// the vector av[] (the same in every lane) against all used entries; returns the position in the used list (every lane)
__device__ __forceinline__ int vq_search_wave(const book_regs *book, const int *av, const int lane)
{
    const int dim = book->dim, used = book->used;
    bool small = book->used_pack != nullptr;
    for (int j = 0; j < dim; j++) small = small && (av[j] >= -32768 && av[j] <= 32767);
    long long best = 0x7fffffffffffffffLL;       // (score << 32) | position: one compare orders by score, then by position
    if (small) {
        typedef short short2v __attribute__((ext_vector_type(2)));
        uint32_t pa[4] = {0u, 0u, 0u, 0u};
        for (int j = 0; j < dim; j++) pa[j >> 1] |= ((uint32_t)av[j] & 0xffffu) << ((j & 1) * 16);
        const uint4 *pk = reinterpret_cast<const uint4 *>(book->used_pack);
        const int *__restrict__ nrm = book->used_norm;
        const int words = (dim + 1) >> 1;
        for (int i = lane; i < used; i += 64) {
            const uint4 v = pk[i];
            int dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.x), __builtin_bit_cast(short2v, pa[0]), 0, false);
            if (words > 1) dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.y), __builtin_bit_cast(short2v, pa[1]), dot, false);
            if (words > 2) {
                dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.z), __builtin_bit_cast(short2v, pa[2]), dot, false);
                dot = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v.w), __builtin_bit_cast(short2v, pa[3]), dot, false);
            }
            const long long key = ((long long)(nrm[i] - 2 * dot) << 32) | (unsigned)i;
            if (key < best) best = key;
        }
    } else {
        const int *pt0 = book->used_point;
        for (int i = lane; i < used; i += 64) {
            const int *pt = pt0 + (size_t)i * dim;
            int dist = 0;
            for (int j = 0; j < dim; j++) {
                const int val = pt[j] - av[j];
                dist += val * val;
            }
            const long long key = ((long long)dist << 32) | (unsigned)i;
            if (key < best) best = key;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long o = __shfl_xor(best, off);
        if (o < best) best = o;
    }
    return (int)(best & 0xffffffffLL);
}

// which channels of submap `sm` take part, and the residue's vector shape
struct res_view {
    const vbm_residue *r;
    int nb;                       // channels in the submap
    int chlist[VBM_MAXCH];
    int used;                     // vectors that are coded (res2: 0 or 1; res0/1: nonzero channels)
    int vch[VBM_MAXCH];           // res0/1: channel of vector j
    int partvals, spp;
};

__device__ __forceinline__ res_view residue_view(const vbm_batch &b, const vbm_map *info, int sm, size_t col0)
{
    res_view v;
    v.r = &b.setup->residue[info->residuesubmap[sm]];
    v.nb = 0;
    v.used = 0;
    int any = 0;
    for (int j = 0; j < b.ch; j++)
        if (info->chmuxlist[j] == sm) {
            const int nz = b.nonzero[col0 + j] ? 1 : 0;
            v.chlist[v.nb++] = j;
            any |= nz;
            if (nz && v.r->type != 2) v.vch[v.used++] = j;
        }
    if (v.r->type == 2) v.used = any ? 1 : 0;
    v.spp = v.r->grouping;
    v.partvals = (v.r->end - v.r->begin) / v.spp;
    return v;
}

#define SBT(sb) ((size_t)((sb) >> 6) * b.sb_slab_words + ((sb) & 63))
#define IWC(cc, x) b.iworkT[(size_t)((col0 + (cc)) >> 6) * b.slab_words + (size_t)(x) * 64 + ((col0 + (cc)) & 63)]
#define PW(jv, iv) partword[((size_t)(jv) * v.partvals + (iv)) * 64]
#define LEN(stg, jv, iv) lenT[(((size_t)(stg) * b.ch + (jv)) * v.partvals + (iv)) * 64]
#define OFF(stg, jv, iv) offT[(((size_t)(stg) * b.ch + (jv)) * v.partvals + (iv)) * 64]

// nonzero[] after coupling (lib/psy.c:5133-5140); couple/quantise reads the flags as floor1_encode left them, so
// this runs after it and before the residue kernels
template <bool BLOBS>
__global__ void k_nonzero_propagate(vbm_batch b)
{
    vbm_blob_enter<BLOBS>(b);
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_map *info = &b.setup->map[b.W];
    const size_t col0 = (size_t)sb * b.ch;
    for (int i = 0; i < info->coupling_steps; i++) {
        const size_t m = col0 + info->coupling_mag[i], a = col0 + info->coupling_ang[i];
        if (b.nonzero[m] || b.nonzero[a]) {
            b.nonzero[m] = 1;
            b.nonzero[a] = 1;
        }
    }
}

// packet header and the floors' bits: needs floor1_encode's values only, so it may run beside couple/quantise
template <bool BLOBS>
__global__ void k_pack_head(vbm_batch b)
{
    vbm_blob_enter<BLOBS>(b);
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const size_t SW = b.slab_words;
    const vbm_setup *s = b.setup;
    const vbm_map *info = &s->map[b.W];
    const int ch = b.ch;
    const size_t col0 = (size_t)sb * ch;
    int i, j, k;

    BitW w;
    w.base = (uint32_t *)b.packetT + (size_t)(sb >> 6) * (b.max_packet_bytes / 4) * 64 + (sb & 63);
    w.nwords = 0;
    w.maxwords = b.max_packet_bytes / 4;
    w.acc = 0;
    w.nbits = 0;

    // packet type, mode number, window flags (lib/mapping0.c:1211-1218)
    bw_write(w, 0, 1);
    bw_write(w, (uint32_t)b.W, s->modebits);
    if (b.W) {
        bw_write(w, b.wflags[sb] & 1, 1);
        bw_write(w, (b.wflags[sb] >> 1) & 1, 1);
    }

    // ---- floors, channel by channel (lib/floor1.c:856-942, :969-972) -----------------------
    for (int c = 0; c < ch; c++) {
        const size_t col = col0 + c;
        const vbm_floor *look = &s->floor[info->floorsubmap[info->chmuxlist[c]]];
        if (!b.post_valid[col]) {
            bw_write(w, 0, 1);
            continue;
        }
#define OUTV(x) b.floor_outT[(size_t)(col >> 6) * SW + (size_t)(x) * 64 + (col & 63)]
        bw_write(w, 1, 1);
        bw_write(w, (uint32_t)OUTV(0), ilog(look->quant_q - 1));
        bw_write(w, (uint32_t)OUTV(1), ilog(look->quant_q - 1));

        for (i = 0, j = 2; i < look->partitions; i++) {
            int cls = look->partitionclass[i];
            int cdim = look->class_dim[cls];
            int csubbits = look->class_subs[cls];
            int csub = 1 << csubbits;
            int bookas[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int cval = 0;
            int cshift = 0;
            int l;
            // the partition's values, then its codewords, are fetched together before any bit is written (the
            // packet stores would otherwise sit between dependent loads: loads and stores retire in order)
            int ov[8];
#pragma unroll
            for (k = 0; k < 8; k++) ov[k] = (k < cdim) ? OUTV(j + k) : 0;

            if (csubbits) {
                int maxval[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (k = 0; k < csub; k++) {
                    int booknum = look->class_subbook[cls][k];
                    if (booknum < 0) maxval[k] = 1;
                    else maxval[k] = s->book[booknum].entries;
                }
#pragma unroll
                for (k = 0; k < 8; k++) {
                    if (k < cdim) {
                        for (l = 0; l < csub; l++) {
                            if (ov[k] < maxval[l]) {
                                bookas[k] = l;
                                break;
                            }
                        }
                        cval |= bookas[k] << cshift;
                        cshift += csubbits;
                    }
                }
            }
            int blen[8];
            uint32_t bcode[8];
#pragma unroll
            for (k = 0; k < 8; k++) {
                blen[k] = 0;
                bcode[k] = 0;
                if (k < cdim) {
                    const int book = look->class_subbook[cls][bookas[k]];
                    if (book >= 0) {
                        const vbm_book *bk = &s->book[book];
                        if (ov[k] >= 0 && ov[k] < bk->entries) {
                            blen[k] = bk->lengthlist[ov[k]];
                            bcode[k] = bk->codelist[ov[k]];
                        }
                    }
                }
            }
            if (csubbits) book_encode(&s->book[look->class_book[cls]], cval, w);
#pragma unroll
            for (k = 0; k < 8; k++)
                if (k < cdim && blen[k] > 0) bw_write(w, bcode[k], blen[k]);
            j += cdim;
        }
#undef OUTV
    }
    b.packet_bits[sb] = bw_finish(w);

}

// classification of partitions [i0, i1) of one stream-block (res*_class) + the res2 interleave
// `reps`: the update sits inside the reference's loop over the packetblobs (lib/mapping0.c:1204, :1296-1304),
// so a managed-bitrate stream applies it PACKETBLOBS times per block
__global__ void k_block_state(vbm_batch b, int reps)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const int sid = b.stream_id[sb];
    const int block_mode = b.block_mode;
    int impadnum = b.st.impadnum[sid];
    int lWbm = b.st.lW_block_mode[sid];
    int lW_no = b.st.lW_no[sid];
    for (int r = 0; r < reps; r++) {
        if (block_mode >= 2) impadnum = 0;
        if ((!lWbm) && (block_mode == 1)) impadnum = 1;
        else if (impadnum && impadnum < 8) impadnum++;
        if (lWbm == block_mode) lW_no++;
        else lW_no = 1;
        lWbm = block_mode;
    }
    b.st.impadnum[sid] = impadnum;
    b.st.lW_no[sid] = lW_no;
    b.st.lW_block_mode[sid] = block_mode;
}

// vorbis_bitrate_addblock, managed branch (lib/bitrate.c:98-226), one lane per stream-block: choose one of the
// PACKETBLOBS packets from the stream's reservoirs, settle its final length (truncated to the ceiling or
// zero-padded to the floor), update the reservoirs.  The chosen packet is gathered by k_blob_gather.
__global__ void k_bitrate_choose(vbm_batch b)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_setup *s = b.setup;
    const int sid = b.stream_id[sb];
    const int *__restrict__ sizes = b.packet_bytes_blob + sb;      // blob k at [k * Ls]
    const size_t Ls = (size_t)b.Ls;
#define BYTES(k) ((long long)sizes[(size_t)(k) * Ls])
    // vorbis_bitrate_init (lib/bitrate.c:28-56)
    const long long ratesamples = s->rate;
    const int halfsamples = s->blocksizes[0] >> 1;
    const long long short_per_long = s->blocksizes[1] / s->blocksizes[0];
    const long long avg_bitsper = (long long)rint(1. * (double)s->bi_avg_rate * halfsamples / (double)ratesamples);
    const long long min_bitsper = (long long)rint(1. * (double)s->bi_min_rate * halfsamples / (double)ratesamples);
    const long long max_bitsper = (long long)rint(1. * (double)s->bi_max_rate * halfsamples / (double)ratesamples);

    long long avg_reservoir = b.st.bm_avg_reservoir[sid], minmax_reservoir = b.st.bm_minmax_reservoir[sid];
    double avgfloat = b.st.bm_avgfloat[sid];

    int choice = (int)rint(avgfloat);
    long long this_bits = BYTES(choice) * 8;
    const long long min_target_bits = (b.W ? min_bitsper * short_per_long : min_bitsper);
    const long long max_target_bits = (b.W ? max_bitsper * short_per_long : max_bitsper);
    const int samples = s->blocksizes[b.W] >> 1;
    const long long desired_fill = (long long)((double)s->bi_reservoir_bits * s->bi_reservoir_bias);
    long long final_bytes;

    if (avg_bitsper > 0) {
        double slew = 0.;
        const long long avg_target_bits = (b.W ? avg_bitsper * short_per_long : avg_bitsper);
        const double slewlimit = 15. / s->bi_slew_damp;
        if (avg_reservoir + (this_bits - avg_target_bits) > desired_fill) {
            while (choice > 0 && this_bits > avg_target_bits && avg_reservoir + (this_bits - avg_target_bits) > desired_fill) {
                choice--;
                this_bits = BYTES(choice) * 8;
            }
        } else if (avg_reservoir + (this_bits - avg_target_bits) < desired_fill) {
            while (choice + 1 < VBM_PACKETBLOBS && this_bits < avg_target_bits &&
                   avg_reservoir + (this_bits - avg_target_bits) < desired_fill) {
                choice++;
                this_bits = BYTES(choice) * 8;
            }
        }
        slew = rint((double)choice - avgfloat) / samples * (double)s->rate;
        if (slew < -slewlimit) slew = -slewlimit;
        if (slew > slewlimit) slew = slewlimit;
        avgfloat += slew / (double)s->rate * samples;
        choice = (int)rint(avgfloat);
        this_bits = BYTES(choice) * 8;
    }

    if (min_bitsper > 0) {
        if (this_bits < min_target_bits) {
            while (minmax_reservoir - (min_target_bits - this_bits) < 0) {
                choice++;
                if (choice >= VBM_PACKETBLOBS) break;
                this_bits = BYTES(choice) * 8;
            }
        }
    }
    if (max_bitsper > 0) {
        if (this_bits > max_target_bits) {
            while (minmax_reservoir + (this_bits - max_target_bits) > s->bi_reservoir_bits) {
                choice--;
                if (choice < 0) break;
                this_bits = BYTES(choice) * 8;
            }
        }
    }

    if (choice < 0) {
        const long long maxsize = (max_target_bits + (s->bi_reservoir_bits - minmax_reservoir)) / 8;
        choice = 0;
        final_bytes = BYTES(choice);
        if (final_bytes > maxsize) final_bytes = maxsize;          // oggpack_writetrunc(maxsize*8)
        this_bits = final_bytes * 8;
    } else {
        long long minsize = (min_target_bits - minmax_reservoir + 7) / 8;
        if (choice >= VBM_PACKETBLOBS) choice = VBM_PACKETBLOBS - 1;
        final_bytes = BYTES(choice);
        minsize -= final_bytes;
        if (minsize > 0) final_bytes += minsize;                    // zero bytes appended
        this_bits = final_bytes * 8;
    }

    if (min_bitsper > 0 || max_bitsper > 0) {
        if (max_target_bits > 0 && this_bits > max_target_bits) {
            minmax_reservoir += (this_bits - max_target_bits);
        } else if (min_target_bits > 0 && this_bits < min_target_bits) {
            minmax_reservoir += (this_bits - min_target_bits);
        } else {
            if (minmax_reservoir > desired_fill) {
                if (max_target_bits > 0) {
                    minmax_reservoir += (this_bits - max_target_bits);
                    if (minmax_reservoir < desired_fill) minmax_reservoir = desired_fill;
                } else {
                    minmax_reservoir = desired_fill;
                }
            } else {
                if (min_target_bits > 0) {
                    minmax_reservoir += (this_bits - min_target_bits);
                    if (minmax_reservoir > desired_fill) minmax_reservoir = desired_fill;
                } else {
                    minmax_reservoir = desired_fill;
                }
            }
        }
    }
    if (avg_bitsper > 0) {
        const long long avg_target_bits = (b.W ? avg_bitsper * short_per_long : avg_bitsper);
        avg_reservoir += this_bits - avg_target_bits;
    }
#undef BYTES
    b.st.bm_avg_reservoir[sid] = avg_reservoir;
    b.st.bm_minmax_reservoir[sid] = minmax_reservoir;
    b.st.bm_avgfloat[sid] = avgfloat;
    b.choice[sb] = choice;
    // a padded packet may not outgrow the buffer (the reference's packer would simply grow)
    b.packet_bytes[sb] = final_bytes <= b.max_packet_bytes ? (int)final_bytes : -1;
}

// packet words of the chosen blob -> dst[sb][max_packet_bytes/4] (vorbis_bitrate_flushpacket, lib/bitrate.c:229-252);
// bytes past the final length are zero (truncation), bytes past the blob's own length already are (padding)
__global__ void k_blob_gather(vbm_batch b, uint32_t *__restrict__ dst)
{
    __shared__ uint32_t tile[64][65];
    const int rows = b.max_packet_bytes / 4;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t blob_stride = (size_t)b.Ls * b.max_packet_bytes;   // bytes per blob
    {
        const int c = c0 + tx;
        const bool live = c < vbm_nsb(b);
        const int k = live ? b.choice[c] : 0;
        const int len = live ? b.packet_bytes[c] : 0;
        const uint32_t *src = (const uint32_t *)(b.packetT_blob + (size_t)k * blob_stride) + (size_t)(c >> 6) * rows * 64 + (c & 63);
        for (int rr = ty; rr < 64; rr += 4) {
            const int r = r0 + rr;
            uint32_t w = 0;
            if (live && r < rows && r * 4 < len) {
                w = src[(size_t)r * 64];
                const int keep = len - r * 4;                      // bytes of this word inside the packet
                if (keep < 4) w &= (1u << (8 * keep)) - 1u;
            }
            tile[rr][tx] = w;
        }
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (c < vbm_nsb(b) && r < rows) dst[(size_t)c * rows + r] = tile[tx][cc];
    }
}

__device__ __forceinline__ void res_classify(const vbm_batch &b, const res_view &v, const int sb, const size_t col0, const int i0, const int i1)
{
    const vbm_residue *r = v.r;
    int *partword = b.partwordT + SBT(sb);
    const int possible_partitions = r->partitions;
    const int rbegin = r->begin;
    const int *__restrict__ classmetric1 = r->classmetric1, *__restrict__ classmetric2 = r->classmetric2;
    int i, j, k;

    if (r->type == 2) {
        // _2class (lib/res0.c:473-526)
        const int nb = v.nb;
        const int lsteps = (v.spp + nb - 1) / nb;   // the source advances l once per nb samples of a partition
        for (i = i0; i < i1; i++) {
            int magmax = 0, angmax = 0;
            int l = rbegin / nb + i * lsteps;
            for (j = 0; j < v.spp; j += nb, l++) {
                int v0 = abs(IWC(v.chlist[0], l));
                if (v0 > magmax) magmax = v0;
                for (k = 1; k < nb; k++) {
                    int vk = abs(IWC(v.chlist[k], l));
                    if (vk > angmax) angmax = vk;
                }
            }
            for (j = 0; j < possible_partitions - 1; j++)
                if (magmax <= classmetric1[j] && angmax <= classmetric2[j]) break;
            PW(0, i) = j;
        }
    } else {
        // _01class (lib/res0.c:406-468): only the nonzero channels take part (:715-745)
        const float scale = (float)(100. / v.spp);
        for (i = i0; i < i1; i++) {
            const int offset = i * v.spp + rbegin;
            for (j = 0; j < v.used; j++) {
                int mx = 0, ent = 0;
                for (k = 0; k < v.spp; k++) {
                    int a = abs(IWC(v.vch[j], offset + k));
                    if (a > mx) mx = a;
                    ent += a;
                }
                ent = (int)((float)ent * scale);
                for (k = 0; k < possible_partitions - 1; k++)
                    if (mx <= classmetric1[k] && (classmetric2[k] < 0 || ent < classmetric2[k])) break;
                PW(j, i) = k;
            }
        }
    }
}

template <bool BLOBS>
__global__ void k_res_vq(vbm_batch b, int sm, int nchunks)
{
    vbm_blob_enter<BLOBS>(b);
    extern __shared__ int vq_lds[];   // [samples per partition][64 lanes]
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_setup *s = b.setup;
    const vbm_map *info = &s->map[b.W];
    const size_t col0 = (size_t)sb * b.ch;
    const res_view v = residue_view(b, info, sm, col0);
    if (!v.used) return;
    const vbm_residue *r = v.r;
    const int i0 = (int)((long)v.partvals * blockIdx.y / nchunks), i1 = (int)((long)v.partvals * (blockIdx.y + 1) / nchunks);
    // this slice's classes first (the same lane reads them back below)
    res_classify(b, v, sb, col0, i0, i1);
    const int *partword = b.partwordT + SBT(sb);
    int *lenT = b.vqlenT + SBT(sb);
    uint64_t *slot = b.vqcodeT + (size_t)(sb >> 6) * b.vq_slab_words + (sb & 63);
    const int veclen = (r->type == 2) ? b.n * v.nb : b.n;
    const size_t stage_slots = (size_t)b.n * b.ch;
    // The samples of one partition are staged in LDS ([sample][lane]: conflict-free) and the cascade runs
    // there: the loads of a partition are all in flight together, and the remainder a stage leaves for the
    // next never goes through global memory (the in-place updates of lib/res0.c:372-375 made every vector's
    // loads wait for the previous vector's stores).  res2's interleaved vector (lib/res0.c:781-787) is
    // formed by the load itself; nothing reads the remainder after the last stage, so it is not written back.
    int *stage = vq_lds + threadIdx.x;
    const int spp = v.spp, nb = v.nb, rbegin = r->begin;

    for (int i = i0; i < i1; i++) {
        const int offset = i * spp + rbegin;
        for (int j = 0; j < v.used; j++) {
            if (r->type == 2) {
                int l = offset / nb, k = offset - l * nb;      // work[x] = in[x % nb][x / nb]
                for (int e = 0; e < spp; e++) {
                    stage[e * 64] = IWC(v.chlist[k], l);
                    if (++k == nb) { k = 0; l++; }
                }
            } else {
                for (int e = 0; e < spp; e++) stage[e * 64] = IWC(v.vch[j], offset + e);
            }
            const int cls = PW(j, i);
            for (int st = 0; st < r->stages; st++) {
                int bits = 0;
                const int bi = (r->secondstages[cls] & (1 << st)) ? r->partbook[cls][st] : -1;
                if (bi >= 0) {
                    const book_regs bk = load_book(&s->book[bi]);
                    const book_regs *book = &bk;
                    const int dim = book->dim;
                    const int step = spp / dim;
                    uint64_t *sl = slot + (st * stage_slots + (size_t)j * veclen + offset) * 64;
                    for (int t = 0; t < step; t++) {
                        int a[VBM_MAX_BOOK_DIM];
                        int *src = stage + t * dim * 64;
                        for (int d = 0; d < dim; d++) a[d] = src[d * 64];
                        const int entry = besterror(book, a);
                        for (int d = 0; d < dim; d++) src[d * 64] = a[d];
                        uint64_t cw = 0;
                        if (entry >= 0 && entry < book->entries) {
                            const int len = book->lengthlist[entry];
                            cw = (uint64_t)book->codelist[entry] | ((uint64_t)(uint32_t)len << 32);
                            bits += len;
                        }
                        sl[(size_t)t * 64] = cw;
                    }
                }
                LEN(st, j, i) = bits;
            }
        }
    }
}

template <bool BLOBS>
__global__ void k_res_offsets(vbm_batch b, int sm)
{
    vbm_blob_enter<BLOBS>(b);
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_setup *s = b.setup;
    const vbm_map *info = &s->map[b.W];
    const size_t col0 = (size_t)sb * b.ch;
    const res_view v = residue_view(b, info, sm, col0);
    const int maxwords = b.max_packet_bytes / 4;
    int pos = b.packet_bits[sb];
    if (v.used) {
        const vbm_residue *r = v.r;
        const int *partword = b.partwordT + SBT(sb);
        const int *lenT = b.vqlenT + SBT(sb);
        int *offT = b.vqoffT + SBT(sb);
        uint32_t *words = (uint32_t *)b.packetT + (size_t)(sb >> 6) * maxwords * 64 + (sb & 63);
        const vbm_book *phrasebook = &s->book[r->groupbook];
        const int partitions_per_word = r->phrase_dim;
        // emission order of _01forward, lib/res0.c:574-636
        if (v.used == 1 && partitions_per_word == 2) {
            // one coded vector (res2) and two partitions per phrase word: eight words' classes, phrase codewords
            // and run lengths are read before any of their offsets / phrase bits is written (the loads would
            // queue behind the stores and atomics of the word before: loads and stores retire in order)
            const signed char *__restrict__ pll = phrasebook->lengthlist;
            const uint32_t *__restrict__ pcl = phrasebook->codelist;
            const int pents = phrasebook->entries, nparts = r->partitions, pv = v.partvals;
            for (int st = 0; st < r->stages; st++) {
                for (int i = 0; i < pv; i += 16) {
                    int plen[8], l0[8], l1[8];
                    uint32_t pcode[8];
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        const int ia = i + 2 * g, ib = ia + 1;
                        plen[g] = 0; pcode[g] = 0;
                        if (st == 0 && ia < pv) {
                            long val = (long)PW(0, ia) * nparts;
                            if (ib < pv) val += PW(0, ib);
                            if (val < pents) { plen[g] = pll[val]; pcode[g] = pcl[val]; }
                        }
                        l0[g] = (ia < pv) ? LEN(st, 0, ia) : 0;
                        l1[g] = (ib < pv) ? LEN(st, 0, ib) : 0;
                    }
#pragma unroll
                    for (int g = 0; g < 8; g++) {
                        const int ia = i + 2 * g, ib = ia + 1;
                        if (ia < pv) {
                            if (st == 0 && plen[g] > 0) {
                                or_bits(words, maxwords, pos, pcode[g], plen[g]);
                                pos += plen[g];
                            }
                            OFF(st, 0, ia) = pos;
                            pos += l0[g];
                            if (ib < pv) {
                                OFF(st, 0, ib) = pos;
                                pos += l1[g];
                            }
                        }
                    }
                }
            }
        } else
        for (int st = 0; st < r->stages; st++) {
            for (int i = 0; i < v.partvals;) {
                if (st == 0) {
                    for (int j = 0; j < v.used; j++) {
                        long val = PW(j, i);
                        for (int k = 1; k < partitions_per_word; k++) {
                            val *= r->partitions;
                            if (i + k < v.partvals) val += PW(j, i + k);
                        }
                        if (val < phrasebook->entries) {
                            const int len = phrasebook->lengthlist[val];
                            or_bits(words, maxwords, pos, phrasebook->codelist[val], len);
                            pos += len;
                        }
                    }
                }
                for (int k = 0; k < partitions_per_word && i < v.partvals; k++, i++)
                    for (int j = 0; j < v.used; j++) {
                        OFF(st, j, i) = pos;
                        pos += LEN(st, j, i);
                    }
            }
        }
        b.packet_bits[sb] = pos;
    }
    b.packet_bytes[sb] = (pos > maxwords * 32) ? -1 : (pos + 7) / 8;
}

template <bool BLOBS>
__global__ void k_res_emit(vbm_batch b, int sm, int nchunks)
{
    vbm_blob_enter<BLOBS>(b);
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_map *info = &b.setup->map[b.W];
    const size_t col0 = (size_t)sb * b.ch;
    const res_view v = residue_view(b, info, sm, col0);
    if (!v.used) return;
    const vbm_residue *r = v.r;
    const int i0 = (int)((long)v.partvals * blockIdx.y / nchunks), i1 = (int)((long)v.partvals * (blockIdx.y + 1) / nchunks);
    const int *lenT = b.vqlenT + SBT(sb);
    const int *offT = b.vqoffT + SBT(sb);
    const uint64_t *slot = b.vqcodeT + (size_t)(sb >> 6) * b.vq_slab_words + (sb & 63);
    const int maxwords = b.max_packet_bytes / 4;
    uint32_t *words = (uint32_t *)b.packetT + (size_t)(sb >> 6) * maxwords * 64 + (sb & 63);
    const int veclen = (r->type == 2) ? b.n * v.nb : b.n;
    const size_t stage_slots = (size_t)b.n * b.ch;

    for (int st = 0; st < r->stages; st++)
        for (int i = i0; i < i1; i++) {
            const int offset = i * v.spp + r->begin;
            for (int j = 0; j < v.used; j++) {
                int remaining = LEN(st, j, i);
                if (remaining <= 0) continue;
                const int pos = OFF(st, j, i);
                const uint64_t *sl = slot + (st * stage_slots + (size_t)j * veclen + offset) * 64;
                int wi = pos >> 5;
                int nbits = pos & 31;   // bits below the run's start stay zero in acc
                uint64_t acc = 0;
                // eight codeword slots are read before any of their bits go out (the loads would otherwise queue
                // behind the atomics of the slot before: loads and stores retire in order)
                for (int t0 = 0; t0 < v.spp && remaining > 0; t0 += 8) {
                    uint64_t cwv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cwv[u] = sl[(size_t)((t0 + u < v.spp) ? t0 + u : v.spp - 1) * 64];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (t0 + u < v.spp && remaining > 0) {
                            const uint64_t cw = cwv[u];
                            const int len = (int)(cw >> 32);
                            if (len) {
                                acc |= (uint64_t)(uint32_t)cw << nbits;
                                nbits += len;
                                remaining -= len;
                                if (nbits >= 32) {
                                    if (wi < maxwords) atomicOr(&words[(size_t)wi * 64], (uint32_t)acc);
                                    acc >>= 32;
                                    nbits -= 32;
                                    wi++;
                                }
                            }
                        }
                    }
                }
                if (nbits > 0 && wi < maxwords) atomicOr(&words[(size_t)wi * 64], (uint32_t)acc);
            }
        }
}


// ---------------------------------------------------------------------------------------------
// Fused packet assembly: ONE wavefront per stream-block does what k_pack_head + k_nonzero_propagate + k_res_vq +
// k_res_offsets + k_res_emit (+ the zero fill before and the layout change after) do for setups whose residue is ONE
// coded vector (stereo coupled res2, mono res1).  Nothing but the packet leaves the wavefront:
//   work    the vector as the residue coder reads it (k_couple_fast leaves it bin-interleaved in res_bm) in LDS, one pad
//           word per partition so that lane i walking partition i meets its own banks
//   floors  a lane per (channel, floor partition): codeword lengths first, a wave scan for the bit positions, then the
//           codewords are OR-ed into the packet words in LDS (floor1_encode's emission order, lib/floor1.c:856-942)
//   class   res*_class per partition, a lane each (lib/res0.c:406-526)
//   stages  _01forward (lib/res0.c:528-640): per stage every lane runs the cascade step of ITS partition
//           (local_book_besterror :316-378 on the remainder in LDS) and leaves (codeword, length) per vector in LDS; the
//           bit position of every partition's run is the scan of the run lengths in emission order (phrase codewords
//           in front of their group at stage 0); then each lane ORs its run in.  Stage by stage, because a stage's
//           positions start where the stage before ended.
//   out     the packet's own words to row sb of the packet buffer ([sb][max_packet_bytes]; nothing is cleared, nothing
//           is transposed: vbm_launch_packets_out copies length bytes and zero-fills the caller's row behind them)
#define PF_WAVE 64
__device__ __forceinline__ void lds_or_bits(uint32_t *words, const int maxwords, const int pos, uint32_t value, const int bits)
{
    if (bits <= 0) return;
    if (bits < 32) value &= (1u << bits) - 1u;
    const uint64_t v = (uint64_t)value << (pos & 31);
    const int wi = pos >> 5;
    if (wi < maxwords) atomicOr(&words[wi], (uint32_t)v);
    if ((v >> 32) && wi + 1 < maxwords) atomicOr(&words[wi + 1], (uint32_t)(v >> 32));
}

// inclusive prefix sum over the wavefront
__device__ __forceinline__ int wave_incl_scan(int v, const int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// One floor item of channel c: it = 0 the channel's lead-in (the "floor present" bit and the two end posts), it >= 1 floor
// partition it - 1 (class codeword + its posts' codewords, lib/floor1.c:873-942).  pos < 0: count bits only; else emit at pos.
__device__ int pf_floor_item(const vbm_batch &b, const vbm_setup *s, const vbm_floor *look, const size_t col, const int it,
                             uint32_t *pkt, const int maxwords, int pos)
{
    const size_t SW = b.slab_words;
#define OUTV(x) b.floor_outT[(size_t)(col >> 6) * SW + (size_t)(x) * 64 + (col & 63)]
    int bits = 0;
    auto put = [&](uint32_t code, int len) {
        if (len <= 0) return;
        if (pos >= 0) { lds_or_bits(pkt, maxwords, pos, code, len); pos += len; }
        bits += len;
    };
    if (it == 0) {
        const int qb = ilog(look->quant_q - 1);
        put(1u, 1);
        put((uint32_t)OUTV(0), qb);
        put((uint32_t)OUTV(1), qb);
        return bits;
    }
    const int i = it - 1;
    int j = 2;
    for (int t = 0; t < i; t++) j += look->class_dim[look->partitionclass[t]];
    const int cls = look->partitionclass[i];
    const int cdim = look->class_dim[cls];
    const int csubbits = look->class_subs[cls];
    const int csub = 1 << csubbits;
    int cval = 0, cshift = 0;
    // which subbook every post takes (:880-893), the class codeword, then the posts' own codewords
    for (int k = 0; k < cdim; k++) {
        const int ov = OUTV(j + k);
        int bk = 0;
        if (csubbits) {
            for (int l = 0; l < csub; l++) {
                const int booknum = look->class_subbook[cls][l];
                const int maxval = booknum < 0 ? 1 : s->book[booknum].entries;
                if (ov < maxval) { bk = l; break; }
            }
            cval |= bk << cshift;
            cshift += csubbits;
        }
    }
    if (csubbits) {
        const vbm_book *cb = &s->book[look->class_book[cls]];
        if (cval >= 0 && cval < cb->entries) put(cb->codelist[cval], cb->lengthlist[cval]);
    }
    cshift = 0;
    for (int k = 0; k < cdim; k++) {
        const int ov = OUTV(j + k);
        const int bk = csubbits ? (cval >> cshift) & (csub - 1) : 0;
        cshift += csubbits;
        const int book = look->class_subbook[cls][bk];
        if (book >= 0) {
            const vbm_book *sb_ = &s->book[book];
            if (ov >= 0 && ov < sb_->entries) put(sb_->codelist[ov], sb_->lengthlist[ov]);
        }
    }
#undef OUTV
    return bits;
}

__global__ __launch_bounds__(PF_WAVE) void k_pack_fused(vbm_batch b)
{
    extern __shared__ int pf_lds[];
    const int sb = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_setup *s = b.setup;
    const vbm_map *info = &s->map[b.W];
    const int ch = b.ch;
    const size_t col0 = (size_t)sb * ch;
    const vbm_residue *r = &s->residue[info->residuesubmap[0]];
    const int spp = r->grouping, pv = (r->end - r->begin) / spp, rbegin = r->begin;
    const int spad = spp + 1;
    const int maxwords = b.max_packet_bytes / 4;
    // LDS: work [pv * spad] | code [pv * spad] | pkt [maxwords] | len bytes [pv * spad] | class bytes [pv] | order bytes [pv]
    int *work = pf_lds;
    uint32_t *code = (uint32_t *)(work + pv * spad);
    uint32_t *pkt = code + pv * spad;
    uint8_t *len = (uint8_t *)(pkt + maxwords);
    uint8_t *pcls = len + pv * spad;

    for (int k = lane; k < maxwords; k += PF_WAVE) pkt[k] = 0u;
    // ---- the vector into LDS (k_couple_fast left it in the coder's order: work[x], x = bin * ch + channel)
    {
        const int *src = b.res_bm + (size_t)sb * b.n * ch + rbegin;
        // (x / spp without a division per element: spp is a power of two in every shipped setup)
        const int sh = (spp & (spp - 1)) == 0 ? __ffs(spp) - 1 : -1;
        if (sh >= 0)
            for (int x = lane; x < pv * spp; x += PF_WAVE) work[x + (x >> sh)] = src[x];
        else
            for (int x = lane; x < pv * spp; x += PF_WAVE) work[x + x / spp] = src[x];
    }
    // ---- nonzero[] after coupling (lib/psy.c:5133-5140)
    int nzbits = 0;
    for (int c = 0; c < ch; c++) nzbits |= (b.nonzero[col0 + c] ? 1 : 0) << c;
    for (int i = 0; i < info->coupling_steps; i++) {
        const int m = info->coupling_mag[i], a = info->coupling_ang[i];
        if (((nzbits >> m) | (nzbits >> a)) & 1) nzbits |= (1 << m) | (1 << a);
    }
    if (lane < ch) b.nonzero[col0 + lane] = (nzbits >> lane) & 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- packet type, mode number, window flags (lib/mapping0.c:1211-1218)
    int pos = 1 + s->modebits + (b.W ? 2 : 0);
    if (lane == 0) {
        uint32_t h = 0;                              // bit 0: packet type 0
        h |= (uint32_t)b.W << 1;
        if (b.W) h |= ((uint32_t)(b.wflags[sb] & 1) << (1 + s->modebits)) | ((uint32_t)((b.wflags[sb] >> 1) & 1) << (2 + s->modebits));
        lds_or_bits(pkt, maxwords, 0, h, pos);
    }
    // ---- floors, channel by channel (lib/floor1.c:856-942): a lane per item, lengths -> scan -> emit
    for (int c = 0; c < ch; c++) {
        const size_t col = col0 + c;
        const vbm_floor *look = &s->floor[info->floorsubmap[info->chmuxlist[c]]];
        if (!b.post_valid[col]) { pos += 1; continue; }            // a zero bit: nothing to OR
        const int items = 1 + look->partitions;                     // <= 32
        int bits = 0;
        if (lane < items) bits = pf_floor_item(b, s, look, col, lane, pkt, maxwords, -1);
        const int incl = wave_incl_scan(bits, lane);
        if (lane < items) (void)pf_floor_item(b, s, look, col, lane, pkt, maxwords, pos + incl - bits);
        pos += __shfl(incl, 63);
    }

    // ---- residue: the channels of the (single) submap as one vector
    const bool used = (r->type == 2) ? (nzbits != 0) : (nzbits & 1);
    if (used) {
        // classification (res*_class): lane i = partition i
        const int nclass = r->partitions;
        const int *__restrict__ classmetric1 = r->classmetric1, *__restrict__ classmetric2 = r->classmetric2;
        for (int i = lane; i < pv; i += PF_WAVE) {
            const int *w = work + i * spad;
            int k;
            if (r->type == 2) {
                // _2class (lib/res0.c:473-526): magnitude = the first channel's samples, angle = all the others
                int magmax = 0, angmax = 0;
                int cc = (rbegin + i * spp) % ch;                  // channel of the partition's first sample
                for (int e = 0; e < spp; e++) {
                    const int a = abs(w[e]);
                    if (cc == 0) { if (a > magmax) magmax = a; }
                    else if (a > angmax) angmax = a;
                    if (++cc == ch) cc = 0;
                }
                for (k = 0; k < nclass - 1; k++)
                    if (magmax <= classmetric1[k] && angmax <= classmetric2[k]) break;
            } else {
                // _01class (lib/res0.c:406-468)
                const float scale = (float)(100. / spp);
                int mx = 0, ent = 0;
                for (int e = 0; e < spp; e++) {
                    const int a = abs(w[e]);
                    if (a > mx) mx = a;
                    ent += a;
                }
                ent = (int)((float)ent * scale);
                for (k = 0; k < nclass - 1; k++)
                    if (mx <= classmetric1[k] && (classmetric2[k] < 0 || ent < classmetric2[k])) break;
            }
            pcls[i] = (uint8_t)k;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // partitions in class order (stable): order[base[k] + rank] = the rank-th partition of class k.  pv <= 64 here
        // (configure()): lane i = partition i.
        uint8_t *order = pcls + pv;
        const int mycls = lane < pv ? pcls[lane] : -1;
        {
            int basek = 0;
            for (int k = 0; k < nclass; k++) {
                const unsigned long long m = __ballot(mycls == k);
                if (mycls == k) order[basek + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                basek += __popcll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        const vbm_book *phrasebook = &s->book[r->groupbook];
        const int pd = r->phrase_dim;
        for (int st = 0; st < r->stages; st++) {
            // ---- the cascade step of every vector of the stage, class by class: all lanes of a pass share the book (its
            //      fields are wave-uniform: scalar loads), a lane takes one vector (partition order[..], vector t)
            {
                int basek = 0;
                for (int k = 0; k < nclass; k++) {
                    const int cnt = __popcll(__ballot(mycls == k));
                    const int bi = (r->secondstages[k] & (1 << st)) ? r->partbook[k][st] : -1;
                    if (bi >= 0 && cnt > 0) {
                        const book_regs bk = load_book(&s->book[bi]);
                        const int dim = bk.dim, nvec = spp / dim, items = cnt * nvec;
                        const int vsh = (nvec & (nvec - 1)) == 0 ? __ffs(nvec) - 1 : -1;     // (wave-uniform)
                        for (int id0 = 0; id0 < items; id0 += PF_WAVE) {        // (wave-uniform trip count)
                            const int id = id0 + lane;
                            const bool act = id < items;
                            const int rnk = vsh >= 0 ? (id >> vsh) : id / nvec, t = id - rnk * nvec;
                            const int i = act ? order[basek + rnk] : 0;
                            const int wo = i * spad + t * dim;                   // where the vector lives in `work`
                            int a[VBM_MAX_BOOK_DIM], pnt[VBM_MAX_BOOK_DIM];
                            int index = -1;
                            bool need = false;
                            if (act) {
                                for (int d = 0; d < dim; d++) a[d] = work[wo + d];
                                need = vq_lattice(&bk, a, index, pnt);
                            }
                            // vectors whose lattice point has no codeword: the nearest entry that has one, found by the whole wavefront
                            for (unsigned long long m = __ballot(need); m; m &= m - 1) {
                                const int src = __ffsll((long long)m) - 1;
                                const int so = __shfl(wo, src);
                                int av[VBM_MAX_BOOK_DIM];
                                for (int d = 0; d < dim; d++) av[d] = work[so + d];
                                const int bj = vq_search_wave(&bk, av, lane);
                                if (lane == src && bk.used > 0) {
                                    const int *pt = bk.used_point + (size_t)bj * dim;
                                    for (int d = 0; d < dim; d++) pnt[d] = pt[d];
                                    index = bk.used_index[bj];
                                }
                            }
                            if (act) {
                                if (index > -1)
                                    for (int d = 0; d < dim; d++) work[wo + d] = a[d] - pnt[d];
                                uint32_t c_ = 0;
                                int l_ = 0;
                                if (index >= 0 && index < bk.entries) { l_ = bk.lengthlist[index]; c_ = bk.codelist[index]; }
                                code[i * spad + t] = c_;
                                len[i * spad + t] = (uint8_t)l_;
                            }
                        }
                    }
                    basek += cnt;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // ---- lane i = partition i: its run's length, the positions in emission order ([phrase of the group] run(i)
            //      run(i + 1) ... per group, lib/res0.c:574-636), then the run is OR-ed into the packet
            {
                const int i = lane;
                const bool mine = i < pv;
                int runbits = 0, phlen = 0, nvec = 0;
                uint32_t phcode = 0;
                if (mine) {
                    if (st == 0 && (i % pd) == 0) {
                        long val = mycls;
                        for (int k = 1; k < pd; k++) {
                            val *= nclass;
                            if (i + k < pv) val += pcls[i + k];
                        }
                        if (val < phrasebook->entries) { phlen = phrasebook->lengthlist[val]; phcode = phrasebook->codelist[val]; }
                    }
                    const int bi = (r->secondstages[mycls] & (1 << st)) ? r->partbook[mycls][st] : -1;
                    if (bi >= 0) {
                        nvec = spp / s->book[bi].dim;
                        const uint8_t *cl = len + i * spad;
                        for (int t = 0; t < nvec; t++) runbits += cl[t];
                    }
                }
                const int mybits = runbits + phlen;
                const int incl = wave_incl_scan(mybits, lane);
                int at = pos + incl - mybits;
                if (mine) {
                    if (phlen > 0) { lds_or_bits(pkt, maxwords, at, phcode, phlen); at += phlen; }
                    const uint32_t *cw = code + i * spad;
                    const uint8_t *cl = len + i * spad;
                    // (a run goes in word by word through a 64-bit accumulator: one LDS atomic per 32 bits, not per codeword)
                    int wi = at >> 5, nb_ = at & 31;
                    uint64_t acc = 0;
                    for (int t = 0; t < nvec; t++) {
                        const int l_ = cl[t];
                        if (l_) {
                            acc |= (uint64_t)cw[t] << nb_;
                            nb_ += l_;
                            if (nb_ >= 32) {
                                if (wi < maxwords) atomicOr(&pkt[wi], (uint32_t)acc);
                                acc >>= 32;
                                nb_ -= 32;
                                wi++;
                            }
                        }
                    }
                    if (nb_ > 0 && wi < maxwords) atomicOr(&pkt[wi], (uint32_t)acc);
                }
                pos += __shfl(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- the packet's own words to its row
    const bool over = pos > maxwords * 32;
    const int bytes = over ? -1 : (pos + 7) / 8;
    if (lane == 0) { b.packet_bytes[sb] = bytes; b.packet_bits[sb] = pos; }
    if (!over) {
        uint32_t *row = (uint32_t *)(b.packetT + (size_t)sb * b.max_packet_bytes);
        const int nw = (bytes + 3) >> 2;
        for (int k = lane; k < nw; k += PF_WAVE) row[k] = pkt[k];
    }
}

// rows of a fused batch to the caller: length bytes copied, the rest of the caller's row zero (16 bytes per lane and step)
__global__ __launch_bounds__(256) void k_rows_out(vbm_batch b, uint8_t *__restrict__ dst, int *__restrict__ dst_bytes)
{
    const int sb = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (sb >= vbm_nsb(b)) return;
    const int bytes = b.packet_bytes[sb];
    if (dst_bytes && lane == 0) dst_bytes[sb] = bytes;
    if (!dst) return;
    const int maxb = b.max_packet_bytes;
    const uint4 *src = (const uint4 *)(b.packetT + (size_t)sb * maxb);
    uint4 *out = (uint4 *)(dst + (size_t)sb * maxb);
    const int live = bytes > 0 ? bytes : 0;
    for (int k = lane; k < maxb / 16; k += 64) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int at = k * 16;
        if (at < live) {
            v = src[k];
            const int keep = live - at;                // bytes of this piece inside the packet
            if (keep < 16) {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
                for (int q = 0; q < 4; q++) {
                    const int kq = keep - 4 * q;
                    if (kq <= 0) w[q] = 0u;
                    else if (kq < 4) w[q] &= (1u << (8 * kq)) - 1u;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        out[k] = v;
    }
}

// res_bm [sb][bin * ch + c] -> rows [sb * ch + c][bin]
__global__ void k_res_bm_rows(vbm_batch b, int *__restrict__ dst)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)vbm_nsb(b) * b.n * b.ch;
    if (idx >= total) return;
    const size_t per = (size_t)b.n * b.ch;
    const size_t sb = idx / per, x = idx - sb * per;
    const int bin = (int)(x / b.ch), c = (int)(x - (size_t)bin * b.ch);
    dst[(sb * b.ch + c) * b.n + bin] = b.res_bm[idx];
}

}  // namespace

extern "C" int vbm_launch_block_state(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_block_state, dim3((unsigned)((b->nsb + 63) / 64)), dim3(64), 0, st, *b, 1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_block_state_managed(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_block_state, dim3((unsigned)((b->nsb + 63) / 64)), dim3(64), 0, st, *b, VBM_PACKETBLOBS);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// after all PACKETBLOBS packets of the batch exist: choice + final lengths into b->choice / b->packet_bytes,
// the chosen packets into d_packets [nsb][max_packet_bytes] (may be NULL)
extern "C" int vbm_launch_bitrate_choose(const vbm_batch *b, uint8_t *d_packets, hipStream_t st)
{
    hipLaunchKernelGGL(k_bitrate_choose, dim3((unsigned)((b->nsb + 63) / 64)), dim3(64), 0, st, *b);
    if (d_packets) {
        const int rows = b->max_packet_bytes / 4;
        hipLaunchKernelGGL(k_blob_gather, dim3((unsigned)((b->nsb + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0, st,
                           *b, (uint32_t *)d_packets);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the packet tiles start out as zeros (the bit runs are OR-ed in); a kernel of our own rather than hipMemsetAsync, so
// that the launch sequence of a round is kernels only (it is replayed as a HIP graph, capi_encoder.cpp)
static __global__ void k_zero_u128(uint4 *__restrict__ p, size_t n, size_t blob_stride)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[(size_t)blockIdx.z * blob_stride + i] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int vbm_launch_pack_head(const vbm_batch *b, hipStream_t st)
{
    const unsigned tiles = (unsigned)((b->nsb + 63) / 64);
    const unsigned nbl = (unsigned)(b->nblobs > 1 ? b->nblobs : 1);      // managed bitrate: a packetblob per blockIdx.z
    {
        const size_t n16 = (size_t)tiles * 64 * b->max_packet_bytes / 16;   // max_packet_bytes is a multiple of 4: 64 * 4 = 256 B per row group
        hipLaunchKernelGGL(k_zero_u128, dim3((unsigned)((n16 + 255) / 256), 1, nbl), dim3(256), 0, st, (uint4 *)b->packetT, n16,
                           (size_t)b->Ls * b->max_packet_bytes / 16);
    }
    if (nbl > 1) hipLaunchKernelGGL(k_pack_head<true>, dim3(tiles, 1, nbl), dim3(64), 0, st, *b);
    else hipLaunchKernelGGL(k_pack_head<false>, dim3(tiles), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_pack_residue(const vbm_batch *b, hipStream_t st)
{
    const unsigned tiles = (unsigned)((b->nsb + 63) / 64);
    const unsigned nbl = (unsigned)(b->nblobs > 1 ? b->nblobs : 1);
    if (nbl > 1) hipLaunchKernelGGL(k_nonzero_propagate<true>, dim3(tiles, 1, nbl), dim3(64), 0, st, *b);
    else hipLaunchKernelGGL(k_nonzero_propagate<false>, dim3(tiles), dim3(64), 0, st, *b);
    for (int sm = 0; sm < b->pack_submaps; sm++) {
        // (a small batch is bound by the length of a slice's walk: one partition per slice there)
        const int most = (b->few || b->nsb <= 1024) ? 256 : 32;
        int nchunks = b->pack_partvals[sm] < most ? b->pack_partvals[sm] : most;
        if (nchunks < 1) nchunks = 1;
        const size_t lds = (size_t)b->pack_spp[sm] * 64 * sizeof(int);
        if (nbl > 1) {
            hipLaunchKernelGGL(k_res_vq<true>, dim3(tiles, (unsigned)nchunks, nbl), dim3(64), lds, st, *b, sm, nchunks);
            hipLaunchKernelGGL(k_res_offsets<true>, dim3(tiles, 1, nbl), dim3(64), 0, st, *b, sm);
            hipLaunchKernelGGL(k_res_emit<true>, dim3(tiles, (unsigned)nchunks, nbl), dim3(64), 0, st, *b, sm, nchunks);
        } else {
            hipLaunchKernelGGL(k_res_vq<false>, dim3(tiles, (unsigned)nchunks), dim3(64), lds, st, *b, sm, nchunks);
            hipLaunchKernelGGL(k_res_offsets<false>, dim3(tiles), dim3(64), 0, st, *b, sm);
            hipLaunchKernelGGL(k_res_emit<false>, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, sm, nchunks);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_pack(const vbm_batch *b, hipStream_t st)
{
    if (b->pack_fused) {
        const vbm_setup *hs = nullptr; (void)hs;
        // LDS: work + code (4 B each per padded slot), the packet words, a length byte per slot, a class byte per partition
        const int spp = b->pack_spp[0], pv = b->pack_partvals[0];
        const size_t slots = (size_t)pv * (spp + 1);
        size_t lds = slots * 8 + (size_t)b->max_packet_bytes + slots + 2 * (size_t)pv;
        lds = (lds + 15) & ~(size_t)15;
        static std::mutex mu;
        static size_t allowed = 48 * 1024;
        {
            std::lock_guard<std::mutex> guard(mu);
            if (lds > allowed) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_pack_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess) return -2;
                allowed = lds;
            }
        }
        hipLaunchKernelGGL(k_pack_fused, dim3((unsigned)b->nsb), dim3(PF_WAVE), lds, st, *b);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    int rc = vbm_launch_pack_head(b, st);
    return rc ? rc : vbm_launch_pack_residue(b, st);
}

extern "C" int vbm_launch_packets_out(const vbm_batch *b, uint8_t *dst, int *dst_bytes, hipStream_t st)
{
    if (!dst && !dst_bytes) return 0;
    if (b->pack_fused) {
        hipLaunchKernelGGL(k_rows_out, dim3((unsigned)((b->nsb + 3) / 4)), dim3(256), 0, st, *b, dst, dst_bytes);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    int rc = 0;
    if (dst) rc = vbm_launch_untranspose_counted((const int *)b->packetT, (int *)dst, b->max_packet_bytes / 4,
                                                 (size_t)(b->max_packet_bytes / 4) * 64, b->nsb, b->d_nsb, st);
    if (!rc && dst_bytes) rc = vbm_launch_copy_counted(dst_bytes, b->packet_bytes, b->nsb, b->d_nsb, st);
    return rc;
}

extern "C" int vbm_launch_res_bm_rows(const vbm_batch *b, int *dst, hipStream_t st)
{
    const size_t total = (size_t)b->nsb * b->n * b->ch;
    if (!total) return 0;
    hipLaunchKernelGGL(k_res_bm_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *b, dst);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
