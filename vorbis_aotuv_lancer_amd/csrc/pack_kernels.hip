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
__global__ void k_nonzero_propagate(vbm_batch b)
{
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
__global__ void k_pack_head(vbm_batch b)
{
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

__global__ void k_res_vq(vbm_batch b, int sm, int nchunks)
{
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

__global__ void k_res_offsets(vbm_batch b, int sm)
{
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

__global__ void k_res_emit(vbm_batch b, int sm, int nchunks)
{
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
static __global__ void k_zero_u128(uint4 *__restrict__ p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int vbm_launch_pack_head(const vbm_batch *b, hipStream_t st)
{
    const unsigned tiles = (unsigned)((b->nsb + 63) / 64);
    {
        const size_t n16 = (size_t)tiles * 64 * b->max_packet_bytes / 16;   // max_packet_bytes is a multiple of 4: 64 * 4 = 256 B per row group
        hipLaunchKernelGGL(k_zero_u128, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (uint4 *)b->packetT, n16);
    }
    hipLaunchKernelGGL(k_pack_head, dim3(tiles), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_pack_residue(const vbm_batch *b, hipStream_t st)
{
    const unsigned tiles = (unsigned)((b->nsb + 63) / 64);
    hipLaunchKernelGGL(k_nonzero_propagate, dim3(tiles), dim3(64), 0, st, *b);
    for (int sm = 0; sm < b->pack_submaps; sm++) {
        // (a small batch is bound by the length of a slice's walk: one partition per slice there)
        const int most = (b->few || b->nsb <= 1024) ? 256 : 32;
        int nchunks = b->pack_partvals[sm] < most ? b->pack_partvals[sm] : most;
        if (nchunks < 1) nchunks = 1;
        hipLaunchKernelGGL(k_res_vq, dim3(tiles, (unsigned)nchunks), dim3(64), (size_t)b->pack_spp[sm] * 64 * sizeof(int), st, *b,
                           sm, nchunks);
        hipLaunchKernelGGL(k_res_offsets, dim3(tiles), dim3(64), 0, st, *b, sm);
        hipLaunchKernelGGL(k_res_emit, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, sm, nchunks);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_pack(const vbm_batch *b, hipStream_t st)
{
    int rc = vbm_launch_pack_head(b, st);
    return rc ? rc : vbm_launch_pack_residue(b, st);
}
