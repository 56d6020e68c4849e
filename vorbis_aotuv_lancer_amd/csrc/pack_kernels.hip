// Packet assembly for gfx950 — one lane per stream-block: header bits, floor-1 entropy coding,
// residue classification, cascaded lattice-VQ encode and the aoTuV block-state update.
//
//   k_pack   mapping0_forward loop C (reference lib/mapping0.c:1204-1313) for the VBR blob:
//            packet type / mode / window bits :1211-1218; floor1_encode's bit emission
//            (lib/floor1.c:856-942); res*_class (_01class lib/res0.c:406-468, _2class :473-526);
//            res*_forward -> _01forward :528-640 -> _encodepart :384-404 ->
//            local_book_besterror :316-378 -> vorbis_book_encode (lib/codebook.c:402-410);
//            block-state update :1297-1305.
// Bits are appended LSb first (libogg oggpack semantics) into byte-major packet buffers
// packetT[byte][Ls]; packet_bytes[sb] = oggpack_bytes().  The nearest-codeword search walks
// the compact list of used entries (ascending entry order, so the reference's lowest-index
// tie rule holds) instead of stepping the lattice odometer through unused entries.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "kernels.h"

namespace {

struct BitW {
    uint8_t *base;     // &packetT[lane]
    size_t stride;     // 64 (tiled layout)
    int nbytes;
    int maxbytes;
    uint64_t acc;
    int nbits;
    int overflow;
};

__device__ __forceinline__ void bw_write(BitW &w, uint32_t value, int bits)
{
    if (bits <= 0) return;
    if (bits < 32) value &= (1u << bits) - 1u;
    w.acc |= (uint64_t)value << w.nbits;
    w.nbits += bits;
    while (w.nbits >= 8) {
        if (w.nbytes < w.maxbytes) w.base[(size_t)w.nbytes * w.stride] = (uint8_t)(w.acc & 0xff);
        else w.overflow = 1;
        w.acc >>= 8;
        w.nbits -= 8;
        w.nbytes++;
    }
}

__device__ __forceinline__ int bw_finish(BitW &w)
{
    int total = w.nbytes + (w.nbits + 7) / 8;
    if (w.nbits > 0) {
        if (w.nbytes < w.maxbytes) w.base[(size_t)w.nbytes * w.stride] = (uint8_t)(w.acc & 0xff);
        else w.overflow = 1;
    }
    return total;
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

// local_book_besterror (lib/res0.c:316-378); a[] is the vector (dim <= 8) in registers
__device__ __forceinline__ int besterror(const vbm_book *book, int *a)
{
    const int dim = book->dim;
    int i, j, o;
    const int minval = book->minval, del = book->delta, qv = book->quantvals;
    const int ze = (qv >> 1);
    int index = 0;
    int p[VBM_MAX_BOOK_DIM] = {0, 0, 0, 0, 0, 0, 0, 0};

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
        const int *pt = book->used_point;
        for (i = 0; i < book->used; i++, pt += dim) {
            int dist = 0;
            for (j = 0; j < dim; j++) {
                int val = pt[j] - a[j];
                dist += val * val;
            }
            if (best == -1 || dist < best) {
                for (j = 0; j < dim; j++) p[j] = pt[j];
                best = dist;
                index = book->used_index[i];
            }
        }
    }

    if (index > -1)
        for (i = 0; i < dim; i++) a[i] -= p[i];
    return index;
}

// _01forward over `nvec` channel vectors; vec[j] + x*stride[j] addresses sample x of vector j
__device__ __forceinline__ void forward01(BitW &w, const vbm_setup *s, const vbm_residue *info, int *const *vec, const size_t *stride,
                          int nvec, const int *partword, size_t pw_stride, int pw_rows)
{
#define PW(j, i) partword[((size_t)(j) * pw_rows + (i)) * pw_stride]
    const int samples_per_partition = info->grouping;
    const int possible_partitions = info->partitions;
    const int partitions_per_word = info->phrase_dim;
    const int n = info->end - info->begin;
    const int partvals = n / samples_per_partition;
    const vbm_book *phrasebook = &s->book[info->groupbook];
    int i, j, k, st;

    for (st = 0; st < info->stages; st++) {
        for (i = 0; i < partvals;) {
            if (st == 0) {
                for (j = 0; j < nvec; j++) {
                    long val = PW(j, i);
                    for (k = 1; k < partitions_per_word; k++) {
                        val *= possible_partitions;
                        if (i + k < partvals) val += PW(j, i + k);
                    }
                    if (val < phrasebook->entries) book_encode(phrasebook, (int)val, w);
                }
            }

            for (k = 0; k < partitions_per_word && i < partvals; k++, i++) {
                long offset = (long)i * samples_per_partition + info->begin;
                for (j = 0; j < nvec; j++) {
                    int cls = PW(j, i);
                    if (info->secondstages[cls] & (1 << st)) {
                        int bi = info->partbook[cls][st];
                        if (bi >= 0) {
                            const vbm_book *book = &s->book[bi];
                            const int dim = book->dim;
                            const int step = samples_per_partition / dim;
                            int *base = vec[j] + (size_t)offset * stride[j];
                            for (int t = 0; t < step; t++) {
                                int a[VBM_MAX_BOOK_DIM];
                                int *vp = base + (size_t)t * dim * stride[j];
                                for (int d = 0; d < dim; d++) a[d] = vp[(size_t)d * stride[j]];
                                int entry = besterror(book, a);
                                for (int d = 0; d < dim; d++) vp[(size_t)d * stride[j]] = a[d];
                                book_encode(book, entry, w);
                            }
                        }
                    }
                }
            }
        }
    }
#undef PW
}

__global__ void k_pack(vbm_batch b)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= b.nsb) return;
    const size_t SW = b.slab_words, SWS = b.sb_slab_words;
    const size_t sbt = (size_t)(sb >> 6) * SWS + (sb & 63);   // this lane in the stream-block slab
    const vbm_setup *s = b.setup;
    const vbm_map *info = &s->map[b.W];
    const int ch = b.ch;
    const int n = b.n;
    const size_t col0 = (size_t)sb * ch;
    const int sid = b.stream_id[sb];
    int i, j, k;

    BitW w;
    w.base = b.packetT + (size_t)(sb >> 6) * b.max_packet_bytes * 64 + (sb & 63);
    w.stride = 64;
    w.nbytes = 0;
    w.maxbytes = b.max_packet_bytes;
    w.acc = 0;
    w.nbits = 0;
    w.overflow = 0;

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

            if (csubbits) {
                int maxval[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (k = 0; k < csub; k++) {
                    int booknum = look->class_subbook[cls][k];
                    if (booknum < 0) maxval[k] = 1;
                    else maxval[k] = s->book[booknum].entries;
                }
                for (k = 0; k < cdim; k++) {
                    for (l = 0; l < csub; l++) {
                        int val = OUTV(j + k);
                        if (val < maxval[l]) {
                            bookas[k] = l;
                            break;
                        }
                    }
                    cval |= bookas[k] << cshift;
                    cshift += csubbits;
                }
                book_encode(&s->book[look->class_book[cls]], cval, w);
            }

            for (k = 0; k < cdim; k++) {
                int book = look->class_subbook[cls][bookas[k]];
                if (book >= 0) {
                    int val = OUTV(j + k);
                    if (val < s->book[book].entries) book_encode(&s->book[book], val, w);
                }
            }
            j += cdim;
        }
#undef OUTV
    }

    // ---- residue, submap by submap (lib/mapping0.c:1273-1295) -------------------------------
    for (int sm = 0; sm < info->submaps; sm++) {
        const vbm_residue *r = &s->residue[info->residuesubmap[sm]];
        int chlist[VBM_MAXCH], zb[VBM_MAXCH];
        int nb = 0;
        for (j = 0; j < ch; j++)
            if (info->chmuxlist[j] == sm) {
                zb[nb] = b.nonzero[col0 + j] ? 1 : 0;
                chlist[nb++] = j;
            }
        const int samples_per_partition = r->grouping;
        const int possible_partitions = r->partitions;
        const int rn = r->end - r->begin;
        const int partvals = rn / samples_per_partition;
        // partword rows: [vector j][partition i], all submaps share the buffer (used one at a time)
        int *partword = b.partwordT + sbt;
        const size_t pw_stride = 64;
        const int pw_rows = partvals;
#define PW(jv, iv) partword[((size_t)(jv) * pw_rows + (iv)) * pw_stride]
#define IWC(cc, x) b.iworkT[(size_t)((col0 + (cc)) >> 6) * SW + (size_t)(x) * 64 + ((col0 + (cc)) & 63)]

        if (r->type == 2) {
            int used = 0;
            for (j = 0; j < nb; j++)
                if (zb[j]) used++;
            if (!used) continue;
            // _2class (lib/res0.c:473-526)
            {
                int l = r->begin / nb;
                for (i = 0; i < partvals; i++) {
                    int magmax = 0, angmax = 0;
                    for (j = 0; j < samples_per_partition; j += nb) {
                        int v0 = abs(IWC(chlist[0], l));
                        if (v0 > magmax) magmax = v0;
                        for (k = 1; k < nb; k++) {
                            int vk = abs(IWC(chlist[k], l));
                            if (vk > angmax) angmax = vk;
                        }
                        l++;
                    }
                    for (j = 0; j < possible_partitions - 1; j++)
                        if (magmax <= r->classmetric1[j] && angmax <= r->classmetric2[j]) break;
                    PW(0, i) = j;
                }
            }
            // res2_forward: interleave into one vector (lib/res0.c:781-787), then _01forward
            int *work = b.workvqT + sbt;
            for (i = 0; i < nb; i++)
                for (j = 0, k = i; j < n; j++, k += nb) work[(size_t)k * 64] = IWC(chlist[i], j);
            int *vec[1] = {work};
            size_t stride[1] = {64};
            forward01(w, s, r, vec, stride, 1, partword, pw_stride, pw_rows);
        } else {
            // res1_class / res1_forward: only the nonzero channels take part (lib/res0.c:715-745)
            int *vec[VBM_MAXCH];
            size_t stride[VBM_MAXCH];
            int used = 0;
            for (j = 0; j < nb; j++)
                if (zb[j]) {
                    vec[used] = &IWC(chlist[j], 0);
                    stride[used] = 64;
                    used++;
                }
            if (!used) continue;
            // _01class (lib/res0.c:406-468)
            {
                float scale = (float)(100. / samples_per_partition);
                for (i = 0; i < partvals; i++) {
                    int offset = i * samples_per_partition + r->begin;
                    for (j = 0; j < used; j++) {
                        int mx = 0, ent = 0;
                        for (k = 0; k < samples_per_partition; k++) {
                            int v = abs(vec[j][(size_t)(offset + k) * stride[j]]);
                            if (v > mx) mx = v;
                            ent += v;
                        }
                        ent = (int)((float)ent * scale);
                        for (k = 0; k < possible_partitions - 1; k++)
                            if (mx <= r->classmetric1[k] && (r->classmetric2[k] < 0 || ent < r->classmetric2[k])) break;
                        PW(j, i) = k;
                    }
                }
            }
            forward01(w, s, r, vec, stride, used, partword, pw_stride, pw_rows);
        }
#undef PW
#undef IWC
    }

    b.packet_bytes[sb] = w.overflow ? -1 : bw_finish(w);

    // ---- aoTuV block-state update (lib/mapping0.c:1297-1305) --------------------------------
    {
        const int block_mode = b.block_mode;
        int impadnum = b.st.impadnum[sid];
        int lWbm = b.st.lW_block_mode[sid];
        int lW_no = b.st.lW_no[sid];
        if (block_mode >= 2) impadnum = 0;
        if ((!lWbm) && (block_mode == 1)) impadnum = 1;
        else if (impadnum && impadnum < 8) impadnum++;
        if (lWbm == block_mode) lW_no++;
        else lW_no = 1;
        b.st.impadnum[sid] = impadnum;
        b.st.lW_no[sid] = lW_no;
        b.st.lW_block_mode[sid] = block_mode;
    }
}

}  // namespace

extern "C" int vbm_launch_pack(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((b->nsb + 63) / 64)), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
