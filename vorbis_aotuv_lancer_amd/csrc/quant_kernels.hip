// Channel coupling + quantisation + noise normalisation for gfx950 — one lane per stream-block.
//
//   k_couple_quantize   _vp_couple_quantize_normalize (reference lib/psy.c:4858-5142) with
//                       flag_lossless :4584-4624, noise_normalize :4732-4854 (ssort :4709),
//                       lossless_coupling(f) :4626-4658, min_indemnity_dipole_hypot :4660-4673,
//                       blob b.blobno (PACKETBLOBS/2 for VBR).
// A partition only depends on its predecessor through aoTuV M6's `side_resdef` (:5032-5034): the
// previous partition's mean magnitude/angle residue difference of the coupled pair.  When the
// coupling steps of the mapping use disjoint channels (stereo: one step) that number depends
// on nothing but the previous partition's mdct / floor values, so
//   k_couple_m6stats   computes it for every partition in the M6 range (one partition per
//                      blockIdx.y) into m6defT, and
//   k_couple_quantize  then runs the partitions sliced over blockIdx.y, reading side_resdef from
//                      the table.
// Mappings whose steps share channels keep the carried value and run as one slice.
// `residue_def` is an order-bound float sum inside a partition; the per-partition scratch
// (raw/quant/floor/res/flag, ch x 32 bins) lives in private memory.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "kernels.h"

#define PART_MAX 32
#define existe(x, y) (x < -y || x >= y)
#define refer_phase(a, b) ((a > 0. && b < 0.) || (b > 0. && a < 0.))
#define VMIN(x, y) ((x) > (y) ? (y) : (x))

namespace {

// Per bin of one partition (start bin i, jn bins): res = mdct / floor and the coupling class of the bin — 1 lossless,
// -1 lossy only if the partner agrees, 0 lossy (what lib/psy.c:4584-4624 decides).  Two magnitude limits apply: below
// the setup's point limit the pre-point pair, above it the post-point pair; the one partition that straddles the limit
// ramps from one pair to the other in jn equal steps.  The ramp is a RUNNING float sum in the source (its rounding is
// part of the result), so it is kept as one here.  The first limit is lowered by the bin's tone peak, never below the
// pre-point value.
__device__ __forceinline__ void flag_lossless(int limit, float prepoint, float postpoint, float prepoint_r, float postpoint_r,
                              float *res, const float *mdct, const float *enpeak, const float *floor, int *flag,
                              int i, int jn)
{
    const int room = limit - i;                         // bins from the partition's start up to the point limit
    const bool below = room > 0, ramp = below && room <= jn;
    float lim1 = below ? prepoint : postpoint, lim2 = below ? prepoint_r : postpoint_r;
    const float step1 = ramp ? (postpoint - prepoint) / jn : 0.f, step2 = ramp ? (postpoint_r - prepoint_r) / jn : 0.f;
    for (int j = 0; j < jn; j++) {
        if (ramp) { lim1 += step1; lim2 += step2; }
        const float q = mdct[j] / floor[j];
        res[j] = q;
        const float mag = fabsf(q);
        float lowered = lim1 - enpeak[j];
        if (lowered < prepoint) lowered = prepoint;
        // (comparisons as the source orders them: an unordered magnitude counts as lossless)
        flag[j] = !(mag < lowered) ? 1 : (mag < lim2 ? 0 : -1);
    }
}

__device__ __forceinline__ void lossless_coupling(int *Mag, int *Ang)
{
    int A = *Mag;
    int B = *Ang;
    if (abs(A) > abs(B)) {
        *Ang = (A > 0 ? A - B : B - A);
    } else {
        *Ang = (B > 0 ? A - B : B - A);
        *Mag = B;
    }
    if (*Ang >= abs(*Mag) * 2) {
        *Ang = -*Ang;
        *Mag = -*Mag;
    }
}

__device__ __forceinline__ void lossless_couplingf(float *Mag, float *Ang)
{
    float A = *Mag;
    float B = *Ang;
    if (fabsf(A) > fabsf(B)) {
        *Ang = (A > 0 ? A - B : B - A);
    } else {
        *Ang = (B > 0 ? A - B : B - A);
        *Mag = B;
    }
    if ((double)*Ang >= fabs((double)*Mag) * 2) {
        *Ang = -*Ang;
        *Mag = -*Mag;
    }
}

__device__ __forceinline__ float min_indemnity_dipole_hypot(const float a, const float b, const float threv)
{
    const float thnor = (float)0.94;
    float a2 = fabsf(a * thnor);
    float b2 = fabsf(b * thnor);
    if (a > 0.) {
        if (b > 0.) return (a2 + b2);
        if (a > -b) return (a2 - b2 * threv);
        return -(b2 - a2 * threv);
    }
    if (b < 0.) return -(a2 + b2);
    if (-a > b) return -(a2 - b2 * threv);
    return (b2 - a2 * threv);
}

// q / out / r / res / f / flags are this channel's partition scratch; out is strided (bin-major)
// loop-invariant psy fields of noise_normalize, read once per kernel (the compiler cannot hoist
// setup loads over the kernel's global stores)
struct nn_consts {
    int normal_p, normal_start;
    double normal_thresh;
};

__device__ __forceinline__ float noise_normalize(const nn_consts *p, const int limit, float *r, float *q, const float *f, float *res,
                                 const int *flags, float acc, const float nepeak, const int i, const int n,
                                 int *out, const size_t ostride)
{
#define OUT(j) out[(size_t)(j) * ostride]
    int sort[PART_MAX];
    int j, k, count = 0;
    int start = (p->normal_p ? p->normal_start - i : n);
    if ((start > n) || ((double)nepeak < -0.5)) start = n;

    acc = 0.f;

    j = 0;
    if (!flags) {
        for (; j < start; j++) OUT(j) = (int)rint((double)res[j]);
    } else {
        for (; j < start; j++) {
            if (flags[j] != 1) {
                float ve = (float)sqrt((double)(q[j] / f[j]));
                if (r[j] < 0) {
                    OUT(j) = (int)-rint((double)ve);
                    res[j] = -ve;
                } else {
                    OUT(j) = (int)rint((double)ve);
                    res[j] = ve;
                }
            }
        }
    }

    if (flags) {
        for (; j < n; j++) {
            float ve;
            if (flags[j] != 1) ve = q[j] / f[j];
            else continue;
            if (ve < .25f && j >= limit - i) {
                acc += ve;
                sort[count++] = j;
                if (r[j] < 0) res[j] = (float)-sqrt((double)ve);
                else res[j] = (float)sqrt((double)ve);
            } else {
                ve = (float)sqrt((double)ve);
                int o;
                if (r[j] < 0) {
                    o = (int)-rint((double)ve);
                    res[j] = -ve;
                } else {
                    o = (int)rint((double)ve);
                    res[j] = ve;
                }
                OUT(j) = o;
                q[j] = o * o * f[j];
            }
        }
    } else {
        for (; j < n; j++) {
            float ve = res[j] * res[j];
            if (ve < .25f) {
                acc += ve;
                sort[count++] = j;
            } else {
                int o = (int)rint((double)res[j]);
                OUT(j) = o;
                q[j] = o * o * f[j];
            }
        }
    }

    acc += acc * nepeak * nepeak;

    if (count) {
        int iacc = ((int)acc) + 1;
        if (iacc > n) iacc = n;
        // ssort: partial selection sort, largest q first (lib/psy.c:4709-4726)
        {
            int bthresh = iacc;
            if (count < bthresh) bthresh = count;
            for (int a = 0; a < bthresh; a++) {
                int large = a;
                for (int bb = a + 1; bb < count; bb++)
                    if (q[sort[large]] < q[sort[bb]]) large = bb;
                int tmp = sort[a];
                sort[a] = sort[large];
                sort[large] = tmp;
            }
        }
        for (k = 0; k < count; k++) {
            int e = sort[k];
            if ((double)acc >= p->normal_thresh) {
                OUT(e) = (int)vbm_unitnorm(r[e]);
                acc -= 1.f;
                q[e] = f[e];
            } else {
                OUT(e) = 0;
                q[e] = 0.f;
            }
        }
    }
    return acc;
#undef OUT
}

// M6 statistics of one partition, lib/psy.c:5010-5034 (the part that does not depend on side_resdef)
template <bool BLOBS>
__global__ void k_couple_m6stats(vbm_batch b)
{
    vbm_blob_enter<BLOBS>(b);
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const size_t SW = b.slab_words;
    const vbm_setup *s = b.setup;
    const vbm_psy *p = &s->psy[b.block_mode];
    const vbm_map *vi = &s->map[b.W];
    const int n = p->n;
    const int partition = (p->normal_p ? p->normal_partition : 16);
    const int pi = blockIdx.y, i = pi * partition;
    const int jn = partition > n - i ? n - i : partition;
    const size_t col0 = (size_t)sb * b.ch;
    float *m6 = b.m6defT + (size_t)(sb >> 6) * b.sb_slab_words + (sb & 63);
#define CT(buf, k, x) (buf)[(size_t)((col0 + (k)) >> 6) * SW + (size_t)(x) * 64 + ((col0 + (k)) & 63)]
    for (int step = 0; step < vi->coupling_steps; step++) {
        const int Mi = vi->coupling_mag[step], Ai = vi->coupling_ang[step];
        const int nzM = b.nonzero[col0 + Mi], nzA = b.nonzero[col0 + Ai];
        float def = -1.f;
        if (nzM || nzA) {
            int rp = 0, pp = 0;
            float residue_def = 0;
            for (int j = 0; j < jn; j++) {
                float resM = 0.f, resA = 0.f, reM = 0.f, reA = 0.f;
                if (nzM) {
                    const float m = CT(b.mdctT, Mi, i + j);
                    resM = m / s->fromdB[CT(b.iworkT, Mi, i + j)];
                    reM = m * m;
                    if (m < 0.f) reM *= -1.f;
                }
                if (nzA) {
                    const float m = CT(b.mdctT, Ai, i + j);
                    resA = m / s->fromdB[CT(b.iworkT, Ai, i + j)];
                    reA = m * m;
                    if (m < 0.f) reA *= -1.f;
                }
                if (existe(resM, 0.5) || existe(resA, 0.5)) {
                    if (refer_phase(reM, reA)) rp++;
                    else pp++;
                    residue_def = (float)((double)residue_def + fabs((double)fabsf(resM) - (double)fabsf(resA)));
                }
            }
            const int ap = rp + pp;
            if (ap != 0) def = residue_def / ap;
        }
        m6[((size_t)pi * vi->coupling_steps + step) * 64] = def;
    }
#undef CT
}

__global__ void k_couple_quantize(vbm_batch b, int nchunks)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const size_t SW = b.slab_words;
    const vbm_setup *s = b.setup;
    const vbm_psy *p = &s->psy[b.block_mode];
    const vbm_map *vi = &s->map[b.W];
    const int ch = b.ch;
    const int blobno = b.blobno;
    const int n = p->n;
    const int partition = (p->normal_p ? p->normal_partition : 16);
    const int limit = s->coupling_pointlimit[p->blockflag][blobno];
    float prepoint = (float)s->stereo_threshholds[s->coupling_prepointamp[blobno]];
    float postpoint = (float)s->stereo_threshholds[s->coupling_postpointamp[blobno]];
    float prepoint_x = (float)s->stereo_threshholds_X[s->coupling_prepointamp[blobno]];
    float postpoint_x = (float)s->stereo_threshholds_X[s->coupling_postpointamp[blobno]];
    float prae;
    const int sliding_lowpass = s->sliding_lowpass[b.W][blobno];
    nn_consts nn;
    nn.normal_p = p->normal_p; nn.normal_start = p->normal_start; nn.normal_thresh = p->normal_thresh;
    const float *__restrict__ fromdB = s->fromdB;
    const int tonefix_end = p->tonefix_end;
    const int coupling_steps = vi->coupling_steps;
    int lowpassr;
    {
        // lib/mapping0.c:778-781
        lowpassr = s->block_lowpassr[b.W ? 1 : 0];
        if (lowpassr % p->normal_partition) lowpassr = (lowpassr / p->normal_partition + 1) * p->normal_partition;
    }

    float raw[VBM_MAXCH][PART_MAX], quant[VBM_MAXCH][PART_MAX], floor[VBM_MAXCH][PART_MAX], res[VBM_MAXCH][PART_MAX];
    int flag[VBM_MAXCH][PART_MAX];
    float mdl[PART_MAX], enp[PART_MAX];
    int nz[VBM_MAXCH], nonzero[VBM_MAXCH];
    float acc[VBM_MAXCH + 16];
    float side_resdef[16];
    int i, pi;

    // columns of this stream-block's channels in the bin-major arrays
    const size_t col0 = (size_t)sb * ch;
#define CT(buf, k, x) (buf)[(size_t)((col0 + (k)) >> 6) * SW + (size_t)(x) * 64 + ((col0 + (k)) & 63)]
#define MD(k, x) CT(b.mdctT, k, x)
#define EP(k, x) CT(b.epeakT, k, x)
#define NP(k, x) CT(b.npeakT, k, x)
#define IW(k, x) CT(b.iworkT, k, x)

    for (i = 0; i < ch; i++) nonzero[i] = b.nonzero[col0 + i];
    for (i = 0; i < ch + vi->coupling_steps; i++) acc[i] = 0.f;

    if (prepoint_x < prepoint) prepoint_x = prepoint;
    if (postpoint_x < prepoint) postpoint_x = prepoint;

    for (i = 0; i < vi->coupling_steps; i++) side_resdef[i] = -1.f;

    if (vi->coupling_steps == 1) prae = (float)0.34;
    else prae = (float)0.825;

    const int nparts = (lowpassr + partition - 1) / partition;
    const int pi0 = (int)((long)nparts * blockIdx.y / nchunks), pi1 = (int)((long)nparts * (blockIdx.y + 1) / nchunks);
    const float *m6 = b.m6defT + (size_t)(sb >> 6) * b.sb_slab_words + (sb & 63);

    for (pi = pi0, i = pi0 * partition; pi < pi1; i += partition, pi++) {
        int k, j, jn = partition > n - i ? n - i : partition;
        int step, track = 0;

        for (k = 0; k < ch; k++) nz[k] = nonzero[k];

        for (k = 0; k < ch; k++)
            for (j = 0; j < partition; j++) flag[k][j] = 0;

        for (k = 0; k < ch; k++) {
            if (nz[k]) {
                for (j = 0; j < jn; j++) {
                    floor[k][j] = fromdB[IW(k, i + j)];
                    mdl[j] = MD(k, i + j);
                    enp[j] = EP(k, i + j);
                }

                flag_lossless(limit, prepoint, postpoint, prepoint_x, postpoint_x, res[k], mdl, enp, floor[k], flag[k],
                              i, jn);

                for (j = 0; j < jn; j++) {
                    quant[k][j] = raw[k][j] = mdl[j] * mdl[j];
                    if (mdl[j] < 0.f) raw[k][j] *= -1.f;
                    floor[k][j] *= floor[k][j];
                }

                acc[track] = noise_normalize(&nn, limit, raw[k], quant[k], floor[k], res[k], nullptr, acc[track],
                                             NP(k, pi), i, jn, &IW(k, i), 64);
            } else {
                for (j = 0; j < jn; j++) {
                    floor[k][j] = 1e-10f;
                    raw[k][j] = 0.f;
                    quant[k][j] = 0.f;
                    res[k][j] = 0.f;
                    flag[k][j] = 0;
                    IW(k, i + j) = 0;
                }
                acc[track] = 0.f;
            }
            track++;
        }

        // coupling
        for (step = 0; step < coupling_steps; step++) {
            int Mi = vi->coupling_mag[step];
            int Ai = vi->coupling_ang[step];
            float *reM = raw[Mi], *reA = raw[Ai];
            float *qeM = quant[Mi], *qeA = quant[Ai];
            float *floorM = floor[Mi], *floorA = floor[Ai];
            float *resM = res[Mi], *resA = res[Ai];
            int *fM = flag[Mi], *fA = flag[Ai];
            int pointflag = 0;

            if (nz[Mi] || nz[Ai]) {
                nz[Mi] = nz[Ai] = 1;

                // M6
                if (tonefix_end > i) {
                    int rp = 0, pp = 0, ap;
                    float residue_def = 0;

                    for (j = 0; j < jn; j++) {
                        if (existe(resM[j], 0.5) || existe(resA[j], 0.5)) {
                            if (refer_phase(reM[j], reA[j])) rp++;
                            else pp++;
                            residue_def = (float)((double)residue_def +
                                                  fabs((double)fabsf(resM[j]) - (double)fabsf(resA[j])));
                        }
                    }
                    ap = rp + pp;

                    if (ap != 0) {
                        float temp_def = residue_def = residue_def / ap;
                        float side = side_resdef[step];
                        if (b.couple_parallel)
                            side = (pi > 0) ? m6[((size_t)(pi - 1) * coupling_steps + step) * 64] : -1.f;
                        if (side > 0)
                            residue_def = (float)((double)temp_def * 0.5 + (double)side * 0.5);
                        side_resdef[step] = temp_def;
                        if (residue_def > 1.f) {
                            for (j = 0; j < jn; j++)
                                if (fM[j] == -1 || fA[j] == -1) fM[j] = 1;
                        }
                        if ((float)rp / ap >= prae) {
                            for (j = 0; j < jn; j++)
                                if ((fM[j] == -1 || fA[j] == -1) && refer_phase(reM[j], reA[j])) fM[j] = 1;
                        }
                    } else
                        side_resdef[step] = -1.f;
                }

                for (j = 0; j < jn; j++) {
                    if (j < sliding_lowpass - i) {
                        if (fM[j] == 1 || fA[j] == 1) {
                            // lossless coupling
                            reM[j] = fabsf(reM[j]) + fabsf(reA[j]);
                            qeM[j] = qeM[j] + qeA[j];
                            fM[j] = fA[j] = 1;

                            lossless_couplingf(&resM[j], &resA[j]);
                            int m = IW(Mi, i + j), a = IW(Ai, i + j);
                            lossless_coupling(&m, &a);
                            IW(Mi, i + j) = m;
                            IW(Ai, i + j) = a;
                        } else {
                            // lossy (point) coupling
                            float hpL, hpH;
                            if (coupling_steps == 1 || step == 3) {
                                hpL = .18f;
                                hpH = .12f;
                            } else {
                                hpL = .18f;
                                hpH = .04f;
                            }
                            if (j < limit - i) reM[j] = min_indemnity_dipole_hypot(reM[j], reA[j], hpL);
                            else reM[j] = min_indemnity_dipole_hypot(reM[j], reA[j], hpH);

                            qeM[j] = fabsf(reM[j]);
                            reA[j] = qeA[j] = 0.f;
                            fA[j] = 1;
                            IW(Ai, i + j) = 0;
                            resA[j] = 0;

                            float npM = NP(Mi, pi), npA = NP(Ai, pi);
                            if (((double)npM < -0.5) || ((double)npA < -0.5)) NP(Mi, pi) = -1;
                            else NP(Mi, pi) = VMIN(npM, npA);

                            pointflag |= 1;
                        }
                    }
                    floorM[j] = floorA[j] = floorM[j] + floorA[j];
                }
                if (pointflag)
                    acc[track] = noise_normalize(&nn, limit, raw[Mi], quant[Mi], floor[Mi], res[Mi], flag[Mi], acc[track],
                                                 NP(Mi, pi), i, jn, &IW(Mi, i), 64);
                track++;
            }
        }
    }

    // bins past the lowpass (sliced like the partitions); the nonzero[] propagation over the coupling
    // steps (lib/psy.c:5133-5140) is applied by k_pack_head, after every slice has read the flags
    if (lowpassr < n) {
        const int z0 = lowpassr + (int)((long)(n - lowpassr) * blockIdx.y / nchunks);
        const int z1 = lowpassr + (int)((long)(n - lowpassr) * (blockIdx.y + 1) / nchunks);
        for (int k = 0; k < ch; k++)
            for (int j = z0; j < z1; j++) IW(k, j) = 0;
    }
#undef MD
#undef CT
#undef EP
#undef NP
#undef IW
}

// ---------------------------------------------------------------------------------------------
// Fast path for 32-bin partitions (long blocks): one LANE per (column, bin) instead of one lane per
// stream-block with per-partition arrays in private memory.  A workgroup of FP x FPC threads owns FPC
// columns x one partition: rows are loaded / stored with the column index fastest (coalesced 128-B
// row segments of the tiled arrays) and transposed through LDS so that the 32 lanes of a half-wave
// hold the 32 bins of one column's partition in registers.  (FPC columns per workgroup.)
//   MODE 0  no coupling: a column is a channel-block (any channel count, e.g. mono, 5.1 q8)
//   MODE 1  stereo, one coupling step: a column is a stream-block, both channels in the lane
// Per-bin work is elementwise.  The order-bound pieces of the source run as 32-step loops over the
// half-wave with shuffles, in bin order: flag_lossless' running point1/point2 (:4597-4600), M6's
// residue_def (:5018-5027).  noise_normalize's tail (sort + unit-norm promotion, :4800-4852) only
// exists when normal_start falls inside the partition; then lane 0 of the half-wave runs the serial
// routine above on LDS copies — the same code as the general kernel, so exact by construction.
#define FP 32      /* bins of a partition */
#ifndef FPC
#define FPC 8      /* columns per workgroup: FP * FPC = 256 threads (measured 8 > 16 > 32 columns, pipelined) */
#endif
struct fast_consts {
    int limit, sliding_lowpass, lowpassr, tonefix_end, n;
    float prepoint, postpoint, prepoint_x, postpoint_x, prae;
    nn_consts nn;
};


// noise_normalize (lib/psy.c:4732-4854) for one 32-bin partition held by the 32 lanes of a half-wave, lane j =
// bin j (the caller has established start < n).  What the source leaves order-bound is kept in its order:
//  * acc, the float sum of the small values in bin order: a 32-step loop over the lanes' values;
//  * ssort's partial selection sort (lib/psy.c:4709-4726, ties included: its swaps decide which of two equal
//    magnitudes comes first): position p of the sort[] array lives in lane p, every step is a 5-step
//    (key, position) maximum over the half-wave followed by the swap of two lanes' entries;
//  * the promotions: acc only falls, so the promoted elements are the first P of the sorted order.
// `slot`: 32 ints of LDS scratch of this half-wave (the inverse permutation goes through it).
__device__ __forceinline__ void nn_wave(const nn_consts &p, const int limit, const float r, float &q, const float f, float &res,
                                        const bool has_flags, const int flag, const float nepeak, const int i, const int n,
                                        const int j, const int base, int *slot, int &out)
{
    int start = (p.normal_p ? p.normal_start - i : n);
    if ((start > n) || ((double)nepeak < -0.5)) start = n;
    const bool inr = j < n;
    bool member = false;
    float vem = 0.f;
    if (inr && j < start) {
        if (!has_flags) {
            out = (int)rint((double)res);
        } else if (flag != 1) {
            const float ve = (float)sqrt((double)(q / f));
            if (r < 0) { out = (int)-rint((double)ve); res = -ve; }
            else { out = (int)rint((double)ve); res = ve; }
        }
    } else if (inr) {
        if (has_flags) {
            if (flag != 1) {
                float ve = q / f;
                if (ve < .25f && j >= limit - i) {
                    member = true;
                    vem = ve;
                    if (r < 0) res = (float)-sqrt((double)ve);
                    else res = (float)sqrt((double)ve);
                } else {
                    ve = (float)sqrt((double)ve);
                    int o;
                    if (r < 0) { o = (int)-rint((double)ve); res = -ve; }
                    else { o = (int)rint((double)ve); res = ve; }
                    out = o;
                    q = o * o * f;
                }
            }
        } else {
            const float ve = res * res;
            if (ve < .25f) {
                member = true;
                vem = ve;
            } else {
                const int o = (int)rint((double)res);
                out = o;
                q = o * o * f;
            }
        }
    }
    const unsigned mask = (unsigned)(__ballot(member) >> base);
    const int count = __popc(mask);
    if (count == 0) return;                              // uniform over the half-wave
    float acc = 0.f;
    for (int t = 0; t < FP; t++) {
        const float v = __shfl(vem, base + t);
        if ((mask >> t) & 1u) acc += v;
    }
    acc += acc * nepeak * nepeak;
    int iacc = ((int)acc) + 1;
    if (iacc > n) iacc = n;
    const int bthresh = count < iacc ? count : iacc;
    // lane p < count: entry p of sort[] = the p-th member in bin order
    int e = 0;
    {
        unsigned m = mask;
        for (int t = 0; t < j && m; t++) m &= m - 1;     // drop the j lowest set bits
        e = m ? __ffs(m) - 1 : 0;
    }
    float key = __shfl(q, base + e);
    const bool live = j < count;
    for (int a = 0; a < bthresh; a++) {
        float bk = (live && j >= a) ? key : -3.0e38f;
        int bp = (live && j >= a) ? j : 64;
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) {
            const float ok = __shfl_xor(bk, sft);
            const int op = __shfl_xor(bp, sft);
            // the scan `if (q[sort[large]] < q[sort[bb]]) large = bb` keeps the earliest of equal maxima
            if (ok > bk || (ok == bk && op < bp)) { bk = ok; bp = op; }
        }
        const int large = bp;                            // uniform
        const int eA = __shfl(e, base + a), eL = __shfl(e, base + large);
        const float kA = __shfl(key, base + a), kL = __shfl(key, base + large);
        if (j == a) { e = eL; key = kL; }
        else if (j == large) { e = eA; key = kA; }
    }
    int P = 0;
    {
        float a2 = acc;
        while (P < count && (double)a2 >= p.normal_thresh) { a2 -= 1.f; P++; }
    }
    if (live) slot[e] = j;                               // element e sits at position j of the sorted order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (member) {
        const int pos = slot[j];
        if (pos < P) { out = (int)vbm_unitnorm(r); q = f; }
        else { out = 0; q = 0.f; }
    }
    __builtin_amdgcn_wave_barrier();
}

struct fast_lds {
    float in_md[2][FP][FPC + 1], in_ep[2][FP][FPC + 1];
    int in_iw[2][FP][FPC + 1];
    int out[FPC][FP];   // nn_wave's scratch (inverse permutation of the partial sort), one row per half-wave
};

// per-channel part of a partition for this lane's bin (lib/psy.c:4952-4991)
__device__ __forceinline__ void fast_channel(const fast_consts &c, fast_lds &L, const float *__restrict__ fromdB,
                                             const int g, const int j, const int i, const int jn, const bool nz,
                                             const float mdl, const float enp, const int iw, const float nepeak,
                                             float &raw, float &quant, float &floor, float &res, int &flag, int &out)
{
    const int lane = threadIdx.x & 63, base = lane & 32;
    if (!nz) {
        floor = 1e-10f; raw = 0.f; quant = 0.f; res = 0.f; flag = 0; out = 0;
        return;
    }
    floor = fromdB[iw];
    // flag_lossless (lib/psy.c:4584-4624): point1/point2 advance by a constant per bin, in bin order
    {
        const int pointlimit = c.limit - i;
        float point1, point2, ps1 = 0.f, ps2 = 0.f;
        int ps = 0;
        if (pointlimit > 0) {
            point1 = c.prepoint;
            point2 = c.prepoint_x;
            if ((pointlimit - jn) <= 0) {
                ps1 = (c.postpoint - c.prepoint) / jn;
                ps2 = (c.postpoint_x - c.prepoint_x) / jn;
                ps = 1;
            }
        } else {
            point1 = c.postpoint;
            point2 = c.postpoint_x;
        }
        if (ps) {
            float p1 = point1, p2 = point2;
            for (int t = 0; t < FP; t++) {
                p1 += ps1;
                p2 += ps2;
                if (t == j) { point1 = p1; point2 = p2; }
            }
        }
        res = mdl / floor;
        const float r = fabsf(res);
        point1 -= enp;
        if (point1 < c.prepoint) point1 = c.prepoint;
        if (r < point1) flag = (r < point2) ? 0 : -1;
        else flag = 1;
    }
    quant = raw = mdl * mdl;
    if (mdl < 0.f) raw *= -1.f;
    floor *= floor;

    // noise_normalize(p, limit, raw, quant, floor, res, NULL, ...) (lib/psy.c:4732-4854)
    int start = (c.nn.normal_p ? c.nn.normal_start - i : jn);
    if ((start > jn) || ((double)nepeak < -0.5)) start = jn;
    if (start >= jn) {
        out = (int)rint((double)res);
    } else {
        nn_wave(c.nn, c.limit, raw, quant, floor, res, false, 0, nepeak, i, jn, j, base, L.out[g], out);
    }
}

template <int MODE, bool BLOBS>
__global__ __launch_bounds__(FP * FPC) void k_couple_fast(vbm_batch b_in)
{
    vbm_batch b = b_in;
    __shared__ fast_lds L;
    if ((int)(blockIdx.x * FPC) >= ((MODE == 1) ? vbm_nsb(b) : vbm_ncb(b))) return;   // (launch bound > device-resident count)
    const vbm_setup *s = b.setup;
    const vbm_psy *p = &s->psy[b.block_mode];
    const vbm_map *vi = &s->map[b.W];
    const size_t SW = b.slab_words;
    const int NCH = (MODE == 1) ? 2 : 1;
    fast_consts c;
    c.n = p->n;
    c.tonefix_end = p->tonefix_end;
    c.prae = (vi->coupling_steps == 1) ? (float)0.34 : (float)0.825;
    c.nn.normal_p = p->normal_p; c.nn.normal_start = p->normal_start; c.nn.normal_thresh = p->normal_thresh;
    {
        int lowpassr = s->block_lowpassr[b.W ? 1 : 0];
        if (lowpassr % p->normal_partition) lowpassr = (lowpassr / p->normal_partition + 1) * p->normal_partition;
        c.lowpassr = lowpassr;
    }
    const float *__restrict__ fromdB = s->fromdB;
    const int ncols = (MODE == 1) ? vbm_nsb(b) : vbm_ncb(b);
    const int pi = blockIdx.y, i = pi * FP;
    const int tid = threadIdx.x;

    // ---- rows i..i+31 of the 32 columns of this workgroup: coalesced loads, LDS transpose -------
    const int lr = tid / FPC, lc = tid % FPC;
    const int colL = blockIdx.x * FPC + lc;
    const bool past = (i >= c.lowpassr);   // partitions past the lowpass only zero the residue (lib/psy.c:5126-5131)
    if (!past && colL < ncols && i + lr < c.n) {
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const size_t cb = (MODE == 1) ? (size_t)colL * 2 + k : (size_t)colL;
            const size_t a = (cb >> 6) * SW + (size_t)(i + lr) * 64 + (cb & 63);
            L.in_md[k][lr][lc] = b.mdctT[a];
            L.in_ep[k][lr][lc] = b.epeakT[a];
        }
    }

    // compute lanes: (column g, bin j)
    const int g = tid >> 5, j = tid & 31;
    const int col = blockIdx.x * FPC + g;
    const int lane = tid & 63, base = lane & 32;
    const int jn = FP > c.n - i ? c.n - i : FP;
    // noise peaks of the partition: read once; a managed block's packetblobs hand them on to one another
    // (_vp_couple_quantize_normalize rewrites NP(Mi) where it point-couples, lib/psy.c:5100-5108, and the reference's
    // loop over the blobs passes the same rows every time, lib/mapping0.c:1249-1260)
    float npk[2] = {0.f, 0.f};
    if (!past && col < ncols)
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const size_t cb = (MODE == 1) ? (size_t)col * 2 + k : (size_t)col;
            npk[k] = b.npeakT[(cb >> 6) * SW + (size_t)pi * 64 + (cb & 63)];
        }

    const int nbl = BLOBS ? b.nblobs : 1;
    for (int kb = 0; kb < nbl; kb++) {
    if (BLOBS) {
        __syncthreads();                 // (the store of the blob before this one is done with in_iw)
        vbm_blob_select(b, kb);
    }
    const int blobno = b.blobno;
    c.limit = s->coupling_pointlimit[p->blockflag][blobno];
    c.prepoint = (float)s->stereo_threshholds[s->coupling_prepointamp[blobno]];
    c.postpoint = (float)s->stereo_threshholds[s->coupling_postpointamp[blobno]];
    c.prepoint_x = (float)s->stereo_threshholds_X[s->coupling_prepointamp[blobno]];
    c.postpoint_x = (float)s->stereo_threshholds_X[s->coupling_postpointamp[blobno]];
    if (c.prepoint_x < c.prepoint) c.prepoint_x = c.prepoint;
    if (c.postpoint_x < c.prepoint) c.postpoint_x = c.prepoint;
    c.sliding_lowpass = s->sliding_lowpass[b.W][blobno];
    if (!past && colL < ncols && i + lr < c.n) {
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const size_t cb = (MODE == 1) ? (size_t)colL * 2 + k : (size_t)colL;
            L.in_iw[k][lr][lc] = b.iworkT[(cb >> 6) * SW + (size_t)(i + lr) * 64 + (cb & 63)];
        }
    }
    __syncthreads();

    // ---- compute: lane = (column g, bin j) ---------------------------------------------------
    int out[2] = {0, 0};
    if (!past && col < ncols) {
        float raw[2], quant[2], floor[2], res[2];
        int flag[2];
        bool nz[2];
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const size_t cb = (MODE == 1) ? (size_t)col * 2 + k : (size_t)col;
            nz[k] = b.nonzero[cb] != 0;
            fast_channel(c, L, fromdB, g, j, i, jn, nz[k], L.in_md[k][j][g], L.in_ep[k][j][g], L.in_iw[k][j][g], npk[k],
                         raw[k], quant[k], floor[k], res[k], flag[k], out[k]);
        }

        if (MODE == 1) {
            const int Mi = vi->coupling_mag[0], Ai = vi->coupling_ang[0];
            // registers of the magnitude / angle channel
            float reM = raw[Mi & 1], reA = raw[Ai & 1], qeM = quant[Mi & 1], qeA = quant[Ai & 1];
            float floorM = floor[Mi & 1], floorA = floor[Ai & 1], resM = res[Mi & 1], resA = res[Ai & 1];
            int fM = flag[Mi & 1], fA = flag[Ai & 1], oM = out[Mi & 1], oA = out[Ai & 1];
            float npM = npk[Mi & 1];
            const float npA = npk[Ai & 1];
            if (nz[0] || nz[1]) {
                // M6 (lib/psy.c:5010-5051)
                if (c.tonefix_end > i) {
                    const bool cond = existe(resM, 0.5) || existe(resA, 0.5);
                    const bool phase = refer_phase(reM, reA);
                    const unsigned mc = (unsigned)(__ballot(cond && j < jn) >> base);
                    const unsigned mp = (unsigned)(__ballot(cond && phase && j < jn) >> base);
                    const int rp = __popc(mp), pp = __popc(mc & ~mp);
                    const double term = fabs((double)fabsf(resM) - (double)fabsf(resA));
                    float residue_def = 0;
                    for (int t = 0; t < FP; t++) {
                        const double tt = __shfl(term, base + t);
                        if ((mc >> t) & 1u) residue_def = (float)((double)residue_def + tt);
                    }
                    const int ap = rp + pp;
                    if (ap != 0) {
                        const float temp_def = residue_def = residue_def / ap;
                        const float *m6 = b.m6defT + (size_t)(col >> 6) * b.sb_slab_words + (col & 63);
                        const float side = (pi > 0) ? m6[(size_t)(pi - 1) * 64] : -1.f;
                        if (side > 0) residue_def = (float)((double)temp_def * 0.5 + (double)side * 0.5);
                        if (residue_def > 1.f) {
                            if (fM == -1 || fA == -1) fM = 1;
                        }
                        if ((float)rp / ap >= c.prae) {
                            if ((fM == -1 || fA == -1) && refer_phase(reM, reA)) fM = 1;
                        }
                    }
                }

                bool lossy = false;
                if (j < c.sliding_lowpass - i) {
                    if (fM == 1 || fA == 1) {
                        // lossless coupling
                        reM = fabsf(reM) + fabsf(reA);
                        qeM = qeM + qeA;
                        fM = fA = 1;
                        lossless_couplingf(&resM, &resA);
                        lossless_coupling(&oM, &oA);
                    } else {
                        // lossy (point) coupling; one step: hpL .18, hpH .12 (lib/psy.c:5075-5081)
                        if (j < c.limit - i) reM = min_indemnity_dipole_hypot(reM, reA, .18f);
                        else reM = min_indemnity_dipole_hypot(reM, reA, .12f);
                        qeM = fabsf(reM);
                        reA = qeA = 0.f;
                        fA = 1;
                        oA = 0;
                        resA = 0;
                        lossy = true;
                    }
                }
                floorM = floorA = floorM + floorA;
                const unsigned ml = (unsigned)(__ballot(lossy && j < jn) >> base);
                if (ml) {
                    // NP(Mi) = -1 or min(NP(Mi), NP(Ai)) — idempotent, the source repeats it per lossy bin
                    if (((double)npM < -0.5) || ((double)npA < -0.5)) npM = -1;
                    else npM = VMIN(npM, npA);
                    npk[Mi & 1] = npM;       // (every lane of the half-wave holds it: the next packetblob starts from here)
                    if (j == 0) {
                        const size_t cbM = (size_t)col * 2 + Mi;
                        b.npeakT[(cbM >> 6) * SW + (size_t)pi * 64 + (cbM & 63)] = npM;
                    }
                    // noise_normalize(p, limit, raw[Mi], quant[Mi], floor[Mi], res[Mi], flag[Mi], ...)
                    int start = (c.nn.normal_p ? c.nn.normal_start - i : jn);
                    if ((start > jn) || ((double)npM < -0.5)) start = jn;
                    if (start >= jn) {
                        if (fM != 1) {
                            const float ve = (float)sqrt((double)(qeM / floorM));
                            if (reM < 0) {
                                oM = (int)-rint((double)ve);
                                resM = -ve;
                            } else {
                                oM = (int)rint((double)ve);
                                resM = ve;
                            }
                        }
                    } else {
                        nn_wave(c.nn, c.limit, reM, qeM, floorM, resM, true, fM, npM, i, jn, j, base, L.out[g], oM);
                    }
                }
            }
            out[Mi & 1] = oM;
            out[Ai & 1] = oA;
        }
    }

    // ---- store.  Fused packet assembly (k_pack_fused): the residue goes out in the residue coder's own order — bin-interleaved
    //      channels, res_bm[column][bin * NCH + channel] (lib/res0.c:781-787) — straight from the lane that holds it: the 32
    //      lanes of a half-wave write 32 * NCH consecutive ints.  Otherwise: tiled rows, back through LDS so that they go out
    //      coalesced.
    if (b.pack_fused) {      // (never a managed setup: one pass)
        if (col < ncols && i + j < c.n) {
            if (MODE == 1) {
                int2 *dst = reinterpret_cast<int2 *>(b.res_bm + (size_t)col * c.n * 2) + (i + j);
                *dst = make_int2(out[0], out[1]);
            } else {
                // MODE 0 runs with one channel here (pack_fused: a single coded vector): column = stream-block
                b.res_bm[(size_t)col * c.n + i + j] = out[0];
            }
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCH; k++) L.in_iw[k][j][g] = out[k];
    __syncthreads();
    if (colL < ncols && i + lr < c.n) {
#pragma unroll
        for (int k = 0; k < NCH; k++) {
            const size_t cb = (MODE == 1) ? (size_t)colL * 2 + k : (size_t)colL;
            b.iworkT[(cb >> 6) * SW + (size_t)(i + lr) * 64 + (cb & 63)] = L.in_iw[k][lr][lc];
        }
    }
    }   // packetblobs
}

}  // namespace

extern "C" int vbm_launch_couple_quantize(const vbm_batch *b, hipStream_t st)
{
    const unsigned tiles = (unsigned)((b->nsb + 63) / 64);
    // fast path: 32-bin partitions with no coupling, or stereo with one coupling step
    if (b->couple_fast == 1) {
        if (b->nblobs > 1)
            hipLaunchKernelGGL((k_couple_fast<0, true>), dim3((unsigned)((b->ncb + FPC - 1) / FPC), (unsigned)(b->n / FP)), dim3(FP * FPC), 0,
                               st, *b);
        else
            hipLaunchKernelGGL((k_couple_fast<0, false>), dim3((unsigned)((b->ncb + FPC - 1) / FPC), (unsigned)(b->n / FP)), dim3(FP * FPC), 0,
                               st, *b);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (b->couple_fast == 2) {
        if (b->nblobs > 1) {
            if (b->couple_m6parts > 0)
                hipLaunchKernelGGL(k_couple_m6stats<true>, dim3(tiles, (unsigned)b->couple_m6parts, (unsigned)b->nblobs), dim3(64), 0, st, *b);
            hipLaunchKernelGGL((k_couple_fast<1, true>), dim3((unsigned)((b->nsb + FPC - 1) / FPC), (unsigned)(b->n / FP)), dim3(FP * FPC),
                               0, st, *b);
        } else {
            if (b->couple_m6parts > 0)
                hipLaunchKernelGGL(k_couple_m6stats<false>, dim3(tiles, (unsigned)b->couple_m6parts), dim3(64), 0, st, *b);
            hipLaunchKernelGGL((k_couple_fast<1, false>), dim3((unsigned)((b->nsb + FPC - 1) / FPC), (unsigned)(b->n / FP)), dim3(FP * FPC),
                               0, st, *b);
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    int nchunks = 1;
    if (b->couple_parallel) {
        if (b->couple_m6parts > 0)
            hipLaunchKernelGGL(k_couple_m6stats<false>, dim3(tiles, (unsigned)b->couple_m6parts), dim3(64), 0, st, *b);   // (general kernel: the host walks the blobs)
        nchunks = b->couple_parts < 32 ? (b->couple_parts > 0 ? b->couple_parts : 1) : 32;
    }
    hipLaunchKernelGGL(k_couple_quantize, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
