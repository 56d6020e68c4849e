// Floor-1 fit and encode for gfx950 — one lane per channel-block (batch.h layout).
//
//   k_floor_fit     floor1_fit (reference lib/floor1.c:597-750): accumulate_fit :427-475 (integer
//                   sums per post segment), fit_line :477-535 (double least squares, rint),
//                   inspect_error :537-586, greedy splitting :646-719, post_Y / render_point
//   k_floor_encode  value part of floor1_encode (lib/floor1.c:774-852, 944-973): quantise posts
//                   by `mult`, predict / wrap the deviations, render the integer floor curve
//                   (render_line0 :397-424) into ilogmask.  The entropy coding of the wrapped
//                   deviations (:856-942) happens in the packet kernel, in channel order.
// Segment bounds, neighbour tables and sort order come from the floor look and are the same
// in every lane; only the greedy split decisions diverge.
//
// floor1_fit looks at the mask only through dBquant(logmask[x]) and the test
// logmdct[x] + twofitatten >= logmask[x].  k_floor_prep evaluates both once per bin and stores
// them as one 16-bit word (quantised level | test << 15) in BLOCK-major rows qf_bm[channel-block][n]
// (tile transpose through LDS).  The fit walks data-dependent bin ranges (inspect_error), so each
// lane reads its own row: consecutive bins of a lane share cache lines, which the bin-major tiles
// cannot offer.  k_floor_fit runs `lpw` lanes per wavefront (launch parameter) to put more
// wavefronts in flight than ncb/64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "batch.h"
#include "kernels.h"

#define T(buf, i) (buf)[tb + (size_t)(i) * 64]   /* tiled bin-major: batch.h */

namespace {

// accumulate_fit's sums of one post segment; the y^2 sums of the source never reach the solution
// (fit_line accumulates but does not read them) and are not kept
struct lsfit_acc {
    int x0, x1;
    int xa, ya, x2a, xya, an;
    int xb, yb, x2b, xyb, bn;
};

// 8-bin register window over a lane's qf row: one 16-byte load per 8 consecutive bins
struct qf_window {
    const uint16_t *row;
    int blk;
    uint64_t lo, hi;
};
__device__ __forceinline__ int qf_get(qf_window &w, int i)
{
    const int blk = i >> 3;
    if (blk != w.blk) {
        w.blk = blk;
        const uint4 v = *reinterpret_cast<const uint4 *>(w.row + (blk << 3));
        w.lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        w.hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    const uint64_t h = (i & 4) ? w.hi : w.lo;
    return (int)((h >> ((i & 3) * 16)) & 0xffff);
}

__device__ __forceinline__ int render_point(int x0, int x1, int y0, int y1, int x)
{
    y0 &= 0x7fff;
    y1 &= 0x7fff;
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = abs(dy);
    int err = ady * (x - x0);
    int off = err / adx;
    if (dy < 0) return (y0 - off);
    return (y0 + off);
}

// lib/floor1.c:294 (scalar): int i = *x*7.3142857f+1023.5f
__device__ __forceinline__ int dBquant(float x)
{
    int i = (int)(x * 7.3142857f + 1023.5f);
    if (i > 1023) return (1023);
    if (i < 0) return (0);
    return i;
}

// accumulate_fit's ten sums per post segment (fields: xa ya x2a xya an xb yb x2b xyb bn), either in the lane's
// private memory or in LDS ([segment * 10 + field][lane]: conflict-free).  The greedy loop re-reads them in
// every fit_line (two per split), a chain of dependent loads: from LDS they come back in tens of cycles
// instead of a scratch round trip.
struct fits_private {
    int v[(VBM_VIF_POSIT + 1) * 10];
    __device__ __forceinline__ int get(int seg, int f) const { return v[seg * 10 + f]; }
    __device__ __forceinline__ void set(int seg, int f, int x) { v[seg * 10 + f] = x; }
};
struct fits_lds {
    int *p;   // base + lane
    __device__ __forceinline__ int get(int seg, int f) const { return p[(seg * 10 + f) * 64]; }
    __device__ __forceinline__ void set(int seg, int f, int x) { p[(seg * 10 + f) * 64] = x; }
};

// fit_line over segments [seg0, seg0 + fits) (lib/floor1.c:477-535); x0 / x1 are the outer posts of the range
template <typename Store>
__device__ __forceinline__ int fit_line(const Store &S, const int *__restrict__ sorted_index, int seg0, int fits, int *y0, int *y1,
                                        const float twofitweight)
{
    double xb = 0, yb = 0, x2b = 0, xyb = 0, bn = 0;
    int i;
    int x0 = sorted_index[seg0];
    int x1 = sorted_index[seg0 + fits];

    for (i = seg0; i < seg0 + fits; i++) {
        const int axa = S.get(i, 0), aya = S.get(i, 1), ax2a = S.get(i, 2), axya = S.get(i, 3), aan = S.get(i, 4);
        const int axb = S.get(i, 5), ayb = S.get(i, 6), ax2b = S.get(i, 7), axyb = S.get(i, 8), abn = S.get(i, 9);
        double weight = (double)((float)(abn + aan) * twofitweight / (float)(aan + 1)) + 1.;

        xb += axb + axa * weight;
        yb += ayb + aya * weight;
        x2b += ax2b + ax2a * weight;
        xyb += axyb + axya * weight;
        bn += abn + aan * weight;
    }

    if (*y0 >= 0) {
        xb += x0;
        yb += *y0;
        x2b += x0 * x0;
        xyb += *y0 * x0;
        bn++;
    }

    if (*y1 >= 0) {
        xb += x1;
        yb += *y1;
        x2b += x1 * x1;
        xyb += *y1 * x1;
        bn++;
    }

    {
        double denom = (bn * x2b - xb * xb);

        if (denom > 0.) {
            double aa = (yb * x2b - xyb * xb) / denom;
            double bb = (bn * xyb - xb * yb) / denom;
            *y0 = (int)rint(aa + bb * x0);
            *y1 = (int)rint(aa + bb * x1);

            if (*y0 > 1023) *y0 = 1023;
            if (*y1 > 1023) *y1 = 1023;
            if (*y0 < 0) *y0 = 0;
            if (*y1 < 0) *y1 = 0;
            return 0;
        } else {
            *y0 = 0;
            *y1 = 0;
            return 1;
        }
    }
}

__device__ __forceinline__ int post_Y(const int *A, const int *B, int pos)
{
    if (A[pos] < 0) return B[pos];
    if (B[pos] < 0) return A[pos];
    return (A[pos] + B[pos]) >> 1;
}

__device__ __forceinline__ int post_Y(const short *A, const short *B, int pos)
{
    if (A[pos] < 0) return B[pos];
    if (B[pos] < 0) return A[pos];
    return (A[pos] + B[pos]) >> 1;
}

// 64 channel-blocks x 64 bins per workgroup of 256 threads
__global__ void k_floor_prep(vbm_batch b)
{
    __shared__ uint16_t tile[64][66];
    const vbm_setup *s = b.setup;
    const vbm_map *map = &s->map[b.W];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    {
        const int c = c0 + tx;
        float att = 0.f;
        if (c < vbm_ncb(b)) att = s->floor[map->floorsubmap[map->chmuxlist[c % b.ch]]].twofitatten;
        const size_t tb = (size_t)(c >> 6) * b.slab_words + (c & 63);
        for (int rr = ty; rr < 64; rr += 4) {
            const int r = r0 + rr;
            uint16_t w = 0;
            if (c < vbm_ncb(b) && r < b.n) {
                const float fl = T(b.logmaskT, r);
                w = (uint16_t)(dBquant(fl) | ((T(b.logmdctT, r) + att >= fl) ? 0x8000 : 0));
            }
            tile[rr][tx] = w;
        }
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, r = r0 + tx;
        if (c < vbm_ncb(b) && r < b.n) b.qf_bm[(size_t)c * b.n + r] = tile[tx][cc];
    }
}

template <bool LDS>
__global__ void k_floor_fit(vbm_batch b, int lpw)
{
    extern __shared__ int fit_lds[];   // LDS: [(posts - 1) * 10][64]
    const int lane = blockIdx.x * lpw + threadIdx.x;
    if ((int)threadIdx.x >= lpw || lane >= vbm_ncb(b)) return;
    const size_t tb = (size_t)(lane >> 6) * b.slab_words + (lane & 63);
    qf_window qf;
    qf.row = b.qf_bm + (size_t)lane * b.n;   // rows are 16-byte aligned: n is a multiple of 8
    qf.blk = -1;
    qf.lo = qf.hi = 0;
    const vbm_setup *s = b.setup;
    const int c = lane % b.ch;
    const vbm_map *map = &s->map[b.W];
    const vbm_floor *look = &s->floor[map->floorsubmap[map->chmuxlist[c]]];
    const vbm_floor *info = look;
    const int n = look->n;
    const int posts = look->posts;
    // loop-invariant look fields in registers (the look is addressed per lane, the compiler cannot
    // hoist these loads over the stores in the loops)
    const float maxover = info->maxover, maxunder = info->maxunder, maxerr = info->maxerr;
    const float twofitweight = info->twofitweight;
    const int imaxover = (int)floorf(maxover), imaxunder = (int)floorf(maxunder);   // both >= 0 in every floor template
    int i, j;
    int nonzero = 0;

    typename std::conditional<LDS, fits_lds, fits_private>::type fits;
    if constexpr (LDS) fits.p = fit_lds + threadIdx.x;
    int fit_valueA[VBM_VIF_POSIT + 2];
    int fit_valueB[VBM_VIF_POSIT + 2];
    int loneighbor[VBM_VIF_POSIT + 2];
    int hineighbor[VBM_VIF_POSIT + 2];
    int memo[VBM_VIF_POSIT + 2];

    for (i = 0; i < posts; i++) {
        fit_valueA[i] = -200;
        fit_valueB[i] = -200;
        loneighbor[i] = 0;
        hineighbor[i] = 1;
        memo[i] = -1;
    }

    // accumulate_fit over every minimal division (lib/floor1.c:427-475, :625-628) in ONE walk over the lane's row,
    // four 16-byte pieces in flight (a lane's loads are a chain of round trips to L2 otherwise: ~130 of them).  A
    // segment takes the bins sorted_index[seg] .. min(sorted_index[seg + 1], n - 1), both ends included: a post's
    // own bin counts for the segment on either side.
    {
        const int nseg = posts - 1;
        for (int seg = 0; seg < nseg; seg++)
            for (int f = 0; f < 10; f++) fits.set(seg, f, 0);
        int seg = 0, end = look->sorted_index[1] >= n ? n - 1 : look->sorted_index[1];
        int xa = 0, ya = 0, x2a = 0, xya = 0, na = 0, xb = 0, yb = 0, x2b = 0, xyb = 0, nb = 0;
        bool live = true;       // false: the bins left lie behind the last post (a floor that ends below n)
        auto flush = [&]() {
            fits.set(seg, 0, xa); fits.set(seg, 1, ya); fits.set(seg, 2, x2a); fits.set(seg, 3, xya); fits.set(seg, 4, na);
            fits.set(seg, 5, xb); fits.set(seg, 6, yb); fits.set(seg, 7, x2b); fits.set(seg, 8, xyb); fits.set(seg, 9, nb);
            nonzero += na;
            xa = ya = x2a = xya = na = xb = yb = x2b = xyb = nb = 0;
        };
        const int last = (n - 1) >> 3;      // (the row has b.n >= n entries, a multiple of 8)
        for (int blk0 = 0; blk0 <= last && live; blk0 += 4) {
            uint4 v4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v4[r] = *reinterpret_cast<const uint4 *>(qf.row + ((blk0 + r <= last ? blk0 + r : last) << 3));
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (blk0 + r > last || !live) continue;
                const uint32_t wd[4] = {v4[r].x, v4[r].y, v4[r].z, v4[r].w};
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    i = ((blk0 + r) << 3) + u;
                    const int w = (int)((wd[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                    const int quantized = w & 0x7fff;
                    if (quantized && live) {
                        if (w & 0x8000) { xa += i; ya += quantized; x2a += i * i; xya += i * quantized; na++; }
                        else { xb += i; yb += quantized; x2b += i * i; xyb += i * quantized; nb++; }
                    }
                    if (i == end) {
                        flush();
                        if (seg + 1 < nseg && look->sorted_index[seg + 1] == i) {
                            seg++;      // the post's bin opens the next segment too
                            const int e = look->sorted_index[seg + 1];
                            end = e >= n ? n - 1 : e;
                            if (quantized) {
                                if (w & 0x8000) { xa += i; ya += quantized; x2a += i * i; xya += i * quantized; na++; }
                                else { xb += i; yb += quantized; x2b += i * i; xyb += i * quantized; nb++; }
                            }
                        } else {
                            live = false;   // the last segment (or one cut short at n - 1) is over
                        }
                    }
                }
            }
        }
    }

    if (!nonzero) {
        b.post_valid[lane] = 0;
        return;
    }

    {
        int y0 = -200;
        int y1 = -200;
        fit_line(fits, look->sorted_index, 0, posts - 1, &y0, &y1, twofitweight);

        fit_valueA[0] = y0;
        fit_valueB[0] = y0;
        fit_valueB[1] = y1;
        fit_valueA[1] = y1;

        for (i = 2; i < posts; i++) {
            int sortpos = look->reverse_index[i];
            int ln = loneighbor[sortpos];
            int hn = hineighbor[sortpos];

            if (memo[ln] != hn) {
                int lsortpos = look->reverse_index[ln];
                int hsortpos = look->reverse_index[hn];
                memo[ln] = hn;
                {
                    int lx = info->postlist[ln];
                    int hx = info->postlist[hn];
                    int ly = post_Y(fit_valueA, fit_valueB, ln);
                    int hy = post_Y(fit_valueA, fit_valueB, hn);
                    // (the reference exit(1)s on ly == -1 || hy == -1, which cannot occur: values are
                    //  -200 or >= 0)

                    // inspect_error (lib/floor1.c:537-586)
                    int split;
                    {
                        int dy = hy - ly;
                        int adx = hx - lx;
                        int ady = abs(dy);
                        int base = dy / adx;
                        int sy = (dy < 0 ? base - 1 : base + 1);
                        int x = lx;
                        int y = ly;
                        int err = 0;
                        int wv = qf_get(qf, x);
                        int val = wv & 0x7fff;
                        int mse = 0;
                        int cnt = 0;
                        split = -1;

                        ady -= abs(base * adx);

                        mse = (y - val);
                        mse *= mse;
                        cnt++;
                        if (wv & 0x8000) {
                            if (y + maxover < val) split = 1;
                            if (y - maxunder > val) split = 1;
                        }
                        if (split < 0 && lx + 1 < hx) {
                            // bins lx+1 .. hx-1 in groups of eight (one 16-byte load each).  y, val are integers,
                            // so  y + maxover < val  <=>  val - y > floor(maxover)  and  y - maxunder > val  <=>
                            // y - val > floor(maxunder)  (thresholds >= 0): integer compares.  After a violation
                            // mse / cnt are never read (split = 1), so the group is simply abandoned.
                            const int xlo = lx + 1, xhi = hx - 1;
                            // (four pieces are fetched before the first is looked at: a lane's loads would be a chain
                            // of round trips otherwise; pieces past the range are fetched again from its last one)
                            const int blast = xhi >> 3;
                            for (int blk0 = xlo >> 3; blk0 <= blast && split < 0; blk0 += 4) {
                                uint4 v4[4];
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    v4[r] = *reinterpret_cast<const uint4 *>(qf.row + ((blk0 + r <= blast ? blk0 + r : blast) << 3));
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int blk = blk0 + r;
                                    if (blk > blast || split >= 0) continue;
                                    const uint32_t wd[4] = {v4[r].x, v4[r].y, v4[r].z, v4[r].w};
#pragma unroll
                                    for (int u = 0; u < 8; u++) {
                                        x = (blk << 3) + u;
                                        if ((unsigned)(x - xlo) <= (unsigned)(xhi - xlo)) {
                                            err = err + ady;
                                            if (err >= adx) {
                                                err -= adx;
                                                y += sy;
                                            } else {
                                                y += base;
                                            }
                                            wv = (int)((wd[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                                            val = wv & 0x7fff;
                                            const int d = y - val;
                                            mse += d * d;
                                            if ((wv & 0x8000) && val && (-d > imaxover || d > imaxunder)) split = 1;
                                        }
                                    }
                                }
                            }
                            cnt += xhi - xlo + 1;
                        }
                        if (split < 0) {
                            if (maxover * maxover / cnt > maxerr) split = 0;
                            else if (maxunder * maxunder / cnt > maxerr) split = 0;
                            else if (mse / cnt > maxerr) split = 1;
                            else split = 0;
                        }
                    }

                    if (split) {
                        int ly0 = -200, ly1 = -200, hy0 = -200, hy1 = -200;
                        int ret0 = fit_line(fits, look->sorted_index, lsortpos, sortpos - lsortpos, &ly0, &ly1, twofitweight);
                        int ret1 = fit_line(fits, look->sorted_index, sortpos, hsortpos - sortpos, &hy0, &hy1, twofitweight);

                        if (ret0) {
                            ly0 = ly;
                            ly1 = hy0;
                        }
                        if (ret1) {
                            hy0 = ly1;
                            hy1 = hy;
                        }

                        if (ret0 && ret1) {
                            fit_valueA[i] = -200;
                            fit_valueB[i] = -200;
                        } else {
                            fit_valueB[ln] = ly0;
                            if (ln == 0) fit_valueA[ln] = ly0;
                            fit_valueA[i] = ly1;
                            fit_valueB[i] = hy0;
                            fit_valueA[hn] = hy1;
                            if (hn == 1) fit_valueB[hn] = hy1;

                            if (ly1 >= 0 || hy0 >= 0) {
                                for (j = sortpos - 1; j >= 0; j--)
                                    if (hineighbor[j] == hn) hineighbor[j] = i;
                                    else break;
                                for (j = sortpos + 1; j < posts; j++)
                                    if (loneighbor[j] == ln) loneighbor[j] = i;
                                    else break;
                            }
                        }
                    } else {
                        fit_valueA[i] = -200;
                        fit_valueB[i] = -200;
                    }
                }
            }
        }

        int *output = b.postT;
        int o0 = post_Y(fit_valueA, fit_valueB, 0);
        int o1 = post_Y(fit_valueA, fit_valueB, 1);
        T(output, 0) = o0;
        T(output, 1) = o1;

        for (i = 2; i < posts; i++) {
            int ln = look->loneighbor[i - 2];
            int hn = look->hineighbor[i - 2];
            int x0 = info->postlist[ln];
            int x1 = info->postlist[hn];
            int y0 = T(output, ln);
            int y1 = T(output, hn);

            int predicted = render_point(x0, x1, y0, y1, info->postlist[i]);
            int vx = post_Y(fit_valueA, fit_valueB, i);

            if (vx >= 0 && predicted != vx) {
                T(output, i) = vx;
            } else {
                T(output, i) = predicted | 0x8000;
            }
        }
    }
    b.post_valid[lane] = 1;
}

// ---------------------------------------------------------------------------------------------
// floor1_fit, cooperative form: FG = 16 lanes per channel-block, four blocks per wavefront, one wavefront per
// workgroup.  What the scalar source does bin by bin is spread over the lanes wherever it is integer arithmetic
// (order free), and kept in the source's order wherever it rounds:
//   accumulate_fit   (lib/floor1.c:427-475) the ten integer sums of every post segment: a lane takes the row eight bins
//                    (one 16-byte load) at a time, sums what falls into one segment in registers and adds it to the
//                    segment's entry in LDS (ds_add: integer, any order).  A post's own bin counts for the segment on
//                    either side, as in the source's [x0, x1] ranges.
//   fit_line         (:477-535) five double-precision accumulations over the segments of a range, each in segment order
//                    (they round): one lane per accumulator, the two fits of a split side by side in the two halves of
//                    the lane group; the 5 x 2 sums then go to every lane, which solves redundantly.
//   inspect_error    (:537-586) the line's y at bin x is closed form (after k steps the walk has taken floor(k ady / adx)
//                    long steps), mse is an integer sum and the over / under tests are integer compares: a lane enters the
//                    line at the start of an eight-bin piece with one division and steps it from there; sums and flags
//                    meet by shuffles.
//   greedy split     (:646-719) serial over the posts, state (fit values, neighbours, memo) in LDS as 16-bit words:
//                    ~1.4 KB per block with the sums — 5.6 KB per wavefront for the 29-post floor of long blocks (the
//                    lane-per-block kernel kept 3.9 KB per LANE in scratch memory, 250 KB per wavefront).
//   output           (:723-748) a post's value needs those of its two static neighbours, which have lower post numbers:
//                    lanes take posts, passes repeat until nothing changes (as many as the neighbour tree is deep).
#define FG 16
struct fitgrp {
    int *sums;                  // [(posts - 1) * 10]
    short *A, *B, *lo, *hi, *memo;   // [posts] each
    // the floor look's tables, copied once (the greedy loop reads them in a chain of dependent loads: from global
    // memory every one of them is a round trip to L2)
    short *sorted, *reverse, *post, *slo, *shi;   // sorted_index, reverse_index, postlist [posts]; static lo / hi neighbours [posts - 2]
};
#define FIT_SHORTS 10           /* 16-bit arrays of `pmax` entries per group */

__device__ __forceinline__ int grp_or(int v)
{
    v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4); v |= __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ int grp_add(int v)
{
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ int grp_max(int v)
{
    v = max(v, __shfl_xor(v, 1)); v = max(v, __shfl_xor(v, 2)); v = max(v, __shfl_xor(v, 4)); v = max(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ int grp_min(int v)
{
    v = min(v, __shfl_xor(v, 1)); v = min(v, __shfl_xor(v, 2)); v = min(v, __shfl_xor(v, 4)); v = min(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ void fit_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Two fit_line calls side by side: half h = (l >> 3) of the lane group fits segments [seg0[h], seg0[h] + nfit[h]) with the
// end values y0[h], y1[h] (in: known value or < 0; out: the fit).  ret[h] as the source's return value.  Lane r = l & 7
// < 5 of a half owns accumulator r (xb, yb, x2b, xyb, bn); nfit[h] = 0: the half idles.  All arguments group-uniform.
__device__ __forceinline__ void fit_line2(const fitgrp &G, const short *__restrict__ sorted_index, const int l, const int gbase,
                                          const int seg0a, const int nfita, const int seg0b, const int nfitb, int &ya0, int &ya1,
                                          int &yb0, int &yb1, int &reta, int &retb, const float twofitweight)
{
    const int h = l >> 3, r = l & 7;
    const int seg0 = h ? seg0b : seg0a, nfit = h ? nfitb : nfita;
    const int y0 = h ? yb0 : ya0, y1 = h ? yb1 : ya1;
    const int x0 = sorted_index[seg0], x1 = sorted_index[seg0 + nfit];
    double acc = 0.;
    if (r < 5) {
        // (the segment's addend of accumulator r, b + a * weight, was formed once after accumulate_fit: see fit_terms)
        const double *T = reinterpret_cast<const double *>(G.sums) + r;
        for (int i = seg0; i < seg0 + nfit; i++) acc += T[i * 5];
        if (y0 >= 0) acc += (r == 0) ? x0 : (r == 1) ? y0 : (r == 2) ? x0 * x0 : (r == 3) ? y0 * x0 : 1;
        if (y1 >= 0) acc += (r == 0) ? x1 : (r == 1) ? y1 : (r == 2) ? x1 * x1 : (r == 3) ? y1 * x1 : 1;
    }
    // the five sums of this half to all of its lanes
    const int src = gbase + (h << 3);
    const double xb = __shfl(acc, src), yb = __shfl(acc, src + 1), x2b = __shfl(acc, src + 2), xyb = __shfl(acc, src + 3),
                 bn = __shfl(acc, src + 4);
    int o0 = 0, o1 = 0, ret = 1;
    {
        const double denom = (bn * x2b - xb * xb);
        if (denom > 0.) {
            const double aa = (yb * x2b - xyb * xb) / denom;
            const double bb = (bn * xyb - xb * yb) / denom;
            o0 = (int)rint(aa + bb * x0);
            o1 = (int)rint(aa + bb * x1);
            if (o0 > 1023) o0 = 1023;
            if (o1 > 1023) o1 = 1023;
            if (o0 < 0) o0 = 0;
            if (o1 < 0) o1 = 0;
            ret = 0;
        }
    }
    // results of both halves to every lane of the group (lane 0 of each half is as good as any)
    ya0 = __shfl(o0, gbase); ya1 = __shfl(o1, gbase); reta = __shfl(ret, gbase);
    yb0 = __shfl(o0, gbase + 8); yb1 = __shfl(o1, gbase + 8); retb = __shfl(ret, gbase + 8);
}

__global__ __launch_bounds__(64) void k_floor_fit_coop(vbm_batch b, const int pmax, const int phases)   // phases: timing experiments (31 = all)
{
    extern __shared__ int fitc_lds[];
    const int lane64 = (int)threadIdx.x, g = lane64 >> 4, l = lane64 & 15, gbase = g << 4;
    const int ncb = vbm_ncb(b);
    if ((int)blockIdx.x * 4 >= ncb) return;       // (the launch covers the batch's bound; the count lives on the device)
    const int blk = (int)blockIdx.x * 4 + g;
    const bool live = blk < ncb;
    const int lane = live ? blk : 0;              // (idle groups of the last wavefront shadow block 0 and write nothing: shuffles stay whole-wave)
    const size_t tb = (size_t)(lane >> 6) * b.slab_words + (lane & 63);
    const vbm_setup *s = b.setup;
    const int c = lane % b.ch;
    const vbm_map *map = &s->map[b.W];
    const vbm_floor *look = &s->floor[map->floorsubmap[map->chmuxlist[c]]];
    const int n = look->n;
    const int posts = look->posts;
    const int nseg = posts - 1;
    const float maxover = look->maxover, maxunder = look->maxunder, maxerr = look->maxerr;
    const float twofitweight = look->twofitweight;
    const int imaxover = (int)floorf(maxover), imaxunder = (int)floorf(maxunder);   // both >= 0 in every floor template
    const uint16_t *__restrict__ row = b.qf_bm + (size_t)lane * b.n;     // 16-byte aligned: n is a multiple of 8

    // per-group LDS: sums, then ten 16-bit arrays
    const int per_grp = ((pmax - 1) * 10 + ((FIT_SHORTS * pmax + 1) >> 1) + 1) & ~1;  // ints, even: the sums become doubles
    fitgrp G;
    G.sums = fitc_lds + g * per_grp;
    G.A = (short *)(G.sums + (pmax - 1) * 10);
    G.B = G.A + pmax; G.lo = G.B + pmax; G.hi = G.lo + pmax; G.memo = G.hi + pmax;
    G.sorted = G.memo + pmax; G.reverse = G.sorted + pmax; G.post = G.reverse + pmax; G.slo = G.post + pmax; G.shi = G.slo + pmax;

    for (int k = l; k < nseg * 10; k += FG) G.sums[k] = 0;
    for (int k = l; k < posts; k += FG) {
        G.A[k] = -200; G.B[k] = -200; G.lo[k] = 0; G.hi[k] = 1; G.memo[k] = -1;
        G.sorted[k] = (short)look->sorted_index[k]; G.reverse[k] = (short)look->reverse_index[k]; G.post[k] = (short)look->postlist[k];
        if (k < posts - 2) { G.slo[k] = (short)look->loneighbor[k]; G.shi[k] = (short)look->hineighbor[k]; }
    }
    fit_lds_sync();
    const short *__restrict__ sorted_index = G.sorted, *__restrict__ reverse_index = G.reverse, *__restrict__ postlist = G.post;

    // ---- accumulate_fit of every minimal division (lib/floor1.c:625-628)
    int any_a = 0;
    {
        const int lastp = sorted_index[posts - 1];
        const int top = (lastp < n - 1) ? lastp : n - 1;          // last bin any segment holds
        int sp = -1;                                              // posts strictly below the lane's current piece, less one
        if (phases & 1)
        for (int c8 = l; (c8 << 3) <= top; c8 += FG) {
            const int i0 = c8 << 3;
            const uint4 v = *reinterpret_cast<const uint4 *>(row + i0);
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
            // segment the piece starts in: sp = (posts strictly below i0) - 1, so that a post AT i0 takes the generic path
            // (pieces come in rising order: the search goes on from where the last one ended)
            while (sp + 1 < posts && sorted_index[sp + 1] < i0) sp++;
            int nxt = (sp + 1 < posts) ? sorted_index[sp + 1] : 0x7fffffff;
            int xa = 0, ya = 0, x2a = 0, xya = 0, na = 0, xb = 0, yb = 0, x2b = 0, xyb = 0, nb = 0;
            auto flush = [&]() {
                if ((na | nb) && sp >= 0 && sp < nseg) {
                    int *S = G.sums + sp * 10;
                    if (na) { atomicAdd(&S[0], xa); atomicAdd(&S[1], ya); atomicAdd(&S[2], x2a); atomicAdd(&S[3], xya); atomicAdd(&S[4], na); }
                    if (nb) { atomicAdd(&S[5], xb); atomicAdd(&S[6], yb); atomicAdd(&S[7], x2b); atomicAdd(&S[8], xyb); atomicAdd(&S[9], nb); }
                }
                xa = ya = x2a = xya = na = xb = yb = x2b = xyb = nb = 0;
            };
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = i0 + u;
                if (i > top) break;
                const int w = (int)((wd[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                const int q = w & 0x7fff;
                const bool cls = (w & 0x8000) != 0;
                if (q && cls) any_a = 1;
                auto add = [&]() {
                    if (q) {
                        if (cls) { xa += i; ya += q; x2a += i * i; xya += i * q; na++; }
                        else { xb += i; yb += q; x2b += i * i; xyb += i * q; nb++; }
                    }
                };
                if (i == nxt) {          // a post's bin: the end of segment sp and the start of segment sp + 1
                    add();
                    flush();
                    sp++;
                    nxt = (sp + 1 < posts) ? sorted_index[sp + 1] : 0x7fffffff;
                }
                add();
            }
            flush();
        }
    }
    any_a = grp_or(any_a);
    fit_lds_sync();
    // fit_line adds, per segment of its range and in segment order, (double)b + (double)a * weight to each of its five
    // sums (lib/floor1.c:490-499), weight = (float)(bn + an) * twofitweight / (float)(an + 1) + 1: the addend does not
    // depend on the range, so every segment's five addends are formed once, in place of its ten integer sums (same 40
    // bytes), and a fit is a chain of additions
    {
        double t[(VBM_VIF_POSIT + 1 + FG - 1) / FG][5];
#pragma unroll
        for (int u = 0; u < (VBM_VIF_POSIT + 1 + FG - 1) / FG; u++) {
            const int sg = l + u * FG;
            if (sg < nseg) {
                const int *S = G.sums + sg * 10;
                const int aan = S[4], abn = S[9];
                const double weight = (double)((float)(abn + aan) * twofitweight / (float)(aan + 1)) + 1.;
#pragma unroll
                for (int r = 0; r < 5; r++) t[u][r] = S[5 + r] + S[r] * weight;
            }
        }
        fit_lds_sync();
#pragma unroll
        for (int u = 0; u < (VBM_VIF_POSIT + 1 + FG - 1) / FG; u++) {
            const int sg = l + u * FG;
            if (sg < nseg) {
                double *T = reinterpret_cast<double *>(G.sums) + sg * 5;
#pragma unroll
                for (int r = 0; r < 5; r++) T[r] = t[u][r];
            }
        }
        fit_lds_sync();
    }
    if (!any_a) {      // (group-uniform) no bin above the two-fit line anywhere: no floor for this channel
        if (live && l == 0) b.post_valid[lane] = 0;
    }
    // (a group without a floor walks on with its empty sums: the shuffles below want whole wavefronts; it writes nothing)
    const bool work = live && any_a;

    int y0 = -200, y1 = -200, d0 = -200, d1 = -200, r0 = 0, r1 = 0;
    fit_line2(G, sorted_index, l, gbase, 0, posts - 1, 0, 0, y0, y1, d0, d1, r0, r1, twofitweight);
    if (l == 0) { G.A[0] = (short)y0; G.B[0] = (short)y0; G.B[1] = (short)y1; G.A[1] = (short)y1; }
    fit_lds_sync();

    if (phases & 2)
    for (int i = 2; i < posts; i++) {
        const int sortpos = reverse_index[i];
        const int ln = G.lo[sortpos], hn = G.hi[sortpos];
        if (G.memo[ln] == hn) continue;               // (group-uniform)
        const int lsortpos = reverse_index[ln], hsortpos = reverse_index[hn];
        const int lx = postlist[ln], hx = postlist[hn];
        const int ly = post_Y(G.A, G.B, ln), hy = post_Y(G.A, G.B, hn);
        fit_lds_sync();                                // every lane has read memo / A / B
        if (l == 0) G.memo[ln] = (short)hn;

        // inspect_error (lib/floor1.c:537-586)
        int split;
        {
            const int dy = hy - ly, adx = hx - lx;
            const int base = (int)((float)dy / (float)adx);      // |dy| < 2^23: truncates like dy / adx
            const int sy = (dy < 0 ? base - 1 : base + 1);
            const int ady = abs(dy) - abs(base * adx);
            int mse = 0, viol = 0;
            if (l == 0) {
                const int wv = row[lx];
                const int val = wv & 0x7fff;
                mse = (ly - val) * (ly - val);
                if (wv & 0x8000) {
                    if (ly + maxover < val) viol = 1;
                    if (ly - maxunder > val) viol = 1;
                }
            }
            const int xlo = lx + 1, xhi = hx - 1;
            if (xlo <= xhi && (phases & 4)) {
                const int blast = xhi >> 3;
                for (int c8 = (xlo >> 3) + l; c8 <= blast; c8 += FG) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(row + (c8 << 3));
                    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                    const int xs = max(c8 << 3, xlo);
                    // the line at bin xs: k steps after lx, q of them long ones (k ady < 2^23: float division is exact enough
                    // to truncate to the integer quotient, see div_trunc in pack_kernels.hip)
                    const int k = xs - lx;
                    const int t = k * ady;
                    const int q = (int)((float)t / (float)adx);
                    int err = t - q * adx;
                    int y = ly + k * base + q * (sy - base);
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int x = (c8 << 3) + u;
                        if (x >= xs && x <= xhi) {
                            const int wv = (int)((wd[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
                            const int val = wv & 0x7fff;
                            const int d = y - val;
                            mse += d * d;
                            if ((wv & 0x8000) && val && (-d > imaxover || d > imaxunder)) viol = 1;
                            err += ady;
                            if (err >= adx) { err -= adx; y += sy; } else y += base;
                        }
                    }
                }
            }
            mse = grp_add(mse);
            viol = grp_or(viol);
            const int cnt = 1 + ((xlo <= xhi) ? xhi - xlo + 1 : 0);
            if (viol) split = 1;
            else if (maxover * maxover / cnt > maxerr) split = 0;
            else if (maxunder * maxunder / cnt > maxerr) split = 0;
            else if (mse / cnt > maxerr) split = 1;
            else split = 0;
        }

        if (split && (phases & 8)) {
            int ly0 = -200, ly1 = -200, hy0 = -200, hy1 = -200, ret0 = 0, ret1 = 0;
            fit_line2(G, sorted_index, l, gbase, lsortpos, sortpos - lsortpos, sortpos, hsortpos - sortpos, ly0, ly1, hy0, hy1,
                      ret0, ret1, twofitweight);
            if (ret0) { ly0 = ly; ly1 = hy0; }
            if (ret1) { hy0 = ly1; hy1 = hy; }
            if (ret0 && ret1) {
                if (l == 0) { G.A[i] = -200; G.B[i] = -200; }
            } else {
                if (l == 0) {
                    G.B[ln] = (short)ly0;
                    if (ln == 0) G.A[ln] = (short)ly0;
                    G.A[i] = (short)ly1;
                    G.B[i] = (short)hy0;
                    G.A[hn] = (short)hy1;
                    if (hn == 1) G.B[hn] = (short)hy1;
                }
                if (ly1 >= 0 || hy0 >= 0) {
                    // for (j = sortpos - 1; j >= 0; j--) if (hineighbor[j] == hn) hineighbor[j] = i; else break;
                    int brk = -1;
                    for (int j = l; j < sortpos; j += FG)
                        if (G.hi[j] != hn) brk = j;              // (ascending: the last one found is the largest)
                    brk = grp_max(brk);
                    // for (j = sortpos + 1; j < posts; j++) if (loneighbor[j] == ln) loneighbor[j] = i; else break;
                    int stop = posts;
                    for (int j = sortpos + 1 + l; j < posts; j += FG)
                        if (G.lo[j] != ln) { stop = j; break; }
                    stop = grp_min(stop);
                    fit_lds_sync();
                    for (int j = brk + 1 + l; j < sortpos; j += FG) G.hi[j] = (short)i;
                    for (int j = sortpos + 1 + l; j < stop; j += FG) G.lo[j] = (short)i;
                }
            }
        } else {
            if (l == 0) { G.A[i] = -200; G.B[i] = -200; }
        }
        fit_lds_sync();
    }

    // ---- output (lib/floor1.c:723-748): posts 0 and 1 as fitted; post i from its static neighbours' outputs.  `memo`
    //      is free now and holds the outputs (16 bits: value | 0x8000)
    short *out = G.memo;
    fit_lds_sync();
    for (int k = l; k < posts; k += FG) out[k] = (k < 2) ? (short)post_Y(G.A, G.B, k) : (short)0;
    fit_lds_sync();
    if (phases & 16)
    for (int pass = 0; pass < posts; pass++) {
        int nv[(VBM_VIF_POSIT + 2 + FG - 1) / FG];
        int changed = 0;
#pragma unroll
        for (int t = 0; t < (VBM_VIF_POSIT + 2 + FG - 1) / FG; t++) {
            const int i = 2 + l + t * FG;
            nv[t] = 0;
            if (i < posts) {
                const int ln = G.slo[i - 2], hn = G.shi[i - 2];
                // render_point (lib/floor1.c:381-395); err = ady (x - x0) < 2^23: float division truncates like the integer one
                const int px0 = postlist[ln], px1 = postlist[hn], py0 = out[ln] & 0x7fff, py1 = out[hn] & 0x7fff;
                const int pdy = py1 - py0, perr = abs(pdy) * (postlist[i] - px0);
                const int poff = (int)((float)perr / (float)(px1 - px0));
                const int predicted = (pdy < 0) ? py0 - poff : py0 + poff;
                const int vx = post_Y(G.A, G.B, i);
                nv[t] = (vx >= 0 && predicted != vx) ? vx : (predicted | 0x8000);
                if (nv[t] != (int)(unsigned short)out[i]) changed = 1;
            }
        }
        fit_lds_sync();
#pragma unroll
        for (int t = 0; t < (VBM_VIF_POSIT + 2 + FG - 1) / FG; t++) {
            const int i = 2 + l + t * FG;
            if (i < posts) out[i] = (short)nv[t];
        }
        fit_lds_sync();
        if (!__any(changed)) break;
    }
    if (work) {
        int *output = b.postT;
        for (int k = l; k < posts; k += FG) T(output, k) = (int)(unsigned short)out[k];
        if (l == 0) b.post_valid[lane] = 1;
    }
}

// floor1_interpolate_fit for the blobs between the three fitted ones (lib/mapping0.c:1169-1181,
// lib/floor1.c:752-771): 16.16 fixed-point blend; a blob has posts only if both of its ends do, and none
// of the extra blobs has when the first fit (blob PACKETBLOBS/2) found nothing
__global__ void k_floor_interp(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = (size_t)(lane >> 6) * b.slab_words + (lane & 63);
    const vbm_setup *s = b.setup;
    const int c = lane % b.ch;
    const vbm_map *map = &s->map[b.W];
    const int posts = s->floor[map->floorsubmap[map->chmuxlist[c]]].posts;
    const int PR = (VBM_VIF_POSIT + 2) * 64, MID = VBM_PACKETBLOBS / 2, LAST = VBM_PACKETBLOBS - 1;
    int *valid = b.post_valid_blob + lane;
    const size_t L = (size_t)b.L;
    const int vm = valid[(size_t)MID * L];
    const int v0 = vm && valid[0], v14 = vm && valid[(size_t)LAST * L];
    valid[0] = v0;
    valid[(size_t)LAST * L] = v14;
    for (int k = 1; k < LAST; k++) {
        if (k == MID) continue;
        const int lo = k < MID;
        const int ok = lo ? v0 : v14;
        valid[(size_t)k * L] = ok;
        if (!ok) continue;
        const int *A = b.postT_blob + (size_t)(lo ? 0 : MID) * PR, *B = b.postT_blob + (size_t)(lo ? MID : LAST) * PR;
        int *out = b.postT_blob + (size_t)k * PR;
        const int del = (lo ? k : k - MID) * 65536 / MID;
        for (int i = 0; i < posts; i++) {
            const int a = T(A, i), bb = T(B, i);
            int o = ((65536 - del) * (a & 0x7fff) + del * (bb & 0x7fff) + 32768) >> 16;
            if ((a & 0x8000) && (bb & 0x8000)) o |= 0x8000;
            T(out, i) = o;
        }
    }
}

template <bool BLOBS>
__global__ void k_floor_encode(vbm_batch b)
{
    vbm_blob_enter<BLOBS>(b);
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = (size_t)(lane >> 6) * b.slab_words + (lane & 63);
    const vbm_setup *s = b.setup;
    const int c = lane % b.ch;
    const vbm_map *map = &s->map[b.W];
    const vbm_floor *look = &s->floor[map->floorsubmap[map->chmuxlist[c]]];
    const vbm_floor *info = look;
    const int posts = look->posts;
    const int n = b.n;
    int *post = b.postT, *out = b.floor_outT, *ilogmask = b.iworkT;
    int i, j;

    if (!b.post_valid[lane]) {
        b.nonzero[lane] = 0;    // k_floor_render zero-fills ilogmask
        return;
    }

    for (i = 0; i < posts; i++) {
        int pv = T(post, i);
        int val = pv & 0x7fff;
        switch (info->mult) {
        case 1: val >>= 2; break;
        case 2: val >>= 3; break;
        case 3: val /= 12; break;
        case 4: val >>= 4; break;
        }
        T(post, i) = val | (pv & 0x8000);
    }

    T(out, 0) = T(post, 0);
    T(out, 1) = T(post, 1);

    for (i = 2; i < posts; i++) {
        int ln = look->loneighbor[i - 2];
        int hn = look->hineighbor[i - 2];
        int x0 = info->postlist[ln];
        int x1 = info->postlist[hn];
        int y0 = T(post, ln);
        int y1 = T(post, hn);
        int pi = T(post, i);

        int predicted = render_point(x0, x1, y0, y1, info->postlist[i]);

        if ((pi & 0x8000) || (predicted == pi)) {
            T(post, i) = predicted | 0x8000;
            T(out, i) = 0;
        } else {
            int headroom = (look->quant_q - predicted < predicted ? look->quant_q - predicted : predicted);
            int val = pi - predicted;

            if (val < 0)
                if (val < -headroom) val = headroom - val - 1;
                else val = -1 - (val << 1);
            else if (val >= headroom) val = val + headroom;
            else val <<= 1;

            T(out, i) = val;
            T(post, ln) &= 0x7fff;
            T(post, hn) &= 0x7fff;
        }
    }

    b.nonzero[lane] = 1;
    (void)n; (void)ilogmask; (void)j;
}

// The quantised floor rendered exactly as the decoder will (lib/floor1.c:944-967, render_line0 :397-424),
// sliced over the bins: slice blockIdx.y draws bins [r0, r1) of every channel-block.  A line that starts
// before the slice is entered at x = r0 with the state its Bresenham walk has there: after k steps the
// walk has taken floor(k * ady / adx) of the long steps and carries err = (k * ady) mod adx.
template <bool BLOBS>
__global__ void k_floor_render(vbm_batch b, int nchunks)
{
    vbm_blob_enter<BLOBS>(b);
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = (size_t)(lane >> 6) * b.slab_words + (lane & 63);
    const vbm_setup *s = b.setup;
    const int c = lane % b.ch;
    const vbm_map *map = &s->map[b.W];
    const vbm_floor *look = &s->floor[map->floorsubmap[map->chmuxlist[c]]];
    const int posts = look->posts, mult = look->mult;
    const int n = b.n;
    const int r0 = (int)((long)n * blockIdx.y / nchunks), r1 = (int)((long)n * (blockIdx.y + 1) / nchunks);
    const int *post = b.postT;
    int *ilogmask = b.iworkT;
    const int *__restrict__ forward_index = look->forward_index, *__restrict__ postlist = look->postlist;

    if (!b.post_valid[lane]) {
        for (int i = r0; i < r1; i++) T(ilogmask, i) = 0;
        return;
    }
    int hx = 0, lx = 0;
    int ly = T(post, 0) * mult;
    for (int j = 1; j < posts && lx < r1; j++) {
        const int current = forward_index[j];
        const int pc = T(post, current);
        int hy = pc & 0x7fff;
        if (hy == pc) {
            hy *= mult;
            hx = postlist[current];
            int nn = n;
            if (nn > hx) nn = hx;            // the line covers [lx, nn)
            if (nn > r0 && lx < r1) {
                const int dy = hy - ly;
                const int adx = hx - lx;
                const int base = dy / adx;
                const int sy = (dy < 0 ? base - 1 : base + 1);
                const int ady = abs(dy) - abs(base * adx);
                int x = lx, y = ly, err = 0;
                if (x < r0) {                // enter at r0
                    const int k = r0 - lx;
                    const int t = k * ady;   // < 2^31: k < 4096, ady < adx <= 4096
                    const int q = t / adx;
                    err = t - q * adx;
                    y = ly + k * base + q * (sy - base);
                    x = r0;
                }
                const int xe = nn < r1 ? nn : r1;
                if (x < xe) T(ilogmask, x) = y;
                while (++x < xe) {
                    err = err + ady;
                    if (err >= adx) {
                        err -= adx;
                        y += sy;
                    } else {
                        y += base;
                    }
                    T(ilogmask, x) = y;
                }
            }
            lx = hx;
            ly = hy;
        }
    }
    // past the last drawn post (lib/floor1.c:966): the walk above may have stopped early (lx >= r1), in which
    // case hx >= r1 and nothing is left to fill here
    for (int j2 = (hx > r0 ? hx : r0); j2 < r1; j2++) T(ilogmask, j2) = ly;
}

}  // namespace

static inline dim3 grid_for(int lanes) { return dim3((unsigned)((lanes + 63) / 64)); }

extern "C" int vbm_launch_floor_fit(const vbm_batch *b, hipStream_t st)
{
    if (!b->mix_makes_qf)   // otherwise k_mix has written qf_bm already (psy_kernels.hip)
        hipLaunchKernelGGL(k_floor_prep, dim3((unsigned)((b->ncb + 63) / 64), (unsigned)((b->n + 63) / 64)), dim3(256), 0,
                           st, *b);
    // Round 3: the cooperative kernel (16 lanes per block, no scratch memory) takes 0.30 ms alone on 32768 long
    // channel-blocks where the lane-per-block kernel takes 0.60 ms — and issues three times the vector instructions
    // for it (139 M against 47 M per launch, rocprofv3 SQ_INSTS_VALU: uniform work is replicated over a group's lanes,
    // a fit keeps 5 of 16 lanes busy).  The step is bound by instruction issue, not by any one kernel's latency
    // (DESIGN.md 4), so the full-size batch is faster with the lean kernel (per-block step 2.80 ms against 3.00,
    // from PCM 4.80 against 4.94), and the cooperative one serves the small batches, whose chain of kernels is what a
    // stream inside a run of short blocks waits for.  VBM_FLOORFIT_COOP: 0 never, 1 small batches (default), 2 always.
    static const int coop = getenv("VBM_FLOORFIT_COOP") ? atoi(getenv("VBM_FLOORFIT_COOP")) : 1;
    // "Small": up to 8192 channel-blocks (VBM_FLOORFIT_COOP_MAX) — the rounds of a 4096-stream pool of the drop-in shim are
    // that size: 27.3-27.9 k streams at 1x through the reference's entry points against 25.2-25.7 k with the limit at 4096;
    // the from-PCM step of the batched boundary does not notice (its small batches carry the `few` flag anyway).
    static const int coop_max = getenv("VBM_FLOORFIT_COOP_MAX") ? atoi(getenv("VBM_FLOORFIT_COOP_MAX")) : 128 * 64;
    if (coop == 2 || (coop == 1 && (b->few || b->ncb <= coop_max))) {
        const int pmax = b->fit_max_posts;
        const size_t lds = (size_t)4 * (((pmax - 1) * 10 + ((10 * pmax + 1) >> 1) + 1) & ~1) * sizeof(int);
        static const int phases = getenv("VBM_FLOORFIT_PHASES") ? atoi(getenv("VBM_FLOORFIT_PHASES")) : 31;   // timing experiments
        hipLaunchKernelGGL(k_floor_fit_coop, dim3((unsigned)((b->ncb + 3) / 4)), dim3(64), lds, st, *b, pmax, phases);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    static const int lpw = [] {
        const char *e = getenv("VBM_FLOORFIT_LPW");   // tuning knob: lanes per wavefront of the greedy fit
        const int v = e ? atoi(e) : 64;
        return (v < 1 || v > 64) ? 64 : v;
    }();
    size_t lds = (size_t)(b->fit_max_posts - 1) * 10 * 64 * sizeof(int);
    static const int force = getenv("VBM_FLOORFIT_LDS") ? 1 : 0;
    static const bool allowed = hipFuncSetAttribute(reinterpret_cast<const void *>(k_floor_fit<true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    (void)allowed;
    if (lds <= 80 * 1024 && lpw == 64 && (force || (coop == 0 && (b->few || b->ncb <= 64 * 64))) && !getenv("VBM_FLOORFIT_PRIVATE"))
        hipLaunchKernelGGL(k_floor_fit<true>, dim3((unsigned)((b->ncb + 63) / 64)), dim3(64), lds, st, *b, 64);
    else
        hipLaunchKernelGGL(k_floor_fit<false>, dim3((unsigned)((b->ncb + lpw - 1) / lpw)), dim3(64), 0, st, *b, lpw);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_floor_interp(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_floor_interp, grid_for(b->ncb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_floor_encode(const vbm_batch *b, hipStream_t st)
{
    const unsigned nbl = (unsigned)(b->nblobs > 1 ? b->nblobs : 1);      // managed bitrate: a packetblob per blockIdx.z
    if (nbl > 1) hipLaunchKernelGGL(k_floor_encode<true>, dim3((unsigned)((b->ncb + 63) / 64), 1, nbl), dim3(64), 0, st, *b);
    else hipLaunchKernelGGL(k_floor_encode<false>, dim3((unsigned)((b->ncb + 63) / 64)), dim3(64), 0, st, *b);
    int nchunks = b->n >= 1024 ? 8 : b->n >= 256 ? 4 : 2;
    if ((b->few || b->ncb <= 1024) && b->n / 16 > nchunks) nchunks = b->n / 16;   // small batch: latency-bound, finer slices
    if (nbl > 1)
        hipLaunchKernelGGL(k_floor_render<true>, dim3((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks, nbl), dim3(64), 0, st, *b, nchunks);
    else
        hipLaunchKernelGGL(k_floor_render<false>, dim3((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
