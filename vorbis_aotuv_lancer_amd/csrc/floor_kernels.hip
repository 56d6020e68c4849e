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

__global__ void k_floor_encode(vbm_batch b)
{
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
__global__ void k_floor_render(vbm_batch b, int nchunks)
{
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
    static const int lpw = [] {
        const char *e = getenv("VBM_FLOORFIT_LPW");   // tuning knob: lanes per wavefront of the greedy fit
        const int v = e ? atoi(e) : 64;
        return (v < 1 || v > 64) ? 64 : v;
    }();
    if (!b->mix_makes_qf)   // otherwise k_mix has written qf_bm already (psy_kernels.hip)
        hipLaunchKernelGGL(k_floor_prep, dim3((unsigned)((b->ncb + 63) / 64), (unsigned)((b->n + 63) / 64)), dim3(256), 0,
                           st, *b);
    // The segment sums in LDS for small batches: the greedy loop's fit_line calls then read them in tens of cycles
    // (fit alone 0.59 -> 0.35 ms).  Not for a full batch: 70 KB per wavefront on every CU keeps the LDS-staged
    // kernels of the other half of the pipeline (MDCT, couple, residue VQ) off the chip while the fit runs —
    // measured 3.91 ms per step against 3.76 (VBM_FLOORFIT_LDS=1 forces it, VBM_FLOORFIT_PRIVATE=1 forbids it).
    // (A variant with 8 / 16 / 32 lanes per block — bin loops split over the lanes, row, addends and fit state in
    // LDS — was built and measured in round 2: 0.48 ms alone against 0.58, but 3.3 ms per step against 2.87 in the
    // pipeline, where its 18 KB of LDS per wavefront keep the other half's kernels off the CUs; the per-block chain
    // of ~27 greedy steps, each a handful of dependent reads and a double-precision solve, is what bounds either.)
    size_t lds = (size_t)(b->fit_max_posts - 1) * 10 * 64 * sizeof(int);
    static const int force = getenv("VBM_FLOORFIT_LDS") ? 1 : 0;
    static const bool allowed = hipFuncSetAttribute(reinterpret_cast<const void *>(k_floor_fit<true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    (void)allowed;
    if (lds <= 80 * 1024 && lpw == 64 && (force || b->few || b->ncb <= 64 * 64) && !getenv("VBM_FLOORFIT_PRIVATE"))
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
    hipLaunchKernelGGL(k_floor_encode, grid_for(b->ncb), dim3(64), 0, st, *b);
    int nchunks = b->n >= 1024 ? 8 : b->n >= 256 ? 4 : 2;
    if ((b->few || b->ncb <= 1024) && b->n / 16 > nchunks) nchunks = b->n / 16;   // small batch: latency-bound, finer slices
    hipLaunchKernelGGL(k_floor_render, dim3((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
