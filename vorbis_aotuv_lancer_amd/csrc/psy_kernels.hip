// Psychoacoustic passes of the per-block encode path for gfx950 — one lane per channel-block,
// 64 independent blocks per wavefront, bin-major buffers (batch.h).
//
// Replaces, for a homogeneous batch of blocks (reference = scalar C path):
//   k_prologue    ampmax tracking of vorbis_analysis_blockout (lib/block.c:649-651,
//                 _vp_ampmax_decay lib/psy.c:4504-4515), global_ampmax of mapping0_forward
//                 (lib/mapping0.c:752, 889-901), _postnoise_detection (lib/psy.c:619-648)
//   k_noisemask   logmdct (lib/mapping0.c:936-950), lb_loudnoise_fix (lib/psy.c:5152-5180),
//                 _vp_noisemask (lib/psy.c:3770-4074) = bark_noise_hybridmp x2 (:3480-3638),
//                 ntfix (:3645-3768), compander, M2 post-echo, M8, M9
//   (_vp_tonemask lives in tone_kernels.hip: a workgroup per group of blocks, seeds in LDS)
//   k_mix         _vp_offset_and_mix (lib/psy.c:4274-4502, set_m3p :4148-4272): offset_select 1 for VBR,
//                 1 / 2 / 0 with bit_managed for managed bitrate, including the aoTuV carried buffers
//                 lastmdct / tempmdct
// The order-bound float accumulations (the five prefix sums of bark_noise_hybridmp, the
// partition sums of M8) stay serial per lane, exactly in source order; table-driven loop
// bounds are identical in every lane, so the wave runs them in lockstep and table reads are
// wave-uniform.  Compiled with -ffp-contract=off; `double` where the C source promotes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "batch.h"
#include "kernels.h"

#define T(buf, i) (buf)[tb + (size_t)(i) * 64]   /* tiled bin-major: batch.h */
#define TB(b, lane) ((size_t)((lane) >> 6) * (b).slab_words + ((lane) & 63))
#define NEGINF -9999.f
#define VMIN(x, y) ((x) > (y) ? (y) : (x))
#define VMAX(x, y) ((x) < (y) ? (y) : (x))

namespace {

// NOTE on setup reads inside loops: the setup lives in global memory behind b.setup, and the compiler
// must assume that the kernels' own global stores may alias it, so `p->field` inside a loop is
// re-loaded (scalar load + wait) every iteration.  Loop-invariant fields and table pointers are
// therefore copied into locals before the hot loops.
__device__ __forceinline__ const vbm_psy *psy_of(const vbm_batch &b)
{
    // psy_look = b->psy + blocktype + (W ? 2 : 0)   (lib/mapping0.c:764) == psy[block_mode]
    return &b.setup->psy[b.block_mode];
}

// ---------------------------------------------------------------------------------------------
__global__ void k_prologue(vbm_batch b)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= vbm_nsb(b)) return;
    const vbm_setup *s = b.setup;
    const int sid = b.stream_id[sb];
    // vorbis_analysis_blockout, lib/block.c:649-651
    float g = b.st.g_ampmax[sid];
    float vbi = b.st.vbi_ampmax[sid];
    if (vbi > g) g = vbi;
    {
        int nn = s->blocksizes[b.W] / 2;
        float secs = (float)nn / s->rate;
        g += secs * s->ampmax_att_per_sec;
        if (g < -9999) g = -9999;
    }
    b.st.g_ampmax[sid] = g;
    // mapping0_forward: global_ampmax = vbi->ampmax, raised by every channel's local maximum
    float global_ampmax = g;
    for (int c = 0; c < b.ch; c++) {
        float la = b.local_ampmax[sb * b.ch + c];
        if (la > global_ampmax) global_ampmax = la;
    }
    b.global_ampmax[sb] = global_ampmax;
    b.st.vbi_ampmax[sid] = global_ampmax;   // lib/mapping0.c:1183

    // _postnoise_detection on the un-windowed block (only trans. blocks after an impulse block)
    const int lw_mode = b.st.lW_block_mode[sid];
    for (int c = 0; c < b.ch; c++) {
        float ret = -1.0f;
        if (b.block_mode == 2 && lw_mode == 0 && b.N >= 2048) {
            const float *pcm = b.pcm + (size_t)(sb * b.ch + c) * b.N;
            int sn = b.N >> 2, mn = sn + sn, en = sn + (b.N >> 1);
            double upt = 0, unt = 0;
            // 16 samples per step as four 16-byte loads (rows and quarter points are 16-byte aligned), summed
            // in sample order
            for (int i = sn; i < mn; i += 16) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(pcm + i + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    upt += fabs((double)v[u].x); upt += fabs((double)v[u].y); upt += fabs((double)v[u].z); upt += fabs((double)v[u].w);
                }
            }
            for (int i = mn; i < en; i += 16) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4 *>(pcm + i + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    unt += fabs((double)v[u].x); unt += fabs((double)v[u].y); unt += fabs((double)v[u].z); unt += fabs((double)v[u].w);
                }
            }
            if (!(unt / sn > 0.01)) {
                upt *= upt;
                unt *= unt;
                unt *= 15;
                if (upt > unt) {
                    ret = (float)(upt - unt);
                    if ((double)ret < 0.1) ret = -1.0f;
                }
            }
        }
        b.poste[sb * b.ch + c] = ret;
    }
}

// ---------------------------------------------------------------------------------------------
// _vp_noisemask is split into launches so that only what the source's float arithmetic forces to
// be serial stays serial:
//   k_nm_prefix<PASS>  the five running sums N, X, XX, Y, XY of bark_noise_hybridmp
//                      (lib/psy.c:3480-3541).  Each sum is an order-bound chain over the bins, but
//                      the five chains are independent of one another: blockIdx.y picks the chain,
//                      so a tile of 64 channel-blocks is walked by five wavefronts.  PASS 1 also
//                      produces logmdct (lib/mapping0.c:936-950) and lb_loudnoise_fix.
//   k_nm_solve<PASS>   the per-bin regression solve (lib/psy.c:3543-3636): every bin only reads the
//                      finished sums, so bins are sliced over blockIdx.y.
//   k_nm_ntfix         aoTuV M7 (short and transition blocks only), serial over <= 256 bins.
//   k_nm_post          compander, M2 post-echo, M8, M9: independent per normal-partition, sliced
//                      over blockIdx.y in units of partitions.
#define HY_PF 16
static_assert(HY_PF == 16, "k_nm_prefix keeps the last row of every HY_PF batch as the checkpoint");
#define NM_C 16     // checkpoint interval of the five running sums (rows NM_C-1, 2*NM_C-1, ... are stored)

struct hy_bounds { int i1, i2, f1, f2; };

// phase limits of the solve loops: table-only conditions, evaluated by the host (setup_host.cpp)
__device__ __forceinline__ hy_bounds hybrid_bounds(const vbm_psy *p, int fixed)
{
    hy_bounds h;
    h.i1 = p->hy_i1; h.i2 = p->hy_i2;
    h.f1 = (fixed > 0) ? p->hy_f1 : 0;
    h.f2 = (fixed > 0) ? p->hy_f2 : 0;
    return h;
}

template <int PASS, int CHAIN>
__device__ __forceinline__ void prefix_chain(const vbm_batch &b, const vbm_psy *p, const size_t tb, const int sid, const int col)
{
    const int n = p->n;
    const float offset = (PASS == 1) ? 140.f : 0.f;
    const float *__restrict__ src = (PASS == 1) ? b.mdctT : b.workT;
    // the five sums of a bin sit next to each other: sumT[(bin * 5 + chain)][64] (the solve reads all five of a
    // window edge at once: one 1280-byte run instead of five rows 256 KB apart)
    float *__restrict__ dst = b.sumT + (size_t)CHAIN * 64;
    float *__restrict__ logmdct = b.logmdctT;
    float acc = 0.f, x = 0.f;
    double hi_th = 0;
    const int n25p = p->n25p, n75p = p->n75p;
    const int rb = p->hy_rb;

    // one bin: logmdct (pass 1), the term of this chain, the running sum
    auto bin = [&](float v, const int k) {
        if (PASS == 1) {
            v = (float)((double)vbm_todB(v) + .345);   // logmdct, lib/mapping0.c:936
            if (CHAIN == 0) {
                T(logmdct, k) = v;
                if (k >= n25p && k < n75p) hi_th += (v > -130) ? (double)v : -130.;
            }
        }
        float y = v + offset;
        if (y < 1.f) y = 1.f;
        float w = y * y;
        if (k == 0) {
            // first element, lib/psy.c:3497-3507: half weight, x = 0 (X takes w, XX and XY nothing)
            w = (float)((double)w * .5);
            if (CHAIN == 0 || CHAIN == 1) acc += w;
            if (CHAIN == 3) acc += w * y;
        } else {
            if (CHAIN == 0) acc += w;
            if (CHAIN == 1) acc += w * x;
            if (CHAIN == 2) acc += w * x * x;
            if (CHAIN == 3) acc += w * y;
            if (CHAIN == 4) acc += w * x * y;
        }
        x += 1.f;
    };
    // rows below rb (a multiple of HY_PF): every row is kept, for the mirrored window edges
    int i = 0;
    for (; i < rb && i < n; i += HY_PF) {
        float fv[HY_PF];
#pragma unroll
        for (int u = 0; u < HY_PF; u++) fv[u] = T(src, (i + u < n) ? i + u : n - 1);   // unconditional, clamped
#pragma unroll
        for (int u = 0; u < HY_PF; u++)
            if (i + u < n) {
                bin(fv[u], i + u);
                T(dst, (i + u) * 5) = acc;
            }
    }
    // the rest: only the last row of a batch is kept, as a checkpoint the solve restarts its running sums from
    // (k_nm_solve).  Loads and stores retire in order, so the next batch's loads are issued BEFORE this batch's
    // stores (a fixed number of them): the wait for the loads then does not include the stores.
    if (i < n) {
        float cur[HY_PF];
#pragma unroll
        for (int u = 0; u < HY_PF; u++) cur[u] = T(src, (i + u < n) ? i + u : n - 1);
        for (; i < n; i += HY_PF) {
            float nxt[HY_PF];
            const int in = (i + HY_PF < n) ? i + HY_PF : i;
#pragma unroll
            for (int u = 0; u < HY_PF; u++) nxt[u] = T(src, (in + u < n) ? in + u : n - 1);
#pragma unroll
            for (int u = 0; u < HY_PF; u++)
                if (i + u < n) bin(cur[u], i + u);
            T(dst, (((i + HY_PF < n) ? i + HY_PF : n) - 1) * 5) = acc;
#pragma unroll
            for (int u = 0; u < HY_PF; u++) cur[u] = nxt[u];
        }
    }

    if (PASS == 1 && CHAIN == 0) {
        // lb_loudnoise_fix (lib/psy.c:5152-5180)
        float noise_compand_level = b.st.lowcomp[col];
        const int lW_block_mode = b.st.lW_block_mode[sid];
        if (p->m_val < 0.5) noise_compand_level = -1;
        else if (p->normal_thresh > .45) noise_compand_level = -1;
        else if ((b.block_mode == 2 && lW_block_mode == 3) || (b.block_mode == 3 && lW_block_mode == 2)) {
            hi_th /= n;
            if (hi_th > -40.) noise_compand_level = -1;
            else if (hi_th < -50.) noise_compand_level = 1.f;
            else noise_compand_level = (float)(1. - ((hi_th + 50) / 10));
        }
        b.st.lowcomp[col] = noise_compand_level;
    }
}

template <int PASS>
__global__ void k_nm_prefix(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = TB(b, lane);
    const vbm_psy *p = psy_of(b);
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    switch (blockIdx.y) {
    case 0: prefix_chain<PASS, 0>(b, p, tb, sid, col); break;
    case 1: prefix_chain<PASS, 1>(b, p, tb, sid, col); break;
    case 2: prefix_chain<PASS, 2>(b, p, tb, sid, col); break;
    case 3: prefix_chain<PASS, 3>(b, p, tb, sid, col); break;
    default: prefix_chain<PASS, 4>(b, p, tb, sid, col); break;
    }
}

// The solve needs the five running sums at the two edges of every bin's window.  The edges only move
// forward with the bin index, so instead of reading stored sums (five floats per edge per bin, 20 loads per
// bin in pass 2: 4.4 GB per step through HBM) each edge keeps ITS OWN running sums in registers and adds the
// terms of the bins it passes — the same additions in the same order as k_nm_prefix, hence the same bits.
// A slice of bins starts its edges from the nearest stored checkpoint row below (every NM_C-th row); rows
// below hy_rb are all stored, for the mirrored edges (-lo) of the first bins, which move backwards.
struct acc5 { float n, x, xx, y, xy; int pos; };   // sums after bin `pos`
struct hy_abd { float A, B, D; };

__device__ __forceinline__ void acc_load(acc5 &a, const float *__restrict__ sum, const size_t tb, const int k)
{
    const float *__restrict__ R = sum + tb + (size_t)k * 5 * 64;
    a.n = R[0]; a.x = R[64]; a.xx = R[128]; a.y = R[192]; a.xy = R[256];
    a.pos = k;
}

// sums after bin `target` (lib/psy.c:3509-3541 for the terms; k >= 1 here: row 0 is always stored)
template <int PASS>
__device__ __forceinline__ void acc_seek(acc5 &a, const int target, const float *__restrict__ src,
                                         const float *__restrict__ sum, const size_t tb, const int rb)
{
    const float offset = (PASS == 1) ? 140.f : 0.f;
    if (target < rb) { acc_load(a, sum, tb, target); return; }
    if (a.pos < 0 || target < a.pos || target - a.pos >= 2 * NM_C) acc_load(a, sum, tb, ((target + 1) & ~(NM_C - 1)) - 1);
    while (a.pos < target) {
        const int k = ++a.pos;
        float y = src[tb + (size_t)k * 64] + offset;
        if (y < 1.f) y = 1.f;
        const float w = y * y, xk = (float)k;
        a.n += w;
        a.x += w * xk;
        a.xx += w * xk * xk;
        a.y += w * y;
        a.xy += w * xk * y;
    }
}

// window sums and the regression terms of one bin (lib/psy.c:3549-3560 mirrored, :3571-3582 plain)
__device__ __forceinline__ hy_abd hybrid_abd(const acc5 &H, const acc5 &L, const bool mirror)
{
    float tN, tX, tXX, tY, tXY;
    if (mirror) {
        tN = H.n + L.n; tX = H.x - L.x; tXX = H.xx + L.xx; tY = H.y + L.y; tXY = H.xy - L.xy;
    } else {
        tN = H.n - L.n; tX = H.x - L.x; tXX = H.xx - L.xx; tY = H.y - L.y; tXY = H.xy - L.xy;
    }
    hy_abd r;
    r.A = tY * tXX - tX * tXY;
    r.B = tN * tXY - tX * tY;
    r.D = tN * tXX - tX * tX;
    return r;
}

// one window: edges hi and lo (mirror: the lower edge is the stored row -lo)
template <int PASS>
__device__ __forceinline__ hy_abd hybrid_window(acc5 &H, acc5 &L, const int lo, const int hi, const bool mirror,
                                                const float *__restrict__ src, const float *__restrict__ sum,
                                                const size_t tb, const int rb)
{
    acc_seek<PASS>(H, hi, src, sum, tb, rb);
    if (mirror) {
        acc5 M;
        acc_load(M, sum, tb, -lo);
        return hybrid_abd(H, M, true);
    }
    acc_seek<PASS>(L, lo, src, sum, tb, rb);
    return hybrid_abd(H, L, false);
}

template <int PASS>
__global__ void k_nm_solve(vbm_batch b, int nchunks)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = TB(b, lane);
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int c0 = (int)((long)n * blockIdx.y / nchunks), c1 = (int)((long)n * (blockIdx.y + 1) / nchunks);
    const float offset = (PASS == 1) ? 140.f : 0.f;
    const int fixed = (PASS == 1) ? -1 : p->noisewindowfixed;
    const hy_bounds h = hybrid_bounds(p, fixed);
    const int *__restrict__ bark_lo = p->bark_lo, *__restrict__ bark_hi = p->bark_hi;   // hoisted: see note at psy_of
    const int rb = p->hy_rb;
    const float *__restrict__ sum = b.sumT;
    float *__restrict__ noise = b.noiseT;
    float *__restrict__ work = b.workT;
    // pass 2 reads `work` along its window edges (bins of other slices), so its own result
    // logmdct - work (lib/psy.c:3812) goes to a second array: logmaskT is free until _vp_offset_and_mix
    float *__restrict__ work2 = b.logmaskT;
    const float *__restrict__ logmdct = b.logmdctT;
    const float *__restrict__ src = (PASS == 1) ? b.logmdctT : b.workT;   // what k_nm_prefix<PASS> summed

    // the tail loops (lib/psy.c:3587-3591, :3631-3635) keep A, B, D of the last bin solved before them
    hy_abd tail; tail.A = 0.f; tail.B = 0.f; tail.D = 1.f;
    if (c1 > h.i2 && h.i2 > 0) {
        const int t = h.i2 - 1;
        acc5 Ht, Lt;
        Ht.pos = Lt.pos = -1;
        tail = hybrid_window<PASS>(Ht, Lt, bark_lo[t], bark_hi[t], t < h.i1, src, sum, tb, rb);
    }
    hy_abd ftail = tail;
    if (fixed > 0 && c1 > h.f2 && h.f2 > 0) {
        const int t = h.f2 - 1, hi = t + fixed / 2, lo = hi - fixed;
        acc5 Ht, Lt;
        Ht.pos = Lt.pos = -1;
        ftail = hybrid_window<PASS>(Ht, Lt, lo, hi, t < h.f1, src, sum, tb, rb);
    }

    acc5 H, L, HF, LF;
    H.pos = L.pos = HF.pos = LF.pos = -1;
    for (int i = c0; i < c1; i++) {
        const float x = (float)i;   // the source's x += 1.f from 0 is exact below 2^24
        hy_abd v = tail;
        if (i < h.i2) v = hybrid_window<PASS>(H, L, bark_lo[i], bark_hi[i], i < h.i1, src, sum, tb, rb);
        float R = (v.A + x * v.B) / v.D;
        if (R < 0.f) R = 0.f;
        float nz = R - offset;
        if (PASS == 1) {
            T(noise, i) = nz;
            T(work, i) = T(logmdct, i) - nz;                    // lib/psy.c:3807
        } else {
            if (fixed > 0) {
                hy_abd w = ftail;
                if (i < h.f2) {
                    const int hi = i + fixed / 2, lo = hi - fixed;
                    w = hybrid_window<PASS>(HF, LF, lo, hi, i < h.f1, src, sum, tb, rb);
                }
                R = (w.A + x * w.B) / w.D;
                if (R - offset < nz) nz = R - offset;
            }
            T(noise, i) = nz;
            T(work2, i) = T(logmdct, i) - T(work, i);           // lib/psy.c:3812
        }
    }
}

// aoTuV M7, lib/psy.c:3645-3768.  temp/inmod: 256-entry per-lane scratch (ntfixT)
__device__ __forceinline__ void ntfix(const vbm_batch &b, const vbm_psy *p, int lane, const float *spectral, float *noise)
{
    const size_t tb = TB(b, lane);
    int i, j, k;
    int n = p->n;
    int nx = p->tonefix_end;
    float *temp = b.ntfixT, *inmod = b.ntfixT + (size_t)256 * 64;
    float limit = fabsf(p->noiseoffset[1][0]);

    if (!nx) return;

    for (i = 0; i < 256; i++) { T(temp, i) = 0.f; T(inmod, i) = 0.f; }

    if (b.block_mode <= 1) {
        const int freq_upc = 3;
        const int freq_unc = 4;
        int nxplus = nx + freq_unc;
        float tolerance = 9.f;
        float strength = .6f;
        if (n == 256) tolerance = 15.f;
        if (nxplus > n) {
            nx = n;
            nxplus = n - freq_unc;
        }

        for (i = 0; i < nxplus; i++) {
            float sp = T(spectral, i);
            if (sp < -70) T(inmod, i) = (float)(-70 + (double)(sp + 70) * .1);
            else T(inmod, i) = sp;
        }
        for (i = freq_unc; i < nx; i++) {
            if ((T(spectral, i) > T(spectral, i - 1)) && (T(spectral, i) > T(spectral, i + 1))) {
                int ps = i - 1;
                int pe = i + 1;
                int upper = i - freq_upc;
                int under = i + freq_unc;
                for (j = ps; j > upper; j--) {
                    if (T(spectral, j + 1) < T(spectral, j)) break;
                    ps = j;
                }
                for (j = pe; j < under; j++) {
                    if (T(spectral, j - 1) < T(spectral, j)) break;
                    pe = j;
                }
                {
                    float ss = T(inmod, i) - T(inmod, ps);
                    ss = VMAX(ss, T(inmod, i) - T(inmod, pe));
                    if (ss > tolerance) {
                        if (T(spectral, i) > T(noise, i)) {
                            ss -= tolerance;
                            ss *= strength;
                        }
                        for (j = ps; j <= pe; j++) {
                            T(temp, j) = VMAX(ss, T(temp, j));
                            if (T(temp, j) < 0) T(temp, j) = 0;
                        }
                    }
                }
                i = pe;
            }
        }
        for (i = freq_unc - 1; i < nx; i++) {
            float test = VMIN(p->ntfix_noiseoffset[i], p->noiseoffset[1][i] + limit);
            if (T(temp, i) > test) T(temp, i) = test;
            T(noise, i) -= T(temp, i);
        }
    } else if (b.block_mode == 2) {
        for (i = 0, k = 0; i < nx; i += 8, k++) {
            double na = 0;
            for (j = 0; j < 8; j++) na += T(noise, i + j);
            na /= 8;
            T(temp, k) = (float)na;
        }
        nx /= 8;
        for (i = 3; i < nx; i++) {
            if ((T(temp, i) > T(temp, i - 1)) && (T(temp, i) > T(temp, i + 1))) {
                int a = 0, bb = 0;
                float thres = 0;
                if (T(temp, i - 1) > T(temp, i - 2)) {
                    thres = T(temp, i - 2);
                    a = i - 3;
                } else {
                    thres = T(temp, i - 1);
                    a = i - 2;
                }
                bb = i + 3;
                thres = T(temp, i) - thres;
                if ((double)thres > 2.) {
                    int eightimes = i * 8;
                    float test = VMIN(p->ntfix_noiseoffset[eightimes], p->noiseoffset[1][eightimes] + limit);
                    thres = VMIN(thres - 2, test);
                    a *= 8;
                    bb *= 8;
                    for (j = a; j <= bb; j++) T(noise, j) -= thres;
                }
            }
        }
    }
}

__global__ void k_nm_ntfix(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    ntfix(b, psy_of(b), lane, b.logmdctT, b.logmaskT);   // pass 2's logmdct - work: see k_nm_solve
}

__global__ void k_nm_post(vbm_batch b, int nchunks)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    const size_t tb = TB(b, lane);
    const vbm_setup *s = b.setup;
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    const int partition = (p->normal_p ? p->normal_partition : 16);
    const int nparts = (n + partition - 1) / partition;
    const int k0 = (int)((long)nparts * blockIdx.y / nchunks), k1 = (int)((long)nparts * (blockIdx.y + 1) / nchunks);
    const int c0 = k0 * partition, c1 = VMIN(k1 * partition, n);
    int i, j, k;

    float *logmdct = b.logmdctT, *logmask = b.noiseT, *work = b.logmaskT /* k_nm_solve<2> */, *epeak = b.epeakT, *npeak = b.npeakT;
    const float noise_compand_level = b.st.lowcomp[col];
    const float *__restrict__ noisecompand = p->noisecompand, *__restrict__ noisecompand_high = p->noisecompand_high;
    const int *__restrict__ stn_compand = s->stn_compand;
    const float *__restrict__ noiseoffset1 = p->noiseoffset[1];
    const int min_nn_lp = p->min_nn_lp;

    // noise compand & aoTuV M5 extension & tone peak, and M9 with it: M9 (lib/psy.c:4058-4072) replaces the tone
    // peak of every bin by a value that depends on that peak, logmdct and lastmdct only — nothing in between
    // reads the intermediate peak — so the final value is stored here and the bins are walked once.  Eight bins'
    // inputs are read before anything is written (loads and stores retire in order).
    {
        const int thter = (noise_compand_level > 0) ? p->n33p : 0;
        const float *lastmdct = b.st.mblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);
        const int m9_end = (b.block_mode > 1) ? p->tonecomp_endp : 0;
        for (i = c0; i < c1; i += 8) {
            float lmk[8], wvv[8], lmd[8], lst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ii = (i + u < c1) ? i + u : c1 - 1;
                lmk[u] = T(logmask, ii);
                wvv[u] = T(work, ii);
                lmd[u] = T(logmdct, ii);
                lst[u] = (ii < m9_end) ? lastmdct[(size_t)ii * 64] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ii = i + u;
                if (ii < c1) {
                    int dB = (int)((double)lmk[u] + .5);
                    if (dB >= VBM_NOISE_COMPAND_LEVELS) dB = VBM_NOISE_COMPAND_LEVELS - 1;
                    if (dB < 0) dB = 0;
                    const float wv = wvv[u];
                    const float ep = wv + stn_compand[dB];
                    if (ii < thter)
                        T(logmask, ii) = wv + noisecompand[dB] -
                                         ((noisecompand[dB] - noisecompand_high[dB]) * noise_compand_level);
                    else
                        T(logmask, ii) = wv + noisecompand[dB];
                    float e = 0.f;
                    if (ii < m9_end) {
                        float temp = lmd[u] - ep;
                        if (temp >= 12.f) {
                            float mi = lmd[u] - lst[u];
                            if (mi >= 1) e = mi;
                        }
                    }
                    T(epeak, ii) = e;
                }
            }
        }
    }

    for (k = k0; k < k1; k++) T(npeak, k) = 0.f;

    // reduction of post-echo (postprocessing of aoTuV M2)
    const float poste = b.poste[lane];
    if (poste > 0) {
        for (k = k0, i = c0; k < k1 && i < min_nn_lp; i += partition, k++) {
            float temp = VMIN(VMIN(poste, 30.f), noiseoffset1[i] + 30.f);
            if (temp <= 0) continue;
            T(npeak, k) = -1.f;
            for (j = 0; j < partition; j++) T(logmask, i + j) -= temp;
        }
    }

    // M8
    for (k = k0, i = c0; k < k1 && i < min_nn_lp; i += partition, k++) {
        const float nt = 4;
        float o = noiseoffset1[i + partition - 1] + 6;
        float me = 0;
        float avge = 0;

        if (o <= 0) continue;
        if ((double)T(npeak, k) < -0.5) continue;

        for (j = 0; j < partition; j++) {
            float temp = T(logmdct, i + j) - T(logmask, i + j);
            if (me < temp) me = temp;
            avge += T(logmdct, i + j);
        }
        if (avge < (-95 * partition)) continue;

        if (me < nt) T(npeak, k) = (VMIN(o, nt - me)) / nt;
    }

}

// (_vp_tonemask: tone_kernels.hip)

// ---------------------------------------------------------------------------------------------
struct mod3 {
    int sw;
    int mdctbuf_flag;
    float noise_rate, noise_rate_low, noise_center, tone_rate;
};

// QF: also leave the floor fit's input word of every bin (floor_kernels.hip: dBquant(logmask) | test << 15) in
// the LDS tile qtile[bin - i0][lane]; the kernel writes the tile out as block-major rows.
// M0 (impulse blocks: one slice, M3 walks tempmdct with dependent read-modify-writes): the lane's tempmdct
// column lives in LDS (tl[row * 64]) for the duration of the kernel.
template <int SEL, bool MANAGED, bool QF, bool M0>
__device__ __forceinline__ void mix_body(const vbm_batch &b, const int nchunks, const int lane, uint16_t (*qtile)[66], float *tl,
                                         const int *bfn_lds)
{
    constexpr bool BUF = (!MANAGED || SEL == 2);   // mp->mdctbuf_flag of set_m3p when the rate is high (lib/psy.c:4165-4173)
    const size_t tb = TB(b, lane);
    const vbm_setup *s = b.setup;
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int i0 = (int)((long)n * blockIdx.y / nchunks), i1 = (int)((long)n * (blockIdx.y + 1) / nchunks);
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    float *lastmdct = b.st.mblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);   // element i at [i*64]
    float *tempmdct = b.st.tblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);
#define LAST(i) lastmdct[(size_t)(i) * 64]
#define TEMP(i) (*(M0 ? &tl[(i) * 64] : &tempmdct[(size_t)(i) * 64]))
    if (M0)
        for (int r = 0; r < 256; r++) tl[r * 64] = tempmdct[(size_t)r * 64];
    const float *noise = b.noiseT, *tone = b.toneT;
    float *logmask = b.logmaskT, *mdct = b.mdctT, *logmdct = b.logmdctT, *npeak = b.npeakT;
    const int offset_select = SEL;
    const int block_mode = b.block_mode;
    const int nW_modenumber = (b.wflags[sb] >> 1) & 1;
    const int lW_block_mode = b.st.lW_block_mode[sid];
    const int lW_no = b.st.lW_no[sid];
    const int impadnum = b.st.impadnum[sid];
    float low_compand = b.st.lowcomp[col];
    const int end_block = s->floor[b.W].info_n;   // vif->n, lib/mapping0.c:1055
    const float twofitatten = s->floor[s->map[b.W].floorsubmap[s->map[b.W].chmuxlist[c]]].twofitatten;

    int i, j, k;
    int hsrate = ((p->rate < 26000) ? 0 : 1);
    int partition = (p->normal_p ? p->normal_partition : 16);
    float m1_de, m1_coeffi;
    float toneatt = p->tone_masteratt[offset_select];

    mod3 mp3;
    mp3.sw = 0; mp3.mdctbuf_flag = 0; mp3.noise_rate = mp3.noise_rate_low = mp3.noise_center = mp3.tone_rate = 0.f;

    int m4_start = p->normal_start;
    int m4_end = p->tonecomp_endp;
    float m4_thres = p->tonecomp_thres;
    int m4_end_block = end_block;

    if (low_compand < 0 || (double)toneatt < 25.) low_compand = 0;
    else low_compand = (float)((double)low_compand * ((double)toneatt - 25.));

    // set_m3p (lib/psy.c:4148-4272)
    if (!hsrate) {
        mp3.sw = 0;
        mp3.mdctbuf_flag = 0;
    } else {
        mp3.mdctbuf_flag = BUF ? 1 : 0;
        if (MANAGED && SEL == 0) {   // high noise scene
            mp3.sw = 0;
        } else if (block_mode) {
            mp3.sw = 0;
        } else if (n == 128 || n == 256) {
            // impulse blocks: the table sits in LDS (M3 PRE reads it in a dependent double loop)
            const int *bfn = M0 ? bfn_lds : ((n == 128) ? s->freq_bfn128 : s->freq_bfn256);
            int count;
            if (n == 128) {
                if (toneatt < 3) count = 2;
                else count = 3;
                if (!lW_block_mode) {
                    if (lW_no < 8) {
                        mp3.noise_rate = (float)(0.7 - (double)((float)(lW_no - 1) / 17));
                        mp3.noise_center = (float)(lW_no * count);
                        mp3.tone_rate = 8 - lW_no;
                    } else {
                        mp3.noise_rate = (float)0.3;
                        mp3.noise_center = 25;
                        mp3.tone_rate = 0;
                        if ((lW_no * count) < 24) mp3.noise_center = lW_no * count;
                    }
                    if (BUF) for (i = 0; i < n; i++) TEMP(i) -= 5;
                } else {
                    mp3.noise_rate = (float)0.7;
                    mp3.noise_center = 0;
                    mp3.tone_rate = 8.f;
                    if (BUF) for (i = 0; i < n; i++) TEMP(i) = LAST(i) - 5;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.125));
                if (BUF) for (i = 0; i < n; i++) {
                    float cell = 75 / (float)bfn[i];
                    for (j = 1; j < bfn[i]; j++) {
                        float freqbuf = T(logmdct, i) - (cell * j);
                        if (TEMP(i + j) < freqbuf) TEMP(i + j) = (float)((double)TEMP(i + j) + (5. / (double)(float)bfn[i + j]));
                    }
                }
            } else {
                if (!lW_block_mode) {
                    count = 6;
                    if (lW_no < 4) {
                        mp3.noise_rate = (float)(0.4 - (double)((float)(lW_no - 1) / 11));
                        mp3.noise_center = (float)(lW_no * count + 12);
                        mp3.tone_rate = 8 - lW_no * 2;
                    } else {
                        mp3.noise_rate = (float)0.2;
                        mp3.noise_center = 30;
                        mp3.tone_rate = 0;
                    }
                    if (BUF) for (i = 0; i < n; i++) TEMP(i) -= 10;
                } else {
                    mp3.noise_rate = (float)0.6;
                    mp3.noise_center = 12;
                    mp3.tone_rate = 8.f;
                    if (BUF) for (i = 0; i < n; i++) TEMP(i) = LAST(i) - 10;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.0625));
                if (BUF) for (i = 0; i < n; i++) {
                    float cell = 75 / (float)bfn[i];
                    for (j = 1; j < bfn[i]; j++) {
                        float freqbuf = T(logmdct, i) - (cell * j);
                        if (TEMP(i + j) < freqbuf) TEMP(i + j) = (float)((double)TEMP(i + j) + (10. / (double)(float)bfn[i + j]));
                    }
                }
            }
        } else {
            mp3.sw = 0;
        }
    }

    // M4 PRE
    m4_end_block += p->normal_partition;
    if (m4_end_block > n) m4_end_block = n;
    if (!hsrate) {
        m4_end = m4_end_block;
    } else {
        if (p->normal_thresh > 1.) m4_start = 9999;
    }

    const float *__restrict__ noiseoffset = p->noiseoffset[offset_select];
    const float noisemaxsupp = p->noisemaxsupp, m_val = p->m_val;
    const int tonecomp_endp = p->tonecomp_endp, m3n0 = p->m3n[0], m3n1 = p->m3n[1], m3n2 = p->m3n[2];
    // Without M3 (every block type but impulse) the bins are independent: eight bins' inputs are read before
    // anything is written, so the loads of a bin do not queue behind the stores of the one before it (loads and
    // stores retire in order) — the kernel is latency-bound, 64 dependent round trips per slice otherwise.
    // M3 SET's plain copies lastmdct[i] = logmdct[i] (lib/psy.c:4463-4501) ride along
    const bool copy_last = mp3.mdctbuf_flag == 1 && !mp3.sw &&
                           ((block_mode <= 1 && !nW_modenumber) || (block_mode == 2 && nW_modenumber) || block_mode == 3);
    if (!mp3.sw) {
        for (i = i0; i < i1; i += 8) {
            float nv[8], tv[8], lv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ii = (i + u < i1) ? i + u : i1 - 1;
                nv[u] = T(noise, ii);
                tv[u] = T(tone, ii);
                lv[u] = T(logmdct, ii);
                if (SEL == 1) mv[u] = T(mdct, ii);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ii = i + u;
                if (ii < i1) {
                    float val = nv[u] + noiseoffset[ii];
                    float tval = tv[u] + toneatt;
                    const float lm = lv[u];
                    if (ii <= m4_start) tval -= low_compand;
                    if (val > noisemaxsupp) val = noisemaxsupp;
                    // M4 MAIN
                    float lmv;
                    if (val > tval) {
                        lmv = val;
                    } else if ((ii > m4_start) && (ii < m4_end)) {
                        if (lm < tval) {
                            if (lm < val) tval -= (tval - val) * m4_thres;
                            else tval = lm;
                        }
                        lmv = tval;
                    } else
                        lmv = tval;
                    T(logmask, ii) = lmv;
                    if (copy_last) LAST(ii) = lm;
                    if (QF) {   // vorbis_dBquant (lib/floor1.c:294) and the two-fit test (lib/floor1.c:452)
                        int q = (int)(lmv * 7.3142857f + 1023.5f);
                        q = q > 1023 ? 1023 : (q < 0 ? 0 : q);
                        qtile[ii - i0][threadIdx.x] = (uint16_t)(q | ((lm + twofitatten >= lmv) ? 0x8000 : 0));
                    }
                    // M1 (offset_select == 1)
                    if (SEL == 1) {
                        m1_coeffi = (float)-17.2;
                        val = val - lm;
                        if (val > m1_coeffi) {
                            m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.005 * (double)m_val));
                            if (m1_de < 0) m1_de = (float)0.0001;
                        } else
                            m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.0003 * (double)m_val));
                        T(mdct, ii) = mv[u] * m1_de;
                    }
                }
            }
        }
    } else
    for (i = i0; i < i1; i++) {
        float val = T(noise, i) + noiseoffset[i];
        float tval = T(tone, i) + toneatt;
        const float lm = T(logmdct, i);
        if (i <= m4_start) tval -= low_compand;
        if (val > noisemaxsupp) val = noisemaxsupp;

        // M3 MAIN
        if (mp3.sw) {
            if (val > tval) {
                const float last = LAST(i);
                if ((val > last) && (lm > (TEMP(i) + mp3.noise_center))) {
                    int toneac = 0;
                    float valmask = 0;
                    float rate_mod;
                    float mainth;

                    if (mp3.mdctbuf_flag == 1) TEMP(i) = lm;
                    if (lm > last) rate_mod = mp3.noise_rate;
                    else rate_mod = mp3.noise_rate_low;
                    if (!impadnum && (i < tonecomp_endp) && ((val - last) > 20.f)) {
                        float dBsub = (lm - last);
                        if (dBsub > 25.f) {
                            toneac = 1;
                            if (tval > -100.f && ((lm - tval) < 48.f)) {
                                float tr_cur = mp3.tone_rate;
                                if (dBsub < 35.f) tr_cur *= ((35.f - dBsub) * .1f);
                                tval -= tr_cur;
                                if (tval < -100.f) tval = -100.f;
                                if ((lm - tval) > 48.f) tval = lm - 48.f;
                            }
                        }
                    }
                    if (i > m3n0) {
                        mainth = 30.f;
                    } else if (i > m3n1) {
                        mainth = 20.f;
                    } else if (i > m3n2) {
                        mainth = 10.f;
                        rate_mod *= .5f;
                    } else {
                        mainth = 10.f;
                        rate_mod *= .3f;
                    }
                    if ((val - tval) > mainth) valmask = ((val - tval - mainth) * .1f + mainth) * rate_mod;
                    else valmask = (val - tval) * rate_mod;

                    if ((val - valmask) > last) val -= valmask;
                    else val = last;

                    if (toneac) {
                        float temp = val - VMAX(last, -140);
                        if (temp > 20.f) val -= (temp - 20.f) * .2f;
                    }

                    if (toneac == 1) T(npeak, i / partition) = -1.f;
                    else if (T(npeak, i / partition) > 0) T(npeak, i / partition) = 0;
                }
            }
        }

        // M4 MAIN
        float lmv;
        if (val > tval) {
            lmv = val;
        } else if ((i > m4_start) && (i < m4_end)) {
            if (lm < tval) {
                if (lm < val) tval -= (tval - val) * m4_thres;
                else tval = lm;
            }
            lmv = tval;
        } else
            lmv = tval;
        T(logmask, i) = lmv;
        if (QF) {   // vorbis_dBquant (lib/floor1.c:294) and the two-fit test (lib/floor1.c:452)
            int q = (int)(lmv * 7.3142857f + 1023.5f);
            q = q > 1023 ? 1023 : (q < 0 ? 0 : q);
            qtile[i - i0][threadIdx.x] = (uint16_t)(q | ((lm + twofitatten >= lmv) ? 0x8000 : 0));
        }

        // M1 (offset_select == 1)
        if (SEL == 1) {
            m1_coeffi = (float)-17.2;
            val = val - lm;
            if (val > m1_coeffi) {
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.005 * (double)m_val));
                if (m1_de < 0) m1_de = (float)0.0001;
            } else
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.0003 * (double)m_val));
            T(mdct, i) *= m1_de;
        }
    }

    // M3 SET lastmdct
    if (mp3.mdctbuf_flag == 1 && !copy_last) {
        const int mag = 8;
        switch (block_mode) {
        case 0:
        case 1:
            if (nW_modenumber) {
                for (i = i0, k = i0 * mag; i < i1; i++, k += mag)
                    for (j = 0; j < mag; j++) LAST(k + j) = T(logmdct, i);
            } else {
                for (i = i0; i < i1; i++) LAST(i) = T(logmdct, i);
            }
            break;
        case 2:
            if (!nW_modenumber) {
                int nsh = n >> 3;
                for (i = i0; i < i1 && i < nsh; i++) {
                    int ni = i * mag;
                    float v = T(logmdct, ni);
                    for (j = 1; j < mag; j++)
                        if (v > T(logmdct, ni + j)) v = T(logmdct, ni + j);
                    LAST(i) = v;
                }
            } else {
                for (i = i0; i < i1; i++) LAST(i) = T(logmdct, i);
            }
            break;
        case 3:
            for (i = i0; i < i1; i++) LAST(i) = T(logmdct, i);
            break;
        default:
            break;
        }
    }
    if (M0 && BUF)
        for (int r = 0; r < 256; r++) tempmdct[(size_t)r * 64] = tl[r * 64];
#undef LAST
#undef TEMP
}

// Bins are independent for every block type except impulse short blocks (M3 reads and rewrites
// tempmdct across bins and updates npeak per partition in bin order), so the launch splits the
// bin range into `nchunks` slices (blockIdx.y) for block modes 1..3 and uses one slice for mode 0.
// QF (slices of at most 64 bins): the slice's floor-fit words go out as block-major rows qf_bm[block][n]
// (what k_floor_prep would compute from logmask and logmdct in a pass of its own).
template <int SEL, bool MANAGED, bool QF, bool M0>
__global__ void k_mix(vbm_batch b, int nchunks)
{
    extern __shared__ float mix_temp[];   // M0: [256][64] tempmdct columns of this wavefront
    __shared__ uint16_t qtile[QF ? 64 : 1][66];
    __shared__ uint8_t colact[64];
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    // managed bitrate: the hi / lo rate passes run only for channels whose first fit exists (lib/mapping0.c:1097)
    const bool active = lane < vbm_ncb(b) && !(MANAGED && SEL != 1 && !b.post_valid_blob[(size_t)(VBM_PACKETBLOBS / 2) * b.L + lane]);
    __shared__ int s_bfn[M0 ? 256 : 1];
    if (M0) {
        const int *g = (b.n == 128) ? b.setup->freq_bfn128 : b.setup->freq_bfn256;
        const int cnt = (b.n == 128) ? 128 : 256;
        for (int t = threadIdx.x; t < cnt; t += 64) s_bfn[t] = g[t];
        __syncthreads();
    }
    if (active) mix_body<SEL, MANAGED, QF, M0>(b, nchunks, lane, qtile, mix_temp + threadIdx.x, s_bfn);
    if (QF) {
        colact[threadIdx.x] = active ? 1 : 0;
        __syncthreads();
        const int n = b.n;
        const int i0 = (int)((long)n * blockIdx.y / nchunks), i1 = (int)((long)n * (blockIdx.y + 1) / nchunks);
        const int k = threadIdx.x, c0 = blockIdx.x * 64;
        if (k < i1 - i0)
            for (int c = 0; c < 64; c++)
                if (colact[c]) b.qf_bm[(size_t)(c0 + c) * n + i0 + k] = qtile[k][c];
    }
}

}  // namespace

static inline dim3 grid_for(int lanes) { return dim3((unsigned)((lanes + 63) / 64)); }
// slices of the bin range for the kernels whose bins are independent (long blocks: 64 bins each)
static inline int bin_chunks(const vbm_batch *b)
{
    static int big = 0;
    if (!big) {
        const char *e = getenv("VBM_BIN_CHUNKS");   // tuning knob: slices of the long-block bin range
        big = e ? atoi(e) : 16;
        if (big < 1 || big > 64) big = 16;
    }
    const int chunks = b->n >= 1024 ? big : b->n >= 512 ? 8 : b->n >= 256 ? 4 : 2;
    // A small batch (the short rounds of the front end: a few wavefronts on an empty chip) is bound by the
    // latency of each wavefront's walk over its bins, not by throughput: slices of 8-16 bins instead of 64.
    if (b->few || b->ncb <= 1024) {
        int fine = b->n / (b->n >= 1024 ? 16 : 8);
        if (fine > 64) fine = 64;
        if (fine > chunks) return fine;
    }
    return chunks;
}

extern "C" int vbm_launch_prologue(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_prologue, grid_for(b->nsb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_noisemask_lds(const vbm_batch *b, hipStream_t st);   // noise_kernels.hip
extern "C" int vbm_launch_noisemask(const vbm_batch *b, hipStream_t st)
{
    static int fused = -1;
    if (fused < 0) fused = getenv("VBM_NOISE_FUSED") ? atoi(getenv("VBM_NOISE_FUSED")) : 1;
    if (fused) return vbm_launch_noisemask_lds(b, st);
    const unsigned tiles = (unsigned)((b->ncb + 63) / 64);
    const int nchunks = bin_chunks(b);
    hipLaunchKernelGGL(k_nm_prefix<1>, dim3(tiles, 5), dim3(64), 0, st, *b);
    hipLaunchKernelGGL(k_nm_solve<1>, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    hipLaunchKernelGGL(k_nm_prefix<2>, dim3(tiles, 5), dim3(64), 0, st, *b);
    hipLaunchKernelGGL(k_nm_solve<2>, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    if (b->block_mode <= 2) hipLaunchKernelGGL(k_nm_ntfix, dim3(tiles), dim3(64), 0, st, *b);
    hipLaunchKernelGGL(k_nm_post, dim3(tiles, (unsigned)nchunks), dim3(64), 0, st, *b, nchunks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
static const size_t kMixTempBytes = (size_t)256 * 64 * sizeof(float);   // impulse blocks: tempmdct columns in LDS

// the impulse-block variants take 64 KB of dynamic LDS on top of their static tiles: above the default limit
template <typename K>
static void allow_big_lds(K kernel)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kMixTempBytes);
}
static void mix_m0_setup()
{
    static bool done = false;
    if (done) return;
    allow_big_lds(k_mix<1, false, false, true>);
    allow_big_lds(k_mix<1, true, false, true>);
    allow_big_lds(k_mix<2, true, false, true>);
    allow_big_lds(k_mix<0, true, false, true>);
    done = true;
}

extern "C" int vbm_launch_mix(const vbm_batch *b, hipStream_t st)
{
    const int nchunks = (b->block_mode == 0) ? 1 : bin_chunks(b);
    const dim3 grid((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks);
    if (b->block_mode == 0) {
        mix_m0_setup();
        hipLaunchKernelGGL((k_mix<1, false, false, true>), grid, dim3(64), kMixTempBytes, st, *b, nchunks);
    }
    else if (b->mix_makes_qf) hipLaunchKernelGGL((k_mix<1, false, true, false>), grid, dim3(64), 0, st, *b, nchunks);
    else hipLaunchKernelGGL((k_mix<1, false, false, false>), grid, dim3(64), 0, st, *b, nchunks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// slices of at most 64 bins fit the 64 x 64 tile k_mix writes the floor-fit words through
extern "C" int vbm_mix_can_make_qf(const vbm_batch *b)
{
    return b->block_mode != 0 && (b->n + bin_chunks(b) - 1) / bin_chunks(b) <= 64;
}
// managed bitrate: offset_select 1 (first fit), 2 (higher rate), 0 (lower rate), lib/mapping0.c:1044-1160
extern "C" int vbm_launch_mix_managed(const vbm_batch *b, int offset_select, hipStream_t st)
{
    const int nchunks = (b->block_mode == 0) ? 1 : bin_chunks(b);
    const dim3 grid((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks);
    if (b->block_mode == 0) {
        mix_m0_setup();
        if (offset_select == 1) hipLaunchKernelGGL((k_mix<1, true, false, true>), grid, dim3(64), kMixTempBytes, st, *b, nchunks);
        else if (offset_select == 2) hipLaunchKernelGGL((k_mix<2, true, false, true>), grid, dim3(64), kMixTempBytes, st, *b, nchunks);
        else hipLaunchKernelGGL((k_mix<0, true, false, true>), grid, dim3(64), kMixTempBytes, st, *b, nchunks);
    } else if (b->mix_makes_qf) {
        if (offset_select == 1) hipLaunchKernelGGL((k_mix<1, true, true, false>), grid, dim3(64), 0, st, *b, nchunks);
        else if (offset_select == 2) hipLaunchKernelGGL((k_mix<2, true, true, false>), grid, dim3(64), 0, st, *b, nchunks);
        else hipLaunchKernelGGL((k_mix<0, true, true, false>), grid, dim3(64), 0, st, *b, nchunks);
    } else {
        if (offset_select == 1) hipLaunchKernelGGL((k_mix<1, true, false, false>), grid, dim3(64), 0, st, *b, nchunks);
        else if (offset_select == 2) hipLaunchKernelGGL((k_mix<2, true, false, false>), grid, dim3(64), 0, st, *b, nchunks);
        else hipLaunchKernelGGL((k_mix<0, true, false, false>), grid, dim3(64), 0, st, *b, nchunks);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
