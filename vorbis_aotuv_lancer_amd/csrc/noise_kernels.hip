// _vp_noisemask and what mapping0_forward does around it (reference lib/mapping0.c:936-950 logmdct,
// lib/psy.c:5152-5180 lb_loudnoise_fix, lib/psy.c:3770-4074 _vp_noisemask = bark_noise_hybridmp x2 (:3480-3638),
// aoTuV M7 ntfix (:3645-3768), noise compander, M2 post-echo reduction, M8, M9) as ONE kernel for gfx950.
//
// A workgroup of 256 threads owns NB consecutive channel-blocks (lanes of one 64-lane tile, batch.h).  The five
// running sums N, X, XX, Y, XY of a block live in LDS (five arrays of n floats: 20 KB for a long block, so several
// workgroups share a CU); logmdct, work and noise of bin i live in the registers of thread i mod 256, for all NB
// blocks.  HBM traffic is the algorithmic one: the block's MDCT row in (block-major, straight from k_window_mdct),
// MDCT (the layout change for k_mix and the couple kernels rides along) / logmdct / logmask / epeak rows and the
// npeak entries out (tiled bin-major: the lane-per-block kernels behind this one read them), the carried lastmdct row
// in for M9.
//
//   terms    every bin's five addends w, w*x, w*x*x, w*y, w*x*y (lib/psy.c:3509-3541) — independent per bin.
//   scan     the running sums themselves are order-bound float chains (the source adds bin after bin; no other
//            association gives the same bits): one lane per (block, sum) walks its n addends in LDS, four per
//            ds_read_b128, thirty-two fetched while the thirty-two before them are added.  5 NB lanes of one wavefront
//            are busy here: the other workgroups of the CU fill the gap with their parallel phases.
//   solve    per bin, the window sums are differences (mirrored: sums) of two finished rows of the five arrays
//            (:3543-3636): a thread takes bin i of every block of the workgroup, so the bark tables are read once.
//   both passes of bark_noise_hybridmp run like this (offset 140 / window "-1", then offset 0 / noisewindowfixed).
//   ntfix    aoTuV M7 (short and transition blocks): a serial walk over <= 256 bins per block, on LDS rows (the sum
//            arrays are free by then).
//   post     compander + M9 per bin, then M2 / M8 per normal-partition (a lane per partition: its sums are
//            order-bound over the partition's bins; the rows are stored skewed by one word per partition, so the
//            lanes hit different banks), then the rows go out.
//
// Workgroups of one tile are made neighbours on the same XCD (its L2 then sees all 64 lanes of a 256-byte row
// within a short time and writes whole lines): blockIdx is remapped so that XCD x takes a contiguous eighth of
// the workgroups.
//
// Exactness: every float / double expression is the scalar source's, evaluated in the source's order
// (-ffp-contract=off); tests compare logmdct, noise, epeak and npeak bit for bit with the oracle for every block
// type (tests/test_pipeline_gpu.py) and the packets behind them (all packet-level tests).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "batch.h"
#include "kernels.h"

#define VMIN(x, y) ((x) > (y) ? (y) : (x))
#define VMAX(x, y) ((x) < (y) ? (y) : (x))
#define NM_PAD 4          /* floats between consecutive LDS arrays: the scan's lanes (one array each) hit different banks */
#define NM_THREADS 256

namespace {

struct hy_abd { float A, B, D; };

// the five addends of bin k (lib/psy.c:3497-3507 first element, :3509-3541 the rest)
__device__ __forceinline__ void nm_terms(const float v, const int k, const float offset, float *__restrict__ S, const int NS)
{
    float y = v + offset;
    if (y < 1.f) y = 1.f;
    float w = y * y;
    float t1, t2, t4;
    if (k == 0) {
        w = (float)((double)w * .5);
        t1 = w; t2 = 0.f; t4 = 0.f;
    } else {
        const float x = (float)k;          // the source's x += 1.f from 0 is exact below 2^24
        t1 = w * x; t2 = w * x * x; t4 = w * x * y;
    }
    S[k] = w;
    S[NS + k] = t1;
    S[2 * NS + k] = t2;
    S[3 * NS + k] = w * y;
    S[4 * NS + k] = t4;
}

// window sums and regression terms (lib/psy.c:3549-3560 mirrored, :3571-3582 plain); mirrored: the lower edge is row -lo
__device__ __forceinline__ hy_abd nm_window(const float *__restrict__ S, const int NS, const int lo, const int hi, const bool mirror)
{
    const int l = mirror ? -lo : lo;
    const float Hn = S[hi], Hx = S[NS + hi], Hxx = S[2 * NS + hi], Hy = S[3 * NS + hi], Hxy = S[4 * NS + hi];
    const float Ln = S[l], Lx = S[NS + l], Lxx = S[2 * NS + l], Ly = S[3 * NS + l], Lxy = S[4 * NS + l];
    float tN, tX, tXX, tY, tXY;
    if (mirror) {
        tN = Hn + Ln; tX = Hx - Lx; tXX = Hxx + Lxx; tY = Hy + Ly; tXY = Hxy - Lxy;
    } else {
        tN = Hn - Ln; tX = Hx - Lx; tXX = Hxx - Lxx; tY = Hy - Ly; tXY = Hxy - Lxy;
    }
    hy_abd r;
    r.A = tY * tXX - tX * tXY;
    r.B = tN * tXY - tX * tY;
    r.D = tN * tXX - tX * tX;
    return r;
}

// one lane: in-place running sum of n addends (n a multiple of 64).  Two register sets of thirty-two addends: while
// one is added up and written back, the other is fetched.  Issue order (a lone wavefront issues in order): a read,
// four adds, a write — the LDS instructions go into the slots the dependent adds leave open.
#define NM_ADD4(v) acc += v.x; v.x = acc; acc += v.y; v.y = acc; acc += v.z; v.z = acc; acc += v.w; v.w = acc;
#define NM_HALF(cur, nxt, kc, kn)                                                   \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                 \
        nxt[u] = *reinterpret_cast<float4 *>(q + (kn) + 4 * u);                     \
        NM_ADD4(cur[u])                                                             \
        *reinterpret_cast<float4 *>(q + (kc) + 4 * u) = cur[u];                     \
    }                                                                               \
    _Pragma("unroll") for (int u = 0; u < 8; u++) {                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                          \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                          \
    }
__device__ __forceinline__ void nm_scan(float *__restrict__ q, const int n)
{
    float acc = 0.f;
    float4 ca[8], cb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) ca[u] = *reinterpret_cast<float4 *>(q + 4 * u);
    for (int k = 0; k < n; k += 64) {
        const int k2 = (k + 64 < n) ? k + 64 : k;
        NM_HALF(ca, cb, k, k + 32)
        NM_HALF(cb, ca, k + 32, k2)
    }
}

// ---- ring form (round 3): the running sums of a block live in a ring of NM_RING rows per array instead of all n.
// A window reaches at most ~137 rows ahead of its bin and ~120 behind (bark tables, checked by the host for every setup
// that takes this path), so bin i can be solved as soon as the scan has passed row i + 137, and rows more than ~380
// behind the scan are dead.  Per iteration t (one barrier each) three things run side by side in a workgroup:
//   P  the addends of chunk t + 1 (64 bins) go into the ring            (the compute wavefront that owns those bins)
//   Q  the scan wavefront (a fifth wavefront, a lane per (block, sum)) carries its sums through chunk t
//   R  chunk t - NM_LAG is solved from finished rows                    (the compute wavefront that owns those bins)
// so the latency chain of the scan hides behind the solves of the SAME workgroup instead of waiting for other
// workgroups to fill the gap, and a block holds 10 KB of LDS instead of 20.
#define NM_RING 512
#define NM_RSTR (NM_RING + NM_PAD)
#define NM_LAG 4

__device__ __forceinline__ void nm_terms_ring(const float v, const int k, const float offset, float *__restrict__ S)
{
    float y = v + offset;
    if (y < 1.f) y = 1.f;
    float w = y * y;
    float t1, t2, t4;
    if (k == 0) {
        w = (float)((double)w * .5);
        t1 = w; t2 = 0.f; t4 = 0.f;
    } else {
        const float x = (float)k;
        t1 = w * x; t2 = w * x * x; t4 = w * x * y;
    }
    const int r = k & (NM_RING - 1);
    S[r] = w;
    S[NM_RSTR + r] = t1;
    S[2 * NM_RSTR + r] = t2;
    S[3 * NM_RSTR + r] = w * y;
    S[4 * NM_RSTR + r] = t4;
}

__device__ __forceinline__ hy_abd nm_window_ring(const float *__restrict__ S, const int lo, const int hi, const bool mirror)
{
    const int l = (mirror ? -lo : lo) & (NM_RING - 1), h = hi & (NM_RING - 1);
    const float Hn = S[h], Hx = S[NM_RSTR + h], Hxx = S[2 * NM_RSTR + h], Hy = S[3 * NM_RSTR + h], Hxy = S[4 * NM_RSTR + h];
    const float Ln = S[l], Lx = S[NM_RSTR + l], Lxx = S[2 * NM_RSTR + l], Ly = S[3 * NM_RSTR + l], Lxy = S[4 * NM_RSTR + l];
    float tN, tX, tXX, tY, tXY;
    if (mirror) {
        tN = Hn + Ln; tX = Hx - Lx; tXX = Hxx + Lxx; tY = Hy + Ly; tXY = Hxy - Lxy;
    } else {
        tN = Hn - Ln; tX = Hx - Lx; tXX = Hxx - Lxx; tY = Hy - Ly; tXY = Hxy - Lxy;
    }
    hy_abd r;
    r.A = tY * tXX - tX * tXY;
    r.B = tN * tXY - tX * tY;
    r.D = tN * tXX - tX * tX;
    return r;
}

// one lane: the running sum carried through 64 more addends, in place (q = the chunk's first row)
__device__ __forceinline__ void nm_scan_chunk(float *__restrict__ q, float &acc)
{
    float4 ca[8], cb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) ca[u] = *reinterpret_cast<float4 *>(q + 4 * u);
    NM_HALF(ca, cb, 0, 32)
    NM_HALF(cb, ca, 32, 0)      // (the second fetch re-reads finished rows: nothing waits for it)
}

// aoTuV M7, lib/psy.c:3645-3768, on LDS rows: spectral = logmdct, noise = pass 2's logmdct - work; temp / inmod: 256 each
__device__ void nm_ntfix(const vbm_psy *p, const int block_mode, const float *spectral, float *noise, float *temp, float *inmod,
                         const int scratch)
{
    int i, j, k;
    const int n = p->n;
    int nx = p->tonefix_end;
    const float limit = fabsf(p->noiseoffset[1][0]);
    const float *__restrict__ ntfix_noiseoffset = p->ntfix_noiseoffset, *__restrict__ noiseoffset1 = p->noiseoffset[1];
    if (!nx) return;
    for (i = 0; i < scratch; i++) { temp[i] = 0.f; inmod[i] = 0.f; }     // the source clears 256; entries from n + 4 on are never read
    if (block_mode <= 1) {
        const int freq_upc = 3;
        const int freq_unc = 4;
        int nxplus = nx + freq_unc;
        float tolerance = 9.f;
        const float strength = .6f;
        if (n == 256) tolerance = 15.f;
        if (nxplus > n) {
            nx = n;
            nxplus = n - freq_unc;
        }
        for (i = 0; i < nxplus; i++) {
            const float sp = spectral[i];
            if (sp < -70) inmod[i] = (float)(-70 + (double)(sp + 70) * .1);
            else inmod[i] = sp;
        }
        for (i = freq_unc; i < nx; i++) {
            if ((spectral[i] > spectral[i - 1]) && (spectral[i] > spectral[i + 1])) {
                int ps = i - 1;
                int pe = i + 1;
                const int upper = i - freq_upc;
                const int under = i + freq_unc;
                for (j = ps; j > upper; j--) {
                    if (spectral[j + 1] < spectral[j]) break;
                    ps = j;
                }
                for (j = pe; j < under; j++) {
                    if (spectral[j - 1] < spectral[j]) break;
                    pe = j;
                }
                {
                    float ss = inmod[i] - inmod[ps];
                    ss = VMAX(ss, inmod[i] - inmod[pe]);
                    if (ss > tolerance) {
                        if (spectral[i] > noise[i]) {
                            ss -= tolerance;
                            ss *= strength;
                        }
                        for (j = ps; j <= pe; j++) {
                            temp[j] = VMAX(ss, temp[j]);
                            if (temp[j] < 0) temp[j] = 0;
                        }
                    }
                }
                i = pe;
            }
        }
        for (i = freq_unc - 1; i < nx; i++) {
            const float test = VMIN(ntfix_noiseoffset[i], noiseoffset1[i] + limit);
            if (temp[i] > test) temp[i] = test;
            noise[i] -= temp[i];
        }
    } else if (block_mode == 2) {
        for (i = 0, k = 0; i < nx; i += 8, k++) {
            double na = 0;
            for (j = 0; j < 8; j++) na += noise[i + j];
            na /= 8;
            temp[k] = (float)na;
        }
        nx /= 8;
        for (i = 3; i < nx; i++) {
            if ((temp[i] > temp[i - 1]) && (temp[i] > temp[i + 1])) {
                int a = 0, bb = 0;
                float thres = 0;
                if (temp[i - 1] > temp[i - 2]) {
                    thres = temp[i - 2];
                    a = i - 3;
                } else {
                    thres = temp[i - 1];
                    a = i - 2;
                }
                bb = i + 3;
                thres = temp[i] - thres;
                if ((double)thres > 2.) {
                    const int eightimes = i * 8;
                    const float test = VMIN(ntfix_noiseoffset[eightimes], noiseoffset1[eightimes] + limit);
                    thres = VMIN(thres - 2, test);
                    a *= 8;
                    bb *= 8;
                    for (j = a; j <= bb; j++) noise[j] -= thres;
                }
            }
        }
    }
}

template <int NB> struct rowseg;
template <> struct rowseg<1> { static __device__ __forceinline__ void put(float *q, const float *v) { q[0] = v[0]; } };
template <> struct rowseg<2> { static __device__ __forceinline__ void put(float *q, const float *v) { *reinterpret_cast<float2 *>(q) = make_float2(v[0], v[1]); } };
template <> struct rowseg<4> { static __device__ __forceinline__ void put(float *q, const float *v) { *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]); } };
template <> struct rowseg<8> {
    static __device__ __forceinline__ void put(float *q, const float *v)
    {
        *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// NB values of tile row i (lanes lane0 .. lane0 + NB - 1 of the tile `q` points into)
template <int NB>
__device__ __forceinline__ void put_row(float *__restrict__ q, const float *v, const int nb)
{
    if (nb == NB) rowseg<NB>::put(q, v);
    else
        for (int u = 0; u < nb; u++) q[u] = v[u];
}

// ROWS = ceil(n / 256): thread t holds bins t, t + 256, ... of all NB blocks
template <int NB, int ROWS, bool RING>
__global__ __launch_bounds__(NM_THREADS + (RING ? 64 : 0)) void k_noisemask(vbm_batch b, const int phases)
{
    extern __shared__ __align__(16) float SU[];      // [NB][5][NS] addends, then running sums (RING: [NB][5][NM_RSTR]); later rows for M7 / M2 / M8
    __shared__ float s_ncl[NB], s_poste[NB];
    __shared__ int s_col[NB], s_need[NB];

    const vbm_setup *s = b.setup;
    const vbm_psy *p = &s->psy[b.block_mode];
    const int n = p->n, NS = n + NM_PAD;
    const int ncb = vbm_ncb(b);
    // XCD x (blockIdx % 8) takes the workgroups [x G/8, (x+1) G/8) in order: a tile's workgroups run side by side
    // (measured: 0.73 ms with, 0.91 ms without — VBM_NOISE_PHASES bit 7 switches it off)
    const int G8 = (int)(gridDim.x >> 3);
    const int wg = (phases & 128) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * G8 + (int)(blockIdx.x >> 3);
    const int lane0 = wg * NB;
    if (lane0 >= ncb) return;
    const int nb = VMIN(NB, ncb - lane0);
    const int tid = (int)threadIdx.x;
    const bool scanw = RING && tid >= NM_THREADS;      // the scan wavefront of the ring form
    const int BS = RING ? 5 * NM_RSTR : 5 * NS;        // floats of LDS per block

    const int i1 = p->hy_i1, i2 = p->hy_i2;
    const int fixed = p->noisewindowfixed;
    const int f1 = (fixed > 0) ? p->hy_f1 : 0, f2 = (fixed > 0) ? p->hy_f2 : 0;
    const size_t tile = (size_t)(lane0 >> 6) * b.slab_words + (lane0 & 63);

    float lm[ROWS][NB], wk[ROWS][NB], nz[ROWS][NB];
    int wlo[ROWS], whi[ROWS];        // the variable window of the thread's bins (bins from i2 on: that of bin i2 - 1)

    // ---- logmdct (lib/mapping0.c:936-950)
    if (!scanw) {
        const float *__restrict__ src = b.mdct_bm + (size_t)lane0 * n;
        const int *__restrict__ bark_lo = p->bark_lo, *__restrict__ bark_hi = p->bark_hi;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            const int iw = (i < i2) ? i : i2 - 1;
            wlo[r] = 0; whi[r] = 0;
            if (i < n && iw >= 0) { wlo[r] = bark_lo[iw]; whi[r] = bark_hi[iw]; }
            float mv[NB];
#pragma unroll
            for (int blk = 0; blk < NB; blk++) {
                float v = 0.f;
                if (i < n && blk < nb) v = src[(size_t)blk * n + i];
                mv[blk] = v;
                lm[r][blk] = (float)((double)vbm_todB(v) + .345);
            }
            // the MDCT rows in the tiled layout for the lane-per-block kernels behind this one (k_mix, couple): this
            // kernel reads the block-major rows anyway, so no transpose pass of its own is needed
            if (i < n) put_row<NB>(b.mdctT + tile + (size_t)i * 64, mv, nb);
        }
    }

    // ---- lb_loudnoise_fix (lib/psy.c:5152-5180); the mean over the middle bins is a double-precision chain in bin
    //      order: a block that needs it (the first block after a change between transition and long blocks) has its
    //      logmdct row put into LDS for one lane to walk
    if (!scanw && tid < nb) {
        const int lane = lane0 + tid;
        const int sb = lane / b.ch, c = lane - sb * b.ch;
        const int sid = b.stream_id[sb];
        const int col = sid * b.ch + c;
        s_col[tid] = col;
        s_poste[tid] = b.poste[lane];
        float noise_compand_level = b.st.lowcomp[col];
        const int lW_block_mode = b.st.lW_block_mode[sid];
        int need = 0;
        if (p->m_val < 0.5) noise_compand_level = -1;
        else if (p->normal_thresh > .45) noise_compand_level = -1;
        else if ((b.block_mode == 2 && lW_block_mode == 3) || (b.block_mode == 3 && lW_block_mode == 2)) need = 1;
        s_ncl[tid] = noise_compand_level;
        s_need[tid] = need;
    }
    __syncthreads();
    {
        int any = 0;
        for (int blk = 0; blk < nb; blk++) any |= s_need[blk];
        if (any) {      // (uniform over the workgroup)
            if (!scanw)
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int i = tid + r * NM_THREADS;
                if (i < n)
#pragma unroll
                    for (int blk = 0; blk < NB; blk++)
                        if (blk < nb) SU[(size_t)blk * BS + i] = lm[r][blk];
            }
            __syncthreads();
            if (!scanw && tid < nb && s_need[tid]) {
                double hi_th = 0;
                const int n25p = p->n25p, n75p = p->n75p;
                const float *row = SU + (size_t)tid * BS;
                for (int k = n25p; k < n75p; k++) {
                    const float v = row[k];
                    hi_th += (v > -130) ? (double)v : -130.;
                }
                hi_th /= n;
                float noise_compand_level;
                if (hi_th > -40.) noise_compand_level = -1;
                else if (hi_th < -50.) noise_compand_level = 1.f;
                else noise_compand_level = (float)(1. - ((hi_th + 50) / 10));
                s_ncl[tid] = noise_compand_level;
            }
            __syncthreads();
        }
    }
    if (!scanw && tid < nb) b.st.lowcomp[s_col[tid]] = s_ncl[tid];

    // ---- the two passes of bark_noise_hybridmp (lib/psy.c:3799-3812)
    if constexpr (RING) {
        const int NCH = n >> 6;                          // chunks of 64 bins (n is a multiple of 256 here)
        const int wv = tid >> 6, ln = tid & 63;
        // the scan wavefront: lane L = (block, sum)
        const int sblk = ln / 5, ssum = ln - sblk * 5;
        const bool slive = scanw && ln < nb * 5;
        float *const sq = SU + (size_t)sblk * BS + ssum * NM_RSTR;
#pragma unroll 1
        for (int pass = 1; pass <= 2; pass++) {
            const float offset = (pass == 1) ? 140.f : 0.f;
            float sacc = 0.f;
            // the solve of bin i = tid + 256 r, r a compile-time constant (the bin's values live in registers lm[r] / wk[r] / nz[r])
            auto solve = [&](auto RC) {
                constexpr int r = decltype(RC)::value;
                const int i = tid + r * NM_THREADS;
                const float x = (float)i;
                const bool have = i2 > 0;
                const bool mirror = ((i < i2) ? i : i2 - 1) < i1;
                const int fw = (i < f2) ? i : f2 - 1;
                const int fhi = fw + fixed / 2, flo = fhi - fixed;
#pragma unroll
                for (int blk = 0; blk < NB; blk++) {
                    if (blk >= nb) continue;
                    const float *S = SU + (size_t)blk * BS;
                    hy_abd v; v.A = 0.f; v.B = 0.f; v.D = 1.f;
                    if (have) v = nm_window_ring(S, wlo[r], whi[r], mirror);
                    float R = (v.A + x * v.B) / v.D;
                    if (R < 0.f) R = 0.f;
                    float nzv = R - offset;
                    if (pass == 1) {
                        wk[r][blk] = lm[r][blk] - nzv;                     // lib/psy.c:3807
                    } else {
                        if (fixed > 0) {
                            hy_abd w = v;
                            if (fw >= 0) w = nm_window_ring(S, flo, fhi, fw < f1);
                            else if (i < i2 && have) w = nm_window_ring(S, p->bark_lo[i2 - 1], p->bark_hi[i2 - 1], i2 - 1 < i1);
                            R = (w.A + x * w.B) / w.D;
                            if (R - offset < nzv) nzv = R - offset;
                        }
                        nz[r][blk] = nzv;
                        wk[r][blk] = lm[r][blk] - wk[r][blk];              // lib/psy.c:3812
                    }
                }
            };
            auto terms = [&](auto RC) {
                constexpr int r = decltype(RC)::value;
                const int i = tid + r * NM_THREADS;
#pragma unroll
                for (int blk = 0; blk < NB; blk++)
                    if (blk < nb) nm_terms_ring(pass == 1 ? lm[r][blk] : wk[r][blk], i, offset, SU + (size_t)blk * BS);
            };
#pragma unroll 1
            for (int t = -1; t < NCH + NM_LAG; t++) {
                if (!scanw) {
                    const int cp = t + 1, cr = t - NM_LAG;
                    if (cp < NCH && (cp & 3) == wv) {               // P: the addends of chunk t + 1
                        switch (cp >> 2) {
                        case 0: terms(std::integral_constant<int, 0>{}); break;
                        case 1: if constexpr (ROWS > 1) terms(std::integral_constant<int, 1>{}); break;
                        case 2: if constexpr (ROWS > 2) terms(std::integral_constant<int, 2>{}); break;
                        case 3: if constexpr (ROWS > 3) terms(std::integral_constant<int, 3>{}); break;
                        }
                    }
                    if (cr >= 0 && cr < NCH && (cr & 3) == wv && (phases & 2)) {   // R: chunk t - NM_LAG from finished rows
                        switch (cr >> 2) {
                        case 0: solve(std::integral_constant<int, 0>{}); break;
                        case 1: if constexpr (ROWS > 1) solve(std::integral_constant<int, 1>{}); break;
                        case 2: if constexpr (ROWS > 2) solve(std::integral_constant<int, 2>{}); break;
                        case 3: if constexpr (ROWS > 3) solve(std::integral_constant<int, 3>{}); break;
                        }
                    }
                } else if (slive && t >= 0 && t < NCH && (phases & 1)) {
                    nm_scan_chunk(sq + ((t << 6) & (NM_RING - 1)), sacc);   // Q: the sums through chunk t
                }
                __syncthreads();
            }
        }
        if (scanw) return;      // the scan wavefront is done (a wavefront that has ended is not waited for at later barriers)
    } else
#pragma unroll
    for (int pass = 1; pass <= 2; pass++) {
        const float offset = (pass == 1) ? 140.f : 0.f;
        // addends of logmdct (pass 1) / work (pass 2)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            if (i < n)
#pragma unroll
                for (int blk = 0; blk < NB; blk++)
                    if (blk < nb) nm_terms(pass == 1 ? lm[r][blk] : wk[r][blk], i, offset, SU + (size_t)blk * 5 * NS, NS);
        }
        __syncthreads();
        if (tid < nb * 5 && (phases & 1)) nm_scan(SU + (size_t)tid * NS, n);
        __syncthreads();
        if (phases & 2) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const int i = tid + r * NM_THREADS;
                if (i >= n) continue;
                const float x = (float)i;                      // the source's x += 1.f from 0 is exact below 2^24
                const bool have = i2 > 0;                      // (bins from i2 on keep A, B, D of the last bin solved, :3587-3591)
                const bool mirror = ((i < i2) ? i : i2 - 1) < i1;
                // fixed window of pass 2 (lib/psy.c:3595-3636): bins from f2 on keep the terms of bin f2 - 1; with no
                // such bin, those the variable window's loops left behind
                const int fw = (i < f2) ? i : f2 - 1;
                const int fhi = fw + fixed / 2, flo = fhi - fixed;
#pragma unroll
                for (int blk = 0; blk < NB; blk++) {
                    if (blk >= nb) continue;
                    const float *S = SU + (size_t)blk * 5 * NS;
                    hy_abd v; v.A = 0.f; v.B = 0.f; v.D = 1.f;
                    if (have) v = nm_window(S, NS, wlo[r], whi[r], mirror);
                    float R = (v.A + x * v.B) / v.D;
                    if (R < 0.f) R = 0.f;
                    float nzv = R - offset;
                    if (pass == 1) {
                        wk[r][blk] = lm[r][blk] - nzv;                     // lib/psy.c:3807
                    } else {
                        if (fixed > 0) {
                            hy_abd w = v;
                            if (fw >= 0) w = nm_window(S, NS, flo, fhi, fw < f1);
                            else if (i < i2 && have) w = nm_window(S, NS, p->bark_lo[i2 - 1], p->bark_hi[i2 - 1], i2 - 1 < i1);
                            R = (w.A + x * w.B) / w.D;
                            if (R - offset < nzv) nzv = R - offset;
                        }
                        nz[r][blk] = nzv;
                        wk[r][blk] = lm[r][blk] - wk[r][blk];              // lib/psy.c:3812
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- aoTuV M7 (short and transition blocks) on rows in LDS: logmdct at [0], work at [NS], scratch behind
    if (b.block_mode <= 2) {
        const int scratch = VMIN(256, NS);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            if (i < n)
#pragma unroll
                for (int blk = 0; blk < NB; blk++)
                    if (blk < nb) {
                        SU[(size_t)blk * BS + i] = lm[r][blk];
                        SU[(size_t)blk * BS + NS + i] = wk[r][blk];
                    }
        }
        __syncthreads();
        if (tid < nb) {
            float *row = SU + (size_t)tid * BS;
            nm_ntfix(p, b.block_mode, row, row + NS, row + 2 * NS, row + 2 * NS + scratch, scratch);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            if (i < n)
#pragma unroll
                for (int blk = 0; blk < NB; blk++)
                    if (blk < nb) wk[r][blk] = SU[(size_t)blk * BS + NS + i];
        }
        __syncthreads();
    }

    // ---- noise compand, aoTuV M5 extension, tone peak with M9 folded in (lib/psy.c:3822-3860, :4058-4072: M9 replaces
    //      every bin's peak by a value of that peak, logmdct and lastmdct alone); logmdct and epeak rows go out, logmdct
    //      and logmask go to LDS for the partitions' lanes (row of partition k shifted by k words)
    const int partition = (p->normal_p ? p->normal_partition : 16);
    // logmask rows start here (a skewed row is n + n / partition <= 2 NS words; the ring form has 5 * NM_RSTR words per block)
    const int SK = RING ? n + n / partition + 4 : 2 * NS;
    {
        const float *__restrict__ noisecompand = p->noisecompand, *__restrict__ noisecompand_high = p->noisecompand_high;
        const int *__restrict__ stn_compand = s->stn_compand;
        const int m9_end = (b.block_mode > 1) ? p->tonecomp_endp : 0;
        const int n33p = p->n33p;
        float *__restrict__ logmdctT = b.logmdctT + tile, *__restrict__ epeakT = b.epeakT + tile;
        if (phases & 4)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            if (i >= n) continue;
            const int sk = i + i / partition;
            float epv[NB];
#pragma unroll
            for (int blk = 0; blk < NB; blk++) {
                epv[blk] = 0.f;
                if (blk < nb) {
                    const float lmk = nz[r][blk], wv = wk[r][blk], lmd = lm[r][blk];
                    const float ncl = s_ncl[blk];
                    const int thter = (ncl > 0) ? n33p : 0;
                    int dB = (int)((double)lmk + .5);
                    if (dB >= VBM_NOISE_COMPAND_LEVELS) dB = VBM_NOISE_COMPAND_LEVELS - 1;
                    if (dB < 0) dB = 0;
                    const float ep = wv + stn_compand[dB];
                    float lk;
                    if (i < thter) lk = wv + noisecompand[dB] - ((noisecompand[dB] - noisecompand_high[dB]) * ncl);
                    else lk = wv + noisecompand[dB];
                    float e = 0.f;
                    if (i < m9_end) {
                        const float temp = lmd - ep;
                        if (temp >= 12.f) {
                            const int col = s_col[blk];
                            const float lst = b.st.mblock[(size_t)(col >> 6) * b.st.slab_words + (col & 63) + (size_t)i * 64];
                            const float mi = lmd - lst;
                            if (mi >= 1) e = mi;
                        }
                    }
                    epv[blk] = e;
                    SU[(size_t)blk * BS + sk] = lmd;
                    SU[(size_t)blk * BS + SK + sk] = lk;
                }
            }
            if (phases & 16) {
                put_row<NB>(logmdctT + (size_t)i * 64, lm[r], nb);
                put_row<NB>(epeakT + (size_t)i * 64, epv, nb);
            }
        }
    }
    __syncthreads();

    // ---- M2 post-echo reduction and M8 per normal-partition (lib/psy.c:3862-3920 ... :4056)
    {
        const int nparts = (n + partition - 1) / partition;
        const float *__restrict__ noiseoffset1 = p->noiseoffset[1];
        const int min_nn_lp = p->min_nn_lp;
        if (phases & 8)
        for (int t = tid; t < nb * nparts; t += NM_THREADS) {
            const int blk = t / nparts, k = t - blk * nparts;
            const int i = k * partition;
            const float *logmdct = SU + (size_t)blk * BS + i + k;
            float *logmask = SU + (size_t)blk * BS + SK + i + k;
            float np = 0.f;
            if (i < min_nn_lp) {
                const float poste = s_poste[blk];
                if (poste > 0) {
                    const float temp = VMIN(VMIN(poste, 30.f), noiseoffset1[i] + 30.f);
                    if (!(temp <= 0)) {
                        np = -1.f;
                        for (int j = 0; j < partition; j++) logmask[j] -= temp;
                    }
                }
                const float nt = 4;
                const float o = noiseoffset1[i + partition - 1] + 6;
                if (!(o <= 0) && !((double)np < -0.5)) {
                    float me = 0;
                    float avge = 0;
                    for (int j = 0; j < partition; j++) {
                        const float temp = logmdct[j] - logmask[j];
                        if (me < temp) me = temp;
                        avge += logmdct[j];
                    }
                    if (!(avge < (-95 * partition)))
                        if (me < nt) np = (VMIN(o, nt - me)) / nt;
                }
            }
            b.npeakT[tile + blk + (size_t)k * 64] = np;
        }
    }
    __syncthreads();

    // ---- logmask rows out
    {
        float *__restrict__ noiseT = b.noiseT + tile;
        if (phases & 16)
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int i = tid + r * NM_THREADS;
            if (i >= n) continue;
            const int sk = i + i / partition;
            float v[NB];
#pragma unroll
            for (int blk = 0; blk < NB; blk++) v[blk] = (blk < nb) ? SU[(size_t)blk * BS + SK + sk] : 0.f;
            put_row<NB>(noiseT + (size_t)i * 64, v, nb);
        }
    }
}

template <int NB, int ROWS, bool RING = false>
int launch(const vbm_batch *b, hipStream_t st)
{
    // (called from several host threads at once: one-time set-up through initialisers of function-local statics)
    static const int phases = getenv("VBM_NOISE_PHASES") ? atoi(getenv("VBM_NOISE_PHASES")) : 31;   // timing experiments
    size_t lds = (size_t)5 * NB * (RING ? NM_RSTR : b->n + NM_PAD) * sizeof(float);
    if (getenv("VBM_NOISE_LDS_PAD")) lds += (size_t)atoi(getenv("VBM_NOISE_LDS_PAD")) * 1024;   // occupancy experiments
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void *>(k_noisemask<NB, ROWS, RING>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
    (void)attr;
    const unsigned wgs = (unsigned)((b->ncb + NB - 1) / NB);
    hipLaunchKernelGGL((k_noisemask<NB, ROWS, RING>), dim3((wgs + 7u) & ~7u), dim3(NM_THREADS + (RING ? 64 : 0)), lds, st, *b, phases);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// Blocks per workgroup: 2 long blocks (n = 1024: 41 KB of LDS, three workgroups per CU), more of the shorter ones
// (the rows going out are NB words wide).  n: 128 .. 4096, a multiple of 16.
extern "C" int vbm_launch_noisemask(const vbm_batch *b, hipStream_t st)
{
    const int n = b->n;
    if ((n & 63) || n < 128 || n > 4096) return -2;     // (M7's scratch rows live in a block's five sum arrays)
    if (n <= 256) return launch<8, 1>(b, st);
    if (n <= 512) return launch<4, 2>(b, st);
    // (occupancy experiments, round 3, tools/gpu_noise_occ.sh / gpu_ab_env.sh — all removed again: the kernel's time alone goes
    //  with 1 / workgroups per CU: 0.73 ms at three, 0.96 at two, 1.74 at one.  A fourth workgroup per CU — sums without
    //  padding = exactly a quarter of the LDS, rows XOR-swizzled per sum so that the ten scan lanes still hit ten banks, the
    //  per-block scalars in registers instead of static LDS — takes 0.68 ms alone and makes the from-PCM step 4.68-5.0 ms
    //  against 4.59-4.62 (LDS the kernels beside this one no longer get); even the register scalars alone cost the padded
    //  form 1 %.  One block per workgroup (seven workgroups per CU, 4-byte row stores): 0.90 ms alone.  At most two
    //  workgroups per CU: from PCM 5.2 ms against 4.7.  Three it is.)
    if (n == 1024 && b->noise_ring) return launch<2, 4, true>(b, st);     // ring form: host-checked window reaches (configure())
    if (n <= 1024) return launch<2, 4>(b, st);
    return launch<1, 16>(b, st);
}
