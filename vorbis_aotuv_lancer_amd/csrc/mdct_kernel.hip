// Window + forward MDCT for gfx950 (MI355X), one wavefront per 2048-sample block
// (or per 8 consecutive 256-sample blocks).
//
// Replaces, for a batch of blocks:  _vorbis_apply_window (reference lib/window.c:2137-2261)
// followed by mdct_forward (lib/mdct.c:1799-1869, scalar branch).  Results are bit-identical
// to that scalar path: every butterfly keeps the source expression shape (two rounded
// products + one rounded sum), the file is compiled with -ffp-contract=off, and the trig
// tables come from the host libm exactly as lib/mdct.c:67-76 computes them.
//
// Data flow per 512-complex "group" (one long block, or 8 short blocks):
//   HBM -> 8 x 16-B loads per lane (fully coalesced, lanes reversed) -> window multiply
//   -> even samples stay in the lane, odd samples cross lanes through LDS (8-B slots)
//   -> fold + pre-twiddle                                   (lib/mdct.c:1819-1851)
//   -> radix-2 stages on index bits 8..6 in registers        (round A)
//   -> LDS transpose -> stages on bits 5,4 in registers      (round B)
//   -> LDS transpose -> 32-point butterflies: top level via one cross-lane swap,
//      16/8-point levels entirely in registers               (round C)
//   -> LDS -> bit-reverse gather + post-twiddle + scale      (lib/mdct.c:1228-1272, 1859-1868)
//   -> 4 x 16-B coalesced stores per lane.
// HBM traffic is exactly the algorithmic 6*N bytes per block; twiddles and windows are
// LDS-resident (staged once per workgroup), the next group's loads are issued before the
// current group's butterflies so every wave keeps 8 KB in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdct_kernel.h"

namespace {

constexpr float K_PI3_8 = .38268343236508977175F;  // lib/mdct.h:44-46
constexpr float K_PI2_8 = .70710678118654752441F;
constexpr float K_PI1_8 = .92387953251128675613F;

constexpr int WAVES_PER_WG = 4;
constexpr int SLOTS = 576;  // 512 complex + 1 pad slot per 8 (bank-conflict-free transposes)

__device__ __forceinline__ int slot_addr(int m) { return m + (m >> 3); }

// wave-level ordering of LDS traffic: DS ops of one wave execute in issue order, so all
// that is needed is that the compiler neither reorders nor caches across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// lo' = rot(up - lo), up' = up + lo      (lib/mdct.c:1044-1049 expression shape)
__device__ __forceinline__ void bfly(float2 &lo, float2 &up, float2 w)
{
    float r0 = up.x - lo.x;
    float r1 = up.y - lo.y;
    up.x += lo.x;
    up.y += lo.y;
    lo.x = r1 * w.y + r0 * w.x;
    lo.y = r1 * w.x - r0 * w.y;
}

// lib/mdct.c:432-452
__device__ __forceinline__ void bfly8(float *x)
{
    float a = x[6] + x[2], b = x[6] - x[2];
    float c = x[4] + x[0], d = x[4] - x[0];
    float e = x[5] - x[1], f = x[7] - x[3];
    float g = x[5] + x[1], h = x[7] + x[3];
    x[6] = a + c;
    x[4] = a - c;
    x[0] = b + e;
    x[2] = b - e;
    x[3] = f + d;
    x[1] = f - d;
    x[7] = h + g;
    x[5] = h - g;
}

// lib/mdct.c:495-528
__device__ __forceinline__ void bfly16(float *x)
{
    float r0, r1;
    r0 = x[1] - x[9];   r1 = x[0] - x[8];
    x[8] += x[0];   x[9] += x[1];
    x[0] = (r0 + r1) * K_PI2_8;
    x[1] = (r0 - r1) * K_PI2_8;
    r0 = x[3] - x[11];  r1 = x[10] - x[2];
    x[10] += x[2];  x[11] += x[3];
    x[2] = r0;  x[3] = r1;
    r0 = x[12] - x[4];  r1 = x[13] - x[5];
    x[12] += x[4];  x[13] += x[5];
    x[4] = (r0 - r1) * K_PI2_8;
    x[5] = (r0 + r1) * K_PI2_8;
    r0 = x[14] - x[6];  r1 = x[15] - x[7];
    x[14] += x[6];  x[15] += x[7];
    x[6] = r0;  x[7] = r1;
    bfly8(x);
    bfly8(x + 8);
}

// Window multiplier for the four samples starting at i (i % 4 == 0) of an n-sample block
// whose neighbours have sizes ln / rn.  lib/window.c:2137-2147, 2247-2258.
// wl / wr: rising half-windows (ln/2, rn/2 floats) in LDS.
__device__ __forceinline__ float4 window4(float4 d, int i, int n, int ln, int rn,
                                          const float *wl, const float *wr)
{
    if (i < (n >> 1)) {
        int lb = (n >> 2) - (ln >> 2);
        if (i < lb) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < lb + (ln >> 1)) {
            float4 w = *reinterpret_cast<const float4 *>(wl + (i - lb));
            return make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
        }
        return d;
    } else {
        int rb = (n >> 1) + (n >> 2) - (rn >> 2);
        if (i < rb) return d;
        if (i < rb + (rn >> 1)) {
            float4 w = *reinterpret_cast<const float4 *>(wr + ((rn >> 1) - 4 - (i - rb)));
            return make_float4(d.x * w.w, d.y * w.z, d.z * w.y, d.w * w.x);
        }
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int N>
struct Geo {
    static constexpr int C = N / 4;          // complex values per block after the fold
    static constexpr int BPG = 512 / C;      // blocks per 512-complex group
    static constexpr int LOG2C = (N == 2048) ? 9 : (N == 1024) ? 8 : (N == 512) ? 7 : (N == 256) ? 6 : 5;
    static constexpr int R2 = 3 * N / 16;    // first pair of fold region 3
    static constexpr int R1 = N / 16;        // first pair of fold region 2
};

// first sample of block `blk`: rows of a block-major array, or (N == 128, envelope search) the
// 128-sample window of search step first[stream] + t0 + t inside a channel's PCM buffer, blk =
// channel * steps + t; nullptr when that step does not exist
template <bool GATHER>
__device__ __forceinline__ const float *block_src(const float *__restrict__ pcm, long blk, int n,
                                                  const vbm_ve_gather &g)
{
    if (!GATHER) return pcm + blk * n;
    const int c = (int)blk / g.steps;              // (fewer than 2^31 blocks per launch)
    const int t = (int)blk - c * g.steps;
    const int s = c / g.ch;
    const int j = g.first[s] + g.t0 + t;
    if (j >= g.last[s]) return nullptr;
    return pcm + (long)g.parity[s] * g.plane + (long)c * g.cap + g.base[s] + (long)j * 64;
}

// group of eight 16-B loads -> registers
// returns whether any block of the group exists (wave-uniform): a group without one is skipped
template <int N, bool GATHER>
__device__ __forceinline__ bool issue_loads(float4 (&v)[8], const float *__restrict__ pcm,
                                            long group, long nblocks, int lane, const vbm_ve_gather &g)
{
    using G = Geo<N>;
    bool any = false;
    // Gathered blocks (the envelope's search MDCTs: 16 blocks of 128 samples per group): where a block starts takes
    // two integer divisions and four loads of per-stream state; lane b resolves block b once and the eight loads of
    // a lane pick their block's address up by shuffle (each lane resolving the block of each of its loads spent
    // more instructions on addresses than the transform has)
    unsigned alo = 0, ahi = 0;
    if (GATHER) {
        const float *mine = nullptr;
        if (lane < G::BPG) {
            const long blk = group * G::BPG + lane;
            if (blk < nblocks) mine = block_src<GATHER>(pcm, blk, N, g);
        }
        alo = (unsigned)(unsigned long long)(uintptr_t)mine;
        ahi = (unsigned)((unsigned long long)(uintptr_t)mine >> 32);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int P = lane + 64 * k;
        int b = P / G::C, p = P % G::C;
        int q0 = (p < G::R2) ? (G::R2 - 1 - p) : (7 * N / 16 - 1 - p);
        long blk = group * G::BPG + b;
        const float *src;
        if (GATHER) {
            const unsigned lo = (unsigned)__shfl((int)alo, b), hi = (unsigned)__shfl((int)ahi, b);
            src = reinterpret_cast<const float *>((uintptr_t)(((unsigned long long)hi << 32) | lo));
        } else {
            src = (blk < nblocks) ? block_src<GATHER>(pcm, blk, N, g) : nullptr;
        }
        if (src) {
            v[k] = *reinterpret_cast<const float4 *>(src + 4 * q0);
            any = true;
        } else
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return __any(any);
}

// Radix rounds A, B, C on one 512-complex group held as c[k] = element lane + 64k: the butterfly stages of
// mdct_butterflies (lib/mdct.c:1105-1135) for index bits 8..4 — those the block size has — and the 32-point
// butterflies; the result is left in sx in natural order (slot 8*lane + k at padded address 9*lane + k).
// LOG2C = log2 of the complex length of the transform the group belongs to (9: a 2048 block = the group;
// 10: one half of a 4096 block after its first stage; 8, 7, 6, 5: two, four, eight, sixteen blocks per group).
template <int LOG2C>
__device__ __forceinline__ void radix_rounds(float2 (&c)[8], float2 *sx, const float *s_trig, const int lane)
{
        // ---------------- round A: index bits 8,7,6 of the blocks that have them ----------
        // trigint of the stage pairing index bit b is 4 << (LOG2C - 1 - b): the first butterfly of a block
        // steps the table by 4, every later stage doubles it (lib/mdct.c:1105-1135)
        if (LOG2C >= 9) {
            constexpr int TI = 4 << (LOG2C >= 9 ? LOG2C - 9 : 0);
#pragma unroll
            for (int k = 0; k < 4; k++) {  // bit 8
                int t = 255 - (lane + 64 * k);
                bfly(c[k], c[k + 4], *reinterpret_cast<const float2 *>(s_trig + TI * t));
            }
        }
        if (LOG2C >= 8) {
            constexpr int TI = 4 << (LOG2C >= 8 ? LOG2C - 8 : 0);
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4)
#pragma unroll
                for (int k = 0; k < 2; k++) {  // bit 7
                    int t = 127 - (lane + 64 * k);
                    bfly(c[kb + k], c[kb + k + 2], *reinterpret_cast<const float2 *>(s_trig + TI * t));
                }
        }
        if (LOG2C >= 7) {
            constexpr int TI = 4 << (LOG2C >= 7 ? LOG2C - 7 : 0);
            int t = 63 - lane;  // bit 6
            float2 w = *reinterpret_cast<const float2 *>(s_trig + TI * t);
#pragma unroll
            for (int kb = 0; kb < 8; kb += 2) bfly(c[kb], c[kb + 1], w);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) sx[slot_addr(lane + 64 * k)] = c[k];
        wave_lds_sync();

        // ---------------- round B: index bits 5,4 ------------------------------------
        {
            const int base = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = sx[slot_addr(base + 8 * k)];
            constexpr int MUL0 = (LOG2C >= 6) ? (4 << (LOG2C >= 6 ? LOG2C - 6 : 0)) : 0;  // trigint of the 64-complex stage
            constexpr int MUL1 = 4 << (LOG2C - 5);  // trigint of the 32-complex stage
            if (LOG2C >= 6) {   // 128-point blocks (32 complex) have no 64-complex stage
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    int t = 31 - ((lane & 7) + 8 * k);
                    bfly(c[k], c[k + 4], *reinterpret_cast<const float2 *>(s_trig + MUL0 * t));
                }
            }
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int t = 15 - ((lane & 7) + 8 * k);
                    bfly(c[kb + k], c[kb + k + 2], *reinterpret_cast<const float2 *>(s_trig + MUL1 * t));
                }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 8; k++) sx[slot_addr(base + 8 * k)] = c[k];
        }
        wave_lds_sync();

        // ---------------- round C: 32-point butterflies (lib/mdct.c:602-658) ----------
        float x[16];
        {
            // slots 8*lane .. 8*lane+7 are contiguous at padded address 9*lane (8-B units);
            // 9*lane*8 bytes is only 8-B aligned, so read as float2
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float2 t = sx[9 * lane + k];
                x[2 * k] = t.x;
                x[2 * k + 1] = t.y;
            }
        }
        {
            float y[16];
#pragma unroll
            for (int j = 0; j < 16; j++) y[j] = __shfl_xor(x[j], 1);
            if (lane & 1) {
                // upper half of the 32-block: x[16+j] += x[j]
#pragma unroll
                for (int j = 0; j < 16; j++) x[j] = x[j] + y[j];
            } else {
                // lower half: differences (upper - lower, or as the source has it) rotated
                float r0, r1;
                r0 = x[0] - y[0];   r1 = x[1] - y[1];
                x[0] = r1 * K_PI3_8 + r0 * K_PI1_8;
                x[1] = r1 * K_PI1_8 - r0 * K_PI3_8;
                r0 = x[2] - y[2];   r1 = x[3] - y[3];
                x[2] = (r1 + r0) * K_PI2_8;
                x[3] = (r1 - r0) * K_PI2_8;
                r0 = x[4] - y[4];   r1 = x[5] - y[5];
                x[4] = r1 * K_PI1_8 + r0 * K_PI3_8;
                x[5] = r1 * K_PI3_8 - r0 * K_PI1_8;
                r0 = y[6] - x[6];   r1 = x[7] - y[7];
                x[6] = r1;  x[7] = r0;
                r0 = y[8] - x[8];   r1 = y[9] - x[9];
                x[8] = r0 * K_PI3_8 - r1 * K_PI1_8;
                x[9] = r1 * K_PI3_8 + r0 * K_PI1_8;
                r0 = y[10] - x[10]; r1 = y[11] - x[11];
                x[10] = (r0 - r1) * K_PI2_8;
                x[11] = (r0 + r1) * K_PI2_8;
                r0 = y[12] - x[12]; r1 = y[13] - x[13];
                x[12] = r0 * K_PI1_8 - r1 * K_PI3_8;
                x[13] = r0 * K_PI3_8 + r1 * K_PI1_8;
                r0 = y[14] - x[14]; r1 = y[15] - x[15];
                x[14] = r0;  x[15] = r1;
            }
        }
        bfly16(x);
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; k++) sx[9 * lane + k] = make_float2(x[2 * k], x[2 * k + 1]);
        wave_lds_sync();

}

template <int N, bool GATHER = false>
__global__ __launch_bounds__(64 * WAVES_PER_WG)
void k_window_mdct(const float *__restrict__ pcm, float *__restrict__ out,
                   const uint8_t *__restrict__ wflags,  // per block: bit0 = lW, bit1 = nW (N == 2048 only; may be null = all long)
                   const float *__restrict__ trig_g,    // N + N/4 floats, lib/mdct.c:67-76
                   const float *__restrict__ win_self,  // rising half-window of size N   (N/2 floats)
                   const float *__restrict__ win_short, // rising half-window of the short size (short_n/2 floats), N == 2048 only
                   int short_n, int apply_window,       // 0 none, 1 Vorbis window (lW/nW shapes), 2 = win_self is a full N-sample table
                   long nblocks, vbm_ve_gather gather,
                   const int *__restrict__ d_live, int live_mult)   // device-resident count: only *d_live * live_mult blocks exist
{
    if (d_live) {
        const long live = (long)*d_live * live_mult;
        if (live < nblocks) nblocks = live;
    }
    using G = Geo<N>;
    constexpr int C = G::C;
    constexpr int NTRIG = N + N / 4;

    __shared__ __attribute__((aligned(16))) float s_trig[NTRIG];
    __shared__ __attribute__((aligned(16))) float s_win[(N == 128) ? N : N / 2];
    __shared__ __attribute__((aligned(16))) float s_wshort[N / 2];   // rising half-window of the short size, long blocks
    __shared__ __attribute__((aligned(16))) float2 s_x[WAVES_PER_WG][SLOTS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    for (int i = tid; i < NTRIG; i += blockDim.x) s_trig[i] = trig_g[i];
    if (apply_window) {
        const int wn = (apply_window == 2 && N == 128) ? N : N / 2;
        for (int i = tid; i < wn; i += blockDim.x) s_win[i] = win_self[i];
        if (wflags)
            for (int i = tid; i < (short_n >> 1); i += blockDim.x) s_wshort[i] = win_short[i];
    }
    __syncthreads();

    float2 *sx = s_x[wave];
    const long ngroups = (nblocks + G::BPG - 1) / G::BPG;
    const long gstride = (long)gridDim.x * WAVES_PER_WG;
    long group = (long)blockIdx.x * WAVES_PER_WG + wave;
    const float scale = 4.f / N;

    float4 v[8];
    bool have = false;
    if (group < ngroups) have = issue_loads<N, GATHER>(v, pcm, group, nblocks, lane, gather);

    for (; group < ngroups; group += gstride) {
        if (!have) {
            // none of the group's blocks exists (the envelope search covers 32 steps per channel and launch, a
            // 1024-sample write brings 16): nothing to transform, nothing to store
            const long nx = group + gstride;
            have = (nx < ngroups) ? issue_loads<N, GATHER>(v, pcm, nx, nblocks, lane, gather) : false;
            continue;
        }
        // ---------------- window + odd-sample exchange -----------------------------
        float2 ev[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int P = lane + 64 * k;
            int b = P / C, p = P % C;
            int q0 = (p < G::R2) ? (G::R2 - 1 - p) : (7 * N / 16 - 1 - p);
            float4 d = v[k];
            if (apply_window == 2 && N == 128) {
                float4 w = *reinterpret_cast<const float4 *>(s_win + 4 * q0);   // lib/envelope.c:124
                d = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
            } else if (apply_window) {
                int ln = N, rn = N;
                const float *wl = s_win, *wr = s_win;
                if (wflags) {
                    long blk = group * G::BPG + b;
                    int f = (blk < nblocks) ? wflags[blk] : 3;
                    if (!(f & 1)) { ln = short_n; wl = s_wshort; }
                    if (!(f & 2)) { rn = short_n; wr = s_wshort; }
                }
                d = window4(d, 4 * q0, N, ln, rn, wl, wr);
            }
            ev[k] = make_float2(d.x, d.z);  // x0[0], x0[2] of this lane's own pair
            // odd samples feed the pair whose x1 pointer lands on this 4-sample group
            int pp = (q0 >= G::R2) ? (q0 - G::R2) : (q0 + G::R1);
            sx[slot_addr(b * C + pp)] = make_float2(d.y, d.w);
        }
        wave_lds_sync();

        // prefetch the next group while this one is transformed
        const long next = group + gstride;
        // ---------------- fold + pre-twiddle (lib/mdct.c:1819-1851) -------------------
        float2 c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int P = lane + 64 * k;
            int b = P / C, p = P % C;
            float2 od = sx[slot_addr(b * C + p)];  // x1[0], x1[2]
            float a0 = ev[k].y, a1 = ev[k].x;      // x0[2], x0[0]
            if (p >= G::R2) { a0 = -a0; a1 = -a1; }
            float r0, r1;
            if (p < G::R1) { r0 = a0 + od.x; r1 = a1 + od.y; }
            else           { r0 = a0 - od.x; r1 = a1 - od.y; }
            float2 T = *reinterpret_cast<const float2 *>(s_trig + (N / 2 - 2 * (p + 1)));
            c[k].x = r1 * T.y + r0 * T.x;
            c[k].y = r1 * T.x - r0 * T.y;
        }
        have = (next < ngroups) ? issue_loads<N, GATHER>(v, pcm, next, nblocks, lane, gather) : false;
        wave_lds_sync();  // exchange slots are reused below

        radix_rounds<G::LOG2C>(c, sx, s_trig, lane);

        // ---------------- bit-reverse gather + post-twiddle + store -------------------
        {
            constexpr int HALFC = C / 2;
            const int U0 = 4 * lane;
            const int b = U0 / HALFC;
            const int u0 = U0 % HALFC;
            const long blk = group * G::BPG + b;
            float oA0[4], oA1[4], oB0[4], oB1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int u = u0 + r;
                int rv = (int)(__brev((unsigned)u) >> (32 - (G::LOG2C - 1)));
                int s1 = 2 * rv;              // bitrev[2u+1] / 2
                int s0 = (C - 1) - 2 * rv;    // bitrev[2u]   / 2
                float2 X0 = sx[slot_addr(b * C + s0)];
                float2 X1 = sx[slot_addr(b * C + s1)];
                float2 T = *reinterpret_cast<const float2 *>(s_trig + N + 2 * u);
                float r0 = X0.y - X1.y;
                float r1 = X0.x + X1.x;
                float r2 = r1 * T.x + r0 * T.y;
                float r3 = r1 * T.y - r0 * T.x;
                float h0 = (X0.y + X1.y) * .5f;
                float h1 = (X0.x - X1.x) * .5f;
                float2 wA = make_float2(h0 + r2, h1 + r3);   // w pair u
                float2 wB = make_float2(h0 - r2, r3 - h1);   // w pair C-1-u
                float2 TA = *reinterpret_cast<const float2 *>(s_trig + N / 2 + 2 * u);
                float2 TB = *reinterpret_cast<const float2 *>(s_trig + N / 2 + 2 * (C - 1 - u));
                oA0[r] = (wA.x * TA.x + wA.y * TA.y) * scale;   // out[u]
                oA1[r] = (wA.x * TA.y - wA.y * TA.x) * scale;   // out[n2-1-u]
                oB0[r] = (wB.x * TB.x + wB.y * TB.y) * scale;   // out[C-1-u]
                oB1[r] = (wB.x * TB.y - wB.y * TB.x) * scale;   // out[C+u]
            }
            if (blk < nblocks) {
                float *o = out + blk * (N / 2);
                *reinterpret_cast<float4 *>(o + u0) = make_float4(oA0[0], oA0[1], oA0[2], oA0[3]);
                *reinterpret_cast<float4 *>(o + C + u0) = make_float4(oB1[0], oB1[1], oB1[2], oB1[3]);
                *reinterpret_cast<float4 *>(o + C - 4 - u0) = make_float4(oB0[3], oB0[2], oB0[1], oB0[0]);
                *reinterpret_cast<float4 *>(o + 2 * C - 4 - u0) = make_float4(oA1[3], oA1[2], oA1[1], oA1[0]);
            }
        }
        wave_lds_sync();
    }
}

// 4096-sample blocks (the long blocks of q < 0 at 44.1/48 kHz): 1024 complex values = two 512-complex
// groups.  One wavefront per block: both halves are loaded, windowed and folded (the odd samples cross
// between the halves through a 1024-slot LDS exchange), the first butterfly stage pairs element m with
// m + 512 — both live in the same lane — and each half then runs the group rounds with the trig strides of
// a 1024-complex transform; the bit-reverse gather reads across both halves.
__global__ __launch_bounds__(64 * WAVES_PER_WG)
void k_window_mdct_4096(const float *__restrict__ pcm, float *__restrict__ out, const uint8_t *__restrict__ wflags,
                        const float *__restrict__ trig_g, const float *__restrict__ win_self,
                        const float *__restrict__ win_short, int short_n, int apply_window, long nblocks,
                        const int *__restrict__ d_live, int live_mult)
{
    if (d_live) {
        const long live = (long)*d_live * live_mult;
        if (live < nblocks) nblocks = live;
    }
    constexpr int N = 4096, C = 1024, R1 = N / 16, R2 = 3 * N / 16, NTRIG = N + N / 4;
    __shared__ __attribute__((aligned(16))) float s_trig[NTRIG];
    __shared__ __attribute__((aligned(16))) float s_win[N / 2];
    __shared__ __attribute__((aligned(16))) float s_wshort[N / 8];
    __shared__ __attribute__((aligned(16))) float2 s_x[WAVES_PER_WG][2][SLOTS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NTRIG; i += blockDim.x) s_trig[i] = trig_g[i];
    if (apply_window) {
        for (int i = tid; i < N / 2; i += blockDim.x) s_win[i] = win_self[i];
        if (wflags)
            for (int i = tid; i < (short_n >> 1) && i < N / 8; i += blockDim.x) s_wshort[i] = win_short[i];
    }
    __syncthreads();

    float2 *sx0 = s_x[wave][0], *sx1 = s_x[wave][1];
#define SLOT(m) (((m) < 512 ? sx0 : sx1)[slot_addr((m) & 511)])
    const float scale = 4.f / N;
    for (long blk = (long)blockIdx.x * WAVES_PER_WG + wave; blk < nblocks; blk += (long)gridDim.x * WAVES_PER_WG) {
        int ln = N, rn = N;
        const float *wl = s_win, *wr = s_win;
        if (apply_window && wflags) {
            const int f = wflags[blk];
            if (!(f & 1)) { ln = short_n; wl = s_wshort; }
            if (!(f & 2)) { rn = short_n; wr = s_wshort; }
        }
        // ---------------- load, window, odd-sample exchange ---------------------------------
        float2 ev[16];
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const int p = lane + 64 * kk;   // pair index 0..1023 (kk < 8: first half)
            const int q0 = (p < R2) ? (R2 - 1 - p) : (7 * N / 16 - 1 - p);
            float4 d = *reinterpret_cast<const float4 *>(pcm + blk * N + 4 * q0);
            if (apply_window) d = window4(d, 4 * q0, N, ln, rn, wl, wr);
            ev[kk] = make_float2(d.x, d.z);
            const int pp = (q0 >= R2) ? (q0 - R2) : (q0 + R1);
            SLOT(pp) = make_float2(d.y, d.w);
        }
        wave_lds_sync();
        // ---------------- fold + pre-twiddle (lib/mdct.c:1819-1851) -------------------------
        float2 c0[8], c1[8];
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const int p = lane + 64 * kk;
            const float2 od = SLOT(p);
            float a0 = ev[kk].y, a1 = ev[kk].x;
            if (p >= R2) { a0 = -a0; a1 = -a1; }
            float r0, r1;
            if (p < R1) { r0 = a0 + od.x; r1 = a1 + od.y; }
            else        { r0 = a0 - od.x; r1 = a1 - od.y; }
            const float2 T = *reinterpret_cast<const float2 *>(s_trig + (N / 2 - 2 * (p + 1)));
            float2 cv;
            cv.x = r1 * T.y + r0 * T.x;
            cv.y = r1 * T.x - r0 * T.y;
            if (kk < 8) c0[kk] = cv;
            else c1[kk - 8] = cv;
        }
        wave_lds_sync();
        // ---------------- first stage: index bit 9, trigint 4 --------------------------------
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = 511 - (lane + 64 * k);
            bfly(c0[k], c1[k], *reinterpret_cast<const float2 *>(s_trig + 4 * t));
        }
        radix_rounds<10>(c0, sx0, s_trig, lane);
        radix_rounds<10>(c1, sx1, s_trig, lane);
        // ---------------- bit-reverse gather + post-twiddle + store --------------------------
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int u0 = 4 * lane + 256 * pass;
            float oA0[4], oA1[4], oB0[4], oB1[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int u = u0 + r;
                const int rv = (int)(__brev((unsigned)u) >> (32 - 9));
                const int s1 = 2 * rv, s0 = (C - 1) - 2 * rv;
                const float2 X0 = SLOT(s0), X1 = SLOT(s1);
                const float2 T = *reinterpret_cast<const float2 *>(s_trig + N + 2 * u);
                const float r0 = X0.y - X1.y;
                const float r1 = X0.x + X1.x;
                const float r2 = r1 * T.x + r0 * T.y;
                const float r3 = r1 * T.y - r0 * T.x;
                const float h0 = (X0.y + X1.y) * .5f;
                const float h1 = (X0.x - X1.x) * .5f;
                const float2 wA = make_float2(h0 + r2, h1 + r3);
                const float2 wB = make_float2(h0 - r2, r3 - h1);
                const float2 TA = *reinterpret_cast<const float2 *>(s_trig + N / 2 + 2 * u);
                const float2 TB = *reinterpret_cast<const float2 *>(s_trig + N / 2 + 2 * (C - 1 - u));
                oA0[r] = (wA.x * TA.x + wA.y * TA.y) * scale;
                oA1[r] = (wA.x * TA.y - wA.y * TA.x) * scale;
                oB0[r] = (wB.x * TB.x + wB.y * TB.y) * scale;
                oB1[r] = (wB.x * TB.y - wB.y * TB.x) * scale;
            }
            float *o = out + blk * (N / 2);
            *reinterpret_cast<float4 *>(o + u0) = make_float4(oA0[0], oA0[1], oA0[2], oA0[3]);
            *reinterpret_cast<float4 *>(o + C + u0) = make_float4(oB1[0], oB1[1], oB1[2], oB1[3]);
            *reinterpret_cast<float4 *>(o + C - 4 - u0) = make_float4(oB0[3], oB0[2], oB0[1], oB0[0]);
            *reinterpret_cast<float4 *>(o + 2 * C - 4 - u0) = make_float4(oA1[3], oA1[2], oA1[1], oA1[0]);
        }
        wave_lds_sync();
    }
#undef SLOT
}

}  // namespace

extern "C" int vbm_launch_window_mdct(const float *d_pcm, float *d_out, const uint8_t *d_wflags,
                                      const float *d_trig, const float *d_win_self,
                                      const float *d_win_short, int n, int short_n,
                                      int apply_window, long nblocks, int max_workgroups,
                                      const int *d_live, int live_mult, hipStream_t stream)
{
    if (nblocks <= 0) return 0;
    if (n == 4096) {
        long wgs4 = (nblocks + WAVES_PER_WG - 1) / WAVES_PER_WG;
        if (max_workgroups <= 0) max_workgroups = 256 * 4;
        if (wgs4 > max_workgroups) wgs4 = max_workgroups;
        hipLaunchKernelGGL(k_window_mdct_4096, dim3((unsigned)wgs4), dim3(64 * WAVES_PER_WG), 0, stream, d_pcm, d_out,
                           d_wflags, d_trig, d_win_self, d_win_short, short_n, apply_window, nblocks, d_live, live_mult);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (n != 2048 && n != 1024 && n != 512 && n != 256) return -1;
    const vbm_ve_gather none = {};
    const int bpg = 2048 / n;
    long ngroups = (nblocks + bpg - 1) / bpg;
    long wgs = (ngroups + WAVES_PER_WG - 1) / WAVES_PER_WG;
    if (max_workgroups <= 0) max_workgroups = 256 * 4;
    if (wgs > max_workgroups) wgs = max_workgroups;
    dim3 grid((unsigned)wgs), block(64 * WAVES_PER_WG);
#define LAUNCH_MDCT(NN)                                                                                        \
    hipLaunchKernelGGL(k_window_mdct<NN>, grid, block, 0, stream, d_pcm, d_out, d_wflags, d_trig, d_win_self, \
                       d_win_short, short_n, apply_window, nblocks, none, d_live, live_mult)
    if (n == 2048) LAUNCH_MDCT(2048);
    else if (n == 1024) LAUNCH_MDCT(1024);
    else if (n == 512) LAUNCH_MDCT(512);
    else LAUNCH_MDCT(256);
#undef LAUNCH_MDCT
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// 128-point search MDCTs of the envelope detector (lib/envelope.c:121-125): blocks are addressed
// through `g` (channel * g.steps + t), out[blk][64]
extern "C" int vbm_launch_ve_mdct(const vbm_ve_gather *g, float *d_out, const float *d_trig, const float *d_win,
                                  long nblocks, hipStream_t stream)
{
    if (nblocks <= 0) return 0;
    long ngroups = (nblocks + 15) / 16;
    long wgs = (ngroups + WAVES_PER_WG - 1) / WAVES_PER_WG;
    if (wgs > 256 * 8) wgs = 256 * 8;
    hipLaunchKernelGGL((k_window_mdct<128, true>), dim3((unsigned)wgs), dim3(64 * WAVES_PER_WG), 0, stream, g->pcm, d_out,
                       (const uint8_t *)nullptr, d_trig, d_win, (const float *)nullptr, 0, 2, nblocks, *g, (const int *)nullptr, 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
