// Window + real FFT + log-power spectrum for gfx950, one wavefront per block.
//
// Replaces, for a batch of blocks, the tonal-estimation half of mapping0_forward loop A
// (reference lib/mapping0.c:825-888): _vorbis_apply_window, drft_forward (FFTPACK radix 4/2,
// lib/smallft.c:5652-5807, 6111-6170), then
//     logfft[0]        = scale_dB + todB(re0)            + .345
//     logfft[(j+1)>>1] = scale_dB + .5f*todB(re^2+im^2)  + .345      (j = 1,3,..,n-3)
//     local_ampmax     = min(0, max logfft)
// Bit-identical to the scalar reference: the FFTPACK pass structure (factors applied last
// to first) and every butterfly expression are kept; only the schedule changes — each pass's
// independent butterflies are spread over the 64 lanes, the block lives in LDS (round 3: ONE
// buffer per wavefront, a pass reads all its inputs into registers before it writes — the
// source's ping-pong between two buffers is kept for 4096-sample blocks only), twiddles (computed on the host with libm exactly
// as drfti1 does, lib/smallft.c:5629-5640) are LDS-resident.  `+ .345` is a double add
// rounded to float, as in the C source.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mdct_kernel.h"

namespace {

constexpr int WAVES_PER_WG = 4;

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// lib/scales.h:43-51
__device__ __forceinline__ float todB(float x)
{
    uint32_t i = __float_as_uint(x) & 0x7fffffffu;
    return (float)((float)i * 7.17711438e-7f - 764.6161886f);
}

// ---- FFTPACK forward passes, CC(i,k,j) = cc[i + ido*(k + l1*j)], CH(i,j,k) = ch[i + ido*(j + R*k)]
template <int IDO, int L1>
__device__ __forceinline__ void radf4_pass(const float *cc, float *ch, const float *wa1, const float *wa2,
                                           const float *wa3, int lane)
{
#define CC(i, k, j) cc[(i) + IDO * ((k) + L1 * (j))]
#define CH(i, j, k) ch[(i) + IDO * ((j) + 4 * (k))]
    for (int k = lane; k < L1; k += 64) {
        float tr1 = CC(0, k, 1) + CC(0, k, 3);
        float tr2 = CC(0, k, 0) + CC(0, k, 2);
        CH(0, 0, k) = tr1 + tr2;
        CH(IDO - 1, 3, k) = tr2 - tr1;
        CH(IDO - 1, 1, k) = CC(0, k, 0) - CC(0, k, 2);
        CH(0, 2, k) = CC(0, k, 3) - CC(0, k, 1);
    }
    if (IDO < 2) return;
    if (IDO > 2) {
        constexpr int PER_K = IDO / 2 - 1;
        constexpr int TOTAL = L1 * PER_K;
        for (int t = lane; t < TOTAL; t += 64) {
            int k = t / PER_K;
            int i = 2 + 2 * (t - k * PER_K);
            int ic = IDO - i;
            float cr2 = wa1[i - 2] * CC(i - 1, k, 1) + wa1[i - 1] * CC(i, k, 1);
            float ci2 = wa1[i - 2] * CC(i, k, 1) - wa1[i - 1] * CC(i - 1, k, 1);
            float cr3 = wa2[i - 2] * CC(i - 1, k, 2) + wa2[i - 1] * CC(i, k, 2);
            float ci3 = wa2[i - 2] * CC(i, k, 2) - wa2[i - 1] * CC(i - 1, k, 2);
            float cr4 = wa3[i - 2] * CC(i - 1, k, 3) + wa3[i - 1] * CC(i, k, 3);
            float ci4 = wa3[i - 2] * CC(i, k, 3) - wa3[i - 1] * CC(i - 1, k, 3);
            float tr1 = cr2 + cr4, tr4 = cr4 - cr2;
            float ti1 = ci2 + ci4, ti4 = ci2 - ci4;
            float ti2 = CC(i, k, 0) + ci3, ti3 = CC(i, k, 0) - ci3;
            float tr2 = CC(i - 1, k, 0) + cr3, tr3 = CC(i - 1, k, 0) - cr3;
            CH(i - 1, 0, k) = tr1 + tr2;
            CH(i, 0, k) = ti1 + ti2;
            CH(ic - 1, 1, k) = tr3 - ti4;
            CH(ic, 1, k) = tr4 - ti3;
            CH(i - 1, 2, k) = ti4 + tr3;
            CH(i, 2, k) = tr4 + ti3;
            CH(ic - 1, 3, k) = tr2 - tr1;
            CH(ic, 3, k) = ti1 - ti2;
        }
        if (IDO & 1) return;
    }
    constexpr float hsqt2 = .70710678118654752f;
    for (int k = lane; k < L1; k += 64) {
        float ti1 = -hsqt2 * (CC(IDO - 1, k, 1) + CC(IDO - 1, k, 3));
        float tr1 = hsqt2 * (CC(IDO - 1, k, 1) - CC(IDO - 1, k, 3));
        CH(IDO - 1, 0, k) = tr1 + CC(IDO - 1, k, 0);
        CH(IDO - 1, 2, k) = CC(IDO - 1, k, 0) - tr1;
        CH(0, 1, k) = ti1 - CC(IDO - 1, k, 2);
        CH(0, 3, k) = ti1 + CC(IDO - 1, k, 2);
    }
#undef CH
#undef CC
}

template <int IDO, int L1>
__device__ __forceinline__ void radf2_pass(const float *cc, float *ch, const float *wa1, int lane)
{
#define CC(i, k, j) cc[(i) + IDO * ((k) + L1 * (j))]
#define CH(i, j, k) ch[(i) + IDO * ((j) + 2 * (k))]
    for (int k = lane; k < L1; k += 64) {
        CH(0, 0, k) = CC(0, k, 0) + CC(0, k, 1);
        CH(IDO - 1, 1, k) = CC(0, k, 0) - CC(0, k, 1);
    }
    if (IDO < 2) return;
    if (IDO > 2) {
        constexpr int PER_K = IDO / 2 - 1;
        constexpr int TOTAL = L1 * PER_K;
        for (int t = lane; t < TOTAL; t += 64) {
            int k = t / PER_K;
            int i = 2 + 2 * (t - k * PER_K);
            int ic = IDO - i;
            float tr2 = wa1[i - 2] * CC(i - 1, k, 1) + wa1[i - 1] * CC(i, k, 1);
            float ti2 = wa1[i - 2] * CC(i, k, 1) - wa1[i - 1] * CC(i - 1, k, 1);
            CH(i, 0, k) = CC(i, k, 0) + ti2;
            CH(ic, 1, k) = ti2 - CC(i, k, 0);
            CH(i - 1, 0, k) = CC(i - 1, k, 0) + tr2;
            CH(ic - 1, 1, k) = CC(i - 1, k, 0) - tr2;
        }
        if (IDO % 2 == 1) return;
    }
    for (int k = lane; k < L1; k += 64) {
        CH(0, 1, k) = -CC(IDO - 1, k, 1);
        CH(IDO - 1, 0, k) = CC(IDO - 1, k, 0);
    }
#undef CH
#undef CC
}

// ---- the same passes IN PLACE (round 3).  A pass is a permutation-with-arithmetic of the whole buffer: every lane first
// reads the inputs of ALL its butterflies into registers (at most 40 floats for n = 2048), then computes and writes.  One
// wavefront owns the buffer and its LDS instructions execute in order, so every read of the pass precedes every write:
// no second buffer.  8 KB of LDS per 2048-sample block instead of 16: three workgroups of four wavefronts per CU instead
// of two.  Every butterfly expression is the one above, operand for operand.
template <int IDO, int L1>
__device__ __forceinline__ void radf4_inplace(float *c, const float *wa1, const float *wa2, const float *wa3, int lane)
{
#define CC(i, k, j) c[(i) + IDO * ((k) + L1 * (j))]
#define CH(i, j, k) c[(i) + IDO * ((j) + 4 * (k))]
    constexpr int KI = (L1 + 63) / 64;
    constexpr int PER_K = IDO > 2 ? IDO / 2 - 1 : 1;
    constexpr int TOTAL = IDO > 2 ? L1 * PER_K : 0;
    constexpr int TI = IDO > 2 ? (TOTAL + 63) / 64 : 1;
    constexpr bool LAST = IDO >= 2 && !(IDO & 1);
    float a[KI][4], z[KI][4], g[TI][8];
    // ---- reads
#pragma unroll
    for (int q = 0; q < KI; q++) {
        const int k = lane + 64 * q;
        if (k < L1) {
#pragma unroll
            for (int j = 0; j < 4; j++) a[q][j] = CC(0, k, j);
            if (LAST)
#pragma unroll
                for (int j = 0; j < 4; j++) z[q][j] = CC(IDO - 1, k, j);
        }
    }
    if (IDO > 2) {
#pragma unroll
        for (int q = 0; q < TI; q++) {
            const int t = lane + 64 * q;
            if (t < TOTAL) {
                const int k = t / PER_K;
                const int i = 2 + 2 * (t - k * PER_K);
#pragma unroll
                for (int j = 0; j < 4; j++) { g[q][2 * j] = CC(i - 1, k, j); g[q][2 * j + 1] = CC(i, k, j); }
            }
        }
    }
    wave_lds_sync();
    // ---- butterflies and writes
#pragma unroll
    for (int q = 0; q < KI; q++) {
        const int k = lane + 64 * q;
        if (k < L1) {
            float tr1 = a[q][1] + a[q][3];
            float tr2 = a[q][0] + a[q][2];
            CH(0, 0, k) = tr1 + tr2;
            CH(IDO - 1, 3, k) = tr2 - tr1;
            CH(IDO - 1, 1, k) = a[q][0] - a[q][2];
            CH(0, 2, k) = a[q][3] - a[q][1];
        }
    }
    if (IDO > 2) {
#pragma unroll
        for (int q = 0; q < TI; q++) {
            const int t = lane + 64 * q;
            if (t < TOTAL) {
                const int k = t / PER_K;
                const int i = 2 + 2 * (t - k * PER_K);
                const int ic = IDO - i;
                const float c0r = g[q][0], c0i = g[q][1], c1r = g[q][2], c1i = g[q][3], c2r = g[q][4], c2i = g[q][5],
                            c3r = g[q][6], c3i = g[q][7];                 // CC(i-1,k,j), CC(i,k,j)
                float cr2 = wa1[i - 2] * c1r + wa1[i - 1] * c1i;
                float ci2 = wa1[i - 2] * c1i - wa1[i - 1] * c1r;
                float cr3 = wa2[i - 2] * c2r + wa2[i - 1] * c2i;
                float ci3 = wa2[i - 2] * c2i - wa2[i - 1] * c2r;
                float cr4 = wa3[i - 2] * c3r + wa3[i - 1] * c3i;
                float ci4 = wa3[i - 2] * c3i - wa3[i - 1] * c3r;
                float tr1 = cr2 + cr4, tr4 = cr4 - cr2;
                float ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                float ti2 = c0i + ci3, ti3 = c0i - ci3;
                float tr2 = c0r + cr3, tr3 = c0r - cr3;
                CH(i - 1, 0, k) = tr1 + tr2;
                CH(i, 0, k) = ti1 + ti2;
                CH(ic - 1, 1, k) = tr3 - ti4;
                CH(ic, 1, k) = tr4 - ti3;
                CH(i - 1, 2, k) = ti4 + tr3;
                CH(i, 2, k) = tr4 + ti3;
                CH(ic - 1, 3, k) = tr2 - tr1;
                CH(ic, 3, k) = ti1 - ti2;
            }
        }
    }
    if (LAST) {
        constexpr float hsqt2 = .70710678118654752f;
#pragma unroll
        for (int q = 0; q < KI; q++) {
            const int k = lane + 64 * q;
            if (k < L1) {
                float ti1 = -hsqt2 * (z[q][1] + z[q][3]);
                float tr1 = hsqt2 * (z[q][1] - z[q][3]);
                CH(IDO - 1, 0, k) = tr1 + z[q][0];
                CH(IDO - 1, 2, k) = z[q][0] - tr1;
                CH(0, 1, k) = ti1 - z[q][2];
                CH(0, 3, k) = ti1 + z[q][2];
            }
        }
    }
#undef CH
#undef CC
}

template <int IDO, int L1>
__device__ __forceinline__ void radf2_inplace(float *c, const float *wa1, int lane)
{
#define CC(i, k, j) c[(i) + IDO * ((k) + L1 * (j))]
#define CH(i, j, k) c[(i) + IDO * ((j) + 2 * (k))]
    constexpr int KI = (L1 + 63) / 64;
    constexpr int PER_K = IDO > 2 ? IDO / 2 - 1 : 1;
    constexpr int TOTAL = IDO > 2 ? L1 * PER_K : 0;
    constexpr int TI = IDO > 2 ? (TOTAL + 63) / 64 : 1;
    constexpr bool LAST = IDO >= 2 && !(IDO & 1);
    float a[KI][2], z[KI][2], g[TI][4];
#pragma unroll
    for (int q = 0; q < KI; q++) {
        const int k = lane + 64 * q;
        if (k < L1) {
            a[q][0] = CC(0, k, 0); a[q][1] = CC(0, k, 1);
            if (LAST) { z[q][0] = CC(IDO - 1, k, 0); z[q][1] = CC(IDO - 1, k, 1); }
        }
    }
    if (IDO > 2) {
#pragma unroll
        for (int q = 0; q < TI; q++) {
            const int t = lane + 64 * q;
            if (t < TOTAL) {
                const int k = t / PER_K;
                const int i = 2 + 2 * (t - k * PER_K);
                g[q][0] = CC(i - 1, k, 0); g[q][1] = CC(i, k, 0); g[q][2] = CC(i - 1, k, 1); g[q][3] = CC(i, k, 1);
            }
        }
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < KI; q++) {
        const int k = lane + 64 * q;
        if (k < L1) {
            CH(0, 0, k) = a[q][0] + a[q][1];
            CH(IDO - 1, 1, k) = a[q][0] - a[q][1];
        }
    }
    if (IDO > 2) {
#pragma unroll
        for (int q = 0; q < TI; q++) {
            const int t = lane + 64 * q;
            if (t < TOTAL) {
                const int k = t / PER_K;
                const int i = 2 + 2 * (t - k * PER_K);
                const int ic = IDO - i;
                float tr2 = wa1[i - 2] * g[q][2] + wa1[i - 1] * g[q][3];
                float ti2 = wa1[i - 2] * g[q][3] - wa1[i - 1] * g[q][2];
                CH(i, 0, k) = g[q][1] + ti2;
                CH(ic, 1, k) = ti2 - g[q][1];
                CH(i - 1, 0, k) = g[q][0] + tr2;
                CH(ic - 1, 1, k) = g[q][0] - tr2;
            }
        }
    }
    if (LAST) {
#pragma unroll
        for (int q = 0; q < KI; q++) {
            const int k = lane + 64 * q;
            if (k < L1) {
                CH(0, 1, k) = -z[q][1];
                CH(IDO - 1, 0, k) = z[q][0];
            }
        }
    }
#undef CH
#undef CC
}

// the pass sequences of fft_forward<N> below, in place (result in c)
template <int N>
__device__ __forceinline__ void fft_forward_inplace(float *c, const float *wa, int lane);

template <>
__device__ __forceinline__ void fft_forward_inplace<2048>(float *c, const float *wa, int lane)
{
    radf4_inplace<1, 512>(c, wa + 2044, wa + 2045, wa + 2046, lane);  wave_lds_sync();
    radf4_inplace<4, 128>(c, wa + 2032, wa + 2036, wa + 2040, lane);  wave_lds_sync();
    radf4_inplace<16, 32>(c, wa + 1984, wa + 2000, wa + 2016, lane);  wave_lds_sync();
    radf4_inplace<64, 8>(c, wa + 1792, wa + 1856, wa + 1920, lane);   wave_lds_sync();
    radf4_inplace<256, 2>(c, wa + 1024, wa + 1280, wa + 1536, lane);  wave_lds_sync();
    radf2_inplace<1024, 1>(c, wa + 0, lane);                          wave_lds_sync();
}

template <>
__device__ __forceinline__ void fft_forward_inplace<1024>(float *c, const float *wa, int lane)
{
    radf4_inplace<1, 256>(c, wa + 1020, wa + 1021, wa + 1022, lane);  wave_lds_sync();
    radf4_inplace<4, 64>(c, wa + 1008, wa + 1012, wa + 1016, lane);   wave_lds_sync();
    radf4_inplace<16, 16>(c, wa + 960, wa + 976, wa + 992, lane);     wave_lds_sync();
    radf4_inplace<64, 4>(c, wa + 768, wa + 832, wa + 896, lane);      wave_lds_sync();
    radf4_inplace<256, 1>(c, wa + 0, wa + 256, wa + 512, lane);       wave_lds_sync();
}

template <>
__device__ __forceinline__ void fft_forward_inplace<512>(float *c, const float *wa, int lane)
{
    radf4_inplace<1, 128>(c, wa + 508, wa + 509, wa + 510, lane);     wave_lds_sync();
    radf4_inplace<4, 32>(c, wa + 496, wa + 500, wa + 504, lane);      wave_lds_sync();
    radf4_inplace<16, 8>(c, wa + 448, wa + 464, wa + 480, lane);      wave_lds_sync();
    radf4_inplace<64, 2>(c, wa + 256, wa + 320, wa + 384, lane);      wave_lds_sync();
    radf2_inplace<256, 1>(c, wa + 0, lane);                           wave_lds_sync();
}

template <>
__device__ __forceinline__ void fft_forward_inplace<256>(float *c, const float *wa, int lane)
{
    radf4_inplace<1, 64>(c, wa + 252, wa + 253, wa + 254, lane);      wave_lds_sync();
    radf4_inplace<4, 16>(c, wa + 240, wa + 244, wa + 248, lane);      wave_lds_sync();
    radf4_inplace<16, 4>(c, wa + 192, wa + 208, wa + 224, lane);      wave_lds_sync();
    radf4_inplace<64, 1>(c, wa + 0, wa + 64, wa + 128, lane);         wave_lds_sync();
}

// drftf1 (lib/smallft.c:6111-6170) unrolled for the two factorisations in use:
//   n = 2048: ifac = {2,4,4,4,4,4}  ->  passes (ip,l1,ido) = (4,512,1) (4,128,4) (4,32,16) (4,8,64) (4,2,256) (2,1,1024)
//   n = 256 : ifac = {4,4,4,4}      ->  passes (4,64,1) (4,16,4) (4,4,16) (4,1,64)
// wa offsets: iw starts at n and drops by (ip-1)*ido per pass; pointers are wa + iw - 1.
template <int N>
__device__ __forceinline__ void fft_forward(float *c, float *ch, const float *wa, int lane);

template <>
__device__ __forceinline__ void fft_forward<2048>(float *c, float *ch, const float *wa, int lane)
{
    // pass 0: ip 4, l1 512, ido 1, iw = 2048-3 = 2045
    radf4_pass<1, 512>(c, ch, wa + 2044, wa + 2045, wa + 2046, lane);
    wave_lds_sync();
    // pass 1: ip 4, l1 128, ido 4, iw = 2045-12 = 2033
    radf4_pass<4, 128>(ch, c, wa + 2032, wa + 2036, wa + 2040, lane);
    wave_lds_sync();
    // pass 2: ip 4, l1 32, ido 16, iw = 2033-48 = 1985
    radf4_pass<16, 32>(c, ch, wa + 1984, wa + 2000, wa + 2016, lane);
    wave_lds_sync();
    // pass 3: ip 4, l1 8, ido 64, iw = 1985-192 = 1793
    radf4_pass<64, 8>(ch, c, wa + 1792, wa + 1856, wa + 1920, lane);
    wave_lds_sync();
    // pass 4: ip 4, l1 2, ido 256, iw = 1793-768 = 1025
    radf4_pass<256, 2>(c, ch, wa + 1024, wa + 1280, wa + 1536, lane);
    wave_lds_sync();
    // pass 5: ip 2, l1 1, ido 1024, iw = 1025-1024 = 1
    radf2_pass<1024, 1>(ch, c, wa + 0, lane);
    wave_lds_sync();
}

template <>
__device__ __forceinline__ void fft_forward<256>(float *c, float *ch, const float *wa, int lane)
{
    // pass 0: l1 64, ido 1, iw = 256-3 = 253
    radf4_pass<1, 64>(c, ch, wa + 252, wa + 253, wa + 254, lane);
    wave_lds_sync();
    // pass 1: l1 16, ido 4, iw = 253-12 = 241
    radf4_pass<4, 16>(ch, c, wa + 240, wa + 244, wa + 248, lane);
    wave_lds_sync();
    // pass 2: l1 4, ido 16, iw = 241-48 = 193
    radf4_pass<16, 4>(c, ch, wa + 192, wa + 208, wa + 224, lane);
    wave_lds_sync();
    // pass 3: l1 1, ido 64, iw = 193-192 = 1
    radf4_pass<64, 1>(ch, c, wa + 0, wa + 64, wa + 128, lane);
    wave_lds_sync();
}

// n = 1024: ifac = {4,4,4,4,4}   ->  passes (4,256,1) (4,64,4) (4,16,16) (4,4,64) (4,1,256); result in ch
// n = 512 : ifac = {2,4,4,4,4}   ->  passes (4,128,1) (4,32,4) (4,8,16) (4,2,64) (2,1,256);  result in ch
template <>
__device__ __forceinline__ void fft_forward<1024>(float *c, float *ch, const float *wa, int lane)
{
    radf4_pass<1, 256>(c, ch, wa + 1020, wa + 1021, wa + 1022, lane);    // iw = 1024-3 = 1021
    wave_lds_sync();
    radf4_pass<4, 64>(ch, c, wa + 1008, wa + 1012, wa + 1016, lane);     // iw = 1021-12 = 1009
    wave_lds_sync();
    radf4_pass<16, 16>(c, ch, wa + 960, wa + 976, wa + 992, lane);       // iw = 1009-48 = 961
    wave_lds_sync();
    radf4_pass<64, 4>(ch, c, wa + 768, wa + 832, wa + 896, lane);        // iw = 961-192 = 769
    wave_lds_sync();
    radf4_pass<256, 1>(c, ch, wa + 0, wa + 256, wa + 512, lane);         // iw = 769-768 = 1
    wave_lds_sync();
}

template <>
__device__ __forceinline__ void fft_forward<512>(float *c, float *ch, const float *wa, int lane)
{
    radf4_pass<1, 128>(c, ch, wa + 508, wa + 509, wa + 510, lane);       // iw = 512-3 = 509
    wave_lds_sync();
    radf4_pass<4, 32>(ch, c, wa + 496, wa + 500, wa + 504, lane);        // iw = 509-12 = 497
    wave_lds_sync();
    radf4_pass<16, 8>(c, ch, wa + 448, wa + 464, wa + 480, lane);        // iw = 497-48 = 449
    wave_lds_sync();
    radf4_pass<64, 2>(ch, c, wa + 256, wa + 320, wa + 384, lane);        // iw = 449-192 = 257
    wave_lds_sync();
    radf2_pass<256, 1>(c, ch, wa + 0, lane);                             // iw = 257-256 = 1
    wave_lds_sync();
}

// n = 4096: ifac = {4,4,4,4,4,4} ->  passes (4,1024,1) (4,256,4) (4,64,16) (4,16,64) (4,4,256) (4,1,1024); result in c
template <>
__device__ __forceinline__ void fft_forward<4096>(float *c, float *ch, const float *wa, int lane)
{
    radf4_pass<1, 1024>(c, ch, wa + 4092, wa + 4093, wa + 4094, lane);   // iw = 4096-3 = 4093
    wave_lds_sync();
    radf4_pass<4, 256>(ch, c, wa + 4080, wa + 4084, wa + 4088, lane);    // iw = 4093-12 = 4081
    wave_lds_sync();
    radf4_pass<16, 64>(c, ch, wa + 4032, wa + 4048, wa + 4064, lane);    // iw = 4081-48 = 4033
    wave_lds_sync();
    radf4_pass<64, 16>(ch, c, wa + 3840, wa + 3904, wa + 3968, lane);    // iw = 4033-192 = 3841
    wave_lds_sync();
    radf4_pass<256, 4>(c, ch, wa + 3072, wa + 3328, wa + 3584, lane);    // iw = 3841-768 = 3073
    wave_lds_sync();
    radf4_pass<1024, 1>(ch, c, wa + 0, wa + 1024, wa + 2048, lane);      // iw = 3073-3072 = 1
    wave_lds_sync();
}

// after an odd number of passes the spectrum sits in the second buffer (drftf1 copies it back, :6163-6169)
template <int N> struct fft_result_in_ch { static constexpr bool value = (N == 1024 || N == 512); };

// waves per workgroup: two LDS buffers of N floats per wave; 4096-sample blocks leave room for two waves
template <int N> struct fft_waves { static constexpr int value = (N == 4096) ? 2 : WAVES_PER_WG; };
// in place (one LDS buffer per wave) for every size but 4096 (64 floats per lane and pass would not stay in registers)
template <int N> struct fft_inplace { static constexpr bool value = (N != 4096); };
template <int N, bool IP> struct fft_run;
template <int N> struct fft_run<N, true> {
    static __device__ __forceinline__ const float *go(float *c, float *, const float *wa, int lane)
    {
        fft_forward_inplace<N>(c, wa, lane);
        return c;
    }
};
template <int N> struct fft_run<N, false> {
    static __device__ __forceinline__ const float *go(float *c, float *ch, const float *wa, int lane)
    {
        fft_forward<N>(c, ch, wa, lane);
        return fft_result_in_ch<N>::value ? ch : c;
    }
};

template <int N>
__global__ __launch_bounds__(64 * fft_waves<N>::value) __attribute__((amdgpu_waves_per_eu(N == 4096 ? 1 : 3)))
void k_window_fft_log(const float *__restrict__ pcm, float *__restrict__ logfft,
                      float *__restrict__ local_ampmax, const uint8_t *__restrict__ wflags,
                      const float *__restrict__ wa_g,      // FFTPACK twiddles, N floats (trigcache + N)
                      const float *__restrict__ win_self, const float *__restrict__ win_short,
                      int short_n, long nblocks, const int *__restrict__ d_live, int live_mult)
{
    if (d_live) {
        const long live = (long)*d_live * live_mult;
        if (live < nblocks) nblocks = live;
    }
    __shared__ __attribute__((aligned(16))) float s_wa[N];
    __shared__ __attribute__((aligned(16))) float s_win[N / 2];
    __shared__ __attribute__((aligned(16))) float s_wshort[N / 4];   // rising half-window of a short block (<= N/2 long)
    constexpr int NW = fft_waves<N>::value;
    constexpr int NBUF = fft_inplace<N>::value ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float s_buf[NW][NBUF][N];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < N; i += blockDim.x) s_wa[i] = wa_g[i];
    for (int i = tid; i < N / 2; i += blockDim.x) s_win[i] = win_self[i];
    if (wflags)
        for (int i = tid; i < (short_n >> 1) && i < N / 4; i += blockDim.x) s_wshort[i] = win_short[i];
    __syncthreads();

    float *c = s_buf[wave][0];
    float *ch = s_buf[wave][NBUF - 1];
    const float scale = 4.f / N;
    const float scale_dB = (float)((double)todB(scale) + .345);  // lib/mapping0.c:795

    for (long blk = (long)blockIdx.x * NW + wave; blk < nblocks; blk += (long)gridDim.x * NW) {
        // ---- load + window (lib/window.c:2137-2258) into LDS ----------------------------
        int ln = N, rn = N;
        const float *wl = s_win, *wr = s_win;
        if (wflags) {
            int f = wflags[blk];
            if (!(f & 1)) { ln = short_n; wl = s_wshort; }
            if (!(f & 2)) { rn = short_n; wr = s_wshort; }
        }
        const float4 *src = reinterpret_cast<const float4 *>(pcm + blk * N);
#pragma unroll
        for (int g = 0; g < N / 256; g++) {
            int q = lane + 64 * g;
            int i = 4 * q;
            float4 d = src[q];
            if (i < N / 2) {
                int lb = N / 4 - (ln >> 2);
                if (i < lb) d = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (i < lb + (ln >> 1)) {
                    float4 w = *reinterpret_cast<const float4 *>(wl + (i - lb));
                    d = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
                }
            } else {
                int rb = N / 2 + N / 4 - (rn >> 2);
                if (i >= rb + (rn >> 1)) d = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (i >= rb) {
                    float4 w = *reinterpret_cast<const float4 *>(wr + ((rn >> 1) - 4 - (i - rb)));
                    d = make_float4(d.x * w.w, d.y * w.z, d.z * w.y, d.w * w.x);
                }
            }
            *reinterpret_cast<float4 *>(c + i) = d;
        }
        wave_lds_sync();

        const float *spec = fft_run<N, fft_inplace<N>::value>::go(c, ch, s_wa, lane);

        // ---- log power spectrum + block maximum (lib/mapping0.c:848-888) -----------------
        float *o = logfft + blk * (N / 2);
        float mx;
        {
            // bin 0 by lane 0; its value seeds the running maximum (:862)
            float v0 = (float)((double)(scale_dB + todB(spec[0])) + .345);
            mx = v0;
            if (lane == 0) o[0] = v0;
        }
        for (int m = 1 + lane; m < N / 2; m += 64) {
            int j = 2 * m - 1;
            float re = spec[j], im = spec[j + 1];
            float temp = re * re + im * im;
            float v = (float)((double)(scale_dB + .5f * todB(temp)) + .345);
            o[m] = v;
            if (v > mx) mx = v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float other = __shfl_xor(mx, off);
            if (other > mx) mx = other;
        }
        if (mx > 0.f) mx = 0.f;
        if (lane == 0) local_ampmax[blk] = mx;
        wave_lds_sync();
    }
}

}  // namespace

extern "C" int vbm_launch_window_fft_log(const float *d_pcm, float *d_logfft, float *d_local_ampmax,
                                         const uint8_t *d_wflags, const float *d_wa, const float *d_win_self,
                                         const float *d_win_short, int n, int short_n, long nblocks,
                                         const int *d_live, int live_mult, hipStream_t stream)
{
    if (nblocks <= 0) return 0;
    if (n != 4096 && n != 2048 && n != 1024 && n != 512 && n != 256) return -1;
    const int nw = (n == 4096) ? 2 : WAVES_PER_WG;
    long wgs = (nblocks + nw - 1) / nw;
    const long per_cu = (n == 4096) ? 1 : (n == 2048) ? 3 : 4;     // workgroups a CU holds (LDS: tables + a buffer per wave)
    if (wgs > 256 * per_cu) wgs = 256 * per_cu;
    dim3 grid((unsigned)wgs), block(64 * nw);
#define LAUNCH_FFT(NN)                                                                                              \
    hipLaunchKernelGGL(k_window_fft_log<NN>, grid, block, 0, stream, d_pcm, d_logfft, d_local_ampmax, d_wflags, d_wa, \
                       d_win_self, d_win_short, short_n, nblocks, d_live, live_mult)
    if (n == 4096) LAUNCH_FFT(4096);
    else if (n == 2048) LAUNCH_FFT(2048);
    else if (n == 1024) LAUNCH_FFT(1024);
    else if (n == 512) LAUNCH_FFT(512);
    else LAUNCH_FFT(256);
#undef LAUNCH_FFT
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
