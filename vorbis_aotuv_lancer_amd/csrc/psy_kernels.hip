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
//   k_tonemask    _vp_tonemask (lib/psy.c:4076-4142): seed_loop/seed_curve (:652-771),
//                 max_seeds/seed_chase (:773-1085)
//   k_mix         _vp_offset_and_mix with offset_select 1, VBR (lib/psy.c:4274-4502, set_m3p
//                 :4148-4272), including the aoTuV carried buffers lastmdct / tempmdct
// The order-bound float accumulations (the five prefix sums of bark_noise_hybridmp, the
// partition sums of M8) stay serial per lane, exactly in source order; table-driven loop
// bounds are identical in every lane, so the wave runs them in lockstep and table reads are
// wave-uniform.  Compiled with -ffp-contract=off; `double` where the C source promotes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "kernels.h"

#define T(buf, i) (buf)[tb + (size_t)(i) * 64]   /* tiled bin-major: batch.h */
#define TB(b, lane) ((size_t)((lane) >> 6) * (b).slab_words + ((lane) & 63))
#define NEGINF -9999.f
#define VMIN(x, y) ((x) > (y) ? (y) : (x))
#define VMAX(x, y) ((x) < (y) ? (y) : (x))

namespace {

__device__ __forceinline__ const vbm_psy *psy_of(const vbm_batch &b)
{
    // psy_look = b->psy + blocktype + (W ? 2 : 0)   (lib/mapping0.c:764) == psy[block_mode]
    return &b.setup->psy[b.block_mode];
}

// ---------------------------------------------------------------------------------------------
__global__ void k_prologue(vbm_batch b)
{
    const int sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= b.nsb) return;
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
            for (int i = sn; i < mn; i++) upt += fabs((double)pcm[i]);
            for (int i = mn; i < en; i++) unt += fabs((double)pcm[i]);
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
// bark_noise_hybridmp, lib/psy.c:3480-3638.  f / noise are bin-major columns of this lane.
// The five prefix sums are strictly sequential per lane (source order); to keep the lane from
// stalling on every element, loads are issued in batches ahead of the dependent arithmetic
// (the input column for the prefix pass, the ten window-edge sums for the solve passes).
// The three solve phases of the source (mirrored window / plain window / tail) are bounded by
// table-only conditions, so their limits are lane-uniform.
#define HY_PF 8
#define HY_SB 4
__device__ __forceinline__ void hybridmp(const vbm_batch &b, const vbm_psy *p, int lane, const float *__restrict__ f,
                         float *__restrict__ noise, const float offset, const int fixed)
{
    const size_t tb = TB(b, lane);
    const int n = p->n;
    float *__restrict__ N = b.sumT;
    float *__restrict__ X = b.sumT + (size_t)n * 64;
    float *__restrict__ XX = b.sumT + (size_t)2 * n * 64;
    float *__restrict__ Y = b.sumT + (size_t)3 * n * 64;
    float *__restrict__ XY = b.sumT + (size_t)4 * n * 64;
    float tN, tX, tXX, tY, tXY;
    int i;
    float R = 0.f, A = 0.f, B = 0.f, D = 1.f;
    float w, x, y;

    tN = tX = tXX = tY = tXY = 0.f;

    y = T(f, 0) + offset;
    if (y < 1.f) y = 1.f;
    w = (float)((double)(y * y) * .5);
    tN += w;
    tX += w;
    tY += w * y;
    T(N, 0) = tN; T(X, 0) = tX; T(XX, 0) = tXX; T(Y, 0) = tY; T(XY, 0) = tXY;

    for (i = 1, x = 1.f; i < n; i += HY_PF) {
        float fv[HY_PF];
#pragma unroll
        for (int u = 0; u < HY_PF; u++) fv[u] = T(f, (i + u < n) ? i + u : n - 1);   // unconditional, clamped
#pragma unroll
        for (int u = 0; u < HY_PF; u++) {
            if (i + u < n) {
                y = fv[u] + offset;
                if (y < 1.f) y = 1.f;
                w = y * y;
                tN += w;
                tX += w * x;
                tXX += w * x * x;
                tY += w * y;
                tXY += w * x * y;
                T(N, i + u) = tN; T(X, i + u) = tX; T(XX, i + u) = tXX; T(Y, i + u) = tY; T(XY, i + u) = tXY;
                x += 1.f;
            }
        }
    }

    // phase limits (lib/psy.c:3543-3544, :3565-3566), identical in every lane
    int i1 = 0;
    for (; i1 < n; i1++) {
        int lo = p->bark_lo[i1], hi = p->bark_hi[i1];
        if (lo >= 0 || -lo >= n) break;
        if (hi >= n) break;
    }
    int i2 = i1;
    for (; i2 < n; i2++) {
        int lo = p->bark_lo[i2], hi = p->bark_hi[i2];
        if (lo < 0 || lo >= n) break;
        if (hi >= n) break;
    }

#define HY_SOLVE(MIRROR, FIXEDPASS, I0, I1)                                                                    \
    for (i = (I0); i < (I1); i += HY_SB) {                                                                      \
        float e[HY_SB][10], prev[HY_SB];                                                                        \
        _Pragma("unroll") for (int u = 0; u < HY_SB; u++) {                                                     \
            {                                                                                                   \
                const int ii = (i + u < (I1)) ? i + u : (I1) - 1;   /* clamped: loads stay unconditional */     \
                int lo, hi;                                                                                     \
                if (FIXEDPASS) { hi = ii + fixed / 2; lo = hi - fixed; }                                        \
                else { lo = p->bark_lo[ii]; hi = p->bark_hi[ii]; }                                              \
                int lo_ = MIRROR ? -lo : lo;                                                                    \
                e[u][0] = T(N, hi);  e[u][1] = T(N, lo_);                                                       \
                e[u][2] = T(X, hi);  e[u][3] = T(X, lo_);                                                       \
                e[u][4] = T(XX, hi); e[u][5] = T(XX, lo_);                                                      \
                e[u][6] = T(Y, hi);  e[u][7] = T(Y, lo_);                                                       \
                e[u][8] = T(XY, hi); e[u][9] = T(XY, lo_);                                                      \
                if (FIXEDPASS) prev[u] = T(noise, ii);                                                          \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < HY_SB; u++) {                                                     \
            if (i + u < (I1)) {                                                                                 \
                x = (float)(i + u);                                                                             \
                if (MIRROR) {                                                                                   \
                    tN = e[u][0] + e[u][1]; tX = e[u][2] - e[u][3]; tXX = e[u][4] + e[u][5];                    \
                    tY = e[u][6] + e[u][7]; tXY = e[u][8] - e[u][9];                                            \
                } else {                                                                                        \
                    tN = e[u][0] - e[u][1]; tX = e[u][2] - e[u][3]; tXX = e[u][4] - e[u][5];                    \
                    tY = e[u][6] - e[u][7]; tXY = e[u][8] - e[u][9];                                            \
                }                                                                                               \
                A = tY * tXX - tX * tXY;                                                                        \
                B = tN * tXY - tX * tY;                                                                         \
                D = tN * tXX - tX * tX;                                                                         \
                R = (A + x * B) / D;                                                                            \
                if (FIXEDPASS) {                                                                                \
                    if (R - offset < prev[u]) T(noise, i + u) = R - offset;                                     \
                } else {                                                                                        \
                    if (R < 0.f) R = 0.f;                                                                       \
                    T(noise, i + u) = R - offset;                                                               \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    }

    HY_SOLVE(true, false, 0, i1)
    HY_SOLVE(false, false, i1, i2)
    for (i = i2; i < n; i++) {   // x is exactly (float)i in the source as well (x += 1.f from 0, i < 2^24)
        x = (float)i;
        R = (A + x * B) / D;
        if (R < 0.f) R = 0.f;
        T(noise, i) = R - offset;
    }

    if (fixed <= 0) return;

    // fixed-width window pass (lib/psy.c:3593-3636)
    int f1 = 0;
    for (; f1 < n; f1++) {
        int hi = f1 + fixed / 2, lo = hi - fixed;
        if (hi >= n) break;
        if (lo >= 0) break;
    }
    int f2 = f1;
    for (; f2 < n; f2++) {
        int hi = f2 + fixed / 2, lo = hi - fixed;
        if (hi >= n) break;
        if (lo < 0) break;
    }
    HY_SOLVE(true, true, 0, f1)
    HY_SOLVE(false, true, f1, f2)
    for (i = f2; i < n; i += HY_PF) {
        float prev[HY_PF];
#pragma unroll
        for (int u = 0; u < HY_PF; u++) prev[u] = T(noise, (i + u < n) ? i + u : n - 1);
#pragma unroll
        for (int u = 0; u < HY_PF; u++) {
            if (i + u < n) {
                x = (float)(i + u);
                R = (A + x * B) / D;
                if (R - offset < prev[u]) T(noise, i + u) = R - offset;
            }
        }
    }
#undef HY_SOLVE
}

// aoTuV M7, lib/psy.c:3645-3768.  temp/inmod: 256-entry per-lane scratch carved from seedT/ampstackT
__device__ __forceinline__ void ntfix(const vbm_batch &b, const vbm_psy *p, int lane, const float *spectral, float *noise)
{
    const size_t tb = TB(b, lane);
    int i, j, k;
    int n = p->n;
    int nx = p->tonefix_end;
    float *temp = b.seedT, *inmod = b.ampstackT;   // both have >= 256 rows (total_octave_lines >= 585)
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

__global__ void k_noisemask(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= b.ncb) return;
    const size_t tb = TB(b, lane);
    const vbm_setup *s = b.setup;
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    const int partition = (p->normal_p ? p->normal_partition : 16);
    int i, j, k;

    float *logmdct = b.logmdctT, *logmask = b.noiseT, *work = b.workT, *epeak = b.epeakT, *npeak = b.npeakT;

    // logmdct[j] = todB(mdct[j]) + .345   (lib/mapping0.c:936)
    for (i = 0; i < n; i++) T(logmdct, i) = (float)((double)vbm_todB(T(b.mdctT, i)) + .345);

    // lb_loudnoise_fix (lib/psy.c:5152-5180)
    float noise_compand_level = b.st.lowcomp[col];
    {
        const int lW_block_mode = b.st.lW_block_mode[sid];
        if (p->m_val < 0.5) noise_compand_level = -1;
        else if (p->normal_thresh > .45) noise_compand_level = -1;
        else if ((b.block_mode == 2 && lW_block_mode == 3) || (b.block_mode == 3 && lW_block_mode == 2)) {
            double hi_th = 0;
            for (i = p->n25p; i < p->n75p; i++) {
                float v = T(logmdct, i);
                if (v > -130) hi_th += v;
                else hi_th += -130;
            }
            hi_th /= n;
            if (hi_th > -40.) noise_compand_level = -1;
            else if (hi_th < -50.) noise_compand_level = 1.f;
            else noise_compand_level = (float)(1. - ((hi_th + 50) / 10));
        }
        b.st.lowcomp[col] = noise_compand_level;
    }

    hybridmp(b, p, lane, logmdct, logmask, 140.f, -1);

    for (i = 0; i < n; i++) T(work, i) = T(logmdct, i) - T(logmask, i);

    hybridmp(b, p, lane, work, logmask, 0.f, p->noisewindowfixed);

    for (i = 0; i < n; i++) T(work, i) = T(logmdct, i) - T(work, i);

    ntfix(b, p, lane, logmdct, work);

    // noise compand & aoTuV M5 extension & pre-store tone peak
    i = 0;
    if (noise_compand_level > 0) {
        int thter = p->n33p;
        for (; i < thter; i++) {
            int dB = (int)((double)T(logmask, i) + .5);
            if (dB >= VBM_NOISE_COMPAND_LEVELS) dB = VBM_NOISE_COMPAND_LEVELS - 1;
            if (dB < 0) dB = 0;
            T(epeak, i) = T(work, i) + s->stn_compand[dB];
            T(logmask, i) = T(work, i) + p->noisecompand[dB] -
                            ((p->noisecompand[dB] - p->noisecompand_high[dB]) * noise_compand_level);
        }
    }
    for (; i < n; i++) {
        int dB = (int)((double)T(logmask, i) + .5);
        if (dB >= VBM_NOISE_COMPAND_LEVELS) dB = VBM_NOISE_COMPAND_LEVELS - 1;
        if (dB < 0) dB = 0;
        T(epeak, i) = T(work, i) + s->stn_compand[dB];
        T(logmask, i) = T(work, i) + p->noisecompand[dB];
    }

    for (i = 0, k = 0; i < n; i += partition, k++) T(npeak, k) = 0.f;

    // reduction of post-echo (postprocessing of aoTuV M2)
    const float poste = b.poste[lane];
    if (poste > 0) {
        for (i = 0, k = 0; i < p->min_nn_lp; i += partition, k++) {
            float temp = VMIN(VMIN(poste, 30.f), p->noiseoffset[1][i] + 30.f);
            if (temp <= 0) continue;
            T(npeak, k) = -1.f;
            for (j = 0; j < partition; j++) T(logmask, i + j) -= temp;
        }
    }

    // M8
    for (k = 0, i = 0; i < p->min_nn_lp; i += partition, k++) {
        const float nt = 4;
        float o = p->noiseoffset[1][i + partition - 1] + 6;
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

    // M9
    {
        i = 0;
        if (b.block_mode > 1) {
            const float *lastmdct = b.st.mblock;
            for (; i < p->tonecomp_endp; i++) {
                float temp = T(logmdct, i) - T(epeak, i);
                T(epeak, i) = 0.f;
                if (temp >= 12.f) {
                    float mi = T(logmdct, i) - lastmdct[(size_t)(col >> 6) * b.st.slab_words + (size_t)i * 64 + (col & 63)];
                    if (mi >= 1) T(epeak, i) = mi;
                }
            }
        }
        for (; i < n; i++) T(epeak, i) = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void seed_curve(const vbm_batch &b, int lane, float *seed, const float *curves /*[P_LEVELS][EHMER_MAX+2]*/,
                           const float amp, const int oc, const int n, const int linesper, const float dBoffset)
{
    const size_t tb = TB(b, lane);
    int i, post1;
    int seedptr;
    const float *posts, *curve;

    int choice = (int)(((double)(amp + dBoffset) - 30.) * (double).1f);   // (amp+dBoffset-P_LEVEL_0)*.1f, P_LEVEL_0 = 30. (double)
    choice = VMAX(choice, 0);
    choice = VMIN(choice, VBM_P_LEVELS - 1);
    posts = curves + choice * (VBM_EHMER_MAX + 2);
    curve = posts + 2;
    post1 = (int)posts[1];
    seedptr = (int)((float)oc + (posts[0] - VBM_EHMER_OFFSET) * linesper - (linesper >> 1));

    for (i = (int)posts[0]; i < post1; i++) {
        if (seedptr > 0) {
            float lin = amp + curve[i];
            if (T(seed, seedptr) < lin) T(seed, seedptr) = lin;
        }
        seedptr += linesper;
        if (seedptr >= n) break;
    }
}

__global__ void k_tonemask(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= b.ncb) return;
    const size_t tb = TB(b, lane);
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int sb = lane / b.ch;
    float *seed = b.seedT, *ampstack = b.ampstackT;
    int *posstack = b.posstackT;
    const float *f = b.logfftT;
    float *flr = b.toneT;
    long i;

    const float global_specmax = b.global_ampmax[sb];
    const float local_specmax = b.local_ampmax[lane];

    float att = local_specmax + p->ath_adjatt;
    for (i = 0; i < p->total_octave_lines; i++) T(seed, i) = NEGINF;
    if (att < p->ath_maxatt) att = p->ath_maxatt;
    for (i = 0; i < n; i++) T(flr, i) = p->ath[i] + att;

    // seed_loop (lib/psy.c:719-771)
    {
        float dBoffset = p->max_curve_dB - global_specmax;
        for (i = 0; i < n; i++) {
            float max = T(f, i);
            long oc = p->octave[i];
            while (i + 1 < n && p->octave[i + 1] == oc) {
                i++;
                if (T(f, i) > max) max = T(f, i);
            }
            if (max + 6.f > T(flr, i)) {
                oc = oc >> p->shiftoc;
                if (oc >= VBM_P_BANDS) oc = VBM_P_BANDS - 1;
                if (oc < 0) oc = 0;
                seed_curve(b, lane, seed, p->tonecurves + (size_t)oc * VBM_P_LEVELS * (VBM_EHMER_MAX + 2), max,
                           p->octave[i] - p->firstoc, p->total_octave_lines, p->eighth_octave_lines, dBoffset);
            }
        }
    }

    // max_seeds (lib/psy.c:936-1085) with seed_chase (:773-934)
    {
        const long tn = p->total_octave_lines;
        const int linesper = p->eighth_octave_lines;
        long stack = 0, pos = 0;
        for (i = 0; i < tn; i++) {
            if (stack < 2) {
                T(posstack, stack) = (int)i;
                T(ampstack, stack++) = T(seed, i);
            } else {
                while (1) {
                    if (T(seed, i) < T(ampstack, stack - 1)) {
                        T(posstack, stack) = (int)i;
                        T(ampstack, stack++) = T(seed, i);
                        break;
                    } else {
                        if (i < T(posstack, stack - 1) + linesper) {
                            if (stack > 1 && T(ampstack, stack - 1) <= T(ampstack, stack - 2) &&
                                i < T(posstack, stack - 2) + linesper) {
                                stack--;
                                continue;
                            }
                        }
                        T(posstack, stack) = (int)i;
                        T(ampstack, stack++) = T(seed, i);
                        break;
                    }
                }
            }
        }
        for (i = 0; i < stack; i++) {
            long endpos;
            if (i < stack - 1 && T(ampstack, i + 1) > T(ampstack, i)) {
                endpos = T(posstack, i + 1);
            } else {
                endpos = T(posstack, i) + linesper + 1;
            }
            if (endpos > tn) endpos = tn;
            for (; pos < endpos; pos++) T(seed, pos) = T(ampstack, i);
        }

        long linpos = 0;
        pos = p->octave[0] - p->firstoc - (linesper >> 1);
        while (linpos + 1 < n) {
            float minV = T(seed, pos);
            long end = ((p->octave[linpos] + p->octave[linpos + 1]) >> 1) - p->firstoc;
            if (minV > p->tone_abs_limit) minV = p->tone_abs_limit;
            while (pos + 1 <= end) {
                pos++;
                float sv = T(seed, pos);
                if ((sv > NEGINF && sv < minV) || minV == NEGINF) minV = sv;
            }
            end = pos + p->firstoc;
            for (; linpos < n && p->octave[linpos] <= end; linpos++)
                if (T(flr, linpos) < minV) T(flr, linpos) = minV;
        }
        {
            float minV = T(seed, tn - 1);
            for (; linpos < n; linpos++)
                if (T(flr, linpos) < minV) T(flr, linpos) = minV;
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct mod3 {
    int sw;
    int mdctbuf_flag;
    float noise_rate, noise_rate_low, noise_center, tone_rate;
};

__global__ void k_mix(vbm_batch b)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= b.ncb) return;
    const size_t tb = TB(b, lane);
    const vbm_setup *s = b.setup;
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    float *lastmdct = b.st.mblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);   // element i at [i*64]
    float *tempmdct = b.st.tblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);
#define LAST(i) lastmdct[(size_t)(i) * 64]
#define TEMP(i) tempmdct[(size_t)(i) * 64]
    const float *noise = b.noiseT, *tone = b.toneT;
    float *logmask = b.logmaskT, *mdct = b.mdctT, *logmdct = b.logmdctT, *npeak = b.npeakT;
    const int offset_select = 1;
    const int block_mode = b.block_mode;
    const int nW_modenumber = (b.wflags[sb] >> 1) & 1;
    const int lW_block_mode = b.st.lW_block_mode[sid];
    const int lW_no = b.st.lW_no[sid];
    const int impadnum = b.st.impadnum[sid];
    float low_compand = b.st.lowcomp[col];
    const int end_block = s->floor[b.W].info_n;   // vif->n, lib/mapping0.c:1055

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

    // set_m3p (lib/psy.c:4148-4272), VBR: bit_managed == 0
    if (!hsrate) {
        mp3.sw = 0;
        mp3.mdctbuf_flag = 0;
    } else {
        mp3.mdctbuf_flag = 1;
        if (block_mode) {
            mp3.sw = 0;
        } else if (n == 128 || n == 256) {
            const int *bfn = (n == 128) ? s->freq_bfn128 : s->freq_bfn256;
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
                    for (i = 0; i < n; i++) TEMP(i) -= 5;
                } else {
                    mp3.noise_rate = (float)0.7;
                    mp3.noise_center = 0;
                    mp3.tone_rate = 8.f;
                    for (i = 0; i < n; i++) TEMP(i) = LAST(i) - 5;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.125));
                for (i = 0; i < n; i++) {
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
                    for (i = 0; i < n; i++) TEMP(i) -= 10;
                } else {
                    mp3.noise_rate = (float)0.6;
                    mp3.noise_center = 12;
                    mp3.tone_rate = 8.f;
                    for (i = 0; i < n; i++) TEMP(i) = LAST(i) - 10;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.0625));
                for (i = 0; i < n; i++) {
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

    for (i = 0; i < n; i++) {
        float val = T(noise, i) + p->noiseoffset[offset_select][i];
        float tval = T(tone, i) + toneatt;
        const float lm = T(logmdct, i);
        if (i <= m4_start) tval -= low_compand;
        if (val > p->noisemaxsupp) val = p->noisemaxsupp;

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
                    if (!impadnum && (i < p->tonecomp_endp) && ((val - last) > 20.f)) {
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
                    if (i > p->m3n[0]) {
                        mainth = 30.f;
                    } else if (i > p->m3n[1]) {
                        mainth = 20.f;
                    } else if (i > p->m3n[2]) {
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
        if (val > tval) {
            T(logmask, i) = val;
        } else if ((i > m4_start) && (i < m4_end)) {
            if (lm < tval) {
                if (lm < val) tval -= (tval - val) * m4_thres;
                else tval = lm;
            }
            T(logmask, i) = tval;
        } else
            T(logmask, i) = tval;

        // M1 (offset_select == 1)
        {
            m1_coeffi = (float)-17.2;
            val = val - lm;
            if (val > m1_coeffi) {
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.005 * (double)p->m_val));
                if (m1_de < 0) m1_de = (float)0.0001;
            } else
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.0003 * (double)p->m_val));
            T(mdct, i) *= m1_de;
        }
    }

    // M3 SET lastmdct
    if (mp3.mdctbuf_flag == 1) {
        const int mag = 8;
        switch (block_mode) {
        case 0:
        case 1:
            if (nW_modenumber) {
                for (i = 0, k = 0; i < n; i++, k += mag)
                    for (j = 0; j < mag; j++) LAST(k + j) = T(logmdct, i);
            } else {
                for (i = 0; i < n; i++) LAST(i) = T(logmdct, i);
            }
            break;
        case 2:
            if (!nW_modenumber) {
                int nsh = n >> 3;
                for (i = 0; i < nsh; i++) {
                    int ni = i * mag;
                    float v = T(logmdct, ni);
                    for (j = 1; j < mag; j++)
                        if (v > T(logmdct, ni + j)) v = T(logmdct, ni + j);
                    LAST(i) = v;
                }
            } else {
                for (i = 0; i < n; i++) LAST(i) = T(logmdct, i);
            }
            break;
        case 3:
            for (i = 0; i < n; i++) LAST(i) = T(logmdct, i);
            break;
        default:
            break;
        }
    }
#undef LAST
#undef TEMP
}

}  // namespace

static inline dim3 grid_for(int lanes) { return dim3((unsigned)((lanes + 63) / 64)); }

extern "C" int vbm_launch_prologue(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_prologue, grid_for(b->nsb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_noisemask(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_noisemask, grid_for(b->ncb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_tonemask(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_tonemask, grid_for(b->ncb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_mix(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_mix, grid_for(b->ncb), dim3(64), 0, st, *b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
