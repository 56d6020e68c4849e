// Psychoacoustic passes of the per-block encode path for gfx950 — one lane per channel-block,
// 64 independent blocks per wavefront, bin-major buffers (batch.h).
//
// Replaces, for a homogeneous batch of blocks (reference = scalar C path):
//   k_prologue    ampmax tracking of vorbis_analysis_blockout (lib/block.c:649-651,
//                 _vp_ampmax_decay lib/psy.c:4504-4515), global_ampmax of mapping0_forward
//                 (lib/mapping0.c:752, 889-901), _postnoise_detection (lib/psy.c:619-648)
//   (_vp_noisemask and _vp_tonemask live in noise_kernels.hip / tone_kernels.hip: a workgroup per group of
//   blocks, working set in LDS)
//   k_mix         _vp_offset_and_mix (lib/psy.c:4274-4502, set_m3p :4148-4272): offset_select 1 for VBR,
//                 1 / 2 / 0 with bit_managed for managed bitrate, including the aoTuV carried buffers
//                 lastmdct / tempmdct
// The order-bound float accumulations stay serial per lane, exactly in source order; table-driven loop
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

// (_vp_noisemask: noise_kernels.hip)
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
    if ((int)(blockIdx.x * blockDim.x) >= vbm_ncb(b)) return;        // (launch bound > device-resident count)
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

// ---------------------------------------------------------------------------------------------
// _vp_offset_and_mix for impulse blocks (block_mode 0, n = 128 / 256) with aoTuV M3 (lib/psy.c:4148-4272 set_m3p,
// :4330-4400 M3 MAIN): ONE wavefront per channel-block, a lane per bin (n / 64 bins each).  Read bin by bin the
// function looks serial — M3's set-up walks tempmdct with read-modify-writes, the main loop reads and rewrites it
// and keeps npeak up to date per partition — but:
//   * the set-up loop (for i, for j < bfn[i]: tempmdct[i + j] conditionally += 5 / bfn[i + j]) touches entry k = i + j
//     only, so entry k is its own chain over i = k - j in rising order: lane k walks it alone;
//   * the main loop reads and writes tempmdct[i] of its own bin only;
//   * the npeak updates of a partition (a tone-like bin sets -1, another qualifying bin clears a positive value) end
//     in a value that does not depend on their order: -1 if any bin was tone-like, else min(value, 0) if any bin
//     qualified.
// Every input row is read once (all of a lane's loads in flight together), every output written once.
template <int SEL, bool MANAGED>
__global__ __launch_bounds__(64) void k_mix_impulse(vbm_batch b)
{
    constexpr bool BUF = (!MANAGED || SEL == 2);   // mp->mdctbuf_flag of set_m3p when the rate is high (lib/psy.c:4165-4173)
    constexpr int RMAX = 4;                         // n <= 256
    __shared__ float s_lm[256];
    __shared__ int s_bfn[256];
    __shared__ int s_tone[64], s_qual[64];          // per partition (n / partition <= 64)
    const int lane = (int)blockIdx.x;               // channel-block
    const int t = (int)threadIdx.x;
    if (lane >= vbm_ncb(b)) return;
    if (MANAGED && SEL != 1 && !b.post_valid_blob[(size_t)(VBM_PACKETBLOBS / 2) * b.L + lane]) return;
    const size_t tb = TB(b, lane);
    const vbm_setup *s = b.setup;
    const vbm_psy *p = psy_of(b);
    const int n = p->n;
    const int R = n >> 6;
    const int sb = lane / b.ch, c = lane - sb * b.ch;
    const int sid = b.stream_id[sb];
    const int col = sid * b.ch + c;
    float *lastmdct = b.st.mblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);   // element i at [i*64]
    float *tempmdct = b.st.tblock + (size_t)(col >> 6) * b.st.slab_words + (col & 63);
    const float *noise = b.noiseT, *tone = b.toneT;
    float *logmask = b.logmaskT, *mdct = b.mdctT, *logmdct = b.logmdctT, *npeak = b.npeakT;
    const int nW_modenumber = (b.wflags[sb] >> 1) & 1;
    const int lW_block_mode = b.st.lW_block_mode[sid];
    const int lW_no = b.st.lW_no[sid];
    const int impadnum = b.st.impadnum[sid];
    float low_compand = b.st.lowcomp[col];
    const int end_block = s->floor[b.W].info_n;   // vif->n, lib/mapping0.c:1055
    const int hsrate = ((p->rate < 26000) ? 0 : 1);
    const int partition = (p->normal_p ? p->normal_partition : 16);
    const float toneatt = p->tone_masteratt[SEL];

    // ---- inputs of the lane's bins
    float nz[RMAX], tn[RMAX], lm[RMAX], md[RMAX], last[RMAX], temp[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        const int i = t + 64 * r;
        nz[r] = tn[r] = lm[r] = md[r] = last[r] = temp[r] = 0.f;
        if (r < R) {
            nz[r] = T(noise, i);
            tn[r] = T(tone, i);
            lm[r] = T(logmdct, i);
            if (SEL == 1) md[r] = T(mdct, i);
            last[r] = lastmdct[(size_t)i * 64];
            temp[r] = tempmdct[(size_t)i * 64];
        }
    }
    {
        const int *g = (n == 128) ? s->freq_bfn128 : s->freq_bfn256;
        for (int i = t; i < n; i += 64) s_bfn[i] = g[i];
    }
#pragma unroll
    for (int r = 0; r < RMAX; r++)
        if (r < R) s_lm[t + 64 * r] = lm[r];
    s_tone[t] = 0; s_qual[t] = 0;

    mod3 mp3;
    mp3.sw = 0; mp3.mdctbuf_flag = 0; mp3.noise_rate = mp3.noise_rate_low = mp3.noise_center = mp3.tone_rate = 0.f;
    int m4_start = p->normal_start;
    int m4_end = p->tonecomp_endp;
    const float m4_thres = p->tonecomp_thres;
    int m4_end_block = end_block;
    if (low_compand < 0 || (double)toneatt < 25.) low_compand = 0;
    else low_compand = (float)((double)low_compand * ((double)toneatt - 25.));

    // set_m3p (lib/psy.c:4148-4272), block_mode 0
    float sub = 0.f;
    double add = 0.;
    if (hsrate) {
        mp3.mdctbuf_flag = BUF ? 1 : 0;
        if (!(MANAGED && SEL == 0)) {
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
                } else {
                    mp3.noise_rate = (float)0.7;
                    mp3.noise_center = 0;
                    mp3.tone_rate = 8.f;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.125));
                sub = 5; add = 5.;
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
                } else {
                    mp3.noise_rate = (float)0.6;
                    mp3.noise_center = 12;
                    mp3.tone_rate = 8.f;
                }
                mp3.noise_rate_low = 0;
                mp3.sw = 1;
                if (impadnum) mp3.noise_rate = (float)((double)mp3.noise_rate * (impadnum * 0.0625));
                sub = 10; add = 10.;
            }
        }
    }
    __syncthreads();

    // ---- M3 set-up of tempmdct (lib/psy.c:4199-4215 / :4246-4262)
    if (mp3.sw && BUF) {
        int bmax = 0;
        for (int i = t; i < n; i += 64) bmax = max(bmax, s_bfn[i]);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) bmax = max(bmax, __shfl_xor(bmax, m));
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            if (r >= R) continue;
            const int k = t + 64 * r;
            float tv = lW_block_mode ? last[r] - sub : temp[r] - sub;
            const double ck = add / (double)(float)s_bfn[k];
            for (int i = (k - bmax + 1 > 0 ? k - bmax + 1 : 0); i < k; i++) {
                const int bf = s_bfn[i];
                const int j = k - i;
                if (j < bf) {
                    const float cell = 75 / (float)bf;
                    const float freqbuf = s_lm[i] - (cell * j);
                    if (tv < freqbuf) tv = (float)((double)tv + ck);
                }
            }
            temp[r] = tv;
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
    const float *__restrict__ noiseoffset = p->noiseoffset[SEL];
    const float noisemaxsupp = p->noisemaxsupp, m_val = p->m_val;
    const int tonecomp_endp = p->tonecomp_endp, m3n0 = p->m3n[0], m3n1 = p->m3n[1], m3n2 = p->m3n[2];

    float lmv[RMAX], mdo[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        lmv[r] = 0.f; mdo[r] = md[r];
        if (r >= R) continue;
        const int i = t + 64 * r;
        float val = nz[r] + noiseoffset[i];
        float tval = tn[r] + toneatt;
        const float l = lm[r];
        if (i <= m4_start) tval -= low_compand;
        if (val > noisemaxsupp) val = noisemaxsupp;

        // M3 MAIN
        if (mp3.sw) {
            if (val > tval) {
                const float lst = last[r];
                if ((val > lst) && (l > (temp[r] + mp3.noise_center))) {
                    int toneac = 0;
                    float valmask = 0;
                    float rate_mod;
                    float mainth;

                    if (mp3.mdctbuf_flag == 1) temp[r] = l;
                    if (l > lst) rate_mod = mp3.noise_rate;
                    else rate_mod = mp3.noise_rate_low;
                    if (!impadnum && (i < tonecomp_endp) && ((val - lst) > 20.f)) {
                        float dBsub = (l - lst);
                        if (dBsub > 25.f) {
                            toneac = 1;
                            if (tval > -100.f && ((l - tval) < 48.f)) {
                                float tr_cur = mp3.tone_rate;
                                if (dBsub < 35.f) tr_cur *= ((35.f - dBsub) * .1f);
                                tval -= tr_cur;
                                if (tval < -100.f) tval = -100.f;
                                if ((l - tval) > 48.f) tval = l - 48.f;
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

                    if ((val - valmask) > lst) val -= valmask;
                    else val = lst;

                    if (toneac) {
                        float tmp = val - VMAX(lst, -140);
                        if (tmp > 20.f) val -= (tmp - 20.f) * .2f;
                    }
                    // npeak of the bin's partition: see the header
                    if (toneac == 1) atomicOr(&s_tone[i / partition], 1);
                    else atomicOr(&s_qual[i / partition], 1);
                }
            }
        }

        // M4 MAIN
        float o;
        if (val > tval) {
            o = val;
        } else if ((i > m4_start) && (i < m4_end)) {
            if (l < tval) {
                if (l < val) tval -= (tval - val) * m4_thres;
                else tval = l;
            }
            o = tval;
        } else
            o = tval;
        lmv[r] = o;

        // M1 (offset_select == 1)
        if (SEL == 1) {
            const float m1_coeffi = (float)-17.2;
            float m1_de;
            val = val - l;
            if (val > m1_coeffi) {
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.005 * (double)m_val));
                if (m1_de < 0) m1_de = (float)0.0001;
            } else
                m1_de = (float)(1.0 - ((double)(val - m1_coeffi) * 0.0003 * (double)m_val));
            mdo[r] = md[r] * m1_de;
        }
    }
    __syncthreads();     // every read of lastmdct / tempmdct / npeak above is done

    // ---- outputs
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        if (r >= R) continue;
        const int i = t + 64 * r;
        T(logmask, i) = lmv[r];
        if (SEL == 1) T(mdct, i) = mdo[r];
        if (mp3.sw && BUF) tempmdct[(size_t)i * 64] = temp[r];
        // M3 SET lastmdct (lib/psy.c:4463-4475): a short block followed by a long one spreads every bin over eight
        if (mp3.mdctbuf_flag == 1) {
            if (nW_modenumber) {
#pragma unroll
                for (int j = 0; j < 8; j++) lastmdct[(size_t)(i * 8 + j) * 64] = lm[r];
            } else {
                lastmdct[(size_t)i * 64] = lm[r];
            }
        }
    }
    if (mp3.sw && t < (n + partition - 1) / partition) {
        if (s_tone[t]) T(npeak, t) = -1.f;
        else if (s_qual[t] && T(npeak, t) > 0) T(npeak, t) = 0;
    }
}

}  // namespace

static inline dim3 grid_for(int lanes) { return dim3((unsigned)((lanes + 63) / 64)); }
// slices of the bin range for the kernels whose bins are independent (long blocks: 64 bins each)
static inline int bin_chunks(const vbm_batch *b)
{
    static const int big = [] {
        const char *e = getenv("VBM_BIN_CHUNKS");   // tuning knob: slices of the long-block bin range
        const int v = e ? atoi(e) : 16;
        return (v < 1 || v > 64) ? 16 : v;
    }();
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
    // (once, whichever host thread comes first: the initialiser of a function-local static)
    static const bool done = [] {
        allow_big_lds(k_mix<1, false, false, true>);
        allow_big_lds(k_mix<1, true, false, true>);
        allow_big_lds(k_mix<2, true, false, true>);
        allow_big_lds(k_mix<0, true, false, true>);
        return true;
    }();
    (void)done;
}

extern "C" int vbm_launch_mix(const vbm_batch *b, hipStream_t st)
{
    const int nchunks = (b->block_mode == 0) ? 1 : bin_chunks(b);
    const dim3 grid((unsigned)((b->ncb + 63) / 64), (unsigned)nchunks);
    if (b->block_mode == 0 && (b->n == 128 || b->n == 256))
        hipLaunchKernelGGL((k_mix_impulse<1, false>), dim3((unsigned)b->ncb), dim3(64), 0, st, *b);
    else if (b->block_mode == 0) {
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
    if (b->block_mode == 0 && (b->n == 128 || b->n == 256)) {
        const dim3 g1((unsigned)b->ncb);
        if (offset_select == 1) hipLaunchKernelGGL((k_mix_impulse<1, true>), g1, dim3(64), 0, st, *b);
        else if (offset_select == 2) hipLaunchKernelGGL((k_mix_impulse<2, true>), g1, dim3(64), 0, st, *b);
        else hipLaunchKernelGGL((k_mix_impulse<0, true>), g1, dim3(64), 0, st, *b);
    } else if (b->block_mode == 0) {
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
