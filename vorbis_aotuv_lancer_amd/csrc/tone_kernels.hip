// _vp_tonemask (reference lib/psy.c:4076-4142) as ONE kernel for gfx950: a workgroup owns NB consecutive
// channel-blocks (a quarter of a 64-lane tile) and keeps their seed arrays — total_octave_lines (585 .. 841)
// floats per block — in LDS from the first stamp to the last read.  HBM traffic = the algorithmic 4 n bytes in
// (block-major rows of k_window_fft_log) + 4 n out (tiled bin-major, batch.h) per channel-block.
//
//   stamp    seed_loop / seed_curve (lib/psy.c:652-771).  One thread per (block, run of equal octave[]): maximum
//            of the run's bins, curve choice, then the curve points go in with ds_max on an order-preserving
//            integer key of the float (max is order free: the stamps of a block may land in any order).
//   compare  everything seed_chase ever asks about two seeds is `seed[j] < seed[j-k]` (k = 1..7) and
//            `seed[j] <= seed[j-d]` (d = 1..6): both only between lines less than eighth_octave_lines (8) apart,
//            because the walk pops an entry only while it lies within that distance of the current line
//            (lib/psy.c:864-868).  One thread per line evaluates them once — 13 float compares — into two bytes.
//   chase    seed_chase's stack walk (lib/psy.c:851-881) on those bytes.  An entry further than 7 lines back can
//            never be popped again, so the walk's whole state is a 7-bit mask "line i-k is still on the stack"
//            (the stack itself is the set of surviving lines; an entry's amplitude is its seed).  One lane walks
//            64 lines + 16 lines of run-in with the state "nothing poppable" assumed at the start; a lane's state
//            at its first own line is then checked against what its left neighbour arrived at, and a lane that
//            guessed wrong re-runs from the true state until all agree (lane 0 starts from the true empty stack,
//            so the fixed point is the serial walk's result; re-runs are rare).  Integer operations on registers only.
//   fill     the tail of seed_chase (lib/psy.c:1012-1026): one wavefront per block, one lane per line: a surviving
//            line's reach ends where the next survivor starts if that one is louder, else 9 lines on; where it
//            starts is the exclusive prefix maximum of the ends before it.  All reads before the first write-back.
//   apply    max_seeds (lib/psy.c:936-1085): per bin the minimum of its seed-line segment (host-built seg_p0/p1)
//            raises the ATH floor.
//
// Exactness: every float expression (curve stamps, ATH) is the scalar source's; after the stamps the data is only
// compared and copied.  tools/ and tests/: the stage output `tone` is compared bit for bit with the oracle for all
// block types and mode packs (tests/test_pipeline_gpu.py, tests/test_frontend_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "batch.h"
#include "kernels.h"

#define NEGINF -9999.f
#define VMIN(x, y) ((x) > (y) ? (y) : (x))
#define VMAX(x, y) ((x) < (y) ? (y) : (x))
#define TM_CHUNKS 16                    /* chase lanes per block (64 lines each): total_octave_lines <= 1024 */

namespace {

__device__ __forceinline__ int seed_key(float f)
{
    int v = __float_as_int(f);
    return v ^ ((v >> 31) & 0x7fffffff);   // monotone float -> int (involution)
}
__device__ __forceinline__ float seed_val(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// One chase lane: lines [i, iend] of a block, starting with window mask m (bit b: line i-1-b is on the stack).
// gl[j] = GE | LE << 8 of line j.  own: the lane's 64 lines start at line `own`; alive gets their final fate,
// m_own / m_next the mask at the start of lines own and own + 64.
__device__ __forceinline__ void chase_run(const unsigned short *__restrict__ gl, int i, const int iend, unsigned m,
                                          const int own, unsigned long long &alive, unsigned &m_own, unsigned &m_next)
{
    unsigned long long H = 0;               // byte b: LE of line i-1-b
#pragma unroll
    for (int b = 0; b < 7; b++) {
        const int q = i - 1 - b;
        if (q >= 0) H |= (unsigned long long)(gl[q] >> 8) << (8 * b);
    }
    alive = 0;
    // four lines per LDS read (i is a multiple of 4 here: run-ins start 16 lines before a multiple of 64)
    for (; i <= iend; i += 4) {
        const unsigned long long g4 = *(const unsigned long long *)(gl + i);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int ii = i + u;
            if (ii <= iend) {
                if (ii == own) m_own = m;
                if (ii == own + 64) m_next = m;
                const unsigned g = (unsigned)(g4 >> (16 * u)) & 0xffffu;
                const unsigned ge = g & 0xffu;
                // lib/psy.c:851-881: pop while the new seed is not below the top (ge) and the top is not above the
                // entry below it (le), both within reach (inside the 7-bit window)
                for (;;) {
                    const unsigned m2 = m & (m - 1);
                    if (!m2) break;
                    const int k0 = __ffs((int)m) - 1, k1 = __ffs((int)m2) - 1;
                    const unsigned c1 = (ge >> k0) & 1u;
                    const unsigned c2 = (unsigned)(H >> (8 * k0 + (k1 - k0 - 1))) & 1u;
                    if (!(c1 & c2)) break;
                    m = m2;
                }
                const int q = ii - 7 - own;     // line ii-7 leaves the window: its fate is final
                if (q >= 0 && q < 64) alive |= (unsigned long long)((m >> 6) & 1u) << q;
                m = ((m << 1) | 1u) & 0x7fu;
                H = (H << 8) | (g >> 8);
            }
        }
    }
    if (iend + 1 == own + 64) m_next = m;
#pragma unroll
    for (int b = 0; b < 7; b++) {           // lines the walk ended on top of
        const int q = iend - b - own;
        if (q >= 0 && q < 64 && iend - b + 7 > iend) alive |= (unsigned long long)((m >> b) & 1u) << q;
    }
}

// TM_NB channel-blocks per workgroup, 32 threads per block
template <int TM_NB>
__global__ __launch_bounds__(TM_NB * 32) void k_tonemask(vbm_batch b, const int phases, const int runin)   // phases: timing experiments (31 = all)
{
    constexpr int TM_THREADS = TM_NB * 32;
    extern __shared__ __align__(16) int tm_lds[];
    __shared__ float s_att[TM_NB], s_dboff[TM_NB];
    __shared__ unsigned long long s_alive[TM_NB][TM_CHUNKS];
    const vbm_psy *p = &b.setup->psy[b.block_mode];
    const int n = p->n, tn = p->total_octave_lines, linesper = p->eighth_octave_lines;
    const int tnp = tn | 1;                         // odd row pitch of the seed rows
    const int glp = (tn + 8) & ~7;                  // row pitch of the compare bytes
    float *seedF = (float *)tm_lds;                 // [NB][tnp] seeds: stamped with the LDS float maximum (ds_max_f32)
    unsigned short *glS = (unsigned short *)(tm_lds + TM_NB * tnp + (TM_NB & 1));   // [NB][glp]
    const int tid = threadIdx.x;
    const int cb0 = blockIdx.x * TM_NB;
    if (cb0 >= vbm_ncb(b)) return;                  // (the launch covers the batch's bound; the count lives on the device)
    const int nblk = (vbm_ncb(b) - cb0 < TM_NB) ? vbm_ncb(b) - cb0 : TM_NB;

    for (int k = tid; k < TM_NB * tnp; k += TM_THREADS) seedF[k] = NEGINF;
    if (tid < TM_NB) {
        const int cb = cb0 + (tid < nblk ? tid : 0);
        float att = b.local_ampmax[cb] + p->ath_adjatt;
        if (att < p->ath_maxatt) att = p->ath_maxatt;
        s_att[tid] = att;
        s_dboff[tid] = p->max_curve_dB - b.global_ampmax[cb / b.ch];
    }
    __syncthreads();

    // ---- stamp ------------------------------------------------------------------------------------------
    if (phases & 1) {
        const int ngroups = p->ngroups;
        const int4 *__restrict__ group_tab = (const int4 *)p->group_tab;
        const float *__restrict__ tonecurves = p->tonecurves;
        const int shiftoc = p->shiftoc, firstoc = p->firstoc;
        // (items = (block, group) pairs dealt round-robin over the threads; block and group advance without a division)
        int blk = 0, g = tid;
        while (g >= ngroups) { g -= ngroups; blk++; }
        for (int item = tid; item < nblk * ngroups; item += TM_THREADS, g += TM_THREADS) {
            while (g >= ngroups) { g -= ngroups; blk++; }
            const float *f = b.logfft_bm + (size_t)(cb0 + blk) * n;
            const int4 rec = group_tab[g];              // first bin, end bin, ath[last], octave[last]
            const int s0 = rec.x, s1 = rec.y;
            float max = f[s0];
            for (int i = s0 + 1; i < s1; i += 4) {              // four bins in flight
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = f[(i + u < s1) ? i + u : s1 - 1];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (v[u] > max) max = v[u];
            }
            if (max + 6.f > __int_as_float(rec.z) + s_att[blk]) {
                long oc = rec.w;
                oc = oc >> shiftoc;
                if (oc >= VBM_P_BANDS) oc = VBM_P_BANDS - 1;
                if (oc < 0) oc = 0;
                const float *curves = tonecurves + (size_t)oc * VBM_P_LEVELS * (VBM_EHMER_MAX + 2);
                const int ocl = rec.w - firstoc;
                int choice = (int)(((double)(max + s_dboff[blk]) - 30.) * (double).1f);   // P_LEVEL_0 = 30. (double)
                choice = VMAX(choice, 0);
                choice = VMIN(choice, VBM_P_LEVELS - 1);
                const float *posts = curves + choice * (VBM_EHMER_MAX + 2);
                const float *curve = posts + 2;
                const int post1 = (int)posts[1];
                int seedptr = (int)((float)ocl + (posts[0] - VBM_EHMER_OFFSET) * linesper - (linesper >> 1));
                float *row = seedF + blk * tnp;
                // curve values eight at a time ahead of their stamps (table reads are global loads)
                for (int i = (int)posts[0]; i < post1 && seedptr < tn; i += 8) {
                    float cv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) cv[u] = curve[(i + u < post1) ? i + u : post1 - 1];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (i + u < post1 && seedptr < tn) {
                            // (ds_max_f32: the maximum of floats is order free; no NaNs, and a sum of two finite floats is never -0)
                            if (seedptr > 0) (void)__hip_atomic_fetch_max(&row[seedptr], max + cv[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            seedptr += linesper;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- compare: GE bit k-1 = !(s[j] < s[j-k]), LE bit d-1 = (s[j] <= s[j-d]) ---------------------------------
    // A thread takes eight consecutive lines: their seeds and the seven before them are read once (15 LDS reads for
    // 8 x 13 comparisons; a line at a time it was 8 reads and the index arithmetic per line: 44 M of the kernel's 209 M
    // vector instructions per launch, rocprofv3 SQ_INSTS_VALU with the phase switched off), the eight results go out as
    // one 16-byte store (the rows are 8-line aligned: glp is a multiple of 8, the row's base a multiple of 32 bytes).
    if (phases & 2) {
        const int ngrp = (tn + 7) >> 3;
        int cblk = 0, cg = tid;
        for (int item = tid; item < nblk * ngrp; item += TM_THREADS, cg += TM_THREADS) {
            while (cg >= ngrp) { cg -= ngrp; cblk++; }
            const int j0 = cg << 3;
            const float *sd = seedF + cblk * tnp;
            float w[15];                               // w[7 + u] = seed of line j0 + u, w[7 - k] = seed of line j0 - k
#pragma unroll
            for (int t = 0; t < 15; t++) {
                int q = j0 - 7 + t;
                q = q < 0 ? 0 : (q < tn ? q : tn - 1);
                w[t] = sd[q];
            }
            unsigned r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + u;
                unsigned ge = 0, le = 0;
#pragma unroll
                for (int k = 1; k <= 7; k++) {
                    if (j - k >= 0) {
                        if (!(w[7 + u] < w[7 + u - k])) ge |= 1u << (k - 1);
                        if (k <= 6 && w[7 + u] <= w[7 + u - k]) le |= 1u << (k - 1);
                    }
                }
                r[u >> 1] |= (ge | (le << 8)) << (16 * (u & 1));
            }
            *reinterpret_cast<uint4 *>(glS + cblk * glp + j0) = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
    __syncthreads();

    // ---- chase: 16 lanes per block, 64 lines each --------------------------------------------------------------
    if (phases & 4) {
        const int blk = tid / TM_CHUNKS, c = tid % TM_CHUNKS;      // threads [0, 16 NB): the first half of the workgroup
        const int own = c * 64;
        const bool active = blk < nblk && own < tn;
        const unsigned short *gl = glS + blk * glp;
        unsigned long long alive = 0;
        unsigned m_own = 0, m_next = 0, used = 0;
        int iend = own + 70;
        if (iend > tn - 1) iend = tn - 1;
        if (active) {
            const int i0 = own >= runin ? own - runin : 0;     // runin: multiple of 4
            chase_run(gl, i0, iend, 0u, own, alive, m_own, m_next);
            used = m_own;
        }
        // a lane whose assumed state differs from what its left neighbour arrived at runs again from the true one
        for (;;) {
            const unsigned truth = __shfl_up(m_next, 1);
            const bool redo = active && c > 0 && truth != used;
            if (!__any(redo)) break;
            if (redo) {
                chase_run(gl, own, iend, truth, own, alive, m_own, m_next);
                used = truth;
            }
        }
        if (blk < nblk) s_alive[blk][c] = active ? alive : 0ull;   // (blk >= TM_NB >= nblk for the second half)
    }
    __syncthreads();

    // ---- fill: wavefront w takes blocks w, w + 4, ...; lane = line within a 64-line chunk ------------------------
    if (phases & 8) {
        const int wave = tid >> 6, lane = tid & 63;
        for (int blk = wave; blk < nblk; blk += TM_THREADS / 64) {
            float *sd = seedF + blk * tnp;
            float am[TM_CHUNKS];
            int st[TM_CHUNKS], en[TM_CHUNKS];
            int carry = 0;                              // largest end among the survivors before this chunk
#pragma unroll
            for (int c = 0; c < TM_CHUNKS; c++) {
                am[c] = 0.f; st[c] = 0; en[c] = 0;
                if (c * 64 < tn) {
                    const unsigned long long w = s_alive[blk][c];
                    const int q = c * 64 + lane;
                    int endpos = 0;
                    if ((w >> lane) & 1ull) {
                        am[c] = sd[q];
                        const unsigned long long above = lane < 63 ? (w >> (lane + 1)) : 0ull;
                        int nq = -1;                    // the next survivor (never more than 8 lines on)
                        if (above) nq = q + 1 + (__ffsll((long long)above) - 1);
                        else if (c + 1 < TM_CHUNKS && (c + 1) * 64 < tn) {
                            const unsigned long long w2 = s_alive[blk][c + 1];
                            if (w2) nq = (c + 1) * 64 + (__ffsll((long long)w2) - 1);
                        }
                        if (nq >= 0 && sd[nq] > am[c]) endpos = nq;
                        else endpos = q + linesper + 1;
                        if (endpos > tn) endpos = tn;
                    }
                    int incl = endpos;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int t = __shfl_up(incl, d);
                        if (lane >= d && t > incl) incl = t;
                    }
                    int excl = __shfl_up(incl, 1);
                    if (lane == 0) excl = 0;
                    if (carry > excl) excl = carry;
                    st[c] = excl;
                    en[c] = endpos;
                    const int tot = __shfl(incl, 63);
                    if (tot > carry) carry = tot;
                }
            }
            // every survivor is in registers: now the lines may be overwritten (LDS operations of a wavefront stay in order)
#pragma unroll
            for (int c = 0; c < TM_CHUNKS; c++)
                for (int x = st[c]; x < en[c]; x++) sd[x] = am[c];
        }
    }
    __syncthreads();

    // ---- apply: tone[i] = max(ath[i] + att, min over the bin's seed segment) -----------------------------------
    if (phases & 16) {
        const int blk = tid % TM_NB, r0 = tid / TM_NB;
        if (blk < nblk) {
            const int cb = cb0 + blk;
            float *out = b.toneT + (size_t)(cb >> 6) * b.slab_words + (cb & 63);
            const float *sd = seedF + blk * tnp;
            const float att = s_att[blk], tone_abs_limit = p->tone_abs_limit;
            const int *__restrict__ seg_p0 = p->seg_p0, *__restrict__ seg_p1 = p->seg_p1;
            const float *__restrict__ ath = p->ath;
            constexpr int STEP = TM_THREADS / TM_NB;
            for (int i0 = r0; i0 < n; i0 += 4 * STEP) {          // four bins' table reads in flight
                int q0[4], q1[4];
                float at[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = (i0 + u * STEP < n) ? i0 + u * STEP : r0;
                    q0[u] = seg_p0[i]; q1[u] = seg_p1[i]; at[u] = ath[i];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * STEP;
                    if (i < n) {
                        float minV;
                        if (q0[u] < 0) {
                            minV = sd[tn - 1];
                        } else {
                            minV = sd[q0[u]];
                            if (minV > tone_abs_limit) minV = tone_abs_limit;
                            for (int pos = q0[u] + 1; pos <= q1[u]; pos++) {
                                const float sv = sd[pos];
                                if ((sv > NEGINF && sv < minV) || minV == NEGINF) minV = sv;
                            }
                        }
                        float v = at[u] + att;
                        if (v < minV) v = minV;
                        out[(size_t)i * 64] = v;
                    }
                }
            }
        }
    }
}

template <int NB>
int launch(const vbm_batch *b, int tn, size_t lds, int phases, hipStream_t st)
{
    // (launchers are called from several host threads at once — one per block type of a round: one-time set-up goes
    // through initialisers of function-local statics, which C++ runs once, and the LDS limit only ever grows)
    static const int runin = [] {
        int r = getenv("VBM_TONE_RUNIN") ? atoi(getenv("VBM_TONE_RUNIN")) : 16;
        r = (r + 3) & ~3;
        return (r < 8 || r > 64) ? 16 : r;
    }();
    {
        static std::mutex mu;
        static size_t allowed = 0;
        std::lock_guard<std::mutex> guard(mu);
        if (lds > allowed) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_tonemask<NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return -2;
            allowed = lds;
        }
    }
    hipLaunchKernelGGL(k_tonemask<NB>, dim3((unsigned)((b->ncb + NB - 1) / NB)), dim3(NB * 32), lds, st, *b, phases, runin);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

size_t lds_bytes(int nb, int tn) { return (size_t)(nb * (tn | 1)) * 4 + (size_t)nb * ((tn + 8) & ~7) * 2 + 64; }

}  // namespace

// tn: total_octave_lines of the batch's psy look (host copy)
extern "C" int vbm_launch_tonemask(const vbm_batch *b, int tn, hipStream_t st)
{
    if (tn > TM_CHUNKS * 64 || tn < 2) return -2;
    static const int phases = getenv("VBM_TONE_PHASES") ? atoi(getenv("VBM_TONE_PHASES")) : 31;
    // 8 blocks per workgroup (256 threads, ~38 KB of LDS).  Measured on MI355X, 16384 stereo streams: alone the kernel
    // takes the same 0.7 ms with 8, 16 or 32 blocks per workgroup, but beside the noise-mask branch and the previous
    // step's back half (MDCT, couple and residue-VQ workgroups want LDS too) the step takes 3.07 / 3.19 / 3.22 ms
    // (round 3, four blocks per workgroup: per-block step 2.93 against 2.91, from PCM 4.83-5.02 against 4.67-4.71).
    static const int force = getenv("VBM_TONE_NB") ? atoi(getenv("VBM_TONE_NB")) : 8;
    if (force == 16) return launch<16>(b, tn, lds_bytes(16, tn), phases, st);
    return launch<8>(b, tn, lds_bytes(8, tn), phases, st);
}
