// Stream front end kernels for gfx950 (frontend.h): PCM intake, LPC extrapolation of stream start
// and end, the envelope detector and the block carve-out decision, for S streams per launch.
//
//   k_fe_append / k_fe_commit       vorbis_analysis_buffer + vorbis_analysis_wrote(vals > 0)
//                                   (reference lib/block.c:405-436, :511-553): pre_amplitude, append
//   k_fe_extrapolate                _preextrapolate_helper (lib/block.c:438-484) and the end-of-stream
//                                   padding of vorbis_analysis_wrote(0) (:520-552), with
//                                   vorbis_lpc_from_data / vorbis_lpc_predict (lib/lpc.c:60-159).
//                                   One lane per channel: Levinson-Durbin in double, in source order.
//   k_fe_ve_range                   which search steps [first, last) _ve_envelope_search still has to
//                                   evaluate (lib/envelope.c:573-577)
//   (128-point search MDCTs: mdct_kernel.hip, vbm_launch_ve_mdct)
//   k_fe_ve_filter                  _ve_amp after its MDCT (lib/envelope.c:127-562, scalar branch) and
//                                   the mark bookkeeping of _ve_envelope_search (:590-625); one lane per
//                                   stream, because ve->stretch couples the channels step by step
//   k_fe_decide                     the cursor walk of _ve_envelope_search (:631-678), _ve_envelope_mark
//                                   (:683-707), vorbis_analysis_blockout (lib/block.c:557-812) without
//                                   its copies, _ve_envelope_shift (:709-728)
//   k_fe_gather                     the block copy of vorbis_analysis_blockout (lib/block.c:653-698)
//                                   into block-major batches for vbm_analysis_batch
//   k_fe_shift                      the memmove of lib/block.c:757-759 as a copy into the other buffer
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "frontend.h"
#include "frontend_kernels.h"

namespace {

__device__ __forceinline__ float *chan_buf(const vbm_fe_state &f, int s, int c)
{
    return f.pcm + (long)f.parity[s] * f.plane + ((long)s * f.ch + c) * f.cap + f.base[s];
}

// ---------------------------------------------------------------------------------------------
__global__ void k_fe_append(vbm_fe_state f, const float *__restrict__ src, int vals, float pre_amplitude)
{
    const int c = blockIdx.x;                       // channel index over S*ch
    const int s = c / f.ch;
    if (f.base[s] + f.pcm_current[s] + vals > f.cap) return;   // full (k_fe_commit counts it): never write past the buffer
    float *dst = f.pcm + (long)f.parity[s] * f.plane + (long)c * f.cap + f.base[s] + f.pcm_current[s];
    const float *in = src + (long)c * vals;
    const int t0 = blockIdx.y * blockDim.x + threadIdx.x, step = gridDim.y * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)in) & 15) == 0 && (vals & 3) == 0) {      // 16 bytes per thread and step
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        for (int i = t0; i < vals / 4; i += step) {
            float4 v = in4[i];
            v.x *= pre_amplitude; v.y *= pre_amplitude; v.z *= pre_amplitude; v.w *= pre_amplitude;   // lib/block.c:514-518
            dst4[i] = v;
        }
        return;
    }
    for (int i = t0; i < vals; i += step)
        dst[i] = in[i] * pre_amplitude;             // lib/block.c:514-518
}

__global__ void k_fe_commit(vbm_fe_state f, int vals)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    if (f.base[s] + f.pcm_current[s] + vals > f.cap) { atomicAdd(f.overflow, 1); return; }
    f.pcm_current[s] += vals;
}

// the same for a subset of the streams: channel c of stream ids[k] comes from src + (by_slot ? ids[k] : k) * stream_stride
// + c * ch_stride (plain [k][ch][vals]: strides ch * vals and vals).  src may be pinned host memory (the drop-in shim's
// staging arena, read over the bus by this kernel: 16 bytes per thread and step when everything is aligned)
__global__ void k_fe_append_ids(vbm_fe_state f, const int *__restrict__ ids, const float *__restrict__ src, int vals,
                                float pre_amplitude, long stream_stride, long ch_stride, int by_slot)
{
    const int kc = blockIdx.x;                      // listed stream k, channel c
    const int k = kc / f.ch, c = kc % f.ch;
    const int s = ids[k];
    if (f.base[s] + f.pcm_current[s] + vals > f.cap) return;
    float *dst = f.pcm + (long)f.parity[s] * f.plane + ((long)s * f.ch + c) * f.cap + f.base[s] + f.pcm_current[s];
    const float *in = src + (long)(by_slot ? s : k) * stream_stride + (long)c * ch_stride;
    const int t0 = blockIdx.y * blockDim.x + threadIdx.x, step = gridDim.y * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)in) & 15) == 0 && (vals & 3) == 0) {
        const float4 *in4 = reinterpret_cast<const float4 *>(in);
        float4 *dst4 = reinterpret_cast<float4 *>(dst);
        for (int i = t0; i < vals / 4; i += step) {
            float4 v = in4[i];
            v.x *= pre_amplitude; v.y *= pre_amplitude; v.z *= pre_amplitude; v.w *= pre_amplitude;   // lib/block.c:514-518
            dst4[i] = v;
        }
        return;
    }
    for (int i = t0; i < vals; i += step)
        dst[i] = in[i] * pre_amplitude;
}

__global__ void k_fe_commit_ids(vbm_fe_state f, const int *__restrict__ ids, int n, int vals)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = ids[k];
    if (f.base[s] + f.pcm_current[s] + vals > f.cap) { atomicAdd(f.overflow, 1); return; }
    f.pcm_current[s] += vals;
}

// vorbis_analysis_init state for the listed streams (a new stream starts in a used slot): lib/block.c:306-344,
// _ve_envelope_init lib/envelope.c:42-87.  One lane per (listed stream, channel); channel 0 also resets the
// per-stream scalars and marks.
__global__ void k_fe_restart(vbm_fe_state f, const int *__restrict__ ids, int n, int long_n)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n * f.ch) return;
    const int s = ids[lane / f.ch], ci = lane % f.ch;
    const long SC = (long)f.S * f.ch, c = (long)s * f.ch + ci;
    // the reference's buffer starts zeroed: the first centerW samples are silence until they are extrapolated
    float *pcm = f.pcm + (long)f.parity[s] * f.plane + c * f.cap;   // (the new stream starts at the buffer's start)
    for (int i = 0; i < long_n / 2; i++) pcm[i] = 0.f;
    for (int k = 0; k < VBM_VE_AMP; k++)
        for (int jb = 0; jb < 16; jb++) f.ve_ampbuf[((long)k * SC + c) * 16 + jb] = 0.f;
    for (int jb = 0; jb < 16; jb++) f.ve_ampptr[c * 16 + jb] = 0;
    for (int k = 0; k < VBM_VE_NEARDC; k++) f.ve_nearDC[(long)k * SC + c] = 0.f;
    f.ve_nearacc[c] = 0.f;
    f.ve_nearacc[SC + c] = 0.f;
    f.ve_nearptr[c] = 0;
    if (ci == 0) {
        f.base[s] = 0;
        f.pcm_current[s] = long_n / 2; f.centerW[s] = long_n / 2;
        f.lW[s] = 0; f.W[s] = 0; f.nW[s] = 0; f.eofflag[s] = 0; f.preextrapolate[s] = 0;
        f.granulepos[s] = 0; f.sequence[s] = 3;
        f.ve_current[s] = 0; f.ve_cursor[s] = long_n / 2; f.ve_curmark[s] = -1; f.ve_stretch[s] = 0;
        for (int k = 0; k < f.marks; k++) f.ve_mark[(long)k * f.S + s] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// vorbis_lpc_from_data (lib/lpc.c:60-130).  data(i) is an accessor; M = order
template <int M, typename Acc>
__device__ void lpc_from_data(Acc data, float *lpci, int n)
{
    double aut[M + 1], lpc[M];
    double error, epsilon;
    int i, j;

    j = M + 1;
    while (j--) {
        double d = 0;
        for (i = j; i < n; i++) d += (double)data(i) * (double)data(i - j);
        aut[j] = d;
    }

    error = aut[0] * (1. + 1e-10);
    epsilon = 1e-9 * aut[0] + 1e-10;

    for (i = 0; i < M; i++) {
        double r = -aut[i + 1];
        if (error < epsilon) {
            for (j = i; j < M; j++) lpc[j] = 0.;
            break;
        }
        for (j = 0; j < i; j++) r -= lpc[j] * aut[i - j];
        r /= error;

        lpc[i] = r;
        for (j = 0; j < i / 2; j++) {
            double tmp = lpc[j];
            lpc[j] += r * lpc[i - 1 - j];
            lpc[i - 1 - j] += r * tmp;
        }
        if (i & 1) lpc[j] += lpc[j] * r;

        error *= 1. - r * r;
    }
    {
        double g = .99;
        double damp = g;
        for (j = 0; j < M; j++) {
            lpc[j] *= damp;
            damp *= g;
        }
    }
    for (j = 0; j < M; j++) lpci[j] = (float)lpc[j];
}

// vorbis_lpc_predict (lib/lpc.c:132-159): prime(i) = the M samples before the first predicted one,
// out(i, y) stores sample i.  The M-sample history slides through registers.
template <int M, typename Prime, typename Out>
__device__ void lpc_predict(const float *coeff, Prime prime, Out out, long n)
{
    float w[M];
#pragma unroll
    for (int i = 0; i < M; i++) w[i] = prime(i);
    for (long i = 0; i < n; i++) {
        float y = 0;
#pragma unroll
        for (int j = 0; j < M; j++) y -= w[j] * coeff[M - 1 - j];
#pragma unroll
        for (int j = 0; j < M - 1; j++) w[j] = w[j + 1];
        w[M - 1] = y;
        out(i, y);
    }
}

// mode 0: start-of-stream pre-extrapolation for every stream that just crossed the threshold of
//         vorbis_analysis_wrote (lib/block.c:547-550); ids == nullptr, lanes cover all channels.
// mode 1: vorbis_analysis_wrote(v, 0) for the listed streams (lib/block.c:520-545).
__global__ void k_fe_extrapolate(vbm_fe_state f, const int *__restrict__ ids, int nids, int mode, int long_n)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = (mode == 0 ? f.S : nids) * f.ch;
    if (lane >= nch) return;
    const int s = mode == 0 ? lane / f.ch : ids[lane / f.ch];
    const int c = lane % f.ch;
    float *pcm = chan_buf(f, s, c);
    const int pc = f.pcm_current[s], centerW = f.centerW[s];

    if (!f.preextrapolate[s] && (mode == 1 || pc - centerW > long_n)) {
        // _preextrapolate_helper: work[j] = pcm[pc-1-j]; LPC on the real data (reversed), then
        // predict the centerW samples in front of it
        const int order = 16;
        if (pc - centerW > order * 2) {
            float lpc[16];
            const int n = pc - centerW;
            lpc_from_data<16>([&](int i) { return pcm[pc - 1 - i]; }, lpc, n);
            lpc_predict<16>(lpc, [&](int i) { return pcm[pc - 1 - (n - order + i)]; },
                            [&](long i, float y) { pcm[pc - 1 - (n + i)] = y; }, centerW);
        }
    }
    if (mode == 1) {
        const int order = 32;
        const int eof = pc, newpc = pc + long_n * 3;
        if (eof > order * 2) {
            float lpc[32];
            int n = eof;
            if (n > long_n) n = long_n;
            const float *d = pcm + eof - n;
            lpc_from_data<32>([&](int i) { return d[i]; }, lpc, n);
            lpc_predict<32>(lpc, [&](int i) { return pcm[eof - order + i]; },
                            [&](long i, float y) { pcm[eof + i] = y; }, newpc - eof);
        } else {
            for (int i = eof; i < newpc; i++) pcm[i] = 0.f;
        }
    }
}

// state part of the two paths above, after every channel has been extrapolated
__global__ void k_fe_extrapolate_commit(vbm_fe_state f, const int *__restrict__ ids, int nids, int mode, int long_n)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= (mode == 0 ? f.S : nids)) return;
    const int s = mode == 0 ? lane : ids[lane];
    if (mode == 0) {
        if (!f.preextrapolate[s] && f.pcm_current[s] - f.centerW[s] > long_n) f.preextrapolate[s] = 1;
    } else {
        f.preextrapolate[s] = 1;
        f.eofflag[s] = f.pcm_current[s];
        f.pcm_current[s] += long_n * 3;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void k_fe_ve_range(vbm_fe_state f)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    const int searchstep = 64;
    int first = f.ve_current[s] / searchstep;
    int last = f.pcm_current[s] / searchstep - VBM_VE_WIN;
    if (first < 0) first = 0;
    // vorbis_analysis_blockout only reaches the search once the start is extrapolated and the stream
    // has not ended (lib/block.c:564-566)
    if (!f.preextrapolate[s] || f.eofflag[s] == -1) last = first;
    f.ve_first[s] = first;
    f.ve_last[s] = last;
}

// lib/scales.h:43-51
__device__ __forceinline__ float fe_todB(float x)
{
    uint32_t i = __float_as_uint(x) & 0x7fffffffu;
    return (float)((float)i * 7.17711438e-7f - 764.6161886f);
}

// steps [first + t0, min(last, first + t0 + CHUNK)) of every stream.  Sixteen lanes per stream (four
// streams per wavefront): lane jb < 12 owns band jb of the detector, lane 0 also the near-DC ring; the
// 32 smoothed spectrum values of a (step, channel) are produced two per lane and shared through LDS.
// The steps of a stream stay in order (ve->stretch couples them), channels and bands run side by side.
//
// The walk is a chain: every (step, channel) reads and updates the band's amplitude ring (VE_AMP entries, up to 13
// of them read back per step) and the near-DC ring.  In HBM a ring's entries lie megabytes apart ([entry][channel]
// [band]), so every read was a round trip of its own, 3 .. 13 in a row per step: ~18 us per (step, channel), 0.57 ms
// per write of 16384 stereo streams.  Both rings and their cursors are therefore brought into LDS once at the start
// (all loads in flight together), walked there, and written back at the end; the spectrum values of the next
// (step, channel) are fetched while the current one is worked on.
// LDS per stream: ch x (VE_AMP x 16 + 16 + 16 + 4) floats (dynamic: 4 streams per workgroup).
__global__ __launch_bounds__(64) void k_fe_ve_filter(vbm_fe_state f, const vbm_setup *__restrict__ setup, int t0)
{
    __shared__ float s_vec[4][32];
    extern __shared__ float s_ring[];
    const int grp = threadIdx.x >> 4, jb = threadIdx.x & 15;
    const int s = blockIdx.x * 4 + grp;
    if (s >= f.S) return;
    const vbm_envelope *ve = &setup->ve;
    const long SC = (long)f.S * f.ch;
    const int first = f.ve_first[s] + t0, last = f.ve_last[s];
    if (first >= last) return;
    const int ch = f.ch;
    const float minV = ve->minenergy;
    const float stretch_penalty = ve->stretch_penalty;
    int ve_stretch = f.ve_stretch[s];
    const int S = f.S;
    const bool band = jb < VBM_VE_BANDS;
    // this lane's band constants
    const int begin = band ? ve->band_begin[jb] : 0, end = band ? ve->band_end[jb] : 0;
    const float total = band ? ve->band_total[jb] : 0.f;
    const float pre_t = band ? ve->preecho_thresh[jb] : 0.f, post_t = band ? ve->postecho_thresh[jb] : 0.f;
    float bw[VBM_VE_MAXBAND];
#pragma unroll
    for (int i = 0; i < VBM_VE_MAXBAND; i++) bw[i] = band ? ve->band_window[jb][i] : 0.f;

    // ---- the stream's rings into LDS: per channel amp[VE_AMP][16], ampptr[16], nearDC[16 (15 used)], near state[4]
    constexpr int PER_CH = VBM_VE_AMP * 16 + 16 + 16 + 4;
    float *R = s_ring + (size_t)grp * ch * PER_CH;
    for (int ci = 0; ci < ch; ci++) {
        const long c = (long)s * ch + ci;
        float *rc = R + ci * PER_CH;
#pragma unroll
        for (int k = 0; k < VBM_VE_AMP; k++) rc[k * 16 + jb] = f.ve_ampbuf[((long)k * SC + c) * 16 + jb];
        reinterpret_cast<int *>(rc + VBM_VE_AMP * 16)[jb] = f.ve_ampptr[c * 16 + jb];
        if (jb < VBM_VE_NEARDC) rc[VBM_VE_AMP * 16 + 16 + jb] = f.ve_nearDC[(long)jb * SC + c];
        if (jb == 0) {
            float *ns = rc + VBM_VE_AMP * 16 + 32;
            ns[0] = f.ve_nearacc[c];
            ns[1] = f.ve_nearacc[SC + c];
            reinterpret_cast<int *>(ns)[2] = f.ve_nearptr[c];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // (the lanes of a stream share a wavefront: its LDS accesses are performed in program order, so the exchanges
    // below need the compiler to keep that order — a wavefront-scope fence — and no wait for memory)
    const int nsteps = (last - first < VBM_FE_CHUNK) ? last - first : VBM_FE_CHUNK;
    const int total_it = nsteps * ch;
    // spectrum values of (step t, channel ci): two per lane, three more for lane 0
    // (step t and channel ci of an item advance without divisions: it = t * ch + ci)
    auto spec_of = [&](int t_, int ci_) { return f.ve_spec + (((long)s * ch + ci_) * VBM_FE_CHUNK + t_) * 64; };
    float na = 0.f, nbq = 0.f, na2 = 0.f, nbq2 = 0.f, nh0 = 0.f, nh1 = 0.f, nh2 = 0.f;
    auto fetch = [&](int t_, int ci_) {
        const float *sp = spec_of(t_, ci_);
        na = sp[2 * jb]; nbq = sp[2 * jb + 1];
        na2 = sp[2 * (jb + 16)]; nbq2 = sp[2 * (jb + 16) + 1];
        if (jb == 0) { nh0 = sp[0]; nh1 = sp[1]; nh2 = sp[2]; }
    };
    fetch(0, 0);

    int ret = 0, stretch = 0;
    float penalty = 0.f;
    int t = 0, ci = 0;
    for (int it = 0; it < total_it; it++, ci++) {
        if (ci == ch) { ci = 0; t++; }
        const int j = first + t;
        if (ci == 0) {
            ret = 0;
            ve_stretch++;
            if (ve_stretch > VBM_VE_MAXSTRETCH * 2) ve_stretch = VBM_VE_MAXSTRETCH * 2;

            // _ve_amp prologue (lib/envelope.c:112-119)
            stretch = ve_stretch / 2;
            if (stretch < VBM_VE_MINSTRETCH) stretch = VBM_VE_MINSTRETCH;
            penalty = stretch_penalty - (ve_stretch / 2 - VBM_VE_MINSTRETCH);
            if (penalty < 0.f) penalty = 0.f;
            if (penalty > stretch_penalty) penalty = stretch_penalty;
        }
        const float a = na, bq = nbq, a2 = na2, bq2 = nbq2, h0 = nh0, h1 = nh1, h2 = nh2;
        if (it + 1 < total_it) fetch(ci + 1 == ch ? t + 1 : t, ci + 1 == ch ? 0 : ci + 1);

        float *rc = R + ci * PER_CH;
        float decay = 0.f;

        // near-DC spreading function (:127-148), lane 0 of the group
        if (jb == 0) {
            float *dc = rc + VBM_VE_AMP * 16 + 16;
            float *ns = rc + VBM_VE_AMP * 16 + 32;
            float temp = (float)((double)(h0 * h0) + (.7 * (double)h1) * (double)h1 + (.2 * (double)h2) * (double)h2);
            int ptr = reinterpret_cast<int *>(ns)[2];
            float acc = ns[0], pacc = ns[1];
            if (ptr == 0) {
                decay = acc = pacc + temp;
                pacc = temp;
            } else {
                decay = acc += temp;
                pacc += temp;
            }
            acc -= dc[ptr];
            dc[ptr] = temp;
            ns[0] = acc;
            ns[1] = pacc;
            decay = (float)((double)decay * (1. / (VBM_VE_NEARDC + 1)));
            ptr++;
            if (ptr >= VBM_VE_NEARDC) ptr = 0;
            reinterpret_cast<int *>(ns)[2] = ptr;
            decay = (float)((double)fe_todB(decay) * .5 - (double)15.f);
        }
        decay = __shfl(decay, threadIdx.x & 48);   // from lane 0 of this 16-lane group

        // spreading, limiting, spectrum smoothing (:151-159): value k uses decay after k subtractions of 8
        // (subtracted one at a time, as the source rounds after each)
        {
            float dk = decay;
            for (int k = 0; k < 32; k++) {
                if (k == jb || k == jb + 16) {
                    const float x = (k == jb) ? a : a2, y = (k == jb) ? bq : bq2;
                    float val = x * x + y * y;
                    val = fe_todB(val) * .5f;
                    if (val < dk) val = dk;
                    if (val < minV) val = minV;
                    s_vec[grp][k] = val;
                }
                dk = dk - 8.f;      // (the source's decay -= 8. in double, rounded back to float: one exact difference, one rounding = the float subtraction)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // preecho / postecho triggering by band (:162-559 scalar), one band per lane
        if (band) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < VBM_VE_MAXBAND; i++)          // (unrolled: bw[] stays in registers with static indices)
                if (i < end) acc += s_vec[grp][i + begin] * bw[i];
            acc *= total;

            float *ampbuf = rc + jb;                               // element k at [k * 16]
            int *aptr = reinterpret_cast<int *>(rc + VBM_VE_AMP * 16) + jb;
            const int cur = *aptr;
            float postmax, postmin, premax = -99999.f, premin = 99999.f;
            int p = cur;
            p--;
            if (p < 0) p += VBM_VE_AMP;
            {
                const float v = ampbuf[p * 16];
                postmax = acc > v ? acc : v;
                postmin = acc < v ? acc : v;
            }
            for (int i = 0; i < stretch; i++) {
                p--;
                if (p < 0) p += VBM_VE_AMP;
                const float v = ampbuf[p * 16];
                premax = premax > v ? premax : v;
                premin = premin < v ? premin : v;
            }
            const float valmin = postmin - premin;
            const float valmax = postmax - premax;

            ampbuf[cur * 16] = acc;
            int np = cur + 1;
            if (np >= VBM_VE_AMP) np = 0;
            *aptr = np;

            if (valmax > pre_t + penalty) ret |= 1 | 4;
            if (valmin < post_t - penalty) ret |= 2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (ci == ch - 1) {
            // OR over the 16 lanes of the group
            ret |= __shfl_xor(ret, 1);
            ret |= __shfl_xor(ret, 2);
            ret |= __shfl_xor(ret, 4);
            ret |= __shfl_xor(ret, 8);

            // mark bookkeeping of _ve_envelope_search (lib/envelope.c:611-624)
            if (jb == 0) {
                f.ve_mark[(long)(j + VBM_VE_POST) * S + s] = 0;
                if (ret & 1) {
                    f.ve_mark[(long)j * S + s] = 1;
                    f.ve_mark[(long)(j + 1) * S + s] = 1;
                }
                if (ret & 2) {
                    f.ve_mark[(long)j * S + s] = 1;
                    if (j > 0) f.ve_mark[(long)(j - 1) * S + s] = 1;
                }
            }
            if (ret & 4) ve_stretch = -1;
        }
    }
    if (jb == 0) {
        f.ve_stretch[s] = ve_stretch;
        // _ve_envelope_search leaves ve->current at the last step it evaluated (lib/envelope.c:627) whether or not a
        // block comes out of the blockout call: a stream that delivers no block in this call (no data yet, or its
        // block type's lane region was full in every round) must not have these steps evaluated again by the next
        if (first + nsteps >= last) f.ve_current[s] = last * 64;
    }

    // ---- rings back
    for (int ci = 0; ci < ch; ci++) {
        const long c = (long)s * ch + ci;
        const float *rc = R + ci * PER_CH;
#pragma unroll
        for (int k = 0; k < VBM_VE_AMP; k++) f.ve_ampbuf[((long)k * SC + c) * 16 + jb] = rc[k * 16 + jb];
        f.ve_ampptr[c * 16 + jb] = reinterpret_cast<const int *>(rc + VBM_VE_AMP * 16)[jb];
        if (jb < VBM_VE_NEARDC) f.ve_nearDC[(long)jb * SC + c] = rc[VBM_VE_AMP * 16 + 16 + jb];
        if (jb == 0) {
            const float *ns = rc + VBM_VE_AMP * 16 + 32;
            f.ve_nearacc[c] = ns[0];
            f.ve_nearacc[SC + c] = ns[1];
            f.ve_nearptr[c] = reinterpret_cast<const int *>(ns)[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// vorbis_analysis_blockout for stream s (lib/block.c:557-812, _ve_envelope_search's cursor walk lib/envelope.c:627-680,
// _ve_envelope_mark :683-707, _ve_envelope_shift :709-728) without its copies.  COMMIT = false evaluates the same
// decision without touching the stream (what WOULD come out): the device-built rounds first look, then assign
// lanes, then commit (k_fe_classify / k_fe_plan / k_fe_commit).
template <bool COMMIT>
__device__ __forceinline__ void fe_decide_one(const vbm_fe_state &f, const vbm_setup *__restrict__ setup, const int s,
                                              vbm_fe_decision &d)
{
    const int S = f.S;
    const int searchstep = 64;
    const int bs0 = setup->blocksizes[0], bs1 = setup->blocksizes[1];
    d.ready = 0; d.lW = d.W = d.nW = 0; d.block_mode = 0; d.eos = 0; d.beginW = 0; d.movement = 0;
    d.granulepos = 0; d.sequence = 0;

    int W = f.W[s], lW = f.lW[s], nW = f.nW[s];
    int centerW = f.centerW[s], pcm_current = f.pcm_current[s], eofflag = f.eofflag[s];
    const int bsW = W ? bs1 : bs0;
    const int beginW = centerW - bsW / 2;

    if (!f.preextrapolate[s] || eofflag == -1) return;

    // _ve_envelope_search, after its evaluation loop (lib/envelope.c:627-680)
    int bp = -1;
    int curmark;
    {
        // ve->current = last * searchstep with last recomputed from the present pcm_current; every step
        // below it has been evaluated by k_fe_ve_filter before this kernel runs
        const int ve_current = (pcm_current / searchstep - VBM_VE_WIN) * searchstep;
        if (COMMIT) f.ve_current[s] = ve_current;
        const int testW = centerW + bsW / 4 + bs1 / 2 + bs0 / 4;
        int j = f.ve_cursor[s];
        int cursor = j;
        curmark = f.ve_curmark[s];
        while (j < ve_current - searchstep) {
            if (j >= testW) { bp = 1; break; }
            cursor = j;
            if (f.ve_mark[(long)(j / searchstep) * S + s]) {
                if (j > centerW) {
                    curmark = j;
                    bp = (j >= testW) ? 1 : 0;
                    break;
                }
            }
            j += searchstep;
        }
        if (COMMIT) {
            f.ve_cursor[s] = cursor;
            f.ve_curmark[s] = curmark;
        }
    }

    if (bp == -1) {
        if (eofflag == 0) return;   // not enough data yet
        nW = 0;
    } else {
        nW = (bs0 == bs1) ? 0 : bp;
    }
    if (COMMIT) f.nW[s] = nW;

    const int bsn = nW ? bs1 : bs0;
    const int centerNext = centerW + bsW / 4 + bsn / 4;
    {
        const int blockbound = centerNext + bsn / 2;
        if (pcm_current < blockbound) return;
    }

    // the block (lib/block.c:606-651)
    d.ready = 1;
    d.lW = lW; d.W = W; d.nW = nW;
    int blocktype;
    if (W) {
        blocktype = (!lW || !nW) ? 0 /* BLOCKTYPE_TRANSITION */ : 1 /* BLOCKTYPE_LONG */;
    } else {
        // _ve_envelope_mark (lib/envelope.c:683-707)
        int b0 = centerW - bs0 / 4 - bs0 / 4;
        int e0 = centerW + bs0 / 4 + bs0 / 4;
        int marked = 0;
        if (curmark >= b0 && curmark < e0) marked = 1;
        else {
            const int first = b0 / searchstep, last = e0 / searchstep;
            for (int i = first; i < last; i++)
                if (f.ve_mark[(long)i * S + s]) { marked = 1; break; }
        }
        blocktype = marked ? 0 /* BLOCKTYPE_IMPULSE */ : 1 /* BLOCKTYPE_PADDING */;
    }
    d.block_mode = blocktype | (W << 1);
    if (!COMMIT) return;
    d.sequence = f.sequence[s]++;
    long long granulepos = f.granulepos[s];
    d.granulepos = granulepos;
    d.beginW = beginW;

    if (eofflag) {
        if (centerW >= eofflag) {
            f.eofflag[s] = -1;
            d.eos = 1;
            return;
        }
    }

    // advance (lib/block.c:742-808)
    {
        const int new_centerNext = bs1 / 2;
        const int movementW = centerNext - new_centerNext;
        if (movementW > 0) {
            // _ve_envelope_shift (lib/envelope.c:709-728)
            {
                int ve_current = f.ve_current[s];
                const int smallsize = ve_current / searchstep + VBM_VE_POST;
                const int smallshift = movementW / searchstep;
                for (int k = 0; k < smallsize - smallshift; k++)
                    f.ve_mark[(long)k * S + s] = f.ve_mark[(long)(k + smallshift) * S + s];
                f.ve_current[s] = ve_current - movementW;
                int cm = f.ve_curmark[s];
                if (cm >= 0) f.ve_curmark[s] = cm - movementW;
                f.ve_cursor[s] -= movementW;
            }
            pcm_current -= movementW;
            f.pcm_current[s] = pcm_current;
            d.movement = movementW;

            f.lW[s] = W;
            f.W[s] = nW;
            f.centerW[s] = new_centerNext;
            centerW = new_centerNext;

            if (eofflag) {
                eofflag -= movementW;
                if (eofflag <= 0) eofflag = -1;
                f.eofflag[s] = eofflag;
                if (centerW >= eofflag) granulepos += movementW - (centerW - eofflag);
                else granulepos += movementW;
            } else {
                granulepos += movementW;
            }
            f.granulepos[s] = granulepos;
        }
    }
}

// hold (may be NULL): streams marked there are left alone in this round (no block, no state change)
__global__ void k_fe_decide(vbm_fe_state f, const vbm_setup *__restrict__ setup, vbm_fe_decision *__restrict__ out,
                            const uint8_t *__restrict__ hold)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    vbm_fe_decision d;
    if (hold && hold[s]) {
        d.ready = 0; d.lW = d.W = d.nW = 0; d.block_mode = 0; d.eos = 0; d.beginW = 0; d.movement = 0;
        d.granulepos = 0; d.sequence = 0;
    } else {
        fe_decide_one<true>(f, setup, s, d);
    }
    out[s] = d;
}

// ---- rounds built on the device ----------------------------------------------------------------------------
// k_fe_classify: which block type would every stream deliver now (-1: none)?  Nothing is changed.
__global__ void k_fe_classify(vbm_fe_state f, const vbm_setup *__restrict__ setup, signed char *__restrict__ type)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    vbm_fe_decision d;
    d.ready = 0;
    fe_decide_one<false>(f, setup, s, d);
    type[s] = d.ready ? (signed char)(d.block_mode & 3) : (signed char)-1;
}

// k_fe_plan (one workgroup): lanes for the blocks of a round.  Block type m owns the fixed lane region
// [lane0[m], lane0[m] + cap[m]); inside it the streams follow in ascending order.  A stream whose type's region is
// full keeps its block for the next round (rounds may be deferred: blocks and packets do not depend on when they
// run).  A stream that has fallen behind (a burst of short blocks takes eight rounds per 1024 samples) delivers a
// block in every round of a call until it has caught up.
__global__ __launch_bounds__(1024) void k_fe_plan(vbm_fe_round r, const signed char *__restrict__ type, int S)
{
    __shared__ int s_part[16][4];
    __shared__ int s_carry[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) s_carry[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < S; base += 1024) {
        const int s = base + threadIdx.x;
        const int t = s < S ? type[s] : -1;
        int incl[4], mine[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            mine[m] = (t == m) ? 1 : 0;
            int v = mine[m];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int u = __shfl_up(v, d);
                if (lane >= d) v += u;
            }
            incl[m] = v;
            if (lane == 63) s_part[wave][m] = v;
        }
        __syncthreads();
        if (s < S) {
            int slot = -1;
            if (t >= 0) {
                int before = s_carry[t];
                for (int w = 0; w < wave; w++) before += s_part[w][t];
                const int rank = before + incl[t] - 1;
                if (rank < r.cap[t]) slot = r.lane0[t] + rank;
            }
            r.slot[s] = slot;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            int tot = s_carry[threadIdx.x];
            for (int w = 0; w < 16; w++) tot += s_part[w][threadIdx.x];
            s_carry[threadIdx.x] = tot;
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const int m = threadIdx.x;
        const int n = s_carry[m] < r.cap[m] ? s_carry[m] : r.cap[m];
        r.count[m] = n;
        atomicAdd(&r.stats[m], (unsigned long long)n);
    }
}

// k_fe_commit: the streams that got a lane deliver their block (state changes as in k_fe_decide) and describe it in
// the lane's entries of the round's lists; every other stream is left alone.
__global__ void k_fe_commit(vbm_fe_state f, const vbm_setup *__restrict__ setup, vbm_fe_round r,
                            vbm_fe_decision *__restrict__ dec)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    vbm_fe_decision d;
    d.ready = 0; d.lW = d.W = d.nW = 0; d.block_mode = 0; d.eos = 0; d.beginW = 0; d.movement = 0;
    d.granulepos = 0; d.sequence = 0;
    const int slot = r.slot[s];
    if (slot >= 0) {
        fe_decide_one<true>(f, setup, s, d);
        // (the look of k_fe_classify and this decision see the same state: d.ready holds)
        r.stream_id[slot] = s;
        r.wflags[slot] = (uint8_t)(d.lW | (d.nW << 1));
        r.begin[slot] = d.beginW;
        vbm_packet_info pi;
        pi.stream = s; pi.block_mode = d.block_mode; pi.lW = d.lW; pi.W = d.W; pi.nW = d.nW; pi.eos = d.eos;
        pi.granulepos = d.granulepos; pi.packetno = d.sequence;
        r.info[slot] = pi;
        if (d.movement > 0) atomicAdd(&r.stats[4], (unsigned long long)d.movement);
    }
    dec[s] = d;
}

// lanes of a region beyond its count: no block (stream -1, length -2)
__global__ void k_fe_blank(vbm_fe_round r, int *__restrict__ packet_bytes, int lanes)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= lanes) return;
    int m = 3;
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (k >= r.lane0[t] && k < r.lane0[t] + r.cap[t]) m = t;
    const bool inside = k >= r.lane0[m] && k < r.lane0[m] + r.cap[m];
    if (!inside || k - r.lane0[m] >= r.count[m]) {
        r.info[k].stream = -1;
        packet_bytes[k] = -2;
    }
}

// ---------------------------------------------------------------------------------------------
// blocks of one block type: dst[k][c][N] <- channel buffers of stream ids[k] at begin[k]
__global__ void k_fe_gather(vbm_fe_state f, const int *__restrict__ ids, const int *__restrict__ begin, int count,
                            int N, float *__restrict__ dst, const int *__restrict__ d_count)
{
    const int kc = blockIdx.x;                      // block k, channel c
    const int k = kc / f.ch, c = kc % f.ch;
    if (d_count) count = *d_count;                  // round built on the device: `count` was the launch bound
    if (k >= count) return;
    const int s = ids[k];
    // the shift of this round has not run yet: parity and contents are those the decision saw
    const float4 *src = reinterpret_cast<const float4 *>(chan_buf(f, s, c) + begin[k]);
    float4 *out = reinterpret_cast<float4 *>(dst + ((long)k * f.ch + c) * N);
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < N / 4; i += gridDim.y * blockDim.x) out[i] = src[i];
}

// every stream whose decision of this round moved its window on: the origin moves up by the same amount; once it
// has passed base_max, the surviving samples are copied to the start of the other buffer (k_fe_flip then switches
// the parity and resets the origin)
__global__ void k_fe_shift(vbm_fe_state f, const vbm_fe_decision *__restrict__ dec)
{
    const int c = blockIdx.x;
    const int s = c / f.ch;
    const int mv = dec[s].movement;
    if (mv <= 0) return;
    const int nb = f.base[s] + mv;
    if (nb <= f.base_max) return;
    const int n = f.pcm_current[s];                 // already reduced by the decision
    const int par = f.parity[s];
    const float4 *src = reinterpret_cast<const float4 *>(f.pcm + (long)par * f.plane + (long)c * f.cap + nb);
    float4 *dst = reinterpret_cast<float4 *>(f.pcm + (long)(par ^ 1) * f.plane + (long)c * f.cap);
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < (n + 3) / 4; i += gridDim.y * blockDim.x) dst[i] = src[i];
}

__global__ void k_fe_flip(vbm_fe_state f, const vbm_fe_decision *__restrict__ dec)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= f.S) return;
    const int mv = dec[s].movement;
    if (mv <= 0) return;
    const int nb = f.base[s] + mv;
    if (nb > f.base_max) {
        f.parity[s] ^= 1;
        f.base[s] = 0;
    } else {
        f.base[s] = nb;
    }
}

}  // namespace

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -2)

extern "C" int vbm_fe_launch_append(const vbm_fe_state *f, const float *d_src, int vals, float pre_amplitude, hipStream_t st)
{
    const unsigned gx = (unsigned)((vals + 1023) / 1024);       // a thread moves 16 bytes per step
    hipLaunchKernelGGL(k_fe_append, dim3((unsigned)(f->S * f->ch), gx ? gx : 1), dim3(256), 0, st, *f, d_src, vals, pre_amplitude);
    hipLaunchKernelGGL(k_fe_commit, dim3((unsigned)((f->S + 255) / 256)), dim3(256), 0, st, *f, vals);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_append_ids(const vbm_fe_state *f, const int *d_ids, int n, const float *d_src, int vals,
                                        float pre_amplitude, long stream_stride, long ch_stride, int by_slot, hipStream_t st)
{
    if (n <= 0) return 0;
    const unsigned gy = (unsigned)((vals + 1023) / 1024);
    hipLaunchKernelGGL(k_fe_append_ids, dim3((unsigned)(n * f->ch), gy ? gy : 1), dim3(256), 0, st, *f, d_ids, d_src, vals,
                       pre_amplitude, stream_stride, ch_stride, by_slot);
    hipLaunchKernelGGL(k_fe_commit_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *f, d_ids, n, vals);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_restart(const vbm_fe_state *f, const int *d_ids, int n, int long_n, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_fe_restart, dim3((unsigned)((n * f->ch + 63) / 64)), dim3(64), 0, st, *f, d_ids, n, long_n);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_extrapolate(const vbm_fe_state *f, const int *d_ids, int nids, int mode, int long_n, hipStream_t st)
{
    const int nstreams = mode == 0 ? f->S : nids;
    if (nstreams <= 0) return 0;
    hipLaunchKernelGGL(k_fe_extrapolate, dim3((unsigned)((nstreams * f->ch + 63) / 64)), dim3(64), 0, st, *f, d_ids, nids, mode, long_n);
    hipLaunchKernelGGL(k_fe_extrapolate_commit, dim3((unsigned)((nstreams + 63) / 64)), dim3(64), 0, st, *f, d_ids, nids, mode, long_n);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_ve_range(const vbm_fe_state *f, hipStream_t st)
{
    hipLaunchKernelGGL(k_fe_ve_range, dim3((unsigned)((f->S + 255) / 256)), dim3(256), 0, st, *f);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_ve_filter(const vbm_fe_state *f, const vbm_setup *d_setup, int t0, hipStream_t st)
{
    const size_t lds = (size_t)4 * f->ch * (VBM_VE_AMP * 16 + 16 + 16 + 4) * sizeof(float);   // the rings of four streams
    if (lds > 60 * 1024) return -2;       // (more than 12 channels: not a Vorbis I mapping this encoder sets up)
    hipLaunchKernelGGL(k_fe_ve_filter, dim3((unsigned)((f->S + 3) / 4)), dim3(64), lds, st, *f, d_setup, t0);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_decide(const vbm_fe_state *f, const vbm_setup *d_setup, vbm_fe_decision *d_out,
                                    const uint8_t *d_hold, hipStream_t st)
{
    hipLaunchKernelGGL(k_fe_decide, dim3((unsigned)((f->S + 63) / 64)), dim3(64), 0, st, *f, d_setup, d_out, d_hold);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_gather(const vbm_fe_state *f, const int *d_ids, const int *d_begin, int count, int N,
                                    float *d_dst, const int *d_count, hipStream_t st)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_fe_gather, dim3((unsigned)(count * f->ch), (unsigned)((N / 4 + 255) / 256)), dim3(256), 0, st, *f,
                       d_ids, d_begin, count, N, d_dst, d_count);
    return CHECK_LAUNCH();
}

// a round built on the device: look, assign lanes, commit, blank the unused lanes (frontend.h: vbm_fe_round)
extern "C" int vbm_fe_launch_round_plan(const vbm_fe_state *f, const vbm_setup *d_setup, const vbm_fe_round *r,
                                        signed char *d_type, vbm_fe_decision *d_dec, int *d_packet_bytes, int lanes,
                                        hipStream_t st)
{
    const dim3 g((unsigned)((f->S + 63) / 64));
    hipLaunchKernelGGL(k_fe_classify, g, dim3(64), 0, st, *f, d_setup, d_type);
    hipLaunchKernelGGL(k_fe_plan, dim3(1), dim3(1024), 0, st, *r, d_type, f->S);
    hipLaunchKernelGGL(k_fe_commit, g, dim3(64), 0, st, *f, d_setup, *r, d_dec);
    hipLaunchKernelGGL(k_fe_blank, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, *r, d_packet_bytes, lanes);
    return CHECK_LAUNCH();
}

extern "C" int vbm_fe_launch_shift(const vbm_fe_state *f, const vbm_fe_decision *d_dec, hipStream_t st)
{
    hipLaunchKernelGGL(k_fe_shift, dim3((unsigned)(f->S * f->ch), 4), dim3(256), 0, st, *f, d_dec);
    hipLaunchKernelGGL(k_fe_flip, dim3((unsigned)((f->S + 255) / 256)), dim3(256), 0, st, *f, d_dec);
    return CHECK_LAUNCH();
}
