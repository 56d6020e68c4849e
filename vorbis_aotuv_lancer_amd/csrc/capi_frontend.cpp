// Stream front end object (include/vorbis_mi355x.h, "stream front end"): PCM in, packets out, for
// all streams of one vbm_encoder.  Batched equivalent of the application loop
//     vorbis_analysis_buffer / vorbis_analysis_wrote;  while (vorbis_analysis_blockout(vd, vb) == 1)
//     { vorbis_analysis(vb, NULL); vorbis_bitrate_addblock(vb); vorbis_bitrate_flushpacket(vd, &op) }
// (reference examples/encoder_example.c:190-235, lib/block.c:405-812, lib/envelope.c).
// Kernels: frontend_kernels.hip; the per-block path behind them is vbm_analysis_batch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "vorbis_mi355x.h"
#include "setup_host.h"
#include "frontend.h"
#include "frontend_kernels.h"
extern "C" void vbm_debug_stamp(hipStream_t st, int tag);   // util_kernels.hip (timing experiments)
#include "mdct_kernel.h"
#include "kernels.h"
#include "vbm_internal.h"

vbm_setup_host *vbm_encoder_setup_host(vbm_encoder *e);
vbm_setup_host *vbm_setup_handle_host(vbm_setup_handle *h);
int vbm_encoder_streams(const vbm_encoder *e);
int vbm_encoder_workspaces(const vbm_encoder *e);
int vbm_encoder_reset_streams_dev(vbm_encoder *e, const int *d_ids, int n, hipStream_t q);
int vbm_encoder_device_round_open(vbm_encoder *e, hipStream_t fork, int *w_out, int **d_stream_id, uint8_t **d_wflags, int *lanes,
                                  int **d_counts);
int vbm_encoder_device_round_run(vbm_encoder *e, int w, const int *lane0, const int *cap, const int *d_count,
                                 const float *d_blocks, uint8_t *d_packets, int *d_packet_bytes, bool first_round,
                                 hipStream_t fork);

struct vbm_frontend {
    vbm_encoder *enc;
    vbm_setup_host *H;
    const vbm_setup *hs;
    int S, ch;
    vbm_fe_state f;                       // device pointers
    std::vector<void *> allocs;
    vbm_fe_decision *d_dec = nullptr, *h_dec = nullptr;   // [S] device / pinned host
    int *d_ids = nullptr, *d_begin = nullptr;             // [S] per-round lists, grouped by block type
    int *h_ids = nullptr, *h_begin = nullptr;             // pinned
    uint8_t *h_flags = nullptr;
    uint8_t *d_hold = nullptr, *h_hold = nullptr;         // [S] streams left alone in the later rounds of a multi-round call
    bool hold_active = false;
    float *d_blocks = nullptr;            // [2][S][ch][blocksizes[1]] block-major batches of one round (two rounds in flight)
    int blocks_turn = 0, nblocks_bufs = 2;
    // host mirrors (bounds checking and skipping rounds that cannot produce a block)
    std::vector<int> pcm_current, W, started, ended;
    // The front end runs on a HIP stream of its own: PCM intake, envelope search, decisions, block gather and the
    // buffer shift of write k+1 depend on nothing the per-block path of write k does, so they (and the host's
    // wait for the decisions) overlap it.  A caller's stream is tied in by events: what it produced before a call
    // is visible to the front end, and it waits for what the call promises (PCM consumed, packets complete).
    hipStream_t q = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // rounds built on the device (vbm_frontend_encode_rounds_device)
    int lane0[4] = {0, 0, 0, 0}, lane_cap[4] = {0, 0, 0, 0}, lanes = 0;
    signed char *d_type = nullptr;        // [S]
    int *d_slot = nullptr, *d_begin_lane = nullptr;       // [S], [lanes]
    unsigned long long *d_stats = nullptr;   // [8]
    float *d_blocks_dev = nullptr;        // [nblocks_bufs][lanes][ch][blocksizes[1]]
    std::vector<long long> written;       // samples written per stream (start-of-stream detection without the mirrors)
    bool mirrors_stale = false;           // device-built rounds ran: pcm_current / W / started are out of date
    bool dirty = false;                   // samples arrived since the envelope was last evaluated
    int pending_steps = 0;                // upper bound of search steps not yet evaluated
};

template <typename T>
static int fe_alloc(vbm_frontend *fe, T **p, size_t count)
{
    hipError_t err = hipMalloc((void **)p, count * sizeof(T) + 256);
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMalloc(front end)");
    fe->allocs.push_back(*p);
    err = hipMemset(*p, 0, count * sizeof(T) + 256);
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemset");
    return 0;
}

extern "C" void vbm_frontend_destroy(vbm_frontend *fe)
{
    if (!fe) return;
    if (fe->q) { (void)hipStreamSynchronize(fe->q); (void)hipStreamDestroy(fe->q); }
    if (fe->ev_in) (void)hipEventDestroy(fe->ev_in);
    if (fe->ev_out) (void)hipEventDestroy(fe->ev_out);
    for (void *p : fe->allocs) (void)hipFree(p);
    if (fe->h_dec) (void)hipHostFree(fe->h_dec);
    if (fe->h_ids) (void)hipHostFree(fe->h_ids);
    if (fe->h_begin) (void)hipHostFree(fe->h_begin);
    if (fe->h_flags) (void)hipHostFree(fe->h_flags);
    if (fe->h_hold) (void)hipHostFree(fe->h_hold);
    delete fe;
}

// the front end's stream sees what `stream` has produced so far
static int fe_enter(vbm_frontend *fe, void *stream)
{
    hipError_t err;
    static int skip = -1;
    if (skip < 0) skip = getenv("VBM_DEBUG_NO_ENTER") ? 1 : 0;   // timing experiments only
    if (skip) return VBM_OK;
    if ((err = hipEventRecord(fe->ev_in, (hipStream_t)stream)) != hipSuccess ||
        (err = hipStreamWaitEvent(fe->q, fe->ev_in, 0)) != hipSuccess) return vbm_set_hip_error(err, "front end stream hand-over");
    return VBM_OK;
}
// `stream` waits for what the front end's stream has been given so far
static int fe_leave(vbm_frontend *fe, void *stream)
{
    hipError_t err;
    if ((err = hipEventRecord(fe->ev_out, fe->q)) != hipSuccess ||
        (err = hipStreamWaitEvent((hipStream_t)stream, fe->ev_out, 0)) != hipSuccess)
        return vbm_set_hip_error(err, "front end stream hand-back");
    return VBM_OK;
}

extern "C" int vbm_frontend_create(vbm_frontend **out, vbm_encoder *enc)
{
    if (!out || !enc) return VBM_EINVAL;
    *out = nullptr;
    vbm_frontend *fe = new vbm_frontend();
    fe->enc = enc;
    fe->H = vbm_encoder_setup_host(enc);
    fe->hs = vbm_setup_host_view(fe->H);
    const vbm_setup *s = fe->hs;
    const int S = fe->S = vbm_encoder_streams(enc), ch = fe->ch = s->channels;
    const int bs1 = s->blocksizes[1];
    vbm_fe_state &f = fe->f;
    memset(&f, 0, sizeof(f));
    f.S = S;
    f.ch = ch;
    // room for the carried window (< 2 long blocks), one write of up to 2 long blocks and the three
    // long blocks of end-of-stream padding (lib/block.c:527-529)
    // PCM buffer per channel, in long blocks.  3 are reserved for the end-of-stream padding (lib/block.c:531);
    // what is left is slack: a stream inside a run of short blocks falls behind the others and catches up
    // later, and the more it may fall behind, the fewer rounds a write forces (VBM_FE_BUFFER_BLOCKS, 8..64;
    // measured at 16384 streams, one 1024-sample write per step: 8 -> 11.9 ms, 12 -> 10.1, 16 -> 9.5, 24 -> 9.5).
    int cap_blocks = 24;
    if (const char *env = getenv("VBM_FE_BUFFER_BLOCKS")) cap_blocks = atoi(env);
    if (cap_blocks < 8) cap_blocks = 8;
    if (cap_blocks > 64) cap_blocks = 64;
    f.cap = (long)bs1 * cap_blocks;
    f.base_max = (int)(bs1 * (cap_blocks / 3));   // a third of the buffer: the origin climbs this far before a compaction
    f.plane = (long)S * ch * f.cap;
    f.marks = (int)(f.cap / 64) + 8;
    int rc = 0;
#define A(field, type, count) do { type *p_; rc = fe_alloc<type>(fe, &p_, (count)); if (rc) { vbm_frontend_destroy(fe); return rc; } field = p_; } while (0)
    const size_t SC = (size_t)S * ch;
    A(f.pcm, float, (size_t)2 * f.plane);
    A(f.parity, int, S);
    A(f.base, int, S);
    A(f.overflow, int, 1);
    A(f.pcm_current, int, S); A(f.centerW, int, S); A(f.lW, int, S); A(f.W, int, S); A(f.nW, int, S);
    A(f.eofflag, int, S); A(f.preextrapolate, int, S);
    A(f.granulepos, long long, S); A(f.sequence, long long, S);
    A(f.ve_current, int, S); A(f.ve_cursor, int, S); A(f.ve_curmark, int, S); A(f.ve_stretch, int, S);
    A(f.ve_mark, int, (size_t)f.marks * S);
    A(f.ve_ampbuf, float, (size_t)16 * VBM_VE_AMP * SC);
    A(f.ve_ampptr, int, (size_t)16 * SC);
    A(f.ve_nearDC, float, (size_t)VBM_VE_NEARDC * SC);
    A(f.ve_nearacc, float, (size_t)2 * SC);
    A(f.ve_nearptr, int, SC);
    A(f.ve_first, int, S); A(f.ve_last, int, S);
    A(f.ve_spec, float, SC * VBM_FE_CHUNK * 64);
    A(fe->d_dec, vbm_fe_decision, S);
    A(fe->d_ids, int, S); A(fe->d_begin, int, S);
    fe->nblocks_bufs = vbm_encoder_workspaces(enc);   // a round's blocks are read until its workspace is handed on
    A(fe->d_blocks, float, (size_t)fe->nblocks_bufs * SC * bs1);
    A(fe->d_hold, uint8_t, (size_t)S);
    if (hipHostMalloc((void **)&fe->h_hold, (size_t)S, hipHostMallocDefault) != hipSuccess) {
        vbm_frontend_destroy(fe);
        g_vbm_err = "hipHostMalloc(hold mask) failed";
        return VBM_EHIP;
    }
#undef A
    if (hipHostMalloc((void **)&fe->h_dec, S * sizeof(vbm_fe_decision), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&fe->h_ids, S * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&fe->h_begin, S * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&fe->h_flags, S, hipHostMallocDefault) != hipSuccess) {
        vbm_frontend_destroy(fe);
        g_vbm_err = "hipHostMalloc(front end) failed";
        return VBM_EHIP;
    }
    // The front end's queue: short kernels the rounds of every block type wait for.  VBM_FE_PRIORITY=1: a high-priority
    // stream (its packets go before those of the batch streams when the command processor picks a queue).
    int plo = 0, phi = 0;
    const bool fe_prio = getenv("VBM_FE_PRIORITY") && atoi(getenv("VBM_FE_PRIORITY")) &&
                         hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && phi < plo;
    if ((fe_prio ? hipStreamCreateWithPriority(&fe->q, hipStreamNonBlocking, phi)
                 : hipStreamCreateWithFlags(&fe->q, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&fe->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&fe->ev_out, hipEventDisableTiming) != hipSuccess) {
        vbm_frontend_destroy(fe);
        g_vbm_err = "hipStreamCreate / hipEventCreate (front end) failed";
        return VBM_EHIP;
    }
    fe->pcm_current.assign(S, 0);
    fe->W.assign(S, 0);
    fe->started.assign(S, 0);
    fe->ended.assign(S, 0);
    fe->written.assign(S, 0);
    rc = vbm_frontend_reset(fe);
    if (rc) { vbm_frontend_destroy(fe); return rc; }
    *out = fe;
    return VBM_OK;
}

// vorbis_analysis_init state (lib/block.c:306-344, _ve_envelope_init lib/envelope.c:42-87)
extern "C" int vbm_frontend_reset(vbm_frontend *fe)
{
    if (!fe) return VBM_EINVAL;
    const vbm_fe_state &f = fe->f;
    const int S = fe->S, bs1 = fe->hs->blocksizes[1];
    const size_t SC = (size_t)S * fe->ch;
    hipError_t err = hipDeviceSynchronize();   // the front end's own stream and the encoder's rounds: nothing in flight
    auto zero = [&](void *p, size_t bytes) { if (err == hipSuccess) err = hipMemset(p, 0, bytes); };
    zero(f.pcm, (size_t)2 * f.plane * sizeof(float));
    zero(f.parity, S * sizeof(int));
    zero(f.base, S * sizeof(int));
    zero(f.overflow, sizeof(int));
    zero(f.lW, S * sizeof(int)); zero(f.W, S * sizeof(int)); zero(f.nW, S * sizeof(int));
    zero(f.eofflag, S * sizeof(int)); zero(f.preextrapolate, S * sizeof(int));
    zero(f.granulepos, S * sizeof(long long));
    zero(f.ve_current, S * sizeof(int)); zero(f.ve_stretch, S * sizeof(int));
    zero(f.ve_mark, (size_t)f.marks * S * sizeof(int));
    zero(f.ve_ampbuf, (size_t)16 * VBM_VE_AMP * SC * sizeof(float));
    zero(f.ve_ampptr, (size_t)16 * SC * sizeof(int));
    zero(f.ve_nearDC, (size_t)VBM_VE_NEARDC * SC * sizeof(float));
    zero(f.ve_nearacc, 2 * SC * sizeof(float));
    zero(f.ve_nearptr, SC * sizeof(int));
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemset(front end)");
    std::vector<int> half(S, bs1 / 2), minus1(S, -1);
    std::vector<long long> seq(S, 3);   // the three header packets come first (lib/block.c:337)
    if ((err = hipMemcpy(f.pcm_current, half.data(), S * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (err = hipMemcpy(f.centerW, half.data(), S * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (err = hipMemcpy(f.ve_cursor, half.data(), S * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (err = hipMemcpy(f.ve_curmark, minus1.data(), S * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (err = hipMemcpy(f.sequence, seq.data(), S * sizeof(long long), hipMemcpyHostToDevice)) != hipSuccess)
        return vbm_set_hip_error(err, "hipMemcpy(front end init)");
    fe->pcm_current.assign(S, bs1 / 2);
    fe->W.assign(S, 0);
    fe->started.assign(S, 0);
    fe->ended.assign(S, 0);
    fe->written.assign(S, 0);
    fe->mirrors_stale = false;
    fe->dirty = false;
    fe->pending_steps = 0;
    return VBM_OK;
}

static int refresh_mirrors(vbm_frontend *fe);

// most samples any stream holds (host mirror of pcm_current): lets the caller bound the rounds it runs
// per write and still drain before the buffers fill
extern "C" int vbm_frontend_max_buffered(const vbm_frontend *fe)
{
    if (!fe) return VBM_EINVAL;
    if (fe->mirrors_stale && refresh_mirrors(const_cast<vbm_frontend *>(fe))) return VBM_EHIP;
    int m = 0;
    for (int i = 0; i < fe->S; i++)
        if (fe->pcm_current[i] > m) m = fe->pcm_current[i];
    return m;
}

extern "C" int vbm_frontend_capacity(const vbm_frontend *fe)
{
    return fe ? (int)(fe->f.cap - 3 * fe->hs->blocksizes[1] - fe->f.base_max) : VBM_EINVAL;
}

extern "C" int vbm_frontend_write(vbm_frontend *fe, const float *d_pcm, int vals, void *stream)
{
    if (!fe || !d_pcm || vals <= 0) return VBM_EINVAL;
    const vbm_setup *s = fe->hs;
    const int bs1 = s->blocksizes[1];
    for (int i = 0; i < fe->S; i++) {
        if (fe->ended[i]) { g_vbm_err = "vbm_frontend_write after vbm_frontend_finish"; return VBM_EINVAL; }
        if (!fe->mirrors_stale && fe->pcm_current[i] + vals > fe->f.cap - 3 * bs1 - fe->f.base_max) {   // OV_EINVAL of lib/block.c:540-541
            g_vbm_err = "PCM buffer full: drain blocks with vbm_frontend_encode_round before writing more";
            return VBM_EINVAL;
        }
    }
    hipStream_t st = fe->q;
    {
        int rc = fe_enter(fe, stream);
        if (rc) return rc;
    }
    vbm_debug_stamp(st, 0);
    vbm_debug_delay_point(VBM_DP_FE_WRITE, st);
    if (vbm_fe_launch_append(&fe->f, d_pcm, vals, s->pre_amplitude, st)) return VBM_EHIP;
    {   // the caller's buffer is free again once the append has run
        int rc = fe_leave(fe, stream);
        if (rc) return rc;
    }
    bool cross = false;
    for (int i = 0; i < fe->S; i++) {
        fe->pcm_current[i] += vals;
        fe->written[i] += vals;
        // vorbis_analysis_wrote: first time more than one long block follows centerW (lib/block.c:547-550); before a
        // stream starts nothing has left its buffer, so the samples written so far decide
        if (!fe->started[i] && fe->written[i] > bs1) { fe->started[i] = 1; cross = true; }
    }
    if (cross && vbm_fe_launch_extrapolate(&fe->f, nullptr, 0, 0, bs1, st)) return VBM_EHIP;
    fe->dirty = true;
    fe->pending_steps += vals / 64 + 1;
    if (cross) fe->pending_steps += (bs1 / 2 + bs1) / 64;   // the first search starts at step 0
    return VBM_OK;
}

static int upload_ids(vbm_frontend *fe, const int *stream_ids, int n, hipStream_t st)
{
    for (int i = 0; i < n; i++)
        if (stream_ids[i] < 0 || stream_ids[i] >= fe->S) return VBM_EINVAL;
    (void)hipStreamSynchronize(st);   // h_ids is reused by the rounds
    memcpy(fe->h_ids, stream_ids, n * sizeof(int));
    hipError_t err = hipMemcpyAsync(fe->d_ids, fe->h_ids, n * sizeof(int), hipMemcpyHostToDevice, st);
    return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipMemcpyAsync(stream ids)");
}

// vorbis_analysis_buffer + vorbis_analysis_wrote for a subset of the streams: d_pcm [n][ch][vals]
extern "C" int vbm_frontend_write_streams(vbm_frontend *fe, const int *stream_ids, int n, const float *d_pcm, int vals,
                                          void *stream)
{
    if (!fe) return VBM_EINVAL;
    return vbm_frontend_write_streams_strided(fe, stream_ids, n, d_pcm, vals, (long)fe->ch * vals, vals, 0, stream);
}

// the same with the source laid out by the caller: channel c of stream_ids[k] at pcm + (by_slot ? stream_ids[k] : k) *
// stream_stride + c * ch_stride floats.  pcm: device memory, or host memory the device can read (hipHostMalloc): the
// append kernel then fetches it over the bus itself — no staging copy, no separate upload.  The call returns when the
// samples have been taken (the source may be rewritten).
extern "C" int vbm_frontend_write_streams_strided(vbm_frontend *fe, const int *stream_ids, int n, const float *d_pcm, int vals,
                                                  long stream_stride, long ch_stride, int by_slot, void *stream)
{
    if (!fe || n < 0 || (n && (!stream_ids || !d_pcm)) || vals <= 0 || ch_stride < vals || stream_stride < 0) return VBM_EINVAL;
    if (n == 0) return VBM_OK;
    const vbm_setup *s = fe->hs;
    const int bs1 = s->blocksizes[1];
    std::vector<char> seen(fe->S, 0);
    for (int k = 0; k < n; k++) {
        const int i = stream_ids[k];
        if (i < 0 || i >= fe->S || seen[i]) return VBM_EINVAL;   // a stream once per call
        seen[i] = 1;
        if (fe->ended[i]) { g_vbm_err = "vbm_frontend_write_streams after vbm_frontend_finish"; return VBM_EINVAL; }
        if (!fe->mirrors_stale && fe->pcm_current[i] + vals > fe->f.cap - 3 * bs1 - fe->f.base_max) {
            g_vbm_err = "PCM buffer full: drain blocks with vbm_frontend_encode_round before writing more";
            return VBM_EINVAL;
        }
    }
    hipStream_t st = fe->q;
    int rc = fe_enter(fe, stream);
    if (rc) return rc;
    rc = upload_ids(fe, stream_ids, n, st);
    if (rc) return rc;
    if (vbm_fe_launch_append_ids(&fe->f, fe->d_ids, n, d_pcm, vals, s->pre_amplitude, stream_stride, ch_stride, by_slot, st)) return VBM_EHIP;
    bool cross = false;
    for (int k = 0; k < n; k++) {
        const int i = stream_ids[k];
        fe->pcm_current[i] += vals;
        fe->written[i] += vals;
        if (!fe->started[i] && fe->written[i] > bs1) { fe->started[i] = 1; cross = true; }
    }
    if (cross && vbm_fe_launch_extrapolate(&fe->f, nullptr, 0, 0, bs1, st)) return VBM_EHIP;
    (void)hipStreamSynchronize(st);   // d_ids / h_ids are free again
    fe->dirty = true;
    fe->pending_steps += vals / 64 + 1;
    if (cross) fe->pending_steps += (bs1 / 2 + bs1) / 64;
    return VBM_OK;
}

// a new stream starts in each of the listed slots (vorbis_analysis_init state for front end and encoder)
extern "C" int vbm_frontend_restart_streams(vbm_frontend *fe, const int *stream_ids, int n, void *stream)
{
    if (!fe || n < 0 || (n && !stream_ids)) return VBM_EINVAL;
    if (n == 0) return VBM_OK;
    hipStream_t st = fe->q;
    const int bs1 = fe->hs->blocksizes[1];
    int rc = fe_enter(fe, stream);
    if (rc) return rc;
    rc = upload_ids(fe, stream_ids, n, st);
    if (rc) return rc;
    if (vbm_fe_launch_restart(&fe->f, fe->d_ids, n, bs1, st)) return VBM_EHIP;
    rc = vbm_encoder_reset_streams_dev(fe->enc, fe->d_ids, n, st);
    if (rc) return rc;
    (void)hipStreamSynchronize(st);
    for (int k = 0; k < n; k++) {
        const int i = stream_ids[k];
        fe->pcm_current[i] = bs1 / 2;
        fe->W[i] = 0;
        fe->started[i] = 0;
        fe->ended[i] = 0;
        fe->written[i] = 0;
    }
    return VBM_OK;
}

extern "C" int vbm_frontend_finish(vbm_frontend *fe, const int *stream_ids, int n, void *stream)
{
    if (!fe || n < 0 || (n && !stream_ids)) return VBM_EINVAL;
    if (n == 0) return VBM_OK;
    const int bs1 = fe->hs->blocksizes[1];
    for (int i = 0; i < n; i++) {
        int s = stream_ids[i];
        if (s < 0 || s >= fe->S || fe->ended[s]) return VBM_EINVAL;
    }
    hipStream_t st = fe->q;
    {
        int rc = fe_enter(fe, stream);
        if (rc) return rc;
    }
    (void)hipStreamSynchronize(st);   // h_ids is reused by the rounds
    memcpy(fe->h_ids, stream_ids, n * sizeof(int));
    hipError_t err = hipMemcpyAsync(fe->d_ids, fe->h_ids, n * sizeof(int), hipMemcpyHostToDevice, st);
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemcpyAsync(finish ids)");
    if (vbm_fe_launch_extrapolate(&fe->f, fe->d_ids, n, 1, bs1, st)) return VBM_EHIP;
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < n; i++) {
        int s = stream_ids[i];
        fe->ended[s] = 1;
        fe->started[s] = 1;
        fe->pcm_current[s] += 3 * bs1;
    }
    fe->dirty = true;
    fe->pending_steps += 3 * bs1 / 64 + (bs1 / 2 + bs1) / 64 + 1;
    return VBM_OK;
}

static vbm_encoder *vbm_frontend_encoder(vbm_frontend *fe) { return fe->enc; }

// Rounds built on the device leave the host mirrors behind: fetch what the host-built rounds need (buffer fill,
// current window size, whether a stream can deliver blocks) once the front end's stream has drained.
static int refresh_mirrors(vbm_frontend *fe)
{
    if (!fe->mirrors_stale) return VBM_OK;
    const int S = fe->S;
    std::vector<int> pre(S), eof(S);
    hipError_t err;
    if ((err = hipStreamSynchronize(fe->q)) != hipSuccess ||
        (err = hipMemcpy(fe->pcm_current.data(), fe->f.pcm_current, S * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess ||
        (err = hipMemcpy(fe->W.data(), fe->f.W, S * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess ||
        (err = hipMemcpy(pre.data(), fe->f.preextrapolate, S * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess ||
        (err = hipMemcpy(eof.data(), fe->f.eofflag, S * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess)
        return vbm_set_hip_error(err, "front end mirrors");
    for (int i = 0; i < S; i++) fe->started[i] = (pre[i] && eof[i] != -1) ? 1 : 0;
    fe->mirrors_stale = false;
    return VBM_OK;
}

static int round_impl(vbm_frontend *fe, uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info, int *nblocks,
                      void *stream, bool defer)
{
    if (!fe || !nblocks || !info) return VBM_EINVAL;
    *nblocks = 0;
    {
        int rcm = refresh_mirrors(fe);
        if (rcm) return rcm;
    }
    const vbm_setup *s = fe->hs;
    const vbm_setup *ds = vbm_setup_device(fe->H);
    const int S = fe->S, ch = fe->ch;
    const int bs0 = s->blocksizes[0], bs1 = s->blocksizes[1];
    hipStream_t st = fe->q;
    void *const fq = (void *)fe->q;
    hipError_t err;
    (void)stream;   // the caller's stream is tied in by the entry points around this (fe_enter / joins)

    // can any stream have a block?  (host mirrors: the smallest possible block bound is a short
    // next window, lib/block.c:593-600)
    bool maybe = false;
    for (int i = 0; i < S && !maybe; i++) {
        if (!fe->started[i]) continue;
        const int bsW = fe->W[i] ? bs1 : bs0;
        if (fe->pcm_current[i] >= bs1 / 2 + bsW / 4 + bs0 / 4 + bs0 / 2) maybe = true;
    }
    if (!maybe) return VBM_OK;

    // evaluate the envelope over everything written since the last round (_ve_envelope_search, first part)
    if (fe->dirty) {
        if (vbm_fe_launch_ve_range(&fe->f, st)) return VBM_EHIP;
        vbm_ve_gather g;
        g.pcm = fe->f.pcm;
        g.first = fe->f.ve_first; g.last = fe->f.ve_last; g.parity = fe->f.parity; g.base = fe->f.base;
        g.ch = ch; g.steps = VBM_FE_CHUNK; g.cap = fe->f.cap; g.plane = fe->f.plane;
        for (int t0 = 0; t0 < fe->pending_steps; t0 += VBM_FE_CHUNK) {
            g.t0 = t0;
            if (vbm_launch_ve_mdct(&g, fe->f.ve_spec, vbm_setup_device_ptrs(fe->H)->ve.mdct_trig,
                                   vbm_setup_device_ptrs(fe->H)->ve.mdct_win, (long)S * ch * VBM_FE_CHUNK, st))
                return VBM_EHIP;
            if (vbm_fe_launch_ve_filter(&fe->f, ds, t0, st)) return VBM_EHIP;
        }
        fe->dirty = false;
        fe->pending_steps = 0;
    }

    if (vbm_fe_launch_decide(&fe->f, ds, fe->d_dec, fe->hold_active ? fe->d_hold : nullptr, st)) return VBM_EHIP;
    if ((err = hipMemcpyAsync(fe->h_dec, fe->d_dec, S * sizeof(vbm_fe_decision), hipMemcpyDeviceToHost, st)) != hipSuccess)
        return vbm_set_hip_error(err, "hipMemcpyAsync(decisions)");
    if ((err = hipStreamSynchronize(st)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamSynchronize");

    // group the ready blocks by block type
    int count[4] = {0, 0, 0, 0}, offset[4];
    for (int i = 0; i < S; i++)
        if (fe->h_dec[i].ready) count[fe->h_dec[i].block_mode & 3]++;
    offset[0] = 0;
    for (int m = 1; m < 4; m++) offset[m] = offset[m - 1] + count[m - 1];
    const int total = offset[3] + count[3];
    if (total == 0) return VBM_OK;
    {
        int fill[4] = {offset[0], offset[1], offset[2], offset[3]};
        for (int i = 0; i < S; i++) {
            const vbm_fe_decision &d = fe->h_dec[i];
            if (!d.ready) continue;
            const int k = fill[d.block_mode & 3]++;
            fe->h_ids[k] = i;
            fe->h_begin[k] = d.beginW;
            fe->h_flags[k] = (uint8_t)(d.lW | (d.nW << 1));
            vbm_packet_info &pi = info[k];
            pi.stream = i; pi.block_mode = d.block_mode; pi.lW = d.lW; pi.W = d.W; pi.nW = d.nW; pi.eos = d.eos;
            pi.granulepos = d.granulepos; pi.packetno = d.sequence;
            // host mirrors
            if (d.movement > 0) {
                fe->pcm_current[i] -= d.movement;
                fe->W[i] = d.nW;
            }
            if (d.eos) fe->started[i] = 0;   // stream over: no further blocks
        }
    }
    if ((err = hipMemcpyAsync(fe->d_ids, fe->h_ids, total * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (err = hipMemcpyAsync(fe->d_begin, fe->h_begin, total * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess)
        return vbm_set_hip_error(err, "hipMemcpyAsync(round lists)");

    // the block buffer of this turn was read by the round before the previous one
    {
        int rc = vbm_analysis_round_wait_workspace(fe->enc, fq);
        if (rc) return rc;
    }
    float *round_blocks = fe->d_blocks + (size_t)fe->blocks_turn * S * ch * bs1;
    fe->blocks_turn = (fe->blocks_turn + 1) % fe->nblocks_bufs;
    for (int m = 0; m < 4; m++) {
        if (!count[m]) continue;
        const int N = (m >> 1) ? bs1 : bs0;
        float *blocks = round_blocks + (size_t)offset[m] * ch * bs1;
        if (vbm_fe_launch_gather(&fe->f, fe->d_ids + offset[m], fe->d_begin + offset[m], count[m], N, blocks, nullptr, st))
            return VBM_EHIP;
    }
    vbm_debug_delay_point(VBM_DP_FE_FORK, st);
    {
        // the batches are forked from the front end's stream (behind the gather); the caller's stream joins them
        int rc = vbm_analysis_round_begin(fe->enc, count, fe->h_ids, fe->h_flags, round_blocks, d_packets, d_packet_bytes, fq);
        if (rc) return rc;
        if (!defer && (rc = vbm_analysis_round_join(fe->enc, stream))) return rc;
    }
    vbm_debug_delay_point(VBM_DP_FE_SHIFT, st);
    if (vbm_fe_launch_shift(&fe->f, fe->d_dec, st)) return VBM_EHIP;
    *nblocks = total;
    return VBM_OK;
}

extern "C" int vbm_frontend_encode_round(vbm_frontend *fe, uint8_t *d_packets, int *d_packet_bytes,
                                         vbm_packet_info *info, int *nblocks, void *stream)
{
    if (!fe) return VBM_EINVAL;
    int rc = fe_enter(fe, stream);
    if (rc) return rc;
    return round_impl(fe, d_packets, d_packet_bytes, info, nblocks, stream, false);
}

// One round for the listed streams only: every other stream is left alone (no block, no state change), as if its
// application had not called vorbis_analysis_blockout yet.
extern "C" int vbm_frontend_encode_round_streams(vbm_frontend *fe, const int *stream_ids, int n, uint8_t *d_packets,
                                                 int *d_packet_bytes, vbm_packet_info *info, int *nblocks, void *stream)
{
    if (!fe || n < 0 || (n && !stream_ids) || !nblocks) return VBM_EINVAL;
    *nblocks = 0;
    if (n == 0) return VBM_OK;
    {
        int rc0 = fe_enter(fe, stream);
        if (rc0) return rc0;
    }
    memset(fe->h_hold, 1, fe->S);
    for (int k = 0; k < n; k++) {
        if (stream_ids[k] < 0 || stream_ids[k] >= fe->S) return VBM_EINVAL;
        fe->h_hold[stream_ids[k]] = 0;
    }
    hipError_t err = hipMemcpyAsync(fe->d_hold, fe->h_hold, fe->S, hipMemcpyHostToDevice, fe->q);
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemcpyAsync(hold mask)");
    fe->hold_active = true;
    const int rc = round_impl(fe, d_packets, d_packet_bytes, info, nblocks, stream, false);
    fe->hold_active = false;
    return rc;
}

// Several rounds in one call, joined at the end: a round only waits for the batches of the round before it that
// its own streams were in, so the short blocks of the streams that need many rounds run beside the long-block
// batch of the previous round.  Rounds stop when none produced a block, after max_rounds, when fewer than
// nstreams output slots are left, or — past min_rounds — once every stream could take `headroom` more samples
// without its buffer passing half full.  Outputs are compact over the rounds, in round order.
static int encode_rounds_impl(vbm_frontend *fe, int min_rounds, int max_rounds, int headroom,
                              uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info,
                              int cap_blocks, int *round_blocks, int *nrounds, void *stream, bool lazy)
{
    if (!fe || !info || !nrounds || max_rounds < 1 || min_rounds < 0 || cap_blocks < 0) return VBM_EINVAL;
    *nrounds = 0;
    const int maxb = vbm_encoder_max_packet_bytes(fe->enc);
    int done = 0, rc = fe_enter(fe, stream);
    if (rc) return rc;
    for (int r = 0; r < max_rounds; r++) {
        if (cap_blocks - done < fe->S) break;
        if (r >= min_rounds && vbm_frontend_max_buffered(fe) + headroom <= vbm_frontend_capacity(fe) / 2) break;
        int n = 0;
        rc = round_impl(fe, d_packets ? d_packets + (size_t)done * maxb : nullptr, d_packet_bytes ? d_packet_bytes + done : nullptr,
                        info + done, &n, stream, true);
        if (rc || n == 0) break;
        if (round_blocks) round_blocks[r] = n;
        if (r == 0 && max_rounds > 1) {
            // The streams of a big batch are left alone for the rest of this call: a block of theirs in a later
            // round would make that round's batch wait for the big one (a stream's blocks stay in order), and
            // with it every stream that shares the batch.  They get their next block in the next call.
            int cnt[4] = {0, 0, 0, 0};
            for (int k = 0; k < n; k++) cnt[info[done + k].block_mode & 3]++;
            int big = 0;
            for (int m = 1; m < 4; m++)
                if (cnt[m] > cnt[big]) big = m;
            if (cnt[big] >= 1024) {
                memset(fe->h_hold, 0, fe->S);
                // ... unless they are behind: a stream whose buffer is filling up keeps its place in every round
                const int half = vbm_frontend_capacity(fe) / 2;
                for (int k = 0; k < n; k++) {
                    const int sid = info[done + k].stream;
                    if ((info[done + k].block_mode & 3) == big && fe->pcm_current[sid] + headroom <= half) fe->h_hold[sid] = 1;
                }
                if (hipMemcpyAsync(fe->d_hold, fe->h_hold, fe->S, hipMemcpyHostToDevice, fe->q) != hipSuccess) {
                    rc = VBM_EHIP;
                    break;
                }
                fe->hold_active = true;
            }
        }
        done += n;
        *nrounds = r + 1;
    }
    fe->hold_active = false;
    int rj = lazy ? vbm_analysis_round_join_lazy(vbm_frontend_encoder(fe), stream)
                  : vbm_analysis_round_join(vbm_frontend_encoder(fe), stream);
    return rc ? rc : rj;
}

extern "C" int vbm_frontend_encode_rounds(vbm_frontend *fe, int min_rounds, int max_rounds, int headroom,
                                          uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info,
                                          int cap_blocks, int *round_blocks, int *nrounds, void *stream)
{
    return encode_rounds_impl(fe, min_rounds, max_rounds, headroom, d_packets, d_packet_bytes, info, cap_blocks, round_blocks,
                              nrounds, stream, false);
}

extern "C" int vbm_frontend_encode_rounds_lazy(vbm_frontend *fe, int min_rounds, int max_rounds, int headroom,
                                               uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info,
                                               int cap_blocks, int *round_blocks, int *nrounds, void *stream)
{
    return encode_rounds_impl(fe, min_rounds, max_rounds, headroom, d_packets, d_packet_bytes, info, cap_blocks, round_blocks,
                              nrounds, stream, true);
}

extern "C" int vbm_packets_compact(const uint8_t *d_packets, const int *d_packet_bytes, int n, int max_packet_bytes,
                                   uint8_t *d_out, long long *d_offsets, void *stream)
{
    if (n < 0 || max_packet_bytes <= 0 || (max_packet_bytes & 3)) return VBM_EINVAL;
    if (n == 0) return VBM_OK;
    if (!d_packets || !d_packet_bytes || !d_out || !d_offsets) return VBM_EINVAL;
    if (((uintptr_t)d_packets & 3) || ((uintptr_t)d_out & 3)) return VBM_EINVAL;
    return vbm_launch_compact(d_packets, d_packet_bytes, n, max_packet_bytes, d_offsets, d_out, (hipStream_t)stream)
               ? VBM_EHIP : VBM_OK;
}

// ---- rounds built on the device -----------------------------------------------------------------------------
// Lane regions of a round, in the order of the block types: the three small types (impulse, padding, transition)
// get a quarter of the stream count each (at least 256 lanes), the long blocks (type 3) one lane per stream, last.
// (The order matters: the block-major arrays of a workspace are addressed lane * n with the type's own n, so the
// regions of the short types have to lie BELOW those of the long ones, as in the host-built rounds.)
// Setups with one block size only deliver types 0 and 1: a full region each.
static void lane_layout(const vbm_setup *s, int S, int *lane0, int *cap, int *lanes)
{
    const int S64 = (S + 63) & ~63;
    if (s->modes < 2) {
        lane0[0] = 0; cap[0] = S; lane0[1] = S64; cap[1] = S;
        lane0[2] = lane0[3] = 2 * S64; cap[2] = cap[3] = 0;
        *lanes = 2 * S64;
        return;
    }
    int q = ((S / 4) + 63) & ~63;
    if (q < 256) q = 256;
    if (q > S64) q = S64;
    const int c = q < S ? q : S;
    lane0[0] = 0; cap[0] = c;
    lane0[1] = q; cap[1] = c;
    lane0[2] = 2 * q; cap[2] = c;
    lane0[3] = 3 * q; cap[3] = S;
    *lanes = 3 * q + S64;
}

extern "C" int vbm_device_round_lanes(const vbm_setup_handle *setup, int nstreams)
{
    if (!setup || nstreams <= 0) return VBM_EINVAL;
    int lane0[4], cap[4], lanes;
    lane_layout(vbm_setup_host_view(vbm_setup_handle_host(const_cast<vbm_setup_handle *>(setup))), nstreams, lane0, cap, &lanes);
    return lanes;
}

extern "C" int vbm_frontend_encode_rounds_device(vbm_frontend *fe, int nrounds, uint8_t *d_packets, int *d_packet_bytes,
                                                 vbm_packet_info *d_info, int *d_counts, int lazy, void *stream)
{
    if (!fe || nrounds < 1 || !d_packet_bytes || !d_info || !d_counts) return VBM_EINVAL;
    const vbm_setup *s = fe->hs;
    const vbm_setup *ds = vbm_setup_device(fe->H);
    const int S = fe->S, ch = fe->ch, bs0 = s->blocksizes[0], bs1 = s->blocksizes[1];
    const int maxb = vbm_encoder_max_packet_bytes(fe->enc);
    hipStream_t q = fe->q;
    hipError_t err;
    int rc = fe_enter(fe, stream);
    if (rc) return rc;
    vbm_debug_stamp(q, 1);
    if (!fe->lanes) {
        lane_layout(s, S, fe->lane0, fe->lane_cap, &fe->lanes);
        const size_t L = (size_t)fe->lanes;
#define A(field, type, count) do { type *p_; rc = fe_alloc<type>(fe, &p_, (count)); if (rc) return rc; field = p_; } while (0)
        A(fe->d_type, signed char, (size_t)S);
        A(fe->d_slot, int, (size_t)S);
        A(fe->d_begin_lane, int, L);
        A(fe->d_stats, unsigned long long, 8);
        A(fe->d_blocks_dev, float, (size_t)fe->nblocks_bufs * L * ch * bs1);
#undef A
        // the zero fills went to the null stream; the front end's own stream is non-blocking and would run on beside
        // them (the first round's gather writing blocks that the fill then wipes): wait here, once
        if ((err = hipDeviceSynchronize()) != hipSuccess) return vbm_set_hip_error(err, "hipDeviceSynchronize");
    }
    // evaluate the envelope over everything written since the last round (_ve_envelope_search, first part)
    if (fe->dirty) {
        if (vbm_fe_launch_ve_range(&fe->f, q)) return VBM_EHIP;
        vbm_ve_gather g;
        g.pcm = fe->f.pcm;
        g.first = fe->f.ve_first; g.last = fe->f.ve_last; g.parity = fe->f.parity; g.base = fe->f.base;
        g.ch = ch; g.steps = VBM_FE_CHUNK; g.cap = fe->f.cap; g.plane = fe->f.plane;
        for (int t0 = 0; t0 < fe->pending_steps; t0 += VBM_FE_CHUNK) {
            g.t0 = t0;
            if (vbm_launch_ve_mdct(&g, fe->f.ve_spec, vbm_setup_device_ptrs(fe->H)->ve.mdct_trig,
                                   vbm_setup_device_ptrs(fe->H)->ve.mdct_win, (long)S * ch * VBM_FE_CHUNK, q))
                return VBM_EHIP;
            if (vbm_fe_launch_ve_filter(&fe->f, ds, t0, q)) return VBM_EHIP;
        }
        fe->dirty = false;
        fe->pending_steps = 0;
    }
    for (int r = 0; r < nrounds; r++) {
        int w, ws_lanes;
        int *d_sid, *counts_ws;
        uint8_t *d_wf;
        rc = vbm_encoder_device_round_open(fe->enc, q, &w, &d_sid, &d_wf, &ws_lanes, &counts_ws);
        if (rc) return rc;
        if (ws_lanes < fe->lanes) {
            g_vbm_err = "encoder workspace too small for rounds built on the device: create it with max_batch >= vbm_device_round_lanes()";
            return VBM_EINVAL;
        }
        float *blocks = fe->d_blocks_dev + (size_t)w * fe->lanes * ch * bs1;
        int *counts_r = d_counts + 4 * r;
        int *bytes_r = d_packet_bytes + (size_t)r * fe->lanes;
        // Lanes per block type in this round.  The long blocks of a call's LATER rounds are those of streams catching
        // up: their region is cut to half the streams (every kernel of a batch is launched for the region's size — a
        // full-size region costs ~40 launches of 16384 mostly empty workgroups; a stream that finds the region full
        // keeps its block for the next round like any other).  A quarter is too little to keep up at the burst rate of
        // the bench signal (a burst leaves a stream ~5 long blocks behind: 1437 extra long blocks per write of 16384
        // streams against 0.33 later rounds per write).
        int caps[4];
        for (int m = 0; m < 4; m++) caps[m] = fe->lane_cap[m];
        if (r > 0 && caps[2] > 0 && caps[3] > 2 * caps[2]) caps[3] = 2 * caps[2];
        vbm_fe_round R;
        for (int m = 0; m < 4; m++) { R.lane0[m] = fe->lane0[m]; R.cap[m] = caps[m]; }
        R.first_round = r == 0;
        R.count = counts_ws;        // the workspace's own counts: what the round's kernels (and graphs) read
        R.slot = fe->d_slot;
        R.stream_id = d_sid;
        R.wflags = d_wf;
        R.begin = fe->d_begin_lane;
        R.info = d_info + (size_t)r * fe->lanes;
        R.stats = fe->d_stats;
        vbm_debug_delay_point(VBM_DP_DEV_PLAN, q);
        if (vbm_fe_launch_round_plan(&fe->f, ds, &R, fe->d_type, fe->d_dec, bytes_r, fe->lanes, q)) return VBM_EHIP;
        if ((err = hipMemcpyAsync(counts_r, counts_ws, 4 * sizeof(int), hipMemcpyDeviceToDevice, q)) != hipSuccess)
            return vbm_set_hip_error(err, "hipMemcpyAsync(counts)");
        for (int m = 0; m < 4; m++) {
            if (!caps[m]) continue;
            const int N = (m >> 1) ? bs1 : bs0;
            if (vbm_fe_launch_gather(&fe->f, d_sid + fe->lane0[m], fe->d_begin_lane + fe->lane0[m], caps[m], N,
                                     blocks + (size_t)fe->lane0[m] * ch * bs1, counts_ws + m, q))
                return VBM_EHIP;
        }
        vbm_debug_stamp(q, 2 + (r ? 1 : 0));
        rc = vbm_encoder_device_round_run(fe->enc, w, fe->lane0, caps, counts_ws, blocks,
                                          d_packets ? d_packets + (size_t)r * fe->lanes * maxb : nullptr, bytes_r, r == 0, q);
        if (rc) return rc;
        vbm_debug_delay_point(VBM_DP_FE_SHIFT, q);
        if (vbm_fe_launch_shift(&fe->f, fe->d_dec, q)) return VBM_EHIP;
    }
    vbm_debug_stamp(q, 4);
    fe->mirrors_stale = true;
    if (lazy == 2) return VBM_OK;          // the consumer joins (vbm_frontend_join on its own stream)
    return lazy ? vbm_analysis_round_join_lazy(fe->enc, stream) : vbm_analysis_round_join(fe->enc, stream);
}

// which block type every stream had ready when the newest device-built round was planned (-1: none) — a stream with a
// type here and no entry in the round's records found its type's lane region full and keeps the block for the next round
extern "C" int vbm_frontend_round_types(vbm_frontend *fe, signed char *out, void *stream)
{
    if (!fe || !out || !fe->d_type) return VBM_EINVAL;
    hipError_t err = hipMemcpyAsync(out, fe->d_type, (size_t)fe->S, hipMemcpyDeviceToHost, (hipStream_t)stream);
    return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipMemcpyAsync(round types)");
}

// running totals of the rounds built on the device: blocks of type 0..3 and the samples all streams advanced by
extern "C" int vbm_frontend_device_stats(vbm_frontend *fe, unsigned long long *out)
{
    if (!fe || !out) return VBM_EINVAL;
    for (int i = 0; i < 6; i++) out[i] = 0;
    hipError_t err;
    int ov = 0;
    if ((err = hipStreamSynchronize(fe->q)) != hipSuccess ||
        (err = hipMemcpy(&ov, fe->f.overflow, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess)
        return vbm_set_hip_error(err, "hipMemcpy(stats)");
    out[5] = (unsigned long long)ov;
    if (!fe->d_stats) return VBM_OK;
    if ((err = hipMemcpy(out, fe->d_stats, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost)) != hipSuccess)
        return vbm_set_hip_error(err, "hipMemcpy(stats)");
    return VBM_OK;
}

// Test instrumentation: the front end's pure scratch (block buffers of the rounds, search spectra, round lists) is
// filled with `byte` while the device is idle; nothing a later call reads may come from there.
extern "C" int vbm_debug_poison_frontend(vbm_frontend *fe, int byte)
{
    if (!fe) return VBM_EINVAL;
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipDeviceSynchronize");
    const size_t SC = (size_t)fe->S * fe->ch, bs1 = (size_t)fe->hs->blocksizes[1];
    if ((err = hipMemset(fe->d_blocks, byte, (size_t)fe->nblocks_bufs * SC * bs1 * sizeof(float))) != hipSuccess ||
        (err = hipMemset(fe->f.ve_spec, byte, SC * VBM_FE_CHUNK * 64 * sizeof(float))) != hipSuccess ||
        (err = hipMemset(fe->d_ids, byte, (size_t)fe->S * sizeof(int))) != hipSuccess ||
        (err = hipMemset(fe->d_begin, byte, (size_t)fe->S * sizeof(int))) != hipSuccess ||
        (fe->d_blocks_dev && (err = hipMemset(fe->d_blocks_dev, byte, (size_t)fe->nblocks_bufs * fe->lanes * fe->ch * bs1 * sizeof(float))) != hipSuccess) ||
        (fe->d_begin_lane && (err = hipMemset(fe->d_begin_lane, byte, (size_t)fe->lanes * sizeof(int))) != hipSuccess))
        return vbm_set_hip_error(err, "hipMemset(poison)");
    err = hipDeviceSynchronize();
    return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipDeviceSynchronize");
}

extern "C" int vbm_frontend_join(vbm_frontend *fe, void *stream)
{
    if (!fe) return VBM_EINVAL;
    return vbm_analysis_round_join(fe->enc, stream);
}
