// Batched encoder object: per-stream carried state + batch workspace on the device, and the
// per-block pipeline (include/vorbis_mi355x.h, "batched analysis").
//
// One call to vbm_analysis_batch() is the batched equivalent of
//     vorbis_analysis(vb, NULL); vorbis_bitrate_addblock(vb); vorbis_bitrate_flushpacket(vd, &op)
// (reference lib/analysis.c:29-63, lib/bitrate.c:88-96, :229-252, VBR) for `nsb` blocks of the same
// block type, one per stream.  Kernel order = mapping0_forward (lib/mapping0.c:738-1322).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include "vorbis_mi355x.h"
#include "setup_host.h"
#include "batch.h"
#include "kernels.h"
#include "mdct_kernel.h"
#include "vbm_internal.h"

extern "C" int vbm_launch_spread_flags(const vbm_batch *b, hipStream_t st);
vbm_setup_host *vbm_setup_handle_host(vbm_setup_handle *h);

static const int kMaxWS = 8;     // (dependency masks hold 4 bits per workspace in 32)

struct vbm_encoder {
    vbm_setup_host *H;
    const vbm_setup *hs;     // host view
    int S, ch, cap;          // streams, channels, max stream-blocks per batch
    int L, Ls;               // leading dimensions for `cap`
    int max_packet_bytes;
    int max_oct, max_partvals;
    std::vector<void *> allocs;
    std::vector<size_t> alloc_bytes;          // size of every entry of `allocs`
    size_t ws_alloc[kMaxWS + 1] = {};         // allocs [ws_alloc[w], ws_alloc[w + 1]) are the scratch of workspace w (vbm_debug_poison_workspace)
    // Two complete batch workspaces (device pointers; the stream state is shared).  Consecutive calls
    // alternate between them, so that vbm_analysis_batch2 can run the second half of call k (floor fit,
    // couple/quantise, packet assembly) on one HIP stream while the first half of call k+1 (transforms,
    // psychoacoustics — everything that touches the carried stream state) runs on another.
    int nws = 2;             // workspaces in rotation (2: a call's back half beside the next call's front half; more let
                             // the short rounds of the front end run on while a big batch still holds its workspace)
    vbm_batch bw[kMaxWS];
    int *d_stream_id[kMaxWS];
    uint8_t *d_wflags[kMaxWS];
    int cur = 0;             // workspace of the last call (vbm_encoder_fetch)
    int next = 0;            // workspace of the next call
    hipEvent_t ev_front[kMaxWS] = {}, ev_back[kMaxWS] = {};
    bool back_pending[kMaxWS] = {};
    // last batch (for vbm_encoder_fetch)
    int last_nsb, last_mode;
    // sub-batches: the stages after the transforms run as `nsplit` tile-aligned slices of the batch,
    // each on its own internal HIP stream (forked from / joined to the caller's stream by events), so
    // the few-wavefront serial kernels of one slice overlap with the wide kernels of the others
    int nsplit = 1;
    std::vector<hipStream_t> sub;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    // pinned staging of stream_ids / wflags (double-buffered; skipped when unchanged)
    int *h_ids[2] = {nullptr, nullptr};
    uint8_t *h_flags[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    int stage_turn = 0;
    std::vector<uint8_t> seen;   // [S] scratch of the duplicate check
    std::vector<int> last_ids[kMaxWS];
    std::vector<uint8_t> last_flags[kMaxWS];
    // optional per-stage timing (HIP events on the stream each kernel is launched on)
    bool profiling = false;
    std::vector<hipEvent_t> events;   // pool; a (begin, end) pair per recorded stage launch
    size_t events_used = 0;
    struct span { int stage; size_t begin, end; };
    std::vector<span> spans;
    int prof_calls = 0, prof_max_calls = 0;
    long long prof_blocks = 0;        // stream-blocks of the profiled batches (rounds: the largest batch of each)
    // rounds with a deferred join (vbm_analysis_round_begin / _join): completion of every block type's batch
    // per workspace, and the batch each stream was part of in the previous round
    hipEvent_t ev_done[kMaxWS][4] = {};
    bool done_pending[kMaxWS][4] = {};     // outputs not yet joined to a caller's stream (vbm_analysis_round_join*)
    bool reuse_pending[kMaxWS][4] = {};    // the workspace slot has not been waited for by a fork stream since it ran
    // Order of a stream's blocks across rounds: every batch records an event after its front half (everything
    // that reads or writes the carried stream state; managed bitrate: after the whole batch, the reservoirs move
    // in the back half) and every stream remembers the batch it was last part of.  A later batch waits for the
    // front events of the batches its own streams come from — nothing else orders rounds against each other, so
    // which HIP stream a round is forked from, or joined to, does not matter.
    hipEvent_t ev_state[kMaxWS][4] = {};
    unsigned epoch[kMaxWS][4] = {};        // bumped every time the slot runs a batch
    signed char slot_queue[kMaxWS][4] = {};// internal HIP stream its front half ran on (sub[] index)
    struct last_batch { signed char w, m; unsigned epoch; };
    std::vector<last_batch> last;          // [S] w = -1: none
    // A big batch runs its front half (up to offset_and_mix: everything that touches the carried stream state)
    // on sub[4] and its back half on sub[5], like the two streams of vbm_analysis_batch2: with a lazy join
    // (vbm_analysis_round_join_lazy) the back half of one write's big batch runs beside the front half of the next.
    int lazy_w = -1, lazy_m = -1;   // the newest big batch (the one a lazy join leaves pending)
    int round_w = -1;                      // workspace of the newest round
    int call_big_w = -1;                   // rounds built on the device: workspace of the current call's big batch
    // ... run as HIP graphs: the ~180 launches of a round cost the host more than the round costs the device, and
    // with device-resident counts every launch of a round has the same arguments each time its workspace comes up.
    // Streams: sub[0..3] the small batches of block type 0..3 of all rounds (high priority), sub[4] the front halves of
    // the big (first round, long blocks) batches, sub[5] their back halves.  A group = the launches of one half of one
    // batch for one workspace, captured the second time it is needed and replayed from then on.
    struct round_graph { hipGraphExec_t exec = nullptr; int uses = 0; };
    // [workspace][block type; 4 = the big batch (type 3, first round): its launches differ from a small type-3 batch's
    // (throughput variants of the kernels, own streams)][0 whole pipeline, 1 front half, 2 back half]
    round_graph gJ[kMaxWS][5][3];
    hipEvent_t ev_state_big[kMaxWS] = {};  // front half of the big batch run in the workspace
    int queue_last_w[4] = {-1, -1, -1, -1};   // workspace of the newest job on sub[0..3]
    bool small_streams_set = false, small_share = false;
    hipEvent_t ev_cap_fork = nullptr, ev_cap_join[4] = {};
    int *d_counts_ws = nullptr;            // [kMaxWS][4] block counts of the round in each workspace
    int use_graphs = -1;
    int device_mode = 0;                   // how the last device-built round ran: 1 graphs, 2 plain launches
    bool device_rounds = false;            // ... have run: the per-stream bookkeeping below knows nothing of them
    // the tone-mask branch of a slice runs on its own stream beside the noise-mask branch
    bool overlap_branches = true;
    std::vector<hipStream_t> aux;
    std::vector<hipEvent_t> ev_aux_fork, ev_aux_join;
};

static const char *const kStageNames[] = {"window_mdct", "window_fft_log", "transpose", "prologue", "noisemask",
                                          "tonemask", "offset_and_mix", "floor_fit", "floor_encode",
                                          "couple_quantize", "pack", "packet_out"};
static const int kNumStages = (int)(sizeof(kStageNames) / sizeof(kStageNames[0]));
static const int kFront = 3;                       // stages launched on the whole batch (caller's stream)
static const int kBack = kNumStages - kFront;      // stages launched per slice

static int round64(int x) { return (x + 63) & ~63; }

// (begin, end) event pair around the launches of one stage, on the stream they go to (profiling only)
struct stage_scope {
    vbm_encoder *e;
    bool on;
    int stage;
    hipStream_t q;
    size_t eb = 0;
    stage_scope(vbm_encoder *e_, bool on_, int stage_, hipStream_t q_) : e(e_), on(on_), stage(stage_), q(q_)
    {
        if (on) { eb = e->events_used++; (void)hipEventRecord(e->events[eb], q); }
    }
    ~stage_scope()
    {
        if (on) {
            const size_t ee = e->events_used++;
            (void)hipEventRecord(e->events[ee], q);
            e->spans.push_back({stage, eb, ee});
        }
    }
};
static const int kBigBatch = 1024;   // stream-blocks from which a round's batch counts as big (own stream; front end holds its streams)

template <typename T>
static int dalloc(vbm_encoder *e, T **p, size_t count, bool zero = true)
{
    hipError_t err = hipMalloc((void **)p, count * sizeof(T) + 256);
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipMalloc(encoder workspace)");
    e->allocs.push_back(*p);
    e->alloc_bytes.push_back(count * sizeof(T) + 256);
    if (zero) {
        err = hipMemset(*p, 0, count * sizeof(T) + 256);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemset");
    }
    return 0;
}

extern "C" void vbm_encoder_destroy(vbm_encoder *e)
{
    if (!e) return;
    for (void *p : e->allocs) (void)hipFree(p);
    for (int i = 0; i < kMaxWS; i++) {
        for (int m = 0; m < 5; m++)
            for (int k = 0; k < 3; k++)
                if (e->gJ[i][m][k].exec) (void)hipGraphExecDestroy(e->gJ[i][m][k].exec);
        if (e->ev_state_big[i]) (void)hipEventDestroy(e->ev_state_big[i]);
    }
    if (e->ev_cap_fork) (void)hipEventDestroy(e->ev_cap_fork);
    for (int i = 0; i < 4; i++)
        if (e->ev_cap_join[i]) (void)hipEventDestroy(e->ev_cap_join[i]);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_join) (void)hipEventDestroy(ev);
    if (e->small_share && e->sub.size() >= 4) { e->sub[1] = nullptr; e->sub[3] = nullptr; }   // (aliases of sub[0] / sub[2])
    for (hipStream_t q : e->sub)
        if (q) (void)hipStreamDestroy(q);
    for (hipStream_t q : e->aux) (void)hipStreamDestroy(q);
    for (hipEvent_t ev : e->ev_aux_fork) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_aux_join) (void)hipEventDestroy(ev);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (int i = 0; i < kMaxWS; i++)
        for (int m = 0; m < 4; m++)
            { if (e->ev_done[i][m]) (void)hipEventDestroy(e->ev_done[i][m]);
              if (e->ev_state[i][m]) (void)hipEventDestroy(e->ev_state[i][m]); }
    for (int i = 0; i < kMaxWS; i++) {
        if (e->ev_front[i]) (void)hipEventDestroy(e->ev_front[i]);
        if (e->ev_back[i]) (void)hipEventDestroy(e->ev_back[i]);
    }
    for (int i = 0; i < 2; i++) {
        if (e->h_ids[i]) (void)hipHostFree(e->h_ids[i]);
        if (e->h_flags[i]) (void)hipHostFree(e->h_flags[i]);
        if (e->ev_stage[i]) (void)hipEventDestroy(e->ev_stage[i]);
    }
    delete e;
}

// vorbis_bitrate_init (reference lib/bitrate.c:28-56): reservoirs at the desired fill, floater in the middle
static int bitrate_state_init(vbm_encoder *e, vbm_stream_state &st)
{
    const vbm_setup *s = e->hs;
    const long long fill = s->managed ? (long long)((double)s->bi_reservoir_bits * s->bi_reservoir_bias) : 0;
    std::vector<long long> r(e->S, fill);
    std::vector<double> f(e->S, (double)(VBM_PACKETBLOBS / 2));
    if (hipMemcpy(st.bm_avg_reservoir, r.data(), e->S * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(st.bm_minmax_reservoir, r.data(), e->S * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(st.bm_avgfloat, f.data(), e->S * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        g_vbm_err = "hipMemcpy(bitrate manager state) failed";
        return VBM_EHIP;
    }
    return VBM_OK;
}

extern "C" int vbm_encoder_create(vbm_encoder **out, vbm_setup_handle *setup, int nstreams, int max_batch)
{
    if (!out || !setup || nstreams <= 0 || max_batch <= 0) return VBM_EINVAL;
    *out = nullptr;
    int ndev = vbm_device_count();
    if (ndev < 0) return ndev;
    if (ndev == 0) {
        g_vbm_err = "no HIP device: the MI355X path has no CPU fallback";
        return VBM_ENODEV;
    }
    vbm_encoder *e = new vbm_encoder();
    e->H = vbm_setup_handle_host(setup);
    int rc = vbm_setup_host_upload(e->H);
    if (rc) { delete e; return rc; }
    e->hs = vbm_setup_host_view(e->H);
    const vbm_setup *s = e->hs;
    e->S = nstreams;
    e->ch = s->channels;
    e->cap = max_batch;
    // lanes: the batch capacity plus padding for vbm_analysis_round, which starts every block type of a
    // round on a tile boundary (4 types -> up to 3 x 63 lanes of padding)
    e->L = round64((max_batch + 256) * e->ch);
    e->Ls = round64(max_batch + 256);
    e->max_packet_bytes = 4096 * ((e->ch + 1) / 2);   // generous: q10 stereo long blocks stay < 3 KB
    e->max_oct = 0;
    for (int i = 0; i < s->psys; i++)
        if (s->psy[i].total_octave_lines > e->max_oct) e->max_oct = s->psy[i].total_octave_lines;
    if (e->max_oct < 256) e->max_oct = 256;
    e->max_partvals = 1;
    for (int i = 0; i < s->residues; i++) {
        int pv = (s->residue[i].end - s->residue[i].begin) / s->residue[i].grouping;
        if (pv > e->max_partvals) e->max_partvals = pv;
    }
    e->last_nsb = 0;
    e->last_mode = -1;

    const int Nmax = s->blocksizes[1], nmax = Nmax / 2;
    const size_t L = e->L, Ls = e->Ls;
    memset(e->bw, 0, sizeof(e->bw));
  {
    vbm_batch &b = e->bw[0];
    b.setup = vbm_setup_device(e->H);
    b.ch = e->ch;
    b.max_packet_bytes = e->max_packet_bytes;
#define A(field, type, count) do { type *p_; rc = dalloc<type>(e, &p_, (count)); if (rc) { vbm_encoder_destroy(e); return rc; } field = p_; } while (0)
    // stream state (lib/codec_internal.h:85-92); -9999 ampmax as vorbis_block_init / _vp_global_look set it
    b.st.S = nstreams;
    b.st.ch = e->ch;
    b.st.Lc = round64(nstreams * e->ch);
    b.st.slab_words = (size_t)(2048 + 256) * 64;
    A(b.st.mblock, float, (size_t)(b.st.Lc / 64) * b.st.slab_words);
    b.st.tblock = b.st.mblock + (size_t)2048 * 64;
    A(b.st.lowcomp, float, (size_t)b.st.Lc);
    A(b.st.g_ampmax, float, (size_t)nstreams);
    A(b.st.vbi_ampmax, float, (size_t)nstreams);
    A(b.st.lW_block_mode, int, (size_t)nstreams);
    A(b.st.lW_no, int, (size_t)nstreams);
    A(b.st.impadnum, int, (size_t)nstreams);
    A(b.st.bm_avg_reservoir, long long, (size_t)nstreams);
    A(b.st.bm_minmax_reservoir, long long, (size_t)nstreams);
    A(b.st.bm_avgfloat, double, (size_t)nstreams);
    rc = bitrate_state_init(e, b.st);
    if (rc) { vbm_encoder_destroy(e); return rc; }
    {
        std::vector<float> init(nstreams, -9999.f);
        (void)hipMemcpy(b.st.g_ampmax, init.data(), nstreams * sizeof(float), hipMemcpyHostToDevice);
        (void)hipMemcpy(b.st.vbi_ampmax, init.data(), nstreams * sizeof(float), hipMemcpyHostToDevice);
    }
  }
  {
      const char *env = getenv("VBM_WORKSPACES");
      e->nws = env ? atoi(env) : 2;
      if (e->nws < 2) e->nws = 2;
      if (e->nws > kMaxWS) e->nws = kMaxWS;
  }
  // managed bitrate: every array the back half writes exists once per packetblob (batch.h, vbm_blob_select)
  const size_t NBL = s->managed ? VBM_PACKETBLOBS : 1;
  for (int w = 0; w < e->nws; w++) {
    vbm_batch &b = e->bw[w];
    e->ws_alloc[w] = e->allocs.size();
    if (w) {
        memset(&b, 0, sizeof(b));
        b.setup = e->bw[0].setup;
        b.ch = e->ch;
        b.max_packet_bytes = e->max_packet_bytes;
        b.st = e->bw[0].st;
    }
    A(e->d_stream_id[w], int, Ls);
    A(e->d_wflags[w], uint8_t, Ls);
    A(b.mdct_bm, float, L * nmax);
    A(b.logfft_bm, float, L * nmax);
    A(b.qf_bm, uint16_t, L * nmax);
    A(b.res_bm, int, L * nmax);
    A(b.local_ampmax, float, L);
    A(b.wflags_cb, uint8_t, L);
    // tiled slabs (batch.h): every per-bin array of a 64-lane tile sits in one contiguous slab
    {
        size_t rows = 0;
        auto take = [&](size_t r) { size_t at = rows; rows += r; return at * 64; };
        const size_t o_mdct = take(nmax), o_logmdct = take(nmax), o_noise = take(nmax),
                     o_tone = take(nmax), o_logmask = take(nmax), o_epeak = take(nmax),
                     o_npeak = take(nmax / 8 + 1), o_post = take((size_t)(VBM_VIF_POSIT + 2) * VBM_PACKETBLOBS),
                     o_fout = take((size_t)(VBM_VIF_POSIT + 2) * NBL), o_iwork = take((size_t)nmax * NBL);
        b.slab_words = rows * 64;
        float *slab;
        A(slab, float, (L / 64) * b.slab_words);
        b.mdctT = slab + o_mdct; b.logmdctT = slab + o_logmdct;
        b.noiseT = slab + o_noise; b.toneT = slab + o_tone; b.logmaskT = slab + o_logmask;
        b.epeakT = slab + o_epeak; b.npeakT = slab + o_npeak;
        b.postT_blob = (int *)(slab + o_post);
        b.postT = b.postT_blob + (size_t)(VBM_PACKETBLOBS / 2) * (VBM_VIF_POSIT + 2) * 64;
        b.floor_outT = (int *)(slab + o_fout); b.iworkT = (int *)(slab + o_iwork);
        b.floor_outT_blob = b.floor_outT; b.iworkT_blob = b.iworkT;
        b.blob_iwork_rows = nmax;

        size_t srows = 0;
        auto stake = [&](size_t r) { size_t at = srows; srows += r; return at * 64; };
        int max_steps = 1;
        for (int i = 0; i < s->modes && i < 2; i++)
            if (s->map[i].coupling_steps > max_steps) max_steps = s->map[i].coupling_steps;
        // (the residue VQ stages its partitions in LDS: no interleaved copy of the residue in HBM any more)
        b.blob_pw_rows = e->max_partvals * e->ch;
        b.blob_m6_rows = (nmax / 8 + 1) * max_steps;
        const size_t o_pw = stake((size_t)b.blob_pw_rows * NBL), o_vq = stake(1),
                     o_m6 = stake((size_t)b.blob_m6_rows * NBL);
        int max_stages = 1;
        for (int i = 0; i < s->residues; i++)
            if (s->residue[i].stages > max_stages) max_stages = s->residue[i].stages;
        b.blob_len_rows = max_stages * e->ch * e->max_partvals;
        const size_t o_len = stake((size_t)b.blob_len_rows * NBL), o_off = stake((size_t)b.blob_len_rows * NBL);
        b.sb_slab_words = srows * 64;
        int *sslab;
        A(sslab, int, (Ls / 64) * b.sb_slab_words);
        b.partwordT = sslab + o_pw;
        b.workvqT = sslab + o_vq;
        b.m6defT = (float *)(sslab + o_m6);
        b.vqlenT = sslab + o_len;
        b.vqoffT = sslab + o_off;
        b.vq_slab_words = (size_t)max_stages * nmax * e->ch * 64;
        b.vq_blob_words = (Ls / 64) * b.vq_slab_words;
        A(b.vqcodeT, uint64_t, b.vq_blob_words * NBL);
        b.partwordT_blob = b.partwordT; b.m6defT_blob = b.m6defT; b.vqlenT_blob = b.vqlenT; b.vqoffT_blob = b.vqoffT;
        b.vqcodeT_blob = b.vqcodeT;
    }
    A(b.poste, float, L);
    A(b.global_ampmax, float, Ls);
    A(b.post_valid_blob, int, L * VBM_PACKETBLOBS);
    b.post_valid = b.post_valid_blob + (size_t)(VBM_PACKETBLOBS / 2) * L;
    A(b.nonzero, int, L * NBL);
    b.nonzero_blob = b.nonzero;
    b.nblobs = 1;
    A(b.packetT, uint8_t, Ls * (size_t)e->max_packet_bytes);   // [sb>>6][max_packet_bytes][64]
    A(b.packet_bytes, int, Ls);
    if (s->managed) {   // one packet buffer per packetblob (lib/mapping0.c:1204)
        A(b.packetT_blob, uint8_t, Ls * (size_t)e->max_packet_bytes * VBM_PACKETBLOBS);
        A(b.packet_bytes_blob, int, Ls * VBM_PACKETBLOBS);
        A(b.choice, int, Ls);
    }
    A(b.packet_bits, int, Ls * NBL);
    b.packet_bits_blob = b.packet_bits;
    b.stream_id = e->d_stream_id[w];
    b.wflags = e->d_wflags[w];
    e->ws_alloc[w + 1] = e->allocs.size();
  }
#undef A
    for (int i = 0; i < kMaxWS; i++)
        if (hipEventCreateWithFlags(&e->ev_front[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_back[i], hipEventDisableTiming) != hipSuccess) {
            vbm_encoder_destroy(e);
            return VBM_EHIP;
        }
    for (int i = 0; i < 2; i++) {
        if (hipHostMalloc((void **)&e->h_ids[i], Ls * sizeof(int), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&e->h_flags[i], Ls, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_stage[i], hipEventDisableTiming) != hipSuccess) {
            vbm_encoder_destroy(e);
            g_vbm_err = "hipHostMalloc / hipEventCreate (staging) failed";
            return VBM_EHIP;
        }
    }
    if (hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) {
        vbm_encoder_destroy(e);
        return VBM_EHIP;
    }
    for (int i = 0; i < kMaxWS; i++)
        for (int m = 0; m < 4; m++)
            if (hipEventCreateWithFlags(&e->ev_done[i][m], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_state[i][m], hipEventDisableTiming) != hipSuccess ||
                (m == 0 && hipEventCreateWithFlags(&e->ev_state_big[i], hipEventDisableTiming) != hipSuccess)) {
                vbm_encoder_destroy(e);
                return VBM_EHIP;
            }
    e->last.assign(nstreams, {-1, -1, 0u});
    {
        const char *env = getenv("VBM_SUB_BATCHES");
        int rc2 = vbm_encoder_set_sub_batches(e, env ? atoi(env) : 1);
        if (rc2) { vbm_encoder_destroy(e); return rc2; }
        env = getenv("VBM_OVERLAP_BRANCHES");
        if (env) e->overlap_branches = atoi(env) != 0;
    }
    // the zero fills above went to the null stream, which the internal (non-blocking) HIP streams do not wait for
    if (hipDeviceSynchronize() != hipSuccess) { vbm_encoder_destroy(e); return VBM_EHIP; }
    *out = e;
    return VBM_OK;
}

extern "C" int vbm_encoder_set_sub_batches(vbm_encoder *e, int n)
{
    if (!e || n < 1 || n > 64) return VBM_EINVAL;
    while ((int)e->sub.size() < n) {
        hipStream_t q;
        hipEvent_t ev;
        hipError_t err = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamCreateWithFlags");
        e->sub.push_back(q);
        err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipEventCreateWithFlags");
        e->ev_join.push_back(ev);
    }
    while ((int)e->aux.size() < n) {
        hipStream_t q;
        hipEvent_t e1, e2;
        hipError_t err = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamCreateWithFlags");
        e->aux.push_back(q);
        if ((err = hipEventCreateWithFlags(&e1, hipEventDisableTiming)) != hipSuccess ||
            (err = hipEventCreateWithFlags(&e2, hipEventDisableTiming)) != hipSuccess)
            return vbm_set_hip_error(err, "hipEventCreateWithFlags");
        e->ev_aux_fork.push_back(e1);
        e->ev_aux_join.push_back(e2);
    }
    e->nsplit = n;
    return VBM_OK;
}

extern "C" int vbm_encoder_sub_batches(const vbm_encoder *e) { return e ? e->nsplit : VBM_EINVAL; }

// internal accessors for the stream front end (capi_frontend.cpp)
vbm_setup_host *vbm_encoder_setup_host(vbm_encoder *e) { return e->H; }
int vbm_encoder_streams(const vbm_encoder *e) { return e->S; }
int vbm_encoder_workspaces(const vbm_encoder *e) { return e->nws; }

extern "C" int vbm_encoder_max_packet_bytes(const vbm_encoder *e) { return e ? e->max_packet_bytes : VBM_EINVAL; }

extern "C" int vbm_encoder_reset(vbm_encoder *e)
{
    if (!e) return VBM_EINVAL;
    // Rounds and back halves may still be in flight on the internal (non-blocking) HIP streams, which the null
    // stream of the copies below does not order against: wait for the device, then forget the bookkeeping.
    hipError_t serr = hipDeviceSynchronize();
    if (serr != hipSuccess) return vbm_set_hip_error(serr, "hipDeviceSynchronize");
    memset(e->done_pending, 0, sizeof(e->done_pending));
    memset(e->reuse_pending, 0, sizeof(e->reuse_pending));
    memset(e->back_pending, 0, sizeof(e->back_pending));
    e->last.assign(e->S, {-1, -1, 0u});
    e->lazy_w = e->lazy_m = e->round_w = -1;
    for (int i = 0; i < kMaxWS; i++) { e->last_ids[i].clear(); e->last_flags[i].clear(); }
    vbm_stream_state &st = e->bw[0].st;
    (void)hipMemset(st.mblock, 0, (size_t)(st.Lc / 64) * st.slab_words * sizeof(float));   // mblock + tblock
    (void)hipMemset(st.lowcomp, 0, (size_t)st.Lc * sizeof(float));
    (void)hipMemset(st.lW_block_mode, 0, e->S * sizeof(int));
    (void)hipMemset(st.lW_no, 0, e->S * sizeof(int));
    (void)hipMemset(st.impadnum, 0, e->S * sizeof(int));
    std::vector<float> init(e->S, -9999.f);
    (void)hipMemcpy(st.g_ampmax, init.data(), e->S * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(st.vbi_ampmax, init.data(), e->S * sizeof(float), hipMemcpyHostToDevice);
    return bitrate_state_init(e, st);
}

// the listed streams start over (a new stream in a used slot); d_ids: device copy of the ids
int vbm_encoder_reset_streams_dev(vbm_encoder *e, const int *d_ids, int n, hipStream_t q)
{
    const vbm_setup *s = e->hs;
    const long long fill = s->managed ? (long long)((double)s->bi_reservoir_bits * s->bi_reservoir_bias) : 0;
    if (vbm_analysis_round_join(e, q)) return VBM_EHIP;
    for (int w = 0; w < e->nws; w++)      // ... including batches a join to another stream has already released
        for (int m = 0; m < 4; m++)
            if (e->reuse_pending[w][m]) {
                if (hipStreamWaitEvent(q, e->ev_done[w][m], 0) != hipSuccess) return VBM_EHIP;
                e->reuse_pending[w][m] = false;
            }
    return vbm_launch_reset_streams(&e->bw[0].st, d_ids, n, fill, q) ? VBM_EHIP : VBM_OK;
}

static void configure(vbm_encoder *e, vbm_batch &b, int block_mode, int nsb, const float *d_pcm, int w)
{
    b = e->bw[w];
    const vbm_setup *s = e->hs;
    b.block_mode = block_mode;
    b.W = block_mode >> 1;
    b.N = s->blocksizes[b.W];
    b.n = b.N / 2;
    b.nsb = nsb;
    b.ncb = nsb * e->ch;
    b.L = e->L;      // fixed leading dimensions: buffers were sized for `cap`
    b.Ls = e->Ls;
    b.pcm = d_pcm;
    b.blobno = VBM_PACKETBLOBS / 2;
    b.fit_max_posts = 2;
    for (int c = 0; c < e->ch; c++) {
        const vbm_map &mp = s->map[b.W];
        const int posts = s->floor[mp.floorsubmap[mp.chmuxlist[c]]].posts;
        if (posts > b.fit_max_posts) b.fit_max_posts = posts;
    }
    b.mix_makes_qf = getenv("VBM_SEPARATE_FLOOR_PREP") ? 0 : vbm_mix_can_make_qf(&b);
    {
        // opt-in (VBM_NOISE_RING=1): measured slower than the plain form — alone 0.93 ms against 0.73, from PCM 5.10 ms per
        // step against 4.51 (profiles/r03/README.md): a barrier per 64-bin chunk puts the scan's chunk on every iteration's
        // critical path, and seven workgroups of five wavefronts per CU do not make up for it
        const char *rv = getenv("VBM_NOISE_RING");      // (read per call: the tests switch it inside one process)
        const int ring = rv ? atoi(rv) : 0;
        b.noise_ring = ring && s->psy[block_mode].hy_ring;
    }
    {
        // partition slicing of couple/quantise (quant_kernels.hip): allowed when no channel takes part
        // in two coupling steps; lowpass rounding as lib/mapping0.c:778-781
        const vbm_map &m = s->map[b.W];
        const vbm_psy &p = s->psy[block_mode];
        unsigned seen = 0;
        b.couple_parallel = 1;
        for (int i = 0; i < m.coupling_steps; i++) {
            unsigned bits = (1u << m.coupling_mag[i]) | (1u << m.coupling_ang[i]);
            if (seen & bits) b.couple_parallel = 0;
            seen |= bits;
        }
        const int partition = p.normal_p ? p.normal_partition : 16;
        int lowpassr = s->block_lowpassr[b.W ? 1 : 0];
        if (lowpassr % p.normal_partition) lowpassr = (lowpassr / p.normal_partition + 1) * p.normal_partition;
        b.couple_parts = (lowpassr + partition - 1) / partition;
        int m6end = p.tonefix_end < lowpassr ? p.tonefix_end : lowpassr;
        b.couple_m6parts = m.coupling_steps ? (m6end + partition - 1) / partition : 0;
        if (b.couple_parts > b.n / 8 + 1) b.couple_parallel = 0;   // table rows were sized for partitions >= 8 bins
        // lane-per-bin kernel (quant_kernels.hip, k_couple_fast): 32-bin partitions only
        b.couple_fast = 0;
        if (partition == 32 && p.normal_partition == 32 && (b.n % 32) == 0 && !getenv("VBM_COUPLE_GENERAL")) {
            if (m.coupling_steps == 0) b.couple_fast = 1;
            else if (m.coupling_steps == 1 && e->ch == 2 && b.couple_parallel) b.couple_fast = 2;
        }
        // fused packet assembly: one submap, its channels one coded vector, couple kernel = the lane-per-bin one (it writes
        // the residue in the coder's own order).  Opt-in (VBM_PACK_FUSED=1): it moves 0.2 GB per step instead of 0.9, issues
        // the same vector instructions as the three-kernel path (136 M against 150 M per step: the cascaded VQ is ~100
        // instructions per vector either way) and holds 21 KB of LDS per stream-block while it does — a wavefront of the
        // lane-per-block kernels holds 8 KB for 64 of them — which keeps the other half of the pipeline off the CUs:
        // pack alone 0.80 ms against 0.64, from PCM 4.81 ms per step against 4.67.
        {
            const char *pf = getenv("VBM_PACK_FUSED");      // (read per call: the tests switch it inside one process)
            const int want = pf ? atoi(pf) : 0;
            const vbm_residue &r0 = s->residue[m.residuesubmap[0]];
            const int nbch = e->ch;
            b.pack_fused = want && !s->managed && m.submaps == 1 && b.couple_fast &&
                           ((r0.type == 2 && (r0.grouping % nbch) == 0 && (r0.begin % nbch) == 0) || e->ch == 1) &&
                           r0.end <= b.n * nbch && r0.phrase_dim >= 1 && r0.phrase_dim <= 8 &&
                           (r0.end - r0.begin) / r0.grouping <= 64;     // a lane per partition
        }
        b.pack_submaps = m.submaps;
        for (int i = 0; i < m.submaps && i < 16; i++) {
            const vbm_residue &r = s->residue[m.residuesubmap[i]];
            b.pack_partvals[i] = (r.end - r.begin) / r.grouping;
            b.pack_spp[i] = r.grouping;
        }
    }
}

// slice [sb0, sb0+nsb) of a configured batch; sb0 is a multiple of 64, so every tiled array starts on
// a tile boundary (channel-block tiles as well: 64*ch channel-blocks)
static vbm_batch slice_of(const vbm_batch &f, int sb0, int nsb, uint8_t *d_packets_unused = nullptr)
{
    (void)d_packets_unused;
    vbm_batch v = f;
    const size_t cb0 = (size_t)sb0 * f.ch;
    const size_t ct = cb0 >> 6, stl = (size_t)sb0 >> 6;
    v.nsb = nsb;
    v.ncb = nsb * f.ch;
    v.stream_id += sb0; v.wflags += sb0;
    v.pcm += cb0 * f.N; v.mdct_bm += cb0 * f.n; v.logfft_bm += cb0 * f.n; v.qf_bm += cb0 * f.n; v.res_bm += cb0 * f.n;
    v.local_ampmax += cb0; v.wflags_cb += cb0; v.poste += cb0; v.post_valid += cb0; v.nonzero += cb0;
    v.post_valid_blob += cb0; v.nonzero_blob += cb0;
    v.global_ampmax += sb0; v.packet_bytes += sb0; v.packet_bits += sb0; v.packet_bits_blob += sb0;
    if (v.packet_bytes_blob) { v.packet_bytes_blob += sb0; v.choice += sb0; v.packetT_blob += stl * 64 * (size_t)f.max_packet_bytes; }
    const size_t co = ct * f.slab_words;
    v.mdctT += co; v.logmdctT += co; v.noiseT += co; v.toneT += co; v.logmaskT += co;
    v.epeakT += co; v.npeakT += co;
    v.postT += co; v.postT_blob += co; v.floor_outT += co; v.iworkT += co;
    v.floor_outT_blob += co; v.iworkT_blob += co;
    const size_t so = stl * f.sb_slab_words;
    v.partwordT += so; v.workvqT += so; v.m6defT += so; v.vqlenT += so; v.vqoffT += so;
    v.partwordT_blob += so; v.m6defT_blob += so; v.vqlenT_blob += so; v.vqoffT_blob += so;
    v.vqcodeT += stl * f.vq_slab_words; v.vqcodeT_blob += stl * f.vq_slab_words;
    v.packetT += stl * 64 * (size_t)f.max_packet_bytes;
    return v;
}

// Managed bitrate (reference lib/mapping0.c:1044-1181): _vp_offset_and_mix + floor1_fit three times
// (offset_select 1, 2, 0 -> blobs PACKETBLOBS/2, PACKETBLOBS-1, 0), the interpolated fits between, and
// the block-state update the reference applies once per packetblob.
static int managed_front(const vbm_batch &v, hipStream_t q)
{
    static const int sel[3] = {1, 2, 0}, slot[3] = {VBM_PACKETBLOBS / 2, VBM_PACKETBLOBS - 1, 0};
    for (int t = 0; t < 3; t++) {
        vbm_batch f = v;
        f.postT = v.postT_blob + (size_t)slot[t] * (VBM_VIF_POSIT + 2) * 64;
        f.post_valid = v.post_valid_blob + (size_t)slot[t] * v.L;
        if (vbm_launch_mix_managed(&f, sel[t], q) || vbm_launch_floor_fit(&f, q)) return -2;
    }
    if (vbm_launch_block_state_managed(&v, q) || vbm_launch_floor_interp(&v, q)) return -2;
    return 0;
}

// loop C once per packetblob (lib/mapping0.c:1204-1313), then vorbis_bitrate_addblock / _flushpacket
// (lib/bitrate.c:98-252): choice, final lengths, the chosen packets
// `before_choose`: called between the blobs' passes and the bitrate manager's choice (the only step of the back half that
// touches state carried from block to block: bm_avg_reservoir / bm_minmax_reservoir / bm_avgfloat of the streams)
static int managed_back(const vbm_batch &v, uint8_t *d_packets, hipStream_t q, const std::function<int()> &before_choose = nullptr)
{
    const char *wv = getenv("VBM_MANAGED_WIDE");        // (read per call: the tests switch it inside one process)
    const int wide = wv ? atoi(wv) : 1;
    if (wide) {
        // all fifteen packetblobs per launch (blob = blockIdx.z); the blobs' coupling passes form a chain through the npeak
        // rows (lib/mapping0.c:1249-1260, lib/psy.c:5100-5108): the lane-per-bin kernel walks them in a loop of its own, the
        // general kernel is launched once per blob
        vbm_batch f = v;
        f.nblobs = VBM_PACKETBLOBS;
        vbm_blob_select(f, 0);
        if (vbm_launch_floor_encode(&f, q)) return -2;
        if (f.couple_fast) {
            if (vbm_launch_couple_quantize(&f, q)) return -2;
        } else {
            for (int k = 0; k < VBM_PACKETBLOBS; k++) {
                vbm_batch g = v;
                vbm_blob_select(g, k);
                if (vbm_launch_couple_quantize(&g, q)) return -2;
            }
        }
        if (vbm_launch_pack(&f, q)) return -2;
        if (before_choose && before_choose()) return -2;
        return vbm_launch_bitrate_choose(&v, d_packets, q);
    }
    for (int k = 0; k < VBM_PACKETBLOBS; k++) {
        vbm_batch f = v;
        vbm_blob_select(f, k);
        if (vbm_launch_floor_encode(&f, q) || vbm_launch_couple_quantize(&f, q) || vbm_launch_pack(&f, q)) return -2;
    }
    if (before_choose && before_choose()) return -2;
    return vbm_launch_bitrate_choose(&v, d_packets, q);
}

extern "C" int vbm_analysis_batch(vbm_encoder *e, int block_mode, int nsb, const int *stream_ids,
                                  const uint8_t *wflags, const float *d_pcm, uint8_t *d_packets,
                                  int *d_packet_bytes, void *stream)
{
    return vbm_analysis_batch2(e, block_mode, nsb, stream_ids, wflags, d_pcm, d_packets, d_packet_bytes, stream, stream);
}

// Two-stream form: the FRONT half (transforms, psychoacoustics, offset_and_mix — every kernel that reads or
// writes the carried stream state) is ordered on `stream_front`, the BACK half (floor fit/encode,
// couple/quantise, packet assembly) on `stream_back`.  With two different streams the back half of one
// call overlaps the front half of the next (consecutive calls use alternate workspaces).
extern "C" int vbm_analysis_batch2(vbm_encoder *e, int block_mode, int nsb, const int *stream_ids,
                                   const uint8_t *wflags, const float *d_pcm, uint8_t *d_packets,
                                   int *d_packet_bytes, void *stream_front, void *stream_back)
{
    void *stream = stream_front;
    if (!e || block_mode < 0 || block_mode > 3 || nsb < 0 || nsb > e->cap) return VBM_EINVAL;
    if (nsb == 0) return VBM_OK;
    if (!stream_ids || !wflags || !d_pcm) return VBM_EINVAL;
    if (((uintptr_t)d_pcm & 15) || ((uintptr_t)d_packets & 3)) return VBM_EINVAL;
    const vbm_setup *s = e->hs;
    if (s->modes < 2 && (block_mode >> 1)) return VBM_EINVAL;
    for (int i = 0; i < nsb; i++)
        if (stream_ids[i] < 0 || stream_ids[i] >= e->S) return VBM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sback = (hipStream_t)stream_back;
    const bool two = (sback != st);
    hipError_t err;
    {   // rounds begun with a deferred join are completed first: their streams may be in this batch
        int rcj = vbm_analysis_round_join(e, stream);
        if (rcj) return rcj;
        for (int ww = 0; ww < e->nws; ww++)
            for (int m = 0; m < 4; m++)
                if (e->reuse_pending[ww][m]) {
                    if ((err = hipStreamWaitEvent((hipStream_t)stream, e->ev_done[ww][m], 0)) != hipSuccess)
                        return vbm_set_hip_error(err, "hipStreamWaitEvent");
                    e->reuse_pending[ww][m] = false;
                }
        e->round_w = -1;
    }
    const int w = e->next;
    e->next = (e->next + 1) % e->nws;
    e->cur = w;
    // this workspace was last read by the back half of the call before the previous one
    if (e->back_pending[w]) {
        if ((err = hipStreamWaitEvent(st, e->ev_back[w], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        e->back_pending[w] = false;
    }
    // stream ids / window flags: through pinned staging so the copy never stalls the host, and not
    // at all when the caller repeats the previous lists (the steady state of a streaming encode)
    if ((int)e->last_ids[w].size() != nsb || memcmp(e->last_ids[w].data(), stream_ids, nsb * sizeof(int)) ||
        memcmp(e->last_flags[w].data(), wflags, nsb)) {
        // a stream once per batch: two blocks of one stream would race on its carried state
        e->seen.assign(e->S, 0);
        for (int i = 0; i < nsb; i++) {
            if (e->seen[stream_ids[i]]) { g_vbm_err = "a stream id appears twice in one batch"; return VBM_EINVAL; }
            e->seen[stream_ids[i]] = 1;
        }
        const int t = e->stage_turn;
        e->stage_turn ^= 1;
        (void)hipEventSynchronize(e->ev_stage[t]);   // the copy that last used this buffer has run
        memcpy(e->h_ids[t], stream_ids, nsb * sizeof(int));
        memcpy(e->h_flags[t], wflags, nsb);
        if ((err = hipMemcpyAsync(e->d_stream_id[w], e->h_ids[t], nsb * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess)
            return vbm_set_hip_error(err, "hipMemcpyAsync(stream_ids)");
        if ((err = hipMemcpyAsync(e->d_wflags[w], e->h_flags[t], nsb, hipMemcpyHostToDevice, st)) != hipSuccess)
            return vbm_set_hip_error(err, "hipMemcpyAsync(wflags)");
        (void)hipEventRecord(e->ev_stage[t], st);
        e->last_ids[w].assign(stream_ids, stream_ids + nsb);
        e->last_flags[w].assign(wflags, wflags + nsb);
    }

    vbm_batch b;
    configure(e, b, block_mode, nsb, d_pcm, w);
    e->last_nsb = nsb;
    e->last_mode = block_mode;

    // slices: multiples of 64 stream-blocks
    const int tiles = (nsb + 63) / 64;
    int nsplit = e->nsplit < tiles ? e->nsplit : tiles;
    if (nsplit < 1) nsplit = 1;
    const int nback = nsplit;             // two-stream form: the sub-batches apply to the back half only
    if (two) nsplit = 1;

    int W = b.W;
    int rc = 0;
    const size_t ev_need = (size_t)2 * (kFront + (size_t)(nsplit > nback ? nsplit : nback) * kBack);
    const bool prof = e->profiling && e->prof_calls < e->prof_max_calls && e->events_used + ev_need <= e->events.size();
    // STAGE(k, q, launch): run `launch` on stream q, bracketed by a (begin, end) event pair when profiling
#define RUN(x) do { rc = (x); if (rc) { g_vbm_err = std::string("launch failed: ") + #x; return VBM_EHIP; } } while (0)
#define STAGE(k, q, launch)                                                                    \
    do {                                                                                       \
        size_t eb_ = 0;                                                                        \
        if (prof) { eb_ = e->events_used++; (void)hipEventRecord(e->events[eb_], (q)); }       \
        launch;                                                                                \
        if (prof) {                                                                            \
            size_t ee_ = e->events_used++;                                                     \
            (void)hipEventRecord(e->events[ee_], (q));                                         \
            e->spans.push_back({(k), eb_, ee_});                                               \
        }                                                                                      \
    } while (0)
    vbm_debug_delay_point(VBM_DP_BATCH_FRONT, st);
    RUN(vbm_launch_spread_flags(&b, st));
    // loop A: window + MDCT, window + FFT + log spectrum (wave per block), whole batch
    STAGE(0, st, RUN(vbm_launch_window_mdct(b.pcm, b.mdct_bm, W ? b.wflags_cb : nullptr,
                                            vbm_setup_device_ptrs(e->H)->mdct_trig[W], vbm_setup_device_ptrs(e->H)->window[W],
                                            vbm_setup_device_ptrs(e->H)->window[0], b.N, s->blocksizes[0], 1, b.ncb, 0, nullptr, 0, st)));
    STAGE(1, st, RUN(vbm_launch_window_fft_log(b.pcm, b.logfft_bm, b.local_ampmax, W ? b.wflags_cb : nullptr,
                                               vbm_setup_device_ptrs(e->H)->fft_wa[W], vbm_setup_device_ptrs(e->H)->window[W],
                                               vbm_setup_device_ptrs(e->H)->window[0], b.N, s->blocksizes[0], b.ncb, nullptr, 0, st)));
    // (stage 2, "transpose": gone — k_noisemask writes the tiled MDCT rows itself)

    // the stages between the transforms and the hand-over: psychoacoustics, offset_and_mix, block state
    auto front_stages = [&](vbm_batch &v, hipStream_t q, int part) -> int {
        // loop B: psychoacoustics + floor fit (lane per channel-block).  The tone-mask branch reads only
        // the log spectrum and the prologue's maxima, the noise-mask branch only the MDCT: they run side
        // by side on two streams and meet at _vp_offset_and_mix.
        STAGE(3, q, RUN(vbm_launch_prologue(&v, q)));
        if (e->overlap_branches) {
            hipStream_t qa = e->aux[part];
            if ((err = hipEventRecord(e->ev_aux_fork[part], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
            if ((err = hipStreamWaitEvent(qa, e->ev_aux_fork[part], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            STAGE(5, qa, RUN(vbm_launch_tonemask(&v, s->psy[v.block_mode].total_octave_lines, qa)));
            if ((err = hipEventRecord(e->ev_aux_join[part], qa)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
            STAGE(4, q, RUN(vbm_launch_noisemask(&v, q)));
            if ((err = hipStreamWaitEvent(q, e->ev_aux_join[part], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        } else {
            STAGE(4, q, RUN(vbm_launch_noisemask(&v, q)));
            STAGE(5, q, RUN(vbm_launch_tonemask(&v, s->psy[v.block_mode].total_octave_lines, q)));
        }
        if (s->managed) STAGE(6, q, RUN(managed_front(v, q)));
        else STAGE(6, q, { RUN(vbm_launch_mix(&v, q)); RUN(vbm_launch_block_state(&v, q)); });
        return 0;
    };
    // floor fit .. packets of stream-blocks [sb0, sb0 + v.nsb)
    auto back_stages = [&](vbm_batch &v, hipStream_t q, int sb0) -> int {
        if (s->managed) {
            STAGE(10, q, RUN(managed_back(v, d_packets ? d_packets + (size_t)sb0 * e->max_packet_bytes : nullptr, q)));
        } else {
            STAGE(7, q, RUN(vbm_launch_floor_fit(&v, q)));
            // loop C: floor encode, couple/quantise, residue + packet assembly
            STAGE(8, q, RUN(vbm_launch_floor_encode(&v, q)));
            STAGE(9, q, RUN(vbm_launch_couple_quantize(&v, q)));
            STAGE(10, q, RUN(vbm_launch_pack(&v, q)));
        }
        STAGE(11, q, {
            if (v.pack_fused) {
                RUN(vbm_launch_packets_out(&v, d_packets ? d_packets + (size_t)sb0 * e->max_packet_bytes : nullptr,
                                           d_packet_bytes ? d_packet_bytes + sb0 : nullptr, q));
            } else {
            if (d_packets && !s->managed)   // word-major tiles -> [nsb][max_packet_bytes] bytes (little-endian words)
                RUN(vbm_launch_untranspose_i32((const int *)v.packetT,
                                               (int *)(d_packets + (size_t)sb0 * e->max_packet_bytes),
                                               e->max_packet_bytes / 4, (size_t)(e->max_packet_bytes / 4) * 64,
                                               v.nsb, q));
            if (d_packet_bytes) {
                if ((err = hipMemcpyAsync(d_packet_bytes + sb0, v.packet_bytes, v.nsb * sizeof(int), hipMemcpyDeviceToDevice, q)) != hipSuccess)
                    return vbm_set_hip_error(err, "hipMemcpyAsync(packet_bytes)");
            }
            }
        });
        return 0;
    };

    if (two) {
        // front half of the whole batch on the caller's front stream; the back half as `nback` tile-aligned
        // slices, the first on the caller's back stream and the others on internal streams beside it (their
        // few-wavefront serial kernels then overlap each other's wide ones), joined back before ev_back
        vbm_batch v = b;
        if ((rc = front_stages(v, st, 0))) return rc;
        if ((err = hipEventRecord(e->ev_front[w], st)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        for (int part = 0; part < nback; part++) {
            const int t0 = (int)((long)tiles * part / nback), t1 = (int)((long)tiles * (part + 1) / nback);
            const int sb0 = t0 * 64, sb1 = (t1 * 64 < nsb) ? t1 * 64 : nsb;
            if (sb1 <= sb0) continue;
            hipStream_t qb = part ? e->sub[part] : sback;
            if ((err = hipStreamWaitEvent(qb, e->ev_front[w], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            vbm_debug_delay_point(VBM_DP_BATCH_BACK, qb);
            vbm_batch vs = nback > 1 ? slice_of(b, sb0, sb1 - sb0) : b;
            if ((rc = back_stages(vs, qb, sb0))) return rc;
            if (part) {
                if ((err = hipEventRecord(e->ev_join[part], qb)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
                if ((err = hipStreamWaitEvent(sback, e->ev_join[part], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            }
        }
        if ((err = hipEventRecord(e->ev_back[w], sback)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        e->back_pending[w] = true;
    } else {
        if (nsplit > 1 && (err = hipEventRecord(e->ev_fork, st)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        for (int part = 0; part < nsplit; part++) {
            const int t0 = (int)((long)tiles * part / nsplit), t1 = (int)((long)tiles * (part + 1) / nsplit);
            const int sb0 = t0 * 64, sb1 = (t1 * 64 < nsb) ? t1 * 64 : nsb;
            if (sb1 <= sb0) continue;
            hipStream_t q = st;
            vbm_batch v = b;
            if (nsplit > 1) {
                q = e->sub[part];
                v = slice_of(b, sb0, sb1 - sb0);
                if ((err = hipStreamWaitEvent(q, e->ev_fork, 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            }
            if ((rc = front_stages(v, q, part))) return rc;
            if ((rc = back_stages(v, q, sb0))) return rc;
            if (nsplit > 1) {
                if ((err = hipEventRecord(e->ev_join[part], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
                if ((err = hipStreamWaitEvent(st, e->ev_join[part], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            }
        }
    }
    if (prof) { e->prof_calls++; e->prof_blocks += nsb; }
#undef STAGE
#undef RUN
    return VBM_OK;
}

// One block type of a round: its whole pipeline on its internal HIP stream(s), forked from ev_fork.
struct type_job {
    int m, w;                    // block type, workspace
    int lane0, bound;            // first stream-block lane of the type's region; blocks in it (d_nsb set: the most it may hold)
    const int *d_nsb;            // device-resident count (rounds built on the device) or NULL
    const float *pcm;            // the region's block PCM [bound][ch][N]
    uint8_t *d_packets;          // outputs of this type, already offset (may be NULL)
    int *d_packet_bytes;
    bool big, timed;
    unsigned depmask;            // bit (ww * 4 + t): slot (ww, t) holds a batch this one's streams may come from
    // rounds run as graphs (device_round_run_graphs): the job is one piece of a group that is forked from / joined
    // to its origin stream by the caller, in or outside a stream capture
    int few = 0;                 // small batch behind a large launch bound (device-built rounds): latency-bound kernel variants
    int part = 0;                // 0 whole pipeline, 1 front half only (up to the block-state update), 2 back half only
    bool grouped = false;        // no ev_fork wait, no dependency waits, no state / done events, no output copy
    hipStream_t q_on = nullptr;  // grouped: the stream to enqueue on
};

static int enqueue_job(vbm_encoder *e, const type_job &j)
{
    const vbm_setup *s = e->hs;
    hipError_t err;
    int rc = 0;
    const int m = j.m, w = j.w;
    hipStream_t q = j.grouped ? j.q_on : (j.big ? e->sub[4] : e->sub[m]);
    const int qid = j.big ? 4 : m;
    if (!j.grouped && (err = hipStreamWaitEvent(q, e->ev_fork, 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
    if (!j.grouped && j.part != 2) vbm_debug_delay_point(j.big ? VBM_DP_JOB_BIG : VBM_DP_JOB_SMALL, q);
    vbm_batch full;
    configure(e, full, m, j.bound, j.pcm, w);
    vbm_batch v = slice_of(full, j.lane0, j.bound);
    v.pcm = j.pcm;
    v.d_nsb = j.d_nsb;
    v.few = j.few;
    const int W = v.W;
    const bool pr = j.timed;
#define RUN(x) do { rc = (x); if (rc) { g_vbm_err = std::string("launch failed: ") + #x; return VBM_EHIP; } } while (0)
#define TIMED(k, qq) stage_scope scope_##k(e, pr, k, qq)
    if (j.part != 2) {
    RUN(vbm_launch_spread_flags(&v, q));
    { TIMED(0, q);
      RUN(vbm_launch_window_mdct(v.pcm, v.mdct_bm, W ? v.wflags_cb : nullptr, vbm_setup_device_ptrs(e->H)->mdct_trig[W],
                                 vbm_setup_device_ptrs(e->H)->window[W], vbm_setup_device_ptrs(e->H)->window[0], v.N,
                                 s->blocksizes[0], 1, v.ncb, 0, v.d_nsb, e->ch, q)); }
    { TIMED(1, q);
      RUN(vbm_launch_window_fft_log(v.pcm, v.logfft_bm, v.local_ampmax, W ? v.wflags_cb : nullptr,
                                    vbm_setup_device_ptrs(e->H)->fft_wa[W], vbm_setup_device_ptrs(e->H)->window[W],
                                    vbm_setup_device_ptrs(e->H)->window[0], v.N, s->blocksizes[0], v.ncb, v.d_nsb, e->ch, q)); }
    // the transforms above read the block's PCM only; from here on the carried stream state is involved: the
    // batches this batch's streams were last part of come first (those on this very HIP stream already do)
    if (!j.grouped)
    for (int ww = 0; ww < e->nws; ww++)
        for (int t = 0; t < 4; t++)
            if (((j.depmask >> (ww * 4 + t)) & 1u) && e->slot_queue[ww][t] != qid &&
                (err = hipStreamWaitEvent(q, e->ev_state[ww][t], 0)) != hipSuccess)
                return vbm_set_hip_error(err, "hipStreamWaitEvent");
    if (!j.grouped) vbm_debug_delay_point(VBM_DP_JOB_STATE, q);
    { TIMED(3, q); RUN(vbm_launch_prologue(&v, q)); }
    // the long-block batch of a round: tone mask (log spectrum + the prologue's maxima) beside the noise mask (MDCT) on a
    // second stream, as in the two-stream form (vbm_analysis_batch2); under a stream capture the fork and the join become
    // edges of the graph.  One such batch is enqueued at a time (rounds are built one after the other): aux[0] is its own.
    static const int round_branches = getenv("VBM_ROUND_BRANCHES") ? atoi(getenv("VBM_ROUND_BRANCHES")) : 0;   // measured: from PCM 5.04 ms per step with, 4.58 without (the two LDS-heavy kernels side by side crowd out the back half of the round before) — off
    if (round_branches && e->overlap_branches && !e->aux.empty() && m == 3 && !j.few && (j.big || (j.grouped && j.part == 1))) {
        hipStream_t qa = e->aux[0];
        if ((err = hipEventRecord(e->ev_aux_fork[0], q)) != hipSuccess || (err = hipStreamWaitEvent(qa, e->ev_aux_fork[0], 0)) != hipSuccess)
            return vbm_set_hip_error(err, "tone-mask branch fork");
        { TIMED(5, qa); RUN(vbm_launch_tonemask(&v, s->psy[v.block_mode].total_octave_lines, qa)); }
        if ((err = hipEventRecord(e->ev_aux_join[0], qa)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        { TIMED(4, q); RUN(vbm_launch_noisemask(&v, q)); }
        if ((err = hipStreamWaitEvent(q, e->ev_aux_join[0], 0)) != hipSuccess) return vbm_set_hip_error(err, "tone-mask branch join");
    } else {
    { TIMED(4, q); RUN(vbm_launch_noisemask(&v, q)); }
    { TIMED(5, q); RUN(vbm_launch_tonemask(&v, s->psy[v.block_mode].total_octave_lines, q)); }
    }
    { TIMED(6, q);
      if (s->managed) RUN(managed_front(v, q));
      else { RUN(vbm_launch_mix(&v, q)); RUN(vbm_launch_block_state(&v, q)); } }
    }   // part != 2
    if (j.part == 1) return VBM_OK;
    if (!j.grouped) {
    if ((err = hipEventRecord(e->ev_state[w][m], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
    if (j.big) {   // hand over to the back-half stream of big batches
        if ((err = hipStreamWaitEvent(e->sub[5], e->ev_state[w][m], 0)) != hipSuccess)
            return vbm_set_hip_error(err, "big batch hand-over");
        q = e->sub[5];
    }
    vbm_debug_delay_point(VBM_DP_JOB_BACK, q);
    }
    if (s->managed) {
        TIMED(10, q);
        // a managed stream's reservoirs move in the bitrate manager's choice at the end of the back half: that step (not
        // the front half, not the blobs' passes) waits for the DONE events of the batches this one's streams were in
        hipStream_t qb = q;
        RUN(managed_back(v, j.d_packets, q, [&]() -> int {
            if (j.grouped) return 0;
            for (int ww = 0; ww < e->nws; ww++)
                for (int t = 0; t < 4; t++)
                    if (((j.depmask >> (ww * 4 + t)) & 1u) && hipStreamWaitEvent(qb, e->ev_done[ww][t], 0) != hipSuccess) return -2;
            return 0;
        }));
    } else {
        { TIMED(7, q); RUN(vbm_launch_floor_fit(&v, q)); }
        { TIMED(8, q); RUN(vbm_launch_floor_encode(&v, q)); }
        { TIMED(9, q); RUN(vbm_launch_couple_quantize(&v, q)); }
        { TIMED(10, q); RUN(vbm_launch_pack(&v, q)); }
    }
    if (j.grouped) return VBM_OK;     // the group's caller copies the outputs and records the events
    vbm_debug_delay_point(VBM_DP_JOB_OUT, q);
    { TIMED(11, q);
      if (v.pack_fused) RUN(vbm_launch_packets_out(&v, j.d_packets, j.d_packet_bytes, q));
      else {
      if (j.d_packets && !s->managed)
          RUN(vbm_launch_untranspose_counted((const int *)v.packetT, (int *)j.d_packets, e->max_packet_bytes / 4,
                                             (size_t)(e->max_packet_bytes / 4) * 64, v.nsb, v.d_nsb, q));
      if (j.d_packet_bytes) RUN(vbm_launch_copy_counted(j.d_packet_bytes, v.packet_bytes, v.nsb, v.d_nsb, q)); } }
#undef TIMED
    if ((err = hipEventRecord(e->ev_done[w][m], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
    return VBM_OK;
}

// outputs of one block type of workspace w to the caller's buffers (the tail of enqueue_job, for grouped jobs)
static int copy_outputs(vbm_encoder *e, const type_job &j, hipStream_t q)
{
    int rc = 0;
    vbm_batch full;
    configure(e, full, j.m, j.bound, j.pcm, j.w);
    vbm_batch v = slice_of(full, j.lane0, j.bound);
    v.d_nsb = j.d_nsb;
    if (e->hs->managed) return VBM_OK;   // (managed_back delivers the chosen packets itself)
    if (v.pack_fused) { RUN(vbm_launch_packets_out(&v, j.d_packets, j.d_packet_bytes, q)); return VBM_OK; }
    if (j.d_packets)
        RUN(vbm_launch_untranspose_counted((const int *)v.packetT, (int *)j.d_packets, e->max_packet_bytes / 4,
                                           (size_t)(e->max_packet_bytes / 4) * 64, v.nsb, v.d_nsb, q));
    if (j.d_packet_bytes) RUN(vbm_launch_copy_counted(j.d_packet_bytes, v.packet_bytes, v.nsb, v.d_nsb, q));
    return VBM_OK;
}
#undef RUN

// One round of blocks of all four block types: counts[m] blocks of type m, described by stream_ids /
// wflags grouped by type (type 0 first); d_pcm: the blocks of type m start at float offset
// (blocks of lower types) * channels * blocksizes[1] and lie block-major [count][channels][N_m].
// The four batches are independent (different streams), so each runs on its own internal HIP stream
// on a tile-aligned slice of the workspace; a handful of short blocks then costs the round no more
// than the big long-block batch it runs beside.  Outputs are compact: packet k of the grouped order.
extern "C" int vbm_analysis_round_join(vbm_encoder *e, void *stream);

// `defer`: the batches are not joined back to `stream`; vbm_analysis_round_join does that.  A deferred round only
// waits for the batches of the previous round that its own streams were part of (a stream's blocks stay in
// order), so the few short blocks of one round can run beside the long-block batch of the round before.
static int analysis_round_impl(vbm_encoder *e, const int *counts, const int *stream_ids, const uint8_t *wflags,
                               const float *d_pcm, uint8_t *d_packets, int *d_packet_bytes, void *stream, bool defer)
{
    if (!e || !counts || !stream_ids || !wflags || !d_pcm) return VBM_EINVAL;
    if (((uintptr_t)d_pcm & 15) || ((uintptr_t)d_packets & 3)) return VBM_EINVAL;
    const vbm_setup *s = e->hs;
    int total = 0, off[4], pad[4], lanes = 0;
    for (int m = 0; m < 4; m++) {
        if (counts[m] < 0) return VBM_EINVAL;
        off[m] = total;
        pad[m] = lanes;
        total += counts[m];
        lanes = (lanes + counts[m] + 63) & ~63;
        if (counts[m] && s->modes < 2 && (m >> 1)) return VBM_EINVAL;
    }
    if (total == 0) return VBM_OK;
    if (total > e->cap || lanes > e->Ls) return VBM_EINVAL;
    e->seen.assign(e->S, 0);
    for (int i = 0; i < total; i++) {
        if (stream_ids[i] < 0 || stream_ids[i] >= e->S) return VBM_EINVAL;
        if (e->seen[stream_ids[i]]) { g_vbm_err = "a stream id appears twice in one round"; return VBM_EINVAL; }
        e->seen[stream_ids[i]] = 1;
    }
    int rc = vbm_encoder_set_sub_batches(e, e->nsplit);   // make sure the internal streams exist
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err;
    const int w = e->next;
    e->next = (e->next + 1) % e->nws;
    e->cur = w;
    if (e->back_pending[w]) {
        if ((err = hipStreamWaitEvent(st, e->ev_back[w], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        e->back_pending[w] = false;
    }
    // the round that last used this workspace has to be done before `st` (and everything forked from it) goes on
    for (int m = 0; m < 4; m++)
        if (e->reuse_pending[w][m]) {
            if ((err = hipStreamWaitEvent(st, e->ev_done[w][m], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            e->reuse_pending[w][m] = false;
        }
    if (e->device_rounds) {   // rounds built on the device ran in between: their streams are unknown here
        for (int ww = 0; ww < e->nws; ww++)
            for (int t = 0; t < 4; t++)
                if (e->reuse_pending[ww][t]) {
                    if ((err = hipStreamWaitEvent(st, e->ev_done[ww][t], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
                    e->reuse_pending[ww][t] = false;
                }
        e->device_rounds = false;
    }
    // Which earlier batches do the streams of each batch of this round come from?  bit (ww * 4 + t) of dep[m]:
    // slot (ww, t) still holds the batch (same epoch) that some stream of batch m was last part of.
    unsigned dep[4] = {0, 0, 0, 0};
    for (int m = 0; m < 4; m++)
        for (int i = 0; i < counts[m]; i++) {
            const vbm_encoder::last_batch &lb = e->last[stream_ids[off[m] + i]];
            if (lb.w >= 0 && e->epoch[(int)lb.w][(int)lb.m] == lb.epoch) dep[m] |= 1u << (lb.w * 4 + lb.m);
        }
    if (getenv("VBM_ROUND_DEBUG"))
        fprintf(stderr, "round w=%d defer=%d counts=[%d %d %d %d] dep masks: %04x %04x %04x %04x\n", w, (int)defer, counts[0], counts[1],
                counts[2], counts[3], dep[0], dep[1], dep[2], dep[3]);
    // ids / flags in the padded lane layout, through the pinned staging (always uploaded: the layout changes
    // from round to round)
    {
        const int t = e->stage_turn;
        e->stage_turn ^= 1;
        (void)hipEventSynchronize(e->ev_stage[t]);
        memset(e->h_ids[t], 0, lanes * sizeof(int));
        memset(e->h_flags[t], 0, lanes);
        for (int m = 0; m < 4; m++) {
            memcpy(e->h_ids[t] + pad[m], stream_ids + off[m], counts[m] * sizeof(int));
            memcpy(e->h_flags[t] + pad[m], wflags + off[m], counts[m]);
        }
        if ((err = hipMemcpyAsync(e->d_stream_id[w], e->h_ids[t], lanes * sizeof(int), hipMemcpyHostToDevice, st)) != hipSuccess ||
            (err = hipMemcpyAsync(e->d_wflags[w], e->h_flags[t], lanes, hipMemcpyHostToDevice, st)) != hipSuccess)
            return vbm_set_hip_error(err, "hipMemcpyAsync(round ids)");
        (void)hipEventRecord(e->ev_stage[t], st);
        e->last_ids[w].clear();
        e->last_flags[w].clear();
    }
    if ((err = hipEventRecord(e->ev_fork, st)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
    e->last_nsb = 0;
#define RUN(x) do { rc = (x); if (rc) { g_vbm_err = std::string("launch failed: ") + #x; return VBM_EHIP; } } while (0)
    // Every block type gets its own internal stream; the largest batch is enqueued first so that its
    // launches are in flight while the host issues the ~35 launches of each small batch.  Measured
    // alternatives on MI355X: big batch on the caller's stream + one side stream, and two
    // streams with disjoint CU masks (hipExtStreamCreateWithCUMask) — both slower: the single-wavefront
    // kernels of the small batches run several times slower beside the wide kernels of the big batch, so
    // their chains are better run next to each other than one after another.
    while ((int)e->sub.size() < 6) {      // one per block type + two for a big batch (front / back half)
        hipStream_t q;
        hipEvent_t ev;
        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return VBM_EHIP;
        e->sub.push_back(q);
        e->ev_join.push_back(ev);
    }
    int order[4] = {0, 1, 2, 3};
    for (int a = 0; a < 4; a++)
        for (int c = a + 1; c < 4; c++)
            if (counts[order[c]] > counts[order[a]]) { int t_ = order[a]; order[a] = order[c]; order[c] = t_; }
    // The ~40 launches of a block type's pipeline cost the host more than the kernels of a small batch cost the
    // device, and the four types are independent: every type is enqueued by its own host thread (the largest
    // batch on the calling thread), each on its own HIP stream.
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    const bool prof = e->profiling && e->prof_calls < e->prof_max_calls && e->events_used + 2 * (size_t)kNumStages <= e->events.size();
    auto enqueue_type = [&](const int m) -> int {
        type_job j;
        j.m = m; j.w = w; j.lane0 = pad[m]; j.bound = counts[m]; j.d_nsb = nullptr;
        j.pcm = d_pcm + (size_t)off[m] * e->ch * s->blocksizes[1];
        j.d_packets = d_packets ? d_packets + (size_t)off[m] * e->max_packet_bytes : nullptr;
        j.d_packet_bytes = d_packet_bytes ? d_packet_bytes + off[m] : nullptr;
        j.big = counts[m] >= kBigBatch;          // a big batch gets streams of its own (front / back half)
        j.timed = prof && m == order[0];         // stage timing covers the round's largest batch (calling thread)
        j.depmask = dep[m];
        return enqueue_job(e, j);
    };
    int rcs[4] = {0, 0, 0, 0};
    std::string msgs[4];
    std::thread workers[4];
    const bool threaded = !getenv("VBM_ROUND_SINGLE_THREAD");
    for (int rank = 3; rank >= 0; rank--) {      // smallest first: their threads start while the big one is enqueued here
        const int m = order[rank];
        if (!counts[m]) continue;
        e->done_pending[w][m] = true;
        e->reuse_pending[w][m] = true;
        e->slot_queue[w][m] = (signed char)(counts[m] >= kBigBatch ? 4 : m);
        if (rank == 0 || !threaded) {
            rcs[m] = enqueue_type(m);
            if (rcs[m]) msgs[m] = g_vbm_err;
        } else {
            workers[m] = std::thread([&, m]() {
                (void)hipSetDevice(dev_id);
                rcs[m] = enqueue_type(m);
                if (rcs[m]) msgs[m] = g_vbm_err;
            });
        }
    }
    for (int m = 0; m < 4; m++)
        if (workers[m].joinable()) workers[m].join();
    for (int m = 0; m < 4; m++)
        if (rcs[m]) { g_vbm_err = msgs[m]; return rcs[m]; }
    if (prof) {
        e->prof_calls++;
        e->prof_blocks += counts[order[0]];
    }
#undef RUN
    for (int m = 0; m < 4; m++)
        if (counts[m] >= kBigBatch) { e->lazy_w = w; e->lazy_m = m; }
    // remember which batch every stream of this round belongs to
    for (int m = 0; m < 4; m++) {
        if (!counts[m]) continue;
        const unsigned ep = ++e->epoch[w][m];
        for (int i = 0; i < counts[m]; i++) e->last[stream_ids[off[m] + i]] = {(signed char)w, (signed char)m, ep};
    }
    e->round_w = w;
    return defer ? VBM_OK : vbm_analysis_round_join(e, stream);
}

// ---- rounds built on the device (capi_frontend.cpp: vbm_frontend_encode_rounds_device) ------------------------------
// The host knows neither which streams deliver a block nor how many: block type m owns the fixed lane region
// [lane0[m], lane0[m] + cap[m]) of workspace w, its kernels are launched for cap[m] blocks and read the count from
// d_count[m].  Order of a stream's blocks: every batch begun earlier may hold the block before one of this round's
// and is waited for (its state event; a finished batch costs nothing) wherever the block types allow the succession.
int vbm_encoder_device_round_open(vbm_encoder *e, hipStream_t fork, int *w_out, int **d_stream_id, uint8_t **d_wflags, int *lanes,
                                  int **d_counts)
{
    int rc = vbm_encoder_set_sub_batches(e, e->nsplit);
    if (rc) return rc;
    if (!e->small_streams_set) {
        // Streams of the device-built rounds.  sub[0..3]: the small batches, chains of ~40 short kernels; beside the
        // big batch every one of them queues behind its wide launches, so they get high priority (their workgroups
        // are few and go first: the chain's latency is what the short-block runs of a stream wait for;
        // VBM_SMALL_PRIORITY=0: plain streams).  sub[4], sub[5]: front and back half of the big batch.  The runtime
        // hands out a limited number of hardware queues per priority (GPU_MAX_HW_QUEUES) in the order the streams
        // are made and lets later streams share: the two big-batch streams are made before the others, so that
        // they do not end up on one queue (the back half of a call has to run beside the front half of the next).
        e->small_streams_set = true;
        const char *env = getenv("VBM_SMALL_PRIORITY");
        int lo = 0, hi = 0;
        const bool prio = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo && (!env || atoi(env));
        while ((int)e->sub.size() < 6) {
            e->sub.push_back(nullptr);
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return VBM_EHIP;
            e->ev_join.push_back(ev);
        }
        // VBM_SMALL_STREAMS=2: types 0 / 1 share one stream, types 2 / 3 another (fewer hardware queues in play)
        const char *ss = getenv("VBM_SMALL_STREAMS");
        e->small_share = ss && atoi(ss) == 2;
        static const int order[6] = {4, 5, 0, 1, 2, 3};
        for (int k = 0; k < 6; k++) {
            const int m = order[k];
            const bool want_prio = prio && m < 4;
            if (e->sub[m] && !want_prio) continue;
            hipStream_t q;
            hipError_t cerr = want_prio ? hipStreamCreateWithPriority(&q, hipStreamNonBlocking, hi)
                                        : hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
            if (cerr != hipSuccess) return vbm_set_hip_error(cerr, "hipStreamCreate");
            if (e->sub[m]) {
                (void)hipStreamSynchronize(e->sub[m]);
                (void)hipStreamDestroy(e->sub[m]);
            }
            e->sub[m] = q;
        }
        if (e->small_share) {
            (void)hipStreamDestroy(e->sub[1]);
            (void)hipStreamDestroy(e->sub[3]);
            e->sub[1] = e->sub[0];
            e->sub[3] = e->sub[2];
        }
    }
    const int w = e->next;
    e->next = (e->next + 1) % e->nws;
    e->cur = w;
    hipError_t err;
    if (e->back_pending[w]) {
        if ((err = hipStreamWaitEvent(fork, e->ev_back[w], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        e->back_pending[w] = false;
    }
    for (int m = 0; m < 4; m++)
        if (e->reuse_pending[w][m]) {
            if ((err = hipStreamWaitEvent(fork, e->ev_done[w][m], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            e->reuse_pending[w][m] = false;
        }
    e->last_ids[w].clear();
    e->last_flags[w].clear();
    if (!e->d_counts_ws) {
        if (hipMalloc((void **)&e->d_counts_ws, kMaxWS * 4 * sizeof(int)) != hipSuccess) return VBM_EHIP;
        e->allocs.push_back(e->d_counts_ws);
        e->alloc_bytes.push_back(kMaxWS * 4 * sizeof(int));
        // (null-stream fill: the front end's non-blocking stream would not wait for it)
        if (hipMemset(e->d_counts_ws, 0, kMaxWS * 4 * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return VBM_EHIP;
    }
    *w_out = w;
    *d_stream_id = e->d_stream_id[w];
    *d_wflags = e->d_wflags[w];
    *lanes = e->Ls;
    *d_counts = e->d_counts_ws + 4 * w;
    return VBM_OK;
}

// A group of a round as a graph: `jobs` (grouped type_jobs) forked from and joined to `origin`; several jobs run side
// by side on sub[job.m].  First use: plain launches (the launchers' one-time set-up runs then); second use: the
// same calls under a stream capture, instantiated; from then on one hipGraphLaunch.
static int run_group(vbm_encoder *e, vbm_encoder::round_graph &g, hipStream_t origin, type_job *jobs, int njobs, int kind = 0)
{
    hipError_t err;
    static int mask = -1;
    if (mask < 0) mask = getenv("VBM_GRAPH_MASK") ? atoi(getenv("VBM_GRAPH_MASK")) : 7;   // debugging: bit 0 small groups, 1 big front, 2 big back
    if (g.exec) {
        if ((err = hipGraphLaunch(g.exec, origin)) != hipSuccess) return vbm_set_hip_error(err, "hipGraphLaunch");
        return VBM_OK;
    }
    const bool capture = e->use_graphs == 1 && g.uses >= 1 && ((mask >> kind) & 1);   // (use_graphs 2: the groups as plain launches, for A/B)
    g.uses++;
    if (capture && (err = hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal)) != hipSuccess)
        return vbm_set_hip_error(err, "hipStreamBeginCapture");
    int rc = VBM_OK;
    if (njobs == 1) {
        jobs[0].q_on = origin;
        rc = enqueue_job(e, jobs[0]);
    } else {
        if ((err = hipEventRecord(e->ev_cap_fork, origin)) != hipSuccess) rc = vbm_set_hip_error(err, "hipEventRecord");
        for (int k = 0; k < njobs && !rc; k++) {
            hipStream_t q = e->sub[jobs[k].m];
            jobs[k].q_on = q;
            if ((err = hipStreamWaitEvent(q, e->ev_cap_fork, 0)) != hipSuccess) { rc = vbm_set_hip_error(err, "hipStreamWaitEvent"); break; }
            rc = enqueue_job(e, jobs[k]);
            if (rc) break;
            if ((err = hipEventRecord(e->ev_cap_join[jobs[k].m], q)) != hipSuccess ||
                (err = hipStreamWaitEvent(origin, e->ev_cap_join[jobs[k].m], 0)) != hipSuccess)
                rc = vbm_set_hip_error(err, "group join");
        }
    }
    if (capture) {
        hipGraph_t graph = nullptr;
        err = hipStreamEndCapture(origin, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamEndCapture");
        err = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (err != hipSuccess) { g.exec = nullptr; return vbm_set_hip_error(err, "hipGraphInstantiate"); }
        if ((err = hipGraphLaunch(g.exec, origin)) != hipSuccess) return vbm_set_hip_error(err, "hipGraphLaunch");
    }
    return rc;
}

// Device-built round with every job's launches replayed as a HIP graph.  Streams as in the plain form: sub[m] takes the
// small batches of block type m of all rounds, in order; sub[4] the front halves of the big (first round, long
// blocks) batches, sub[5] their back halves.  What a job waits for besides its own stream is decided by the block
// types alone (lib/block.c:620-638: a long block with both neighbours long is type 3, with a short neighbour type 2;
// short blocks are 0 / 1): a stream's consecutive blocks go 3->3, 3->2, 2->3, 2->2, 2->0/1, 0/1->0/1, 0/1->2, so
// types 0 / 1 never follow type 3 directly and type 3 never follows them: short batches and long batches only meet
// through the transition batches.  A job waits for the newest state event of every type it may follow, as they stood
// before the round's own jobs went in; for the long types that includes the front half of the newest big batch (the
// call's own in its later rounds: a stream that has fallen behind delivers a long block in every round).
extern "C" void vbm_debug_stamp(hipStream_t st, int tag);   // util_kernels.hip (timing experiments)
static int device_round_run_graphs(vbm_encoder *e, int w, const int *lane0, const int *cap, const float *d_blocks,
                                   uint8_t *d_packets, int *d_packet_bytes, bool first_round, hipStream_t fork)
{
    const vbm_setup *s = e->hs;
    hipError_t err;
    const int *d_count = e->d_counts_ws + 4 * w;
    if ((err = hipEventRecord(e->ev_fork, fork)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
    e->last_nsb = 0;
    const bool has_big = first_round && cap[3] > 0;
    auto job_of = [&](int m, int part) {
        type_job j;
        j.m = m; j.w = w; j.lane0 = lane0[m]; j.bound = cap[m]; j.d_nsb = d_count + m;
        j.pcm = d_blocks + (size_t)lane0[m] * e->ch * s->blocksizes[1];
        j.d_packets = d_packets ? d_packets + (size_t)lane0[m] * e->max_packet_bytes : nullptr;
        j.d_packet_bytes = d_packet_bytes ? d_packet_bytes + lane0[m] : nullptr;
        j.big = false; j.timed = false; j.depmask = 0;
        j.part = part; j.grouped = true;
        return j;
    };
    // newest state event of every queue before this round: (workspace, type) of the last job on sub[0..3] and of
    // the newest big batch (first round: the previous call's; later rounds: this call's)
    int last_w[4];
    const int prev_big_w = e->call_big_w;
    for (int t = 0; t < 4; t++) last_w[t] = e->queue_last_w[t];
    static const unsigned follows[4] = {0x7u, 0x7u, 0xfu, 0xcu};   // bit t: a block of this type may follow one of type t
    auto wait_preds = [&](hipStream_t q, int m, int own_queue) -> int {
        for (int t = 0; t < 4; t++) {
            if (!((follows[m] >> t) & 1u)) continue;
            if (t != own_queue && last_w[t] >= 0 &&
                (err = hipStreamWaitEvent(q, e->ev_state[last_w[t]][t], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            if (t == 3 && own_queue != 4 && prev_big_w >= 0 &&
                (err = hipStreamWaitEvent(q, e->ev_state_big[prev_big_w], 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        }
        return VBM_OK;
    };
    int rc;
    if (has_big) {      // the long pole first
        hipStream_t qF = e->sub[4], qB = e->sub[5];
        if ((err = hipStreamWaitEvent(qF, e->ev_fork, 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        if ((rc = wait_preds(qF, 3, 4))) return rc;
        type_job jf = job_of(3, 1), jb = job_of(3, 2);
        vbm_debug_stamp(qF, 10);
        vbm_debug_delay_point(VBM_DP_DEV_BIG_FRONT, qF);
        if ((rc = run_group(e, e->gJ[w][4][1], qF, &jf, 1, 1))) return rc;
        vbm_debug_stamp(qF, 11);
        if ((err = hipEventRecord(e->ev_state_big[w], qF)) != hipSuccess ||
            (err = hipStreamWaitEvent(qB, e->ev_state_big[w], 0)) != hipSuccess) return vbm_set_hip_error(err, "big batch hand-over");
        vbm_debug_stamp(qB, 12);
        vbm_debug_delay_point(VBM_DP_DEV_BIG_BACK, qB);
        if ((rc = run_group(e, e->gJ[w][4][2], qB, &jb, 1, 2))) return rc;
        vbm_debug_delay_point(VBM_DP_DEV_OUT, qB);
        if ((rc = copy_outputs(e, jb, qB))) return rc;
        vbm_debug_stamp(qB, 13);
        if ((err = hipEventRecord(e->ev_done[w][3], qB)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        e->done_pending[w][3] = true;
        e->reuse_pending[w][3] = true;
        e->slot_queue[w][3] = 4;
        e->epoch[w][3]++;
    }
    for (int m = 0; m < 4; m++) {
        if (!cap[m] || (m == 3 && has_big)) continue;
        hipStream_t q = e->sub[m];
        if ((err = hipStreamWaitEvent(q, e->ev_fork, 0)) != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
        if ((rc = wait_preds(q, m, m))) return rc;
        // two graphs, the state event between them: what follows this batch on another stream waits for its front
        // half only (the chain short blocks -> transition -> long blocks of the next call is what the big batch of
        // the next call waits for)
        type_job j = job_of(m, 1), j2 = job_of(m, 2);
        j.few = j2.few = 1;
        vbm_debug_stamp(q, 20 + m);
        vbm_debug_delay_point(VBM_DP_DEV_SMALL_FRONT, q);
        if ((rc = run_group(e, e->gJ[w][m][1], q, &j, 1, 0))) return rc;
        if ((err = hipEventRecord(e->ev_state[w][m], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        vbm_debug_delay_point(VBM_DP_DEV_SMALL_BACK, q);
        if ((rc = run_group(e, e->gJ[w][m][2], q, &j2, 1, 0))) return rc;
        vbm_debug_delay_point(VBM_DP_DEV_OUT, q);
        if ((rc = copy_outputs(e, j2, q))) return rc;
        vbm_debug_stamp(q, 30 + m);
        if ((err = hipEventRecord(e->ev_done[w][m], q)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
        e->done_pending[w][m] = true;
        e->reuse_pending[w][m] = true;
        e->slot_queue[w][m] = (signed char)m;
        e->epoch[w][m]++;
        e->queue_last_w[m] = w;
    }
    if (has_big) { e->call_big_w = w; e->lazy_w = w; e->lazy_m = 3; }
    e->round_w = w;
    e->device_rounds = true;
    return VBM_OK;
}

int vbm_encoder_device_round_run(vbm_encoder *e, int w, const int *lane0, const int *cap, const int *d_count,
                                 const float *d_blocks, uint8_t *d_packets, int *d_packet_bytes, bool first_round,
                                 hipStream_t fork)
{
    const vbm_setup *s = e->hs;
    hipError_t err;
    for (int m = 0; m < 4; m++)
        if (cap[m] < 0 || (lane0[m] & 63) || lane0[m] + cap[m] > e->Ls || (cap[m] && s->modes < 2 && (m >> 1))) return VBM_EINVAL;
    if (d_count != e->d_counts_ws + 4 * w) return VBM_EINVAL;
    if (e->use_graphs < 0) {
        const char *env = getenv("VBM_DEVICE_GRAPHS");     // 0: plain launches, 1 (default): HIP graphs, 2: groups without capture
        e->use_graphs = env ? atoi(env) : 1;
    }
    // Stage timing needs the plain launches (events between the kernels); a managed-bitrate back half writes the
    // chosen packets straight to the caller's buffers, which change from call to call: plain launches as well.
    const int mode = (e->use_graphs && !e->profiling && !s->managed) ? 1 : 2;
    if (e->device_mode && e->device_mode != mode) {
        // the two ways use different internal streams: everything in flight is waited for at the switch
        for (int ww = 0; ww < e->nws; ww++)
            for (int t = 0; t < 4; t++)
                if (e->reuse_pending[ww][t] && (err = hipStreamWaitEvent(fork, e->ev_done[ww][t], 0)) != hipSuccess)
                    return vbm_set_hip_error(err, "hipStreamWaitEvent");
    }
    e->device_mode = mode;
    if (mode == 1) return device_round_run_graphs(e, w, lane0, cap, d_blocks, d_packets, d_packet_bytes, first_round, fork);
    if ((err = hipEventRecord(e->ev_fork, fork)) != hipSuccess) return vbm_set_hip_error(err, "hipEventRecord");
    e->last_nsb = 0;
    // every slot that may still be running, except this round's own
    unsigned depmask = 0;
    for (int ww = 0; ww < e->nws; ww++)
        for (int t = 0; t < 4; t++)
            if (ww != w && e->reuse_pending[ww][t]) depmask |= 1u << (ww * 4 + t);
    const bool prof = e->profiling && e->prof_calls < e->prof_max_calls && e->events_used + 2 * (size_t)kNumStages <= e->events.size();
    int dev_id = 0;
    (void)hipGetDevice(&dev_id);
    int rcs[4] = {0, 0, 0, 0};
    std::string msgs[4];
    std::thread workers[4];
    const bool threaded = !getenv("VBM_ROUND_SINGLE_THREAD");
    const int bigm = first_round ? 3 : -1;
    auto job_of = [&](int m) {
        type_job j;
        j.m = m; j.w = w; j.lane0 = lane0[m]; j.bound = cap[m]; j.d_nsb = d_count + m;
        j.pcm = d_blocks + (size_t)lane0[m] * e->ch * s->blocksizes[1];
        j.d_packets = d_packets ? d_packets + (size_t)lane0[m] * e->max_packet_bytes : nullptr;
        j.d_packet_bytes = d_packet_bytes ? d_packet_bytes + lane0[m] : nullptr;
        j.big = m == bigm;
        j.timed = prof && m == 3 && first_round;
        // (short blocks never follow a type-3 block directly: no wait for the newest big batch)
        j.depmask = (m < 2 && e->call_big_w >= 0) ? depmask & ~(1u << (e->call_big_w * 4 + 3)) : depmask;
        return j;
    };
    for (int m = 0; m < 4; m++) {
        if (!cap[m] || m == 3) continue;
        e->done_pending[w][m] = true;
        e->reuse_pending[w][m] = true;
        e->slot_queue[w][m] = (signed char)m;
        e->epoch[w][m]++;
        if (!threaded) {
            rcs[m] = enqueue_job(e, job_of(m));
            if (rcs[m]) msgs[m] = g_vbm_err;
        } else {
            workers[m] = std::thread([&, m]() {
                (void)hipSetDevice(dev_id);
                rcs[m] = enqueue_job(e, job_of(m));
                if (rcs[m]) msgs[m] = g_vbm_err;
            });
        }
    }
    if (cap[3]) {      // the long blocks on the calling thread
        e->done_pending[w][3] = true;
        e->reuse_pending[w][3] = true;
        e->slot_queue[w][3] = (signed char)(bigm == 3 ? 4 : 3);
        e->epoch[w][3]++;
        rcs[3] = enqueue_job(e, job_of(3));
        if (rcs[3]) msgs[3] = g_vbm_err;
    }
    for (int m = 0; m < 4; m++)
        if (workers[m].joinable()) workers[m].join();
    for (int m = 0; m < 4; m++)
        if (rcs[m]) { g_vbm_err = msgs[m]; return rcs[m]; }
    if (prof && first_round) e->prof_calls++;
    if (first_round) { e->call_big_w = w; e->lazy_w = w; e->lazy_m = 3; }
    e->round_w = w;
    // (per-stream bookkeeping of the host-built rounds does not apply: forget it, so that a later host-built round
    // waits for everything instead of trusting stale entries)
    e->device_rounds = true;
    return VBM_OK;
}

extern "C" int vbm_analysis_round(vbm_encoder *e, const int *counts, const int *stream_ids, const uint8_t *wflags,
                                  const float *d_pcm, uint8_t *d_packets, int *d_packet_bytes, void *stream)
{
    return analysis_round_impl(e, counts, stream_ids, wflags, d_pcm, d_packets, d_packet_bytes, stream, false);
}

extern "C" int vbm_analysis_round_begin(vbm_encoder *e, const int *counts, const int *stream_ids, const uint8_t *wflags,
                                        const float *d_pcm, uint8_t *d_packets, int *d_packet_bytes, void *stream)
{
    return analysis_round_impl(e, counts, stream_ids, wflags, d_pcm, d_packets, d_packet_bytes, stream, true);
}

// `stream` waits for every batch of the rounds begun so far
extern "C" int vbm_analysis_round_join(vbm_encoder *e, void *stream)
{
    if (!e) return VBM_EINVAL;
    for (int w = 0; w < e->nws; w++)
        for (int m = 0; m < 4; m++)
            if (e->done_pending[w][m]) {
                hipError_t err = hipStreamWaitEvent((hipStream_t)stream, e->ev_done[w][m], 0);
                if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
                e->done_pending[w][m] = false;
            }
    return VBM_OK;
}

// like vbm_analysis_round_join, but the newest big batch and the batches of the newest round stay pending: a later
// round's batches wait for the batches their own streams were in (analysis_round_impl), nothing else needs them
// yet, so the back half of one write's big batch runs beside the front half of the next.  Their outputs are
// complete on `stream` after the next lazy join (or a join).
extern "C" int vbm_analysis_round_join_lazy(vbm_encoder *e, void *stream)
{
    if (!e) return VBM_EINVAL;
    for (int w = 0; w < e->nws; w++)
        for (int m = 0; m < 4; m++)
            if (e->done_pending[w][m] && !(w == e->lazy_w && m == e->lazy_m) && w != e->round_w) {
                hipError_t err = hipStreamWaitEvent((hipStream_t)stream, e->ev_done[w][m], 0);
                if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
                e->done_pending[w][m] = false;
            }
    return VBM_OK;
}

// `stream` waits until the workspace the next round will use is free (buffers handed to that round may then
// be rewritten on `stream`)
extern "C" int vbm_analysis_round_wait_workspace(vbm_encoder *e, void *stream)
{
    if (!e) return VBM_EINVAL;
    const int w = e->next;
    for (int m = 0; m < 4; m++)
        if (e->reuse_pending[w][m]) {
            hipError_t err = hipStreamWaitEvent((hipStream_t)stream, e->ev_done[w][m], 0);
            if (err != hipSuccess) return vbm_set_hip_error(err, "hipStreamWaitEvent");
            e->reuse_pending[w][m] = false;
        }
    return VBM_OK;
}

// Test instrumentation: every scratch array of workspace w (-1: all) is filled with `byte` while the device is idle.
// A batch writes what it reads, so its packets do not depend on what the workspace held before (tests/test_ordering_gpu.py).
extern "C" int vbm_debug_poison_workspace(vbm_encoder *e, int w, int byte)
{
    if (!e || w >= e->nws) return VBM_EINVAL;
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) return vbm_set_hip_error(err, "hipDeviceSynchronize");
    for (int ww = 0; ww < e->nws; ww++) {
        if (w >= 0 && ww != w) continue;
        for (size_t k = e->ws_alloc[ww]; k < e->ws_alloc[ww + 1]; k++)
            if ((err = hipMemset(e->allocs[k], byte, e->alloc_bytes[k])) != hipSuccess) return vbm_set_hip_error(err, "hipMemset(poison)");
        e->last_ids[ww].clear();        // (the id / flag lists of the workspace are gone too: upload them again)
        e->last_flags[ww].clear();
    }
    err = hipDeviceSynchronize();
    return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipDeviceSynchronize");
}

extern "C" int vbm_encoder_profile_begin(vbm_encoder *e, int max_calls)
{
    if (!e || max_calls <= 0) return VBM_EINVAL;
    size_t need = (size_t)max_calls * 2 * (kFront + (size_t)e->nsplit * kBack);
    while (e->events.size() < need) {
        hipEvent_t ev;
        hipError_t err = hipEventCreate(&ev);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipEventCreate");
        e->events.push_back(ev);
    }
    e->events_used = 0;
    e->spans.clear();
    e->prof_calls = 0;
    e->prof_blocks = 0;
    e->prof_max_calls = max_calls;
    e->profiling = true;
    return VBM_OK;
}

extern "C" int vbm_encoder_profile_end(vbm_encoder *e, float *stage_ms, int *ncalls)
{
    if (!e || !stage_ms) return VBM_EINVAL;
    e->profiling = false;
    for (int k = 0; k < kNumStages; k++) stage_ms[k] = 0.f;
    if (!e->spans.empty()) {
        hipError_t err = hipDeviceSynchronize();
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipDeviceSynchronize");
        for (const vbm_encoder::span &sp : e->spans) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e->events[sp.begin], e->events[sp.end]);
            stage_ms[sp.stage] += ms;
        }
    }
    if (ncalls) *ncalls = e->prof_calls;
    e->events_used = 0;
    e->spans.clear();
    e->prof_calls = 0;
    return VBM_OK;
}

// stream-blocks the profiled launches covered (a round: its largest batch), summed over the covered calls
extern "C" long long vbm_encoder_profile_blocks(const vbm_encoder *e) { return e ? e->prof_blocks : 0; }

extern "C" int vbm_encoder_stage_count(void) { return kNumStages; }
extern "C" const char *vbm_encoder_stage_name(int k) { return (k >= 0 && k < kNumStages) ? kStageNames[k] : ""; }

// Stage intermediates of the LAST batch, converted to block-major rows, for parity tests.
// managed bitrate: packetblob k of the last batch as it was before the bitrate manager chose
// (packets [nsb][max_packet_bytes], lengths [nsb])
extern "C" int vbm_encoder_fetch_blob(vbm_encoder *e, int k, uint8_t *d_packets, int *d_packet_bytes, void *stream)
{
    if (!e || e->last_nsb <= 0 || k < 0 || k >= VBM_PACKETBLOBS) return VBM_EINVAL;
    if (!e->hs->managed) { g_vbm_err = "not a managed-bitrate setup"; return VBM_EINVAL; }
    vbm_batch b;
    configure(e, b, e->last_mode, e->last_nsb, nullptr, e->cur);
    hipStream_t st = (hipStream_t)stream;
    if (d_packets &&
        vbm_launch_untranspose_i32((const int *)(b.packetT_blob + (size_t)k * b.Ls * b.max_packet_bytes), (int *)d_packets,
                                   e->max_packet_bytes / 4, (size_t)(e->max_packet_bytes / 4) * 64, b.nsb, st))
        return VBM_EHIP;
    if (d_packet_bytes) {
        hipError_t err = hipMemcpyAsync(d_packet_bytes, b.packet_bytes_blob + (size_t)k * b.Ls, (size_t)b.nsb * sizeof(int),
                                        hipMemcpyDeviceToDevice, st);
        if (err != hipSuccess) return vbm_set_hip_error(err, "hipMemcpyAsync(fetch_blob)");
    }
    return VBM_OK;
}

extern "C" int vbm_encoder_fetch(vbm_encoder *e, const char *name, void *d_out, long *rows_out, char *kind,
                                 void *stream)
{
    if (!e || !name || e->last_nsb <= 0) return VBM_EINVAL;
    vbm_batch b;
    configure(e, b, e->last_mode, e->last_nsb, nullptr, e->cur);
    hipStream_t st = (hipStream_t)stream;
    const vbm_setup *s = e->hs;
    const vbm_psy &p = s->psy[b.block_mode];
    const int partition = p.normal_p ? p.normal_partition : 16;
    struct Ent { const char *name; const void *ptr; int rows; char kind; int lanes; };
    const Ent table[] = {
        {"mdct_raw", b.mdct_bm, -1, 'f', b.ncb},        // block-major already
        {"logfft", b.logfft_bm, -1, 'f', b.ncb},
        {"mdct", b.mdctT, b.n, 'f', b.ncb},             // after M1 rescale
        {"logmdct", b.logmdctT, b.n, 'f', b.ncb},
        {"noise", b.noiseT, b.n, 'f', b.ncb},
        {"tone", b.toneT, b.n, 'f', b.ncb},
        {"logmask", b.logmaskT, b.n, 'f', b.ncb},
        {"epeak", b.epeakT, b.n, 'f', b.ncb},
        {"npeak", b.npeakT, b.n / partition, 'f', b.ncb},
        {"post", b.postT, VBM_VIF_POSIT + 2, 'i', b.ncb},
        {"floor_out", b.floor_outT, VBM_VIF_POSIT + 2, 'i', b.ncb},
        {"residue", b.iworkT, b.n, 'i', b.ncb},
    };
    if (b.pack_fused && !strcmp(name, "residue")) {   // the couple kernel wrote the coder's interleaved order (res_bm), not the tiles
        if (rows_out) *rows_out = b.n;
        if (kind) *kind = 'i';
        if (!d_out) return VBM_OK;
        return vbm_launch_res_bm_rows(&b, (int *)d_out, st) ? VBM_EHIP : VBM_OK;
    }
    for (const Ent &t : table) {
        if (strcmp(t.name, name)) continue;
        if (rows_out) *rows_out = t.rows < 0 ? b.n : t.rows;
        if (kind) *kind = t.kind;
        if (!d_out) return VBM_OK;
        if (t.rows < 0) {
            hipError_t err = hipMemcpyAsync(d_out, t.ptr, (size_t)b.ncb * b.n * 4, hipMemcpyDeviceToDevice, st);
            return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipMemcpyAsync(fetch)");
        }
        int rc = (t.kind == 'f')
                     ? vbm_launch_untranspose_f32((const float *)t.ptr, (float *)d_out, t.rows, b.slab_words, t.lanes, st)
                     : vbm_launch_untranspose_i32((const int *)t.ptr, (int *)d_out, t.rows, b.slab_words, t.lanes, st);
        return rc ? VBM_EHIP : VBM_OK;
    }
    struct Vec { const char *name; const void *ptr; char kind; int count; };
    const Vec vecs[] = {
        {"local_ampmax", b.local_ampmax, 'f', b.ncb}, {"global_ampmax", b.global_ampmax, 'f', b.nsb},
        {"post_valid", b.post_valid, 'i', b.ncb},     {"nonzero", b.nonzero, 'i', b.ncb},
        {"poste", b.poste, 'f', b.ncb},               {"packet_bytes", b.packet_bytes, 'i', b.nsb},
        {"choice", b.choice, 'i', b.choice ? b.nsb : 0},            // managed bitrate: bm->choice per stream-block
    };
    for (const Vec &t : vecs) {
        if (strcmp(t.name, name) || !t.ptr) continue;
        if (rows_out) *rows_out = 1;
        if (kind) *kind = t.kind;
        if (!d_out) return VBM_OK;
        hipError_t err = hipMemcpyAsync(d_out, t.ptr, (size_t)t.count * 4, hipMemcpyDeviceToDevice, st);
        return err == hipSuccess ? VBM_OK : vbm_set_hip_error(err, "hipMemcpyAsync(fetch)");
    }
    g_vbm_err = std::string("unknown intermediate: ") + name;
    return VBM_EINVAL;
}
