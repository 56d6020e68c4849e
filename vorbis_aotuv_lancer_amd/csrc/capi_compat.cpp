// The reference's own entry points (include/vorbis_compat.h): vorbis_analysis_init / _buffer / _wrote /
// _blockout, vorbis_analysis, vorbis_bitrate_addblock / _flushpacket and what surrounds them, on the public
// struct layouts of the reference's include/vorbis/codec.h:27-149 — thin adapters over the batched device
// path (vbm_frontend_* / vbm_encoder_*, include/vorbis_mi355x.h).
//
// A vorbis_dsp_state is one slot of a POOL of device streams of its encoder class.  Samples are queued by
// vorbis_analysis_wrote; a vorbis_analysis_blockout that its stream's queue cannot answer uploads what every
// stream of the pool has queued and runs ONE blockout round of the device front end for all of them (envelope
// search, carve-out and the whole per-block path: at most one block per stream), files the packets per stream,
// and answers from there.  vorbis_analysis / vorbis_bitrate_addblock / vorbis_bitrate_flushpacket then follow
// the reference's bookkeeping (lib/analysis.c:29-63, lib/bitrate.c:73-96, :229-252) on that packet.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <atomic>
#include <chrono>
#include <deque>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "vorbis_compat.h"
#include "vorbis_mi355x.h"
#include "vbm_internal.h"

#define VBM_PACKETBLOBS_HALF 7       /* PACKETBLOBS / 2, lib/backends.h */

namespace {

struct cpool;
struct cclass;

struct cblock {                      // one carved and encoded block (vb->internal)
    vbm_packet_info info;
    std::vector<uint8_t> pkt;
};

struct cwrite {                      // one vorbis_analysis_wrote, not yet on the device
    int vals;
    bool in_arena = false;           // the samples sit in the stream's slot of the pool's staging arena (no copy was made)
    std::vector<float> data;         // [ch][vals] otherwise
};

struct cstream {                     // vd->backend_state
    cpool *pool = nullptr;
    int slot = -1;
    std::vector<float> hostbuf;      // what vorbis_analysis_buffer hands out when the arena cannot serve: [ch][buf_vals]
    int buf_vals = 0;
    bool buf_in_arena = false;       // the pointers handed out last point into the pool's arena
    std::atomic<bool> arena_pending{false};   // the slot's arena region holds a write that has not gone up yet
    std::vector<float *> ptrs;
    std::deque<cwrite> writes;
    bool eof_asked = false;          // vorbis_analysis_wrote(v, 0) seen, not yet on the device
    bool eof_sent = false;           // ... on the device
    bool over = false;               // the e_o_s block has been carved
    bool dirty = false;              // something arrived since a round last had no block for this stream
    int given_since_write = 0;       // blocks handed out since the stream's last vorbis_analysis_wrote (VORBIS_MI355X_DEFER_BLOCKS)
    long writes_seen = 0;
    std::deque<cblock> ready;        // carved, not yet handed out by vorbis_analysis_blockout
    cblock cur;                      // the block vorbis_analysis_blockout handed out last
    bool cur_valid = false, cur_analysed = false;
    cblock parked;                   // bm->vb of the reference: added, not yet flushed
    bool parked_valid = false;
    cblock out;                      // storage of the packet vorbis_bitrate_flushpacket returned last
    std::vector<uint8_t> seam_pkt;   // storage of the packet vbm_mapping0_forward produced last
    std::vector<uint8_t> hdr[3];     // header packets handed out by vorbis_analysis_headerout
};

struct cpool {
    std::mutex mu;                   // everything below and the slots' streams: calls on streams of different pools run side by side
    cclass *cls = nullptr;
    vbm_encoder *enc = nullptr;
    vbm_frontend *fe = nullptr;
    int S = 0, maxb = 0;
    std::vector<cstream *> slots;
    std::vector<char> used_before;   // a stream has lived in the slot: the next one restarts it
    hipStream_t q = nullptr;
    float *h_pcm = nullptr, *d_pcm = nullptr;   // staging of one group of writes (those that did not go through the arena)
    // Staging arena: pinned host memory the device reads directly, [slot][ch][arena_vals].  vorbis_analysis_buffer hands
    // out the stream's own region (as the reference hands out its own PCM buffer, lib/block.c:405-436), so that
    // vorbis_analysis_wrote copies nothing and the upload is the append kernel fetching over the bus.
    float *arena = nullptr;
    int arena_vals = 0;
    size_t pcm_floats = 0;
    uint8_t *d_pkt = nullptr, *d_cmp = nullptr, *h_cmp = nullptr;
    int *d_len = nullptr, *h_len = nullptr;
    long long *d_off = nullptr, *h_off = nullptr;
    // rounds built on the device (vbm_frontend_encode_rounds_device: no decision comes back to the host in the middle of a
    // round, the kernels replay as HIP graphs): `lanes` output rows per round, records and the streams' ready types read back
    int lanes = 0;                   // 0: host-built rounds
    vbm_packet_info *d_info = nullptr, *h_info = nullptr;
    int *d_counts = nullptr;
    signed char *h_types = nullptr;
    size_t h_cmp_bytes = 0;
    std::vector<vbm_packet_info> info;
    int capacity = 0;                // samples a stream's device buffer may hold (vbm_frontend_capacity)
    std::vector<int> buffered;       // host mirror: samples in each slot's device buffer
};

struct cclass {                      // vi->codec_setup; one per mode pack, shared by every vorbis_info that names it
    std::mutex mu;                   // the pool list and the hand-out of slots
    std::string leaf;
    int refs = 0;
    long bitrate[3] = {0, 0, 0};     // nominal, lower, upper
    long reservoir_bits = 0;         // bitrate_manager_info (managed setups)
    double reservoir_bias = 0., damping = 0.;
    double lowpass_khz = 0.;         // hi.lowpass_kHz of the setup
    vbm_setup_handle *setup = nullptr;
    int ch = 0, managed = 0, bs[2] = {0, 0};
    long rate = 0;
    std::vector<cpool *> pools;
};

std::mutex g_mu;                     // the class registry and the knobs below
std::map<std::string, cclass *> g_classes;
int g_pool_streams = 0;
int g_carve_ahead = 1;
int g_defer_blocks = -1;             // -1: environment VORBIS_MI355X_DEFER_BLOCKS, default 0
std::string g_data_dir;
long long g_rounds = 0;
std::set<const vorbis_info *> g_unsealed;   // vorbis_encode_setup_vbr / _managed called, vorbis_encode_setup_init not yet
// where the host's time goes (VORBIS_MI355X_TIMES): seconds summed over all threads —
// 0 vorbis_analysis_wrote's copy, 1 staging copies of an upload, 2 upload: H2D + append kernels + waits, 3 the device
// round (decisions read back, kernels enqueued), 4 packet compaction + D2H waits, 5 filing packets per stream
std::atomic<long long> g_ns[8];
struct span_timer {
    int k; std::chrono::steady_clock::time_point t0;
    explicit span_timer(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {}
    ~span_timer() { g_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

std::string data_dir()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_data_dir.empty()) return g_data_dir;
    if (const char *env = getenv("VORBIS_MI355X_DATA")) return env;
    Dl_info di;
    if (dladdr((const void *)&vorbis_mi355x_ctl, &di) && di.dli_fname) {
        std::string p = di.dli_fname;
        const size_t at = p.rfind('/');
        return (at == std::string::npos ? std::string(".") : p.substr(0, at)) + "/data";
    }
    return "data";
}

int pool_streams()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pool_streams > 0) return g_pool_streams;
    if (const char *env = getenv("VORBIS_MI355X_POOL_STREAMS")) {
        const int n = atoi(env);
        if (n > 0) return n;
    }
    return 64;
}

void pool_destroy(cpool *p)
{
    if (!p) return;
    if (p->q) (void)hipStreamSynchronize(p->q);
    if (p->fe) vbm_frontend_destroy(p->fe);
    if (p->enc) vbm_encoder_destroy(p->enc);
    if (p->h_pcm) (void)hipHostFree(p->h_pcm);
    if (p->arena) (void)hipHostFree(p->arena);
    if (p->d_pcm) (void)hipFree(p->d_pcm);
    if (p->d_pkt) (void)hipFree(p->d_pkt);
    if (p->d_cmp) (void)hipFree(p->d_cmp);
    if (p->h_cmp) (void)hipHostFree(p->h_cmp);
    if (p->d_len) (void)hipFree(p->d_len);
    if (p->h_len) (void)hipHostFree(p->h_len);
    if (p->d_off) (void)hipFree(p->d_off);
    if (p->h_off) (void)hipHostFree(p->h_off);
    if (p->d_info) (void)hipFree(p->d_info);
    if (p->h_info) (void)hipHostFree(p->h_info);
    if (p->d_counts) (void)hipFree(p->d_counts);
    if (p->h_types) (void)hipHostFree(p->h_types);
    if (p->q) (void)hipStreamDestroy(p->q);
    delete p;
}

cpool *pool_create(cclass *c)
{
    cpool *p = new cpool();
    p->cls = c;
    p->S = pool_streams();
    {
        // VORBIS_MI355X_DEVICE_ROUNDS=1: the pool's rounds are built on the device (no decision read back in the middle of
        // a round, HIP graphs).  Off by default: a pool round here is a few thousand blocks at most and often a handful
        // (the reference's delivery asks for a round per short block), and a device-built round launches every block type's
        // whole lane region whatever it holds — measured 4 x 4096 streams: 19 k streams at 1x against 21-25 k (deferred
        // delivery), 5.3 k against 10.3 k (reference delivery).
        const char *env = getenv("VORBIS_MI355X_DEVICE_ROUNDS");
        if (env && atoi(env)) p->lanes = vbm_device_round_lanes(c->setup, p->S);
        if (p->lanes < 0) p->lanes = 0;
    }
    if (vbm_encoder_create(&p->enc, c->setup, p->S, p->lanes ? p->lanes : p->S) || vbm_frontend_create(&p->fe, p->enc)) {
        pool_destroy(p);
        return nullptr;
    }
    p->maxb = vbm_encoder_max_packet_bytes(p->enc);
    p->capacity = vbm_frontend_capacity(p->fe);
    const size_t S = (size_t)(p->lanes ? p->lanes : p->S);      // output rows of a round
    if (p->lanes &&
        (hipMalloc((void **)&p->d_info, S * sizeof(vbm_packet_info)) != hipSuccess ||
         hipHostMalloc((void **)&p->h_info, S * sizeof(vbm_packet_info), hipHostMallocDefault) != hipSuccess ||
         hipMalloc((void **)&p->d_counts, 4 * sizeof(int)) != hipSuccess ||
         hipHostMalloc((void **)&p->h_types, (size_t)p->S, hipHostMallocDefault) != hipSuccess)) {
        pool_destroy(p);
        return nullptr;
    }
    if (hipStreamCreateWithFlags(&p->q, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&p->d_pkt, S * p->maxb) != hipSuccess || hipMalloc((void **)&p->d_cmp, S * p->maxb) != hipSuccess ||
        hipMalloc((void **)&p->d_len, S * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void **)&p->h_len, S * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&p->d_off, (S + 1) * sizeof(long long)) != hipSuccess ||
        hipHostMalloc((void **)&p->h_off, (S + 1) * sizeof(long long), hipHostMallocDefault) != hipSuccess) {
        pool_destroy(p);
        return nullptr;
    }
    {
        // VORBIS_MI355X_ARENA_VALS: samples per channel a stream's arena region holds (default 1024 = the write size of the
        // reference's examples; larger requests are served from ordinary memory and copied); 0: no arena
        const char *env = getenv("VORBIS_MI355X_ARENA_VALS");
        p->arena_vals = env ? atoi(env) : 1024;
        if (p->arena_vals > 0 &&
            hipHostMalloc((void **)&p->arena, (size_t)p->S * c->ch * (size_t)p->arena_vals * sizeof(float), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            p->arena = nullptr;
            p->arena_vals = 0;
        }
    }
    p->slots.assign(p->S, nullptr);
    p->used_before.assign(p->S, 0);
    p->info.resize(S);
    p->buffered.assign(p->S, c->bs[1] / 2);     // centerW of a fresh stream (lib/block.c:330)
    return p;
}

int pool_stage(cpool *p, size_t floats)
{
    if (floats <= p->pcm_floats) return 0;
    (void)hipStreamSynchronize(p->q);
    if (p->h_pcm) (void)hipHostFree(p->h_pcm);
    if (p->d_pcm) (void)hipFree(p->d_pcm);
    p->h_pcm = nullptr; p->d_pcm = nullptr; p->pcm_floats = 0;
    if (hipHostMalloc((void **)&p->h_pcm, floats * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&p->d_pcm, floats * sizeof(float)) != hipSuccess) return OV_EFAULT;
    p->pcm_floats = floats;
    return 0;
}

// one device round (for all streams, or for `only`), packets to the streams' queues; *got_only: the round had a
// block for `only`
int pool_round(cpool *p, cstream *only, bool restrict_to_only, bool *got_only)
{
    int n = 0, rc;
    const bool dev = p->lanes > 0 && !restrict_to_only;     // the round is built on the device: rows = lanes, empty ones marked -2
    const vbm_packet_info *info = dev ? p->h_info : p->info.data();
    {
    span_timer tm(3);
    if (restrict_to_only) {
        const int id = only->slot;
        rc = vbm_frontend_encode_round_streams(p->fe, &id, 1, p->d_pkt, p->d_len, p->info.data(), &n, p->q);
    } else if (dev) {
        rc = vbm_frontend_encode_rounds_device(p->fe, 1, p->d_pkt, p->d_len, p->d_info, p->d_counts, 0, p->q);
        n = p->lanes;
    } else {
        rc = vbm_frontend_encode_round(p->fe, p->d_pkt, p->d_len, p->info.data(), &n, p->q);
    }
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_rounds++;
    }
    if (got_only) *got_only = false;
    if (rc) return rc;
    // a stream the round had no block for has nothing more to give until it is written to again: its next
    // vorbis_analysis_blockout is answered without a device round.  (Device-built rounds: unless it had a block ready and
    // found its type's lane region full — h_types says so — in which case the next round delivers it.)
    auto settle = [&](int nb) {
        if (restrict_to_only) return;
        std::vector<char> had(p->S, 0);
        for (int k = 0; k < nb; k++)
            if (!dev || p->h_len[k] != -2) had[info[k].stream] = 1;
        for (int i = 0; i < p->S; i++)
            if (p->slots[i] && !had[i] && p->slots[i]->writes.empty() && !(dev && p->h_types[i] >= 0)) p->slots[i]->dirty = false;
    };
    if (n == 0) { settle(0); return 0; }
    std::unique_ptr<span_timer> tm4(new span_timer(4));
    if (vbm_packets_compact(p->d_pkt, p->d_len, n, p->maxb, p->d_cmp, p->d_off, p->q)) return OV_EFAULT;
    if (hipMemcpyAsync(p->h_len, p->d_len, n * sizeof(int), hipMemcpyDeviceToHost, p->q) != hipSuccess ||
        hipMemcpyAsync(p->h_off, p->d_off, (n + 1) * sizeof(long long), hipMemcpyDeviceToHost, p->q) != hipSuccess ||
        (dev && (hipMemcpyAsync(p->h_info, p->d_info, (size_t)n * sizeof(vbm_packet_info), hipMemcpyDeviceToHost, p->q) != hipSuccess ||
                 vbm_frontend_round_types(p->fe, p->h_types, p->q))) ||
        hipStreamSynchronize(p->q) != hipSuccess) return OV_EFAULT;
    const size_t total = (size_t)p->h_off[n];
    if (total > p->h_cmp_bytes) {
        if (p->h_cmp) (void)hipHostFree(p->h_cmp);
        p->h_cmp = nullptr;
        p->h_cmp_bytes = 0;
        const size_t want = total + total / 2 + 4096;
        if (hipHostMalloc((void **)&p->h_cmp, want, hipHostMallocDefault) != hipSuccess) return OV_EFAULT;
        p->h_cmp_bytes = want;
    }
    if (total && (hipMemcpyAsync(p->h_cmp, p->d_cmp, total, hipMemcpyDeviceToHost, p->q) != hipSuccess ||
                  hipStreamSynchronize(p->q) != hipSuccess)) return OV_EFAULT;
    tm4.reset();
    span_timer tm5(5);
    const int bs0 = p->cls->bs[0], bs1 = p->cls->bs[1];
    for (int k = 0; k < n; k++) {
        if (dev && p->h_len[k] == -2) continue;       // a lane without a block
        const vbm_packet_info &pi = info[k];
        if (pi.stream < 0 || pi.stream >= p->S) return OV_EFAULT;
        cstream *s = p->slots[pi.stream];
        if (p->h_len[k] < 0) return OV_EFAULT;        // a packet outgrew max_packet_bytes: never silently truncated
        // host mirror of the buffer fill: the buffer moves down by the distance between block centres
        // (lib/block.c:745-759), not at all after the e_o_s block
        if (!pi.eos) p->buffered[pi.stream] -= (pi.W ? bs1 : bs0) / 4 + (pi.nW ? bs1 : bs0) / 4;
        if (!s) continue;                             // slot released with blocks in flight: dropped
        cblock b;
        b.info = pi;
        b.pkt.assign(p->h_cmp + p->h_off[k], p->h_cmp + p->h_off[k] + p->h_len[k]);
        if (pi.eos) s->over = true;
        s->ready.push_back(std::move(b));
        if (s == only && got_only) *got_only = true;
    }
    settle(n);
    return 0;
}

// queued writes -> device.  `only` != nullptr: that stream's writes alone.  Writes of equal length go up together
// (vbm_frontend_write_streams); a write that would overrun a stream's device buffer waits for rounds to drain it.
int pool_upload(cpool *p, cstream *only)
{
    const int ch = p->cls->ch;
    for (;;) {
        // the front write of every stream that has one and room for it, grouped by length
        std::map<int, std::vector<cstream *>> groups;
        bool blocked = false, any = false;
        for (cstream *s : p->slots) {
            if (!s || s->writes.empty() || (only && s != only)) continue;
            any = true;
            const int vals = s->writes.front().vals;
            if (vals > p->capacity) return OV_EINVAL;                 // can never fit (lib/block.c:540-541)
            if (p->buffered[s->slot] + vals > p->capacity) { blocked = true; continue; }
            groups[vals].push_back(s);
        }
        if (!any) return 0;
        if (groups.empty() && blocked) {
            // every pending write waits for room: carve blocks (they go to the queues) and try again
            bool got = false;
            int rc = pool_round(p, only, only != nullptr, &got);
            if (rc) return rc;
            bool still = true;
            for (cstream *s : p->slots)
                if (s && !s->writes.empty() && (!only || s == only) &&
                    p->buffered[s->slot] + s->writes.front().vals <= p->capacity) still = false;
            if (still) return OV_EINVAL;                              // the round freed nothing: give up, not spin
            continue;
        }
        for (auto &g : groups) {
            const int vals = g.first;
            std::vector<cstream *> direct, copied;
            for (cstream *s : g.second) (s->writes.front().in_arena ? direct : copied).push_back(s);
            if (!direct.empty()) {
                // straight from the arena: the append kernel reads the slots' regions over the bus
                span_timer tm2(2);
                std::vector<int> ids(direct.size());
                for (size_t k = 0; k < direct.size(); k++) ids[k] = direct[k]->slot;
                int rc = vbm_frontend_write_streams_strided(p->fe, ids.data(), (int)ids.size(), p->arena, vals,
                                                            (long)ch * p->arena_vals, p->arena_vals, 1, p->q);
                if (rc) return rc;
                for (cstream *s : direct) {          // (the call returns when the samples have been taken)
                    p->buffered[s->slot] += vals;
                    s->writes.pop_front();
                    s->arena_pending = false;
                }
            }
            if (copied.empty()) continue;
            std::vector<cstream *> &ss = copied;
            const size_t per = (size_t)ch * vals;
            int rc = pool_stage(p, per * ss.size());
            if (rc) return rc;
            std::vector<int> ids(ss.size());
            (void)hipStreamSynchronize(p->q);                         // the staging buffer is free again
            {
            span_timer tm(1);
            for (size_t k = 0; k < ss.size(); k++) {
                ids[k] = ss[k]->slot;
                memcpy(p->h_pcm + k * per, ss[k]->writes.front().data.data(), per * sizeof(float));
            }
            }
            span_timer tm2(2);
            if (hipMemcpyAsync(p->d_pcm, p->h_pcm, per * ss.size() * sizeof(float), hipMemcpyHostToDevice, p->q) != hipSuccess)
                return OV_EFAULT;
            rc = vbm_frontend_write_streams(p->fe, ids.data(), (int)ids.size(), p->d_pcm, vals, p->q);
            if (rc) return rc;
            for (cstream *s : ss) {
                p->buffered[s->slot] += vals;
                s->writes.pop_front();
            }
        }
    }
}

// vorbis_analysis_wrote(v, 0) for stream s: everything it wrote goes up, then the end is declared on the device at
// once — blocks the application has not asked for yet stay in the buffer, and the end-of-stream extrapolation is
// fitted to exactly that buffer, as in the reference (lib/block.c:497-537; its own test writes 2048 samples and
// declares the end before the first vorbis_analysis_blockout, test/write_read.c:95-99)
int stream_send_eof(cpool *p, cstream *s)
{
    int carve;
    { std::lock_guard<std::mutex> lk(g_mu); carve = g_carve_ahead; }
    int rc = pool_upload(p, carve ? nullptr : s);
    if (rc) return rc;
    const int id = s->slot;
    rc = vbm_frontend_finish(p->fe, &id, 1, p->q);
    if (rc) return rc;
    p->buffered[s->slot] += 3 * p->cls->bs[1];
    s->eof_asked = false;
    s->eof_sent = true;
    s->dirty = true;
    return 0;
}

cstream *stream_of(vorbis_dsp_state *v) { return v ? (cstream *)v->backend_state : nullptr; }

void fill_op(ogg_packet *op, const cblock &b)
{
    op->packet = const_cast<unsigned char *>(b.pkt.data());
    op->bytes = (long)b.pkt.size();
    op->b_o_s = 0;
    op->e_o_s = b.info.eos;
    op->granulepos = b.info.granulepos;
    op->packetno = b.info.packetno;
}

int class_open(vorbis_info *vi, long channels, long rate, const std::string &leaf)
{
    if (!vi) return OV_EINVAL;
    const std::string dir = data_dir();
    std::lock_guard<std::mutex> lk(g_mu);
    cclass *c = nullptr;
    auto it = g_classes.find(dir + "/" + leaf);
    if (it != g_classes.end()) {
        c = it->second;
    } else {
        const std::string mode = dir + "/" + leaf, common = dir + "/common.vpk";
        FILE *f = fopen(mode.c_str(), "rb");
        if (!f) return OV_EIMPL;                    // no shipped mode pack for this (channels, rate, quality)
        fclose(f);
        vbm_setup_handle *h = nullptr;
        if (vbm_setup_create(&h, common.c_str(), mode.c_str())) return OV_EFAULT;
        const void *data;
        long count;
        char kind;
        if (vbm_setup_table(h, "info", &data, &count, &kind) || count < 4) { vbm_setup_destroy(h); return OV_EFAULT; }
        const int *inf = (const int *)data;
        c = new cclass();
        c->leaf = mode;
        c->setup = h;
        c->ch = (int)channels;
        c->rate = rate;
        c->bs[0] = inf[2];
        c->bs[1] = inf[3];
        if (vbm_setup_table(h, "bitrate", &data, &count, &kind) == 0 && kind == 'd' && count >= 7) {
            const double *bi = (const double *)data;   // managed, nominal, lower, upper, reservoir bits, bias, damping
            c->managed = bi[0] != 0.;
            for (int k = 0; k < 3; k++) c->bitrate[k] = (long)bi[1 + k];
            c->reservoir_bits = (long)bi[4];
            c->reservoir_bias = bi[5];
            c->damping = bi[6];
        }
        if (vbm_setup_table(h, "lowpass_kHz", &data, &count, &kind) == 0 && kind == 'd' && count >= 1)
            c->lowpass_khz = ((const double *)data)[0];
        g_classes[mode] = c;
    }
    c->refs++;
    vi->version = 0;
    vi->channels = (int)channels;
    vi->rate = rate;
    vi->bitrate_nominal = c->bitrate[0];
    vi->bitrate_lower = c->bitrate[1];
    vi->bitrate_upper = c->bitrate[2];
    vi->codec_setup = c;
    return 0;
}

}  // namespace

// ---- knobs ---------------------------------------------------------------------------------------
extern "C" int vorbis_mi355x_ctl(int request, void *arg)
{
    if (!arg) return OV_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    switch (request) {
    case VORBIS_MI355X_POOL_STREAMS:
        if (*(int *)arg <= 0) return OV_EINVAL;
        g_pool_streams = *(int *)arg;
        return 0;
    case VORBIS_MI355X_CARVE_AHEAD: g_carve_ahead = *(int *)arg != 0; return 0;
    case VORBIS_MI355X_DEFER_BLOCKS: g_defer_blocks = *(int *)arg != 0; return 0;
    case VORBIS_MI355X_DATA_DIR: g_data_dir = (const char *)arg; return 0;
    case VORBIS_MI355X_ROUNDS: *(long long *)arg = g_rounds; return 0;
    case VORBIS_MI355X_TIMES:
        for (int i = 0; i < 8; i++) ((double *)arg)[i] = 1e-9 * (double)g_ns[i].load();
        return 0;
    }
    return OV_EINVAL;
}

// ---- vorbis_info / vorbis_comment (reference lib/info.c:57-185) --------------------------------------
extern "C" void vorbis_info_init(vorbis_info *vi)
{
    if (vi) memset(vi, 0, sizeof(*vi));
}

extern "C" void vorbis_info_clear(vorbis_info *vi)
{
    if (!vi) return;
    cclass *c = (cclass *)vi->codec_setup;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_unsealed.erase(vi);
    }
    if (c) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (--c->refs <= 0) {                        // the last vorbis_info of the class: pools and tables go
            g_classes.erase(c->leaf);
            for (cpool *p : c->pools) pool_destroy(p);
            if (c->setup) vbm_setup_destroy(c->setup);
            delete c;
        }
    }
    memset(vi, 0, sizeof(*vi));
}

extern "C" int vorbis_info_blocksize(vorbis_info *vi, int zo)
{
    cclass *c = vi ? (cclass *)vi->codec_setup : nullptr;
    return c ? c->bs[zo ? 1 : 0] : -1;
}

extern "C" void vorbis_comment_init(vorbis_comment *vc)
{
    if (vc) memset(vc, 0, sizeof(*vc));
}

extern "C" void vorbis_comment_add(vorbis_comment *vc, const char *comment)
{
    if (!vc || !comment) return;
    const int n = vc->comments;
    vc->user_comments = (char **)realloc(vc->user_comments, (n + 2) * sizeof(char *));
    vc->comment_lengths = (int *)realloc(vc->comment_lengths, (n + 2) * sizeof(int));
    vc->comment_lengths[n] = (int)strlen(comment);
    vc->user_comments[n] = strdup(comment);
    vc->comments = n + 1;
    vc->user_comments[n + 1] = nullptr;
}

extern "C" void vorbis_comment_add_tag(vorbis_comment *vc, const char *tag, const char *contents)
{
    if (!tag || !contents) return;
    const std::string s = std::string(tag) + "=" + contents;
    vorbis_comment_add(vc, s.c_str());
}

extern "C" char *vorbis_comment_query(vorbis_comment *vc, const char *tag, int count)
{
    if (!vc || !tag) return nullptr;
    const size_t tl = strlen(tag);
    int found = 0;
    for (int i = 0; i < vc->comments; i++) {
        const char *c = vc->user_comments[i];
        if (!strncasecmp(c, tag, tl) && c[tl] == '=') {
            if (found == count) return vc->user_comments[i] + tl + 1;
            found++;
        }
    }
    return nullptr;
}

extern "C" int vorbis_comment_query_count(vorbis_comment *vc, const char *tag)
{
    if (!vc || !tag) return 0;
    const size_t tl = strlen(tag);
    int found = 0;
    for (int i = 0; i < vc->comments; i++)
        if (!strncasecmp(vc->user_comments[i], tag, tl) && vc->user_comments[i][tl] == '=') found++;
    return found;
}

extern "C" void vorbis_comment_clear(vorbis_comment *vc)
{
    if (!vc) return;
    for (int i = 0; i < vc->comments; i++) free(vc->user_comments[i]);
    free(vc->user_comments);
    free(vc->comment_lengths);
    free(vc->vendor);
    memset(vc, 0, sizeof(*vc));
}

extern "C" const char *vorbis_version_string(void) { return vbm_version(); }

extern "C" double vorbis_granule_time(vorbis_dsp_state *v, ogg_int64_t granulepos)
{
    if (!v || !v->vi || granulepos < 0 || v->vi->rate <= 0) return -1.;
    return (double)granulepos / (double)v->vi->rate;
}

// ---- libvorbisenc's two one-call initialisers: pick the shipped mode pack ------------------------------
extern "C" int vorbis_encode_init_vbr(vorbis_info *vi, long channels, long rate, float base_quality)
{
    char leaf[128];
    double q = floor((double)base_quality * 1000. + .5) / 1000.;     // 0.1f is 0.100000001…: the pack is named q0.1
    if (q == 0.) q = 0.;                                             // no "-0"
    snprintf(leaf, sizeof(leaf), "mode_%ldch_%ld_q%g.vpk", channels, rate, q);
    return class_open(vi, channels, rate, leaf);
}

extern "C" int vorbis_encode_init(vorbis_info *vi, long channels, long rate, long max_bitrate, long nominal_bitrate,
                                  long min_bitrate)
{
    if (nominal_bitrate <= 0) return OV_EIMPL;
    std::string leaf = "mode_" + std::to_string(channels) + "ch_" + std::to_string(rate) + "_b" + std::to_string(nominal_bitrate);
    if (max_bitrate > 0) leaf += "_max" + std::to_string(max_bitrate);
    if (min_bitrate > 0) leaf += "_min" + std::to_string(min_bitrate);
    return class_open(vi, channels, rate, leaf + ".vpk");
}

// ---- the three-step setup (reference lib/vorbisenc.c:977-1260) ------------------------------------------------
extern "C" int vorbis_encode_setup_vbr(vorbis_info *vi, long channels, long rate, float quality)
{
    if (rate <= 0) return OV_EINVAL;
    int rc = vorbis_encode_init_vbr(vi, channels, rate, quality);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    g_unsealed.insert(vi);
    return 0;
}

extern "C" int vorbis_encode_setup_managed(vorbis_info *vi, long channels, long rate, long max_bitrate, long nominal_bitrate,
                                           long min_bitrate)
{
    if (rate <= 0) return OV_EINVAL;
    int rc = vorbis_encode_init(vi, channels, rate, max_bitrate, nominal_bitrate, min_bitrate);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    g_unsealed.insert(vi);
    return 0;
}

extern "C" int vorbis_encode_setup_init(vorbis_info *vi)
{
    if (!vi || !vi->codec_setup) return OV_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    g_unsealed.erase(vi);        // (sealing twice is harmless, as in the reference: set_in_stone stays set)
    return 0;
}

extern "C" int vorbis_encode_ctl(vorbis_info *vi, int number, void *arg)
{
    if (!vi || !vi->codec_setup) return OV_EINVAL;
    const cclass *c = (const cclass *)vi->codec_setup;
    const bool set = (number & 0xf) != 0;                 // a read request has a low nibble of 0 (lib/vorbisenc.c:1076)
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (set && !g_unsealed.count(vi)) return OV_EINVAL;   // set_in_stone (:1078)
    }
    const long kmin = c->bitrate[1] > 0 ? c->bitrate[1] / 1000 : 0, kmax = c->bitrate[2] > 0 ? c->bitrate[2] / 1000 : 0;
    switch (number) {
    case OV_ECTL_RATEMANAGE2_GET: {
        ovectl_ratemanage2_arg *ai = (ovectl_ratemanage2_arg *)arg;
        if (!ai) return OV_EINVAL;
        ai->management_active = c->managed;
        ai->bitrate_limit_min_kbps = kmin;
        ai->bitrate_limit_max_kbps = kmax;
        ai->bitrate_average_kbps = c->bitrate[0] / 1000;
        ai->bitrate_average_damping = c->damping;
        ai->bitrate_limit_reservoir_bits = c->reservoir_bits;
        ai->bitrate_limit_reservoir_bias = c->reservoir_bias;
        return 0;
    }
    case OV_ECTL_RATEMANAGE2_SET: {
        const ovectl_ratemanage2_arg *ai = (const ovectl_ratemanage2_arg *)arg;
        if (!ai) return c->managed ? OV_EIMPL : 0;       // "no management": what a VBR pack is already
        if (!!ai->management_active != !!c->managed) return OV_EIMPL;
        if (!c->managed) return 0;
        const bool same = ai->bitrate_limit_min_kbps == kmin && ai->bitrate_limit_max_kbps == kmax &&
                          ai->bitrate_average_kbps == c->bitrate[0] / 1000 && ai->bitrate_average_damping == c->damping &&
                          ai->bitrate_limit_reservoir_bits == c->reservoir_bits &&
                          ai->bitrate_limit_reservoir_bias == c->reservoir_bias;
        return same ? 0 : OV_EIMPL;
    }
    case OV_ECTL_LOWPASS_GET:
        if (!arg) return OV_EINVAL;
        *(double *)arg = c->lowpass_khz;
        return 0;
    case OV_ECTL_LOWPASS_SET:
        if (!arg) return OV_EINVAL;
        return *(const double *)arg == c->lowpass_khz ? 0 : OV_EIMPL;
    case OV_ECTL_IBLOCK_GET:
        if (!arg) return OV_EINVAL;
        *(double *)arg = 0.;                              // hi.impulse_noisetune of a fresh setup
        return 0;
    case OV_ECTL_IBLOCK_SET:
        if (!arg) return OV_EINVAL;
        return *(const double *)arg == 0. ? 0 : OV_EIMPL;
    case OV_ECTL_COUPLING_GET:
        if (!arg) return OV_EINVAL;
        *(int *)arg = 1;                                  // hi.coupling_p
        return 0;
    case OV_ECTL_COUPLING_SET:
        if (!arg) return OV_EINVAL;
        return *(const int *)arg ? 0 : OV_EIMPL;
    }
    return OV_EIMPL;
}

extern "C" int vorbis_commentheader_out(vorbis_comment *vc, ogg_packet *op)
{
    if (!op) return OV_EFAULT;
    const int nc = vc ? vc->comments : 0;
    const char *const *cm = vc ? (const char *const *)vc->user_comments : nullptr;
    long len = 0;
    if (vbm_comment_packet(nullptr, cm, nc, nullptr, 0, &len)) return OV_EIMPL;
    unsigned char *buf = (unsigned char *)malloc((size_t)len);
    if (!buf) return OV_EFAULT;
    if (vbm_comment_packet(nullptr, cm, nc, buf, len, &len)) { free(buf); return OV_EIMPL; }
    memset(op, 0, sizeof(*op));
    op->packet = buf;
    op->bytes = len;
    op->b_o_s = 0;
    op->e_o_s = 0;
    op->granulepos = 0;
    op->packetno = 1;
    return 0;
}

// ---- stream life cycle (reference lib/block.c:84-107, :306-409) ----------------------------------------
extern "C" int vorbis_analysis_init(vorbis_dsp_state *v, vorbis_info *vi)
{
    if (!v || !vi || !vi->codec_setup) return 1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_unsealed.count(vi)) return 1;      // vorbis_encode_setup_init has not been called
    }
    cclass *c = (cclass *)vi->codec_setup;
    memset(v, 0, sizeof(*v));
    std::lock_guard<std::mutex> lk(c->mu);
    cpool *p = nullptr;
    int slot = -1;
    for (cpool *cand : c->pools) {
        for (int i = 0; i < cand->S && slot < 0; i++)
            if (!cand->slots[i]) { p = cand; slot = i; }
        if (slot >= 0) break;
    }
    if (slot < 0) {
        p = pool_create(c);                      // fails without a HIP device: there is no CPU path
        if (!p) return 1;
        c->pools.push_back(p);
        slot = 0;
    }
    std::lock_guard<std::mutex> lkp(p->mu);
    if (p->used_before[slot]) {
        if (vbm_frontend_restart_streams(p->fe, &slot, 1, p->q)) return 1;
        p->buffered[slot] = c->bs[1] / 2;
    }
    p->used_before[slot] = 1;
    cstream *s = new cstream();
    s->pool = p;
    s->slot = slot;
    s->ptrs.assign(c->ch, nullptr);
    p->slots[slot] = s;
    v->analysisp = 1;
    v->vi = vi;
    v->W = 0;
    v->centerW = c->bs[1] / 2;
    v->pcm_current = (int)v->centerW;
    v->sequence = 3;                             // the three header packets come first (lib/block.c:337)
    v->backend_state = s;
    return 0;
}

extern "C" void vorbis_dsp_clear(vorbis_dsp_state *v)
{
    cstream *s = stream_of(v);
    if (s) {
        cclass *c = s->pool->cls;
        std::lock_guard<std::mutex> lk(c->mu);
        std::lock_guard<std::mutex> lkp(s->pool->mu);
        s->pool->slots[s->slot] = nullptr;
        delete s;
    }
    if (v) memset(v, 0, sizeof(*v));
}

extern "C" int vorbis_block_init(vorbis_dsp_state *v, vorbis_block *vb)
{
    if (!vb) return OV_EINVAL;
    memset(vb, 0, sizeof(*vb));
    vb->vd = v;
    return 0;
}

extern "C" int vorbis_block_clear(vorbis_block *vb)
{
    if (vb) memset(vb, 0, sizeof(*vb));
    return 0;
}

extern "C" int vorbis_analysis_headerout(vorbis_dsp_state *v, vorbis_comment *vc, ogg_packet *op, ogg_packet *op_comm,
                                         ogg_packet *op_code)
{
    cstream *s = stream_of(v);
    if (!s || !op || !op_comm || !op_code) return OV_EFAULT;
    cclass *c = s->pool->cls;
    long lens[3];
    const int nc = vc ? vc->comments : 0;
    const char *const *cm = vc ? (const char *const *)vc->user_comments : nullptr;
    if (vbm_header_packets(c->setup, nullptr, cm, nc, nullptr, 0, lens)) return OV_EIMPL;
    std::vector<uint8_t> all(lens[0] + lens[1] + lens[2]);
    if (vbm_header_packets(c->setup, nullptr, cm, nc, all.data(), (long)all.size(), lens)) return OV_EIMPL;
    long at = 0;
    ogg_packet *ops[3] = {op, op_comm, op_code};
    for (int i = 0; i < 3; i++) {
        s->hdr[i].assign(all.begin() + at, all.begin() + at + lens[i]);
        at += lens[i];
        memset(ops[i], 0, sizeof(ogg_packet));
        ops[i]->packet = s->hdr[i].data();
        ops[i]->bytes = lens[i];
        ops[i]->b_o_s = i == 0;
        ops[i]->packetno = i;
    }
    return 0;
}

// ---- PCM in (reference lib/block.c:411-553) -----------------------------------------------------------
extern "C" float **vorbis_analysis_buffer(vorbis_dsp_state *v, int vals)
{
    cstream *s = stream_of(v);
    if (!s || vals < 0) return nullptr;
    cpool *p = s->pool;
    const int ch = p->cls->ch;
    if (p->arena && vals <= p->arena_vals && !s->arena_pending.load()) {
        // the stream's own region of the pool's pinned staging arena (free: its last write has gone up)
        for (int c = 0; c < ch; c++) s->ptrs[c] = p->arena + ((size_t)s->slot * ch + c) * p->arena_vals;
        s->buf_in_arena = true;
        v->pcmret = s->ptrs.data();
        v->pcm_storage = p->arena_vals;
        return v->pcmret;
    }
    s->buf_in_arena = false;
    if (vals > s->buf_vals) {
        s->buf_vals = vals * 2 > 1024 ? vals * 2 : 1024;
        s->hostbuf.assign((size_t)ch * s->buf_vals, 0.f);
    }
    for (int c = 0; c < ch; c++) s->ptrs[c] = s->hostbuf.data() + (size_t)c * s->buf_vals;
    v->pcmret = s->ptrs.data();
    v->pcm_storage = s->buf_vals;
    return v->pcmret;
}

extern "C" int vorbis_analysis_wrote(vorbis_dsp_state *v, int vals)
{
    cstream *s = stream_of(v);
    if (!s) return OV_EINVAL;
    cpool *p = s->pool;
    cclass *c = p->cls;
    std::lock_guard<std::mutex> lk(p->mu);
    if (s->eof_asked || s->eof_sent) return OV_EINVAL;
    if (vals <= 0) {
        s->eof_asked = true;
        s->dirty = true;
        v->eofflag = v->pcm_current;
        // the end is declared on the device now: what the buffer holds at THIS moment decides the extrapolation
        // (lib/block.c:497-537)
        int rc = stream_send_eof(p, s);
        return rc ? OV_EINVAL : 0;
    }
    if (vals > (s->buf_in_arena ? p->arena_vals : s->buf_vals)) return OV_EINVAL;    // more than vorbis_analysis_buffer handed out (lib/block.c:540-541)
    span_timer tm(0);
    cwrite w;
    w.vals = vals;
    if (s->buf_in_arena && !s->arena_pending.load()) {
        w.in_arena = true;                       // the samples stay where the application put them
        s->arena_pending = true;
    } else {
        w.data.resize((size_t)c->ch * vals);
        for (int k = 0; k < c->ch; k++) memcpy(w.data.data() + (size_t)k * vals, s->ptrs[k], vals * sizeof(float));
    }
    s->writes.push_back(std::move(w));
    s->dirty = true;
    s->given_since_write = 0;
    s->writes_seen++;
    v->pcm_current += vals;
    v->preextrapolate = 1;
    return 0;
}

// ---- blocks out (reference lib/block.c:557-812) --------------------------------------------------------
extern "C" int vorbis_analysis_blockout(vorbis_dsp_state *v, vorbis_block *vb)
{
    cstream *s = stream_of(v);
    if (!s || !vb) return 0;
    cpool *p = s->pool;
    cclass *c = p->cls;
    std::lock_guard<std::mutex> lk(p->mu);
    int carve, defer;
    {
        std::lock_guard<std::mutex> lk2(g_mu);
        carve = g_carve_ahead;
        if (g_defer_blocks < 0) g_defer_blocks = getenv("VORBIS_MI355X_DEFER_BLOCKS") ? atoi(getenv("VORBIS_MI355X_DEFER_BLOCKS")) != 0 : 0;
        defer = g_defer_blocks;
    }
    // VORBIS_MI355X_DEFER_BLOCKS: a stream that has had its block for this write (two every third write) is told "more
    // data needed" although the device might carve another one — a stream inside a run of short blocks would otherwise
    // make the whole pool run up to eight more rounds per write, each for a handful of blocks.  The blocks come later
    // (latest when the stream's buffer is half full, or at end of stream): same packets, later delivery.
    if (defer && !s->eof_sent && !s->eof_asked && s->ready.empty() &&
        s->given_since_write >= 1 + (s->writes_seen % 3 == 0) && p->buffered[s->slot] <= p->capacity / 2)
        return 0;
    if (s->ready.empty() && s->dirty && !(s->over)) {
        if (pool_upload(p, carve ? nullptr : s)) return 0;
        if (s->ready.empty()) {
            bool got = false;
            if (pool_round(p, s, !carve, &got)) return 0;
            if (!got) s->dirty = false;          // nothing more until more PCM (or the end) arrives
        }
    }
    if (s->ready.empty()) return 0;
    s->cur = std::move(s->ready.front());
    s->ready.pop_front();
    s->given_since_write++;
    s->cur_valid = true;
    s->cur_analysed = false;
    const vbm_packet_info &pi = s->cur.info;
    memset(vb, 0, sizeof(*vb));
    vb->lW = pi.lW;
    vb->W = pi.W;
    vb->nW = pi.nW;
    vb->pcmend = c->bs[pi.W ? 1 : 0];
    vb->mode = pi.W;
    vb->eofflag = pi.eos;
    vb->granulepos = pi.granulepos;
    vb->sequence = pi.packetno;
    vb->vd = v;
    vb->internal = &s->cur;
    // the stream state as the reference leaves it after the block (lib/block.c:730-808)
    v->lW = pi.W;
    v->W = pi.nW;
    v->nW = 0;
    v->sequence = pi.packetno + 1;
    v->granulepos = pi.granulepos;
    if (pi.eos) v->eofflag = -1;
    return 1;
}

// ---- the per-block path's own entry points ---------------------------------------------------------------
// vorbis_analysis (lib/analysis.c:29-63).  The device ran mapping0_forward for this block in the round that
// carved it; here its result is bound to vb (vb->opb describes the packet) and, for op != NULL, handed out.
extern "C" int vorbis_analysis(vorbis_block *vb, ogg_packet *op)
{
    if (!vb || !vb->vd) return OV_EINVAL;
    cstream *s = stream_of(vb->vd);
    if (!s || !s->cur_valid || vb->internal != &s->cur) return OV_EINVAL;
    vb->glue_bits = vb->time_bits = vb->floor_bits = vb->res_bits = 0;
    s->cur_analysed = true;
    vb->opb.buffer = vb->opb.ptr = s->cur.pkt.data();
    vb->opb.endbyte = (long)s->cur.pkt.size();
    vb->opb.endbit = 0;
    vb->opb.storage = (long)s->cur.pkt.size();
    if (op) {
        if (s->pool->cls->managed) return OV_EINVAL;     // bit-managed mode without the bitrate interface (:50-53)
        fill_op(op, s->cur);
    }
    return 0;
}

// vorbis_bitrate_addblock (lib/bitrate.c:73-227): the block is parked until flushpacket claims it.  (Managed
// streams: the reservoirs chose one of the 15 packetblobs on the device, k_bitrate_choose, in block order.)
extern "C" int vorbis_bitrate_addblock(vorbis_block *vb)
{
    if (!vb || !vb->vd) return OV_EINVAL;
    cstream *s = stream_of(vb->vd);
    if (!s || !s->cur_valid || vb->internal != &s->cur || !s->cur_analysed) return OV_EINVAL;
    if (s->parked_valid && !s->pool->cls->managed) return -1;          // one submitted without being claimed (:92)
    s->parked = s->cur;               // (vb keeps describing s->cur: a copy, as the reference's bm->vb keeps the block)
    s->parked_valid = true;
    return 0;
}

// vorbis_bitrate_flushpacket (lib/bitrate.c:229-252)
extern "C" int vorbis_bitrate_flushpacket(vorbis_dsp_state *vd, ogg_packet *op)
{
    cstream *s = stream_of(vd);
    if (!s || !s->parked_valid) return 0;
    if (op) {
        s->out = std::move(s->parked);
        fill_op(op, s->out);
    }
    s->parked_valid = false;
    return 1;
}

// ---- the reference's internal plugin seam (lib/backends.h:121-128, lib/mapping0.c:1500-1506, lib/registry.c:42-44) --
// mapping0_forward(vorbis_block *vb) for a block the CALLER carved (vb->pcm on the host): a batch of one through
// vbm_analysis_batch on the stream's device slot.  The caller keeps its own blockout and bitrate management.
extern "C" int vbm_mapping0_forward(vorbis_block *vb)
{
    if (!vb || !vb->vd || !vb->pcm || !vb->internal) return OV_EINVAL;
    cstream *s = stream_of(vb->vd);
    if (!s) return OV_EINVAL;
    cpool *p = s->pool;
    cclass *c = p->cls;
    const vorbis_block_internal *vbi = (const vorbis_block_internal *)vb->internal;
    const int W = vb->W ? 1 : 0, N = c->bs[W], ch = c->ch;
    if (vb->pcmend != N || (vbi->blocktype & ~1) || (c->bs[0] == c->bs[1] && W)) return OV_EINVAL;
    for (int k = 0; k < ch; k++)
        if (!vb->pcm[k]) return OV_EINVAL;
    std::lock_guard<std::mutex> lk(p->mu);
    const size_t per = (size_t)ch * N;
    if (pool_stage(p, per)) return OV_EFAULT;
    (void)hipStreamSynchronize(p->q);                    // the staging buffer is free
    for (int k = 0; k < ch; k++) memcpy(p->h_pcm + (size_t)k * N, vb->pcm[k], (size_t)N * sizeof(float));
    const int block_mode = vbi->blocktype | (W << 1);    // psy_look = b->psy + blocktype + (W ? 2 : 0), lib/mapping0.c:764
    const int id = s->slot;
    const uint8_t wf = (uint8_t)((vb->lW ? 1 : 0) | (vb->nW ? 2 : 0));
    if (hipMemcpyAsync(p->d_pcm, p->h_pcm, per * sizeof(float), hipMemcpyHostToDevice, p->q) != hipSuccess) return OV_EFAULT;
    if (vbm_analysis_batch(p->enc, block_mode, 1, &id, &wf, p->d_pcm, p->d_pkt, p->d_len, p->q)) return OV_EFAULT;
    if (hipMemcpyAsync(p->h_len, p->d_len, sizeof(int), hipMemcpyDeviceToHost, p->q) != hipSuccess ||
        hipStreamSynchronize(p->q) != hipSuccess) return OV_EFAULT;
    const int bytes = p->h_len[0];
    if (bytes < 0 || bytes > p->maxb) return OV_EFAULT;  // a packet outgrew max_packet_bytes: never silently truncated
    s->seam_pkt.resize((size_t)bytes);
    if (bytes && hipMemcpy(s->seam_pkt.data(), p->d_pkt, (size_t)bytes, hipMemcpyDeviceToHost) != hipSuccess) return OV_EFAULT;
    vb->mode = W;
    vb->glue_bits = vb->time_bits = vb->floor_bits = vb->res_bits = 0;
    vb->opb.buffer = vb->opb.ptr = s->seam_pkt.data();
    vb->opb.endbyte = bytes;
    vb->opb.endbit = 0;
    vb->opb.storage = bytes;
    oggpack_buffer *o = vbi->packetblob[VBM_PACKETBLOBS_HALF];
    if (o) {                                              // where mapping0_forward leaves the packet (lib/mapping0.c:1204-1313)
        if (!o->buffer || o->storage < bytes + 1) {
            unsigned char *nb = (unsigned char *)realloc(o->buffer, (size_t)bytes + 256);
            if (!nb) return OV_EFAULT;
            o->buffer = nb;
            o->storage = bytes + 256;
        }
        if (bytes) memcpy(o->buffer, s->seam_pkt.data(), (size_t)bytes);
        o->buffer[bytes] = 0;
        o->ptr = o->buffer + bytes;
        o->endbyte = bytes;
        o->endbit = 0;
    }
    return 0;
}

#if !defined(__HIP_DEVICE_COMPILE__)    // (this file goes through hipcc: a host function's address means nothing to the device pass)
extern "C" const vorbis_func_mapping mapping0_exportbundle_mi355x = {nullptr, nullptr, nullptr, &vbm_mapping0_forward, nullptr};
#endif

// ---- page framing under libogg's names (include/vorbis_compat.h) -------------------------------------------
extern "C" int ogg_stream_init(ogg_stream_state *os, int serialno)
{
    if (!os) return -1;
    vbm_ogg_stream *h = nullptr;
    if (vbm_ogg_stream_create(&h, serialno)) return -1;
    os->impl = h;
    return 0;
}

extern "C" int ogg_stream_clear(ogg_stream_state *os)
{
    if (os && os->impl) vbm_ogg_stream_destroy((vbm_ogg_stream *)os->impl);
    if (os) os->impl = nullptr;
    return 0;
}

extern "C" int ogg_stream_packetin(ogg_stream_state *os, ogg_packet *op)
{
    if (!os || !os->impl || !op) return -1;
    return vbm_ogg_stream_packetin((vbm_ogg_stream *)os->impl, op->packet, op->bytes, (int)op->e_o_s, op->granulepos) ? -1 : 0;
}

static int page_out(ogg_stream_state *os, ogg_page *og, int flush)
{
    if (!os || !os->impl || !og) return 0;
    const uint8_t *page;
    long bytes;
    if (vbm_ogg_stream_pageout((vbm_ogg_stream *)os->impl, flush, &page, &bytes) != 1) return 0;
    const long hl = 27 + page[26];               // fixed header + segment table (doc/framing.html)
    og->header = const_cast<unsigned char *>(page);
    og->header_len = hl;
    og->body = const_cast<unsigned char *>(page) + hl;
    og->body_len = bytes - hl;
    return 1;
}

extern "C" int ogg_stream_pageout(ogg_stream_state *os, ogg_page *og) { return page_out(os, og, 0); }
extern "C" int ogg_stream_flush(ogg_stream_state *os, ogg_page *og) { return page_out(os, og, 1); }
extern "C" int ogg_page_eos(const ogg_page *og) { return og && og->header ? (og->header[5] & 4) != 0 : 0; }
