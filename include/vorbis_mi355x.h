/* vorbis_mi355x — C ABI of the MI355X-native batched Vorbis (aoTuV) encode path.
 *
 * Drop-in boundary for the per-block encode path of spvkgn/vorbis-aotuv-lancer
 * (libvorbis 1.3.7 + aoTuV b6.03): vorbis_analysis() -> mapping0_forward() -> {window,
 * mdct_forward, drft_forward, _vp_* psy, floor1_fit/encode, couple/quantise, residue VQ}
 * -> vorbis_bitrate_addblock()/flushpacket().  Plain pointers and sizes only; device
 * pointers are HIP device addresses, `stream` is a hipStream_t passed as void*.
 *
 * Every entry point names the reference interface it replaces (file:line relative to
 * the reference tree).  All functions return 0 on success, a negative VBM_E* otherwise,
 * and never fall back to a CPU path: without a HIP device they fail with VBM_ENODEV.
 *
 * Results are bit-identical to the reference's SCALAR C path (the `#else` branches of
 * `#ifdef __SSE__`), see DESIGN.md.
 */
#ifndef VORBIS_MI355X_H
#define VORBIS_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBM_OK       0
#define VBM_EINVAL  (-131) /* same value as OV_EINVAL, include/vorbis/codec.h:228 */
#define VBM_EFAULT  (-129) /* OV_EFAULT */
#define VBM_EIMPL   (-130) /* OV_EIMPL  */
#define VBM_ENODEV  (-1000)
#define VBM_EHIP    (-1001)

/* library / device ------------------------------------------------------------------ */
const char *vbm_version(void);
/* number of HIP devices visible; <0 on error.  Does not create a context. */
int vbm_device_count(void);
/* last HIP error string seen by this library on the calling thread ("" if none) */
const char *vbm_last_error(void);

/* ---- MDCT ------------------------------------------------------------------------
 * vbm_mdct_plan replaces mdct_lookup (lib/mdct.h:55-73) + mdct_init (lib/mdct.c:54-92):
 * trig tables are computed on the host with libm in double and rounded to float exactly
 * as lib/mdct.c:67-76 does, then kept device-resident.  It also carries the two
 * rising half-windows the block size can meet (lib/window.c:29-2122; b->window[W],
 * lib/block.c:218-219) so that windowing fuses into the transform.
 *
 *   n        transform size: 256, 512, 1024, 2048 or 4096 (the block sizes of the shipped mode families)
 *   short_n  size of the short block of the same mode pair (for n==2048: 256); the
 *            left/right window halves of a long block next to a short one use it
 *   win_n / win_short: host pointers to the rising half-windows (n/2 and short_n/2
 *            floats); may be NULL if only vbm_mdct_forward_batch() will be used.
 */
typedef struct vbm_mdct_plan vbm_mdct_plan;

int vbm_mdct_plan_create(vbm_mdct_plan **plan, int n, int short_n,
                         const float *win_n, const float *win_short);
void vbm_mdct_plan_destroy(vbm_mdct_plan *plan);
/* host copy of the trig table (n + n/4 floats) — for parity tests */
const float *vbm_mdct_plan_trig(const vbm_mdct_plan *plan);

/* Batched mdct_forward(mdct_lookup*, float *in, float *out) — lib/mdct.c:1799 / lib/mdct.h:78.
 *   d_in : nblocks x n   floats, block-major, device
 *   d_out: nblocks x n/2 floats, device
 */
int vbm_mdct_forward_batch(const vbm_mdct_plan *plan, const float *d_in, float *d_out,
                           long nblocks, void *stream);

/* Batched _vorbis_apply_window(pcm, winno, blocksizes, lW, W, nW) + mdct_forward —
 * lib/window.c:2137 + lib/mdct.c:1799 as called from mapping0_forward, lib/mapping0.c:825-843.
 *   d_pcm    : nblocks x n floats (un-windowed block PCM, block-major)
 *   d_wflags : one byte per block, bit0 = lW, bit1 = nW (vb->lW / vb->nW; only read for
 *              long blocks; NULL means every neighbour is long).  Short blocks always
 *              use the full short window (lib/window.c:2139-2140).
 */
int vbm_window_mdct_batch(const vbm_mdct_plan *plan, const float *d_pcm, float *d_out,
                          const uint8_t *d_wflags, long nblocks, void *stream);

/* Batched tonal-estimation transform of mapping0_forward loop A (lib/mapping0.c:825-888):
 * _vorbis_apply_window + drft_forward (lib/smallft.c:6770, FFTPACK scalar path) + the
 * log-power conversion
 *     logfft[0] = scale_dB + todB(re0) + .345 ;  logfft[k] = scale_dB + .5f*todB(re_k^2+im_k^2) + .345
 * and the per-block maximum local_ampmax = min(0, max_k logfft[k]) (:862-888).
 *   d_pcm          : nblocks x n floats (un-windowed)
 *   d_logfft       : nblocks x n/2 floats
 *   d_local_ampmax : nblocks floats
 */
int vbm_window_fft_log_batch(const vbm_mdct_plan *plan, const float *d_pcm, float *d_logfft,
                             float *d_local_ampmax, const uint8_t *d_wflags, long nblocks, void *stream);
/* host copy of the FFTPACK twiddle table (n floats = drft_lookup.trigcache + n) — for parity tests */
const float *vbm_mdct_plan_fft_twiddles(const vbm_mdct_plan *plan);

/* ---- encoder setup ----------------------------------------------------------------------
 * vbm_setup replaces codec_setup_info + the looks of private_state that vorbis_analysis_init()
 * builds (lib/block.c:181-331: _vp_psy_init, floor1_look, res0_look, vorbis_book_init_encode,
 * mdct_init, drft_init).  It is created from two VPK packs shipped with the package:
 *   common_vpk : static tables (windows, ATH, tone masks, dB lookup …)   data/common.vpk
 *   mode_vpk   : the (channels, rate, quality) class, i.e. the output of
 *                vorbis_encode_init_vbr (lib/vorbisenc.c:977)           data/mode_*.vpk
 * Creation is host-only; tables are uploaded when an encoder is created on a device. */
typedef struct vbm_setup_handle vbm_setup_handle;
int vbm_setup_create(vbm_setup_handle **setup, const char *common_vpk, const char *mode_vpk);
void vbm_setup_destroy(vbm_setup_handle *setup);
/* Introspection of the derived look tables by name ("psy/3/ath", "floor/1/sorted_index",
 * "book/7/codelist", "residue/0/partbook", "info", …) for parity checks; *kind is one of
 * 'f' float32, 'i' int32, 'u' uint32, 'b' int8, 'd' float64.  The pointer stays valid until the
 * next call on the same thread (scalars) or until the setup is destroyed (arrays). */
int vbm_setup_table(const vbm_setup_handle *setup, const char *name, const void **data, long *count,
                    char *kind);

/* ---- batched analysis ---------------------------------------------------------------------
 * vbm_encoder holds, on the device, the carried per-stream encoder state of `nstreams` streams
 * (private_state's aoTuV fields mblock/tblock/lownoise_compand_level/lW_block_mode/lW_no/impadnum,
 * lib/codec_internal.h:85-92; vorbis_block_internal.ampmax and vorbis_look_psy_global.ampmax)
 * plus the workspace for batches of up to `max_batch` blocks.
 *
 * vbm_analysis_batch() is the batched form of the reference's per-block sequence
 *     vorbis_analysis(vb, NULL)            lib/analysis.c:29   -> mapping0_forward lib/mapping0.c:738
 *     vorbis_bitrate_addblock(vb)          lib/bitrate.c:73    (VBR: parks the block)
 *     vorbis_bitrate_flushpacket(vd, &op)  lib/bitrate.c:229   (VBR: packetblob[PACKETBLOBS/2])
 * for `nsb` blocks that share one block type, at most one block per stream and call, in the
 * stream's block order (blocks of one stream are order dependent).
 *   block_mode   0 impulse short, 1 padding short, 2 transition long, 3 long
 *                (= blocktype | W<<1, lib/mapping0.c:768-775; vbi->blocktype from lib/block.c:620-638)
 *   stream_ids   host array [nsb]: which stream each block belongs to
 *   wflags       host array [nsb]: bit0 = vb->lW, bit1 = vb->nW
 *   d_pcm        device, [nsb][channels][blocksize] floats: vb->pcm as vorbis_analysis_blockout
 *                hands it over (un-windowed, lib/block.c:653-698)
 *   d_packets    device, [nsb][vbm_encoder_max_packet_bytes()] bytes: op->packet of every block
 *   d_packet_bytes device, [nsb] ints: op->bytes, or -1 if a packet outgrew max_packet_bytes — check it
 *                before using the row (its bytes are then incomplete)
 * A stream id may appear ONCE per call: the blocks of a stream are order dependent and two of them in one
 * batch would race on the carried state; duplicates are rejected with VBM_EINVAL.
 * Managed-bitrate setups (15 packetblobs + reservoirs) go through the same calls: see "Managed bitrate" below. */
typedef struct vbm_encoder vbm_encoder;
int vbm_encoder_create(vbm_encoder **enc, vbm_setup_handle *setup, int nstreams, int max_batch);
void vbm_encoder_destroy(vbm_encoder *enc);
int vbm_encoder_reset(vbm_encoder *enc); /* back to the state after vorbis_analysis_init */
int vbm_encoder_max_packet_bytes(const vbm_encoder *enc);
int vbm_analysis_batch(vbm_encoder *enc, int block_mode, int nsb, const int *stream_ids,
                       const uint8_t *wflags, const float *d_pcm, uint8_t *d_packets,
                       int *d_packet_bytes, void *stream);
/* Two-stream form of vbm_analysis_batch.  The FRONT half of the path (window, MDCT, FFT,
 * psychoacoustics, _vp_offset_and_mix: every kernel that reads or writes the carried per-stream
 * state) is ordered on `stream_front`, the BACK half (floor1_fit/_encode,
 * _vp_couple_quantize_normalize, residue VQ, packet assembly) on `stream_back`; d_pcm must be ready
 * on stream_front, packets are ready on stream_back.  Consecutive calls use alternate workspaces,
 * so with two different streams the back half of one call runs while the front half of the next
 * one does: the back half's few-wavefront kernels (floor fit) and the front half's wide ones
 * share the GPU.  Results are those of vbm_analysis_batch.  stream_back == stream_front is
 * vbm_analysis_batch. */
int vbm_analysis_batch2(vbm_encoder *enc, int block_mode, int nsb, const int *stream_ids,
                        const uint8_t *wflags, const float *d_pcm, uint8_t *d_packets,
                        int *d_packet_bytes, void *stream_front, void *stream_back);

/* One ROUND: blocks of all four block types at once (what one pass of vorbis_analysis_blockout over
 * all streams yields).  counts[m] blocks of type m; stream_ids / wflags (host) grouped by type, type
 * 0 first; d_pcm: the blocks of type m start at float offset (number of blocks of lower types) *
 * channels * blocksizes[1] and lie [count][channels][N_m].  A stream may appear once per round.
 * The four batches run side by side on internal HIP streams (forked from and joined back to
 * `stream`), so the few
 * short blocks of a round cost no more than the long-block batch beside them.  Packets come back in
 * the grouped order: d_packets[k][max_packet_bytes], d_packet_bytes[k]. */
int vbm_analysis_round(vbm_encoder *enc, const int *counts, const int *stream_ids, const uint8_t *wflags,
                       const float *d_pcm, uint8_t *d_packets, int *d_packet_bytes, void *stream);
/* Rounds with a deferred join.  vbm_analysis_round_begin enqueues a round like vbm_analysis_round but does not
 * make `stream` wait for its batches; a later round only waits for those batches of the round before it that
 * its own streams were part of (the blocks of a stream stay in order), so the handful of short blocks of one
 * round runs beside the long-block batch of the previous one.  Outputs are complete on `stream` after
 * vbm_analysis_round_join.  Buffers handed to a round (d_pcm, outputs) may be reused once `stream` has passed
 * vbm_analysis_round_wait_workspace before the round after the next, or a join.  (Reference: the application
 * loop of examples/encoder_example.c:211-235 run for many streams at once; the packets do not depend on how the
 * rounds are scheduled.) */
int vbm_analysis_round_begin(vbm_encoder *enc, const int *counts, const int *stream_ids, const uint8_t *wflags,
                             const float *d_pcm, uint8_t *d_packets, int *d_packet_bytes, void *stream);
int vbm_analysis_round_join(vbm_encoder *enc, void *stream);
/* ... leaving the newest big batch (>= 1024 blocks of one type) pending: only its front half is waited for; its
 * back half (floor, couple/quantise, packets: nothing the next round reads) then runs beside the front half of
 * the next write's big batch.  Its outputs are complete on `stream` after the next lazy join or a full join. */
int vbm_analysis_round_join_lazy(vbm_encoder *enc, void *stream);
int vbm_analysis_round_wait_workspace(vbm_encoder *enc, void *stream);
/* Stage intermediates of the LAST batch as block-major rows ([channel-block][rows]) for parity
 * tests: "mdct_raw" "logfft" "logmdct" "noise" "tone" "logmask" "mdct" "epeak" "npeak" "post"
 * "floor_out" "residue", and the vectors "local_ampmax" "global_ampmax" "post_valid" "nonzero"
 * "poste" "packet_bytes".  d_out may be NULL to query rows/kind ('f' float32, 'i' int32). */
int vbm_encoder_fetch(vbm_encoder *enc, const char *name, void *d_out, long *rows, char *kind, void *stream);

/* Managed bitrate (setups made by vorbis_encode_init: reference lib/vorbisenc.c:997-1070).  With such a
 * setup vbm_analysis_batch / _batch2 / _round and the front end do per block what vorbis_analysis(vb, NULL)
 * + vorbis_bitrate_addblock + vorbis_bitrate_flushpacket do (lib/analysis.c:30-62, lib/bitrate.c:73-252):
 * all PACKETBLOBS (15) packets of the block are produced (lib/mapping0.c:1097-1181, loop :1204), the
 * stream's reservoirs pick one, and that packet — truncated to the ceiling or zero-padded to the floor when
 * max / min rates are set — is the one handed out.  Nothing in the call sequence changes.  For parity tests:
 * packetblob k of the LAST batch as produced, before the choice (d_packets [nsb][max_packet_bytes],
 * d_packet_bytes [nsb]; either may be NULL), and the vector "choice" of vbm_encoder_fetch. */
int vbm_encoder_fetch_blob(vbm_encoder *enc, int k, uint8_t *d_packets, int *d_packet_bytes, void *stream);

/* Sub-batches.  The transforms run on the whole batch; the stages after them run as `n` slices of
 * the batch (multiples of 64 stream-blocks), each on an internal HIP stream forked from and joined
 * back to the caller's stream with events, so that the serial few-wavefront kernels of one slice
 * overlap with the wide kernels of the others.  Results do not depend on n.  Default 1
 * (environment VBM_SUB_BATCHES overrides at create); n = 1 launches everything on the caller's
 * stream.  In the two-stream form (vbm_analysis_batch2 with two different streams) the slices apply to the
 * back half only: the first on stream_back, the others on internal streams joined to it.  Measured on MI355X:
 * every HIP stream beyond four shares a hardware queue with another and costs more than the slices gain, so 1
 * stays the default. */
int vbm_encoder_set_sub_batches(vbm_encoder *enc, int n);
int vbm_encoder_sub_batches(const vbm_encoder *enc);

/* ---- stream front end --------------------------------------------------------
 * What libvorbis does between the application's PCM and vorbis_analysis(), for all `nstreams` of
 * an encoder at once and on the device:
 *   vorbis_analysis_buffer + vorbis_analysis_wrote   (reference include/vorbis/codec.h:192-193,
 *                                                     lib/block.c:405-553: pre_amplitude, LPC
 *                                                     extrapolation of stream start and end)
 *   vorbis_analysis_blockout                          (codec.h:194, lib/block.c:557-812) with the
 *                                                     envelope detector that picks the block sizes
 *                                                     (lib/envelope.c:101-728)
 * followed by vbm_analysis_batch on the blocks that came out.  Usage, mirroring
 * examples/encoder_example.c:190-235:
 *     vbm_frontend_write(fe, d_pcm, vals, q);                       // every stream gets `vals` samples
 *     do vbm_frontend_encode_round(fe, d_pkt, d_len, info, &n, q);  // <= 1 block per stream per round
 *     while (n > 0);
 *     ... vbm_frontend_finish(fe, ids, k, q) = vorbis_analysis_wrote(vd, 0), then rounds until n == 0.
 * Rounds may be deferred (a stream inside a run of short blocks yields up to 8 blocks per 1024
 * samples; blocks and packets do not depend on when the rounds run), but drain completely before
 * vbm_frontend_finish: like the reference (lib/block.c:531-541) it fits the end-of-stream LPC to the
 * samples the buffer holds at that moment.
 * The block sequence (lW, W, nW, block type, granulepos, packetno, e_o_s) and the packets are those
 * of the reference's scalar build for the same PCM.  d_pcm: device float, [nstreams][channels][vals].
 * encode_round: packets of the round in d_packets[k][max_packet_bytes] / d_packet_bytes[k],
 * k < *nblocks, described by info[k] (host array of nstreams entries); blocks are grouped by block
 * type, streams ascending inside a group.  The call synchronises `stream` once (it reads the
 * block decisions back to choose the batches).  Errors: VBM_EINVAL for writes that would overrun
 * the PCM buffer (the reference's OV_EINVAL, lib/block.c:540) or follow finish. */
typedef struct vbm_frontend vbm_frontend;
typedef struct vbm_packet_info {
    int stream;                 /* stream index */
    int block_mode;             /* 0 impulse, 1 padding, 2 transition, 3 long */
    int lW, W, nW;              /* vb->lW, vb->W, vb->nW */
    int eos;                    /* op.e_o_s */
    long long granulepos;       /* op.granulepos */
    long long packetno;         /* op.packetno (audio packets start at 3) */
} vbm_packet_info;
int vbm_frontend_create(vbm_frontend **fe, vbm_encoder *enc);
void vbm_frontend_destroy(vbm_frontend *fe);
int vbm_frontend_reset(vbm_frontend *fe);
int vbm_frontend_write(vbm_frontend *fe, const float *d_pcm, int vals, void *stream);
int vbm_frontend_finish(vbm_frontend *fe, const int *stream_ids, int n, void *stream);
/* Streams that do not move in lock step: write `vals` samples to the listed streams only (d_pcm:
 * [n][channels][vals], one entry per listed stream, a stream at most once per call), and start a
 * new stream in listed slots (state of vorbis_analysis_init for the front end and the encoder;
 * typically after the slot's previous stream has delivered its e_o_s packet). */
int vbm_frontend_write_streams(vbm_frontend *fe, const int *stream_ids, int n, const float *d_pcm, int vals,
                               void *stream);
/* The same with the caller's own layout: channel c of stream_ids[k] at pcm + (by_slot ? stream_ids[k] : k) * stream_stride
 * + c * ch_stride floats (ch_stride >= vals).  pcm may be host memory the device can read (hipHostMalloc): the append
 * kernel fetches it over the bus itself, no staging copy and no separate upload (what the drop-in shim's
 * vorbis_analysis_buffer hands out, reference lib/block.c:405-436).  Returns when the samples have been taken. */
int vbm_frontend_write_streams_strided(vbm_frontend *fe, const int *stream_ids, int n, const float *pcm, int vals,
                                       long stream_stride, long ch_stride, int by_slot, void *stream);
int vbm_frontend_restart_streams(vbm_frontend *fe, const int *stream_ids, int n, void *stream);
/* Buffer occupancy, for callers that do not drain completely after every write (a stream inside a
 * run of short blocks yields up to 8 blocks per 1024 samples, each in its own round): the most
 * samples any stream holds now, and the occupancy a write may not exceed (VBM_EINVAL beyond it).  The buffer
 * is 24 long blocks per channel (environment VBM_FE_BUFFER_BLOCKS, 8..64, read at create): 3 of them reserved
 * for the end-of-stream padding, a third for the origin of the samples to climb before they are moved back to the
 * start (one copy per several blocks where the reference memmoves after every block), the rest is slack that lets
 * such a stream fall behind and catch up, instead of forcing extra rounds on every write.  (The reference grows its buffer on demand: lib/block.c:424-430.) */
int vbm_frontend_max_buffered(const vbm_frontend *fe);
int vbm_frontend_capacity(const vbm_frontend *fe);
int vbm_frontend_encode_round(vbm_frontend *fe, uint8_t *d_packets, int *d_packet_bytes,
                              vbm_packet_info *info, int *nblocks, void *stream);
/* One round for the listed streams only: every other stream is left alone (no block, no state change), as if
 * its application had not asked vorbis_analysis_blockout yet.  Outputs as vbm_frontend_encode_round. */
int vbm_frontend_encode_round_streams(vbm_frontend *fe, const int *stream_ids, int n, uint8_t *d_packets,
                                      int *d_packet_bytes, vbm_packet_info *info, int *nblocks, void *stream);
/* Packets for a host consumer: the rows d_packets[k][max_packet_bytes] with d_packet_bytes[k] bytes used (k < n)
 * as one byte run in d_out, packet k at d_offsets[k] (4-byte aligned, exclusive prefix sum of the padded
 * lengths), d_offsets[n] = bytes of d_out used; negative lengths count as 0.  d_out needs n * max_packet_bytes
 * bytes in the worst case, d_offsets n + 1 entries.  One D2H copy of d_offsets[n] bytes then replaces n row
 * copies (what vorbis_bitrate_flushpacket's ogg_packet needs on the host, reference lib/bitrate.c:229-252). */
int vbm_packets_compact(const uint8_t *d_packets, const int *d_packet_bytes, int n, int max_packet_bytes,
                        uint8_t *d_out, long long *d_offsets, void *stream);
/* Up to max_rounds rounds in one call (deferred joins: vbm_analysis_round_begin), results complete on `stream`
 * when the call's work has run: packets / lengths / infos of all rounds, compact, in round order;
 * round_blocks[r] = blocks of round r, *nrounds = rounds that produced blocks.  Rounds stop when one produces
 * nothing, when fewer than `nstreams` of the cap_blocks output slots are left, or — after min_rounds — as
 * soon as every stream could take `headroom` more samples with its buffer at most half full.  Same packets
 * as any other schedule of vbm_frontend_encode_round calls. */
int vbm_frontend_encode_rounds(vbm_frontend *fe, int min_rounds, int max_rounds, int headroom,
                               uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info, int cap_blocks,
                               int *round_blocks, int *nrounds, void *stream);
/* The same with a lazy join (vbm_analysis_round_join_lazy): the outputs of the call's big batch are complete on
 * `stream` only after the NEXT vbm_frontend_encode_rounds_lazy call, or vbm_frontend_join; all other outputs as
 * above.  The caller keeps the output buffers of a call untouched until then.  For throughput: the back half of
 * one write's long-block batch overlaps the next write. */
int vbm_frontend_encode_rounds_lazy(vbm_frontend *fe, int min_rounds, int max_rounds, int headroom,
                                    uint8_t *d_packets, int *d_packet_bytes, vbm_packet_info *info, int cap_blocks,
                                    int *round_blocks, int *nrounds, void *stream);
int vbm_frontend_join(vbm_frontend *fe, void *stream);

/* Rounds built on the device: the throughput form of the loop above with NO host in it.  The device decides which
 * streams deliver a block (k_fe_classify), assigns them lanes (k_fe_plan), carves the blocks and runs the per-block
 * path on device-resident counts; the call only enqueues work and returns.  Nothing is read back: the host learns
 * what came out from the outputs, when it chooses to look.
 *   Layout: a round has vbm_device_round_lanes() output slots ("lanes").  Block type m owns a fixed region of
 *   them (the impulse, padding and transition blocks a quarter of the stream count each, then the long blocks with
 *   one lane per stream — half of them in the later rounds of a call, whose long blocks are those of streams catching
 *   up); inside a region the blocks follow in ascending stream order.  A stream whose type's region is
 *   full keeps its block for the next round (rounds may be deferred: blocks and packets do not depend on when they
 *   run); a stream that has fallen behind its input delivers a block in every round of a call until it has caught up.
 *   nrounds rounds per call; round r writes
 *     d_packets      [r][lane][max_packet_bytes]   (may be NULL)
 *     d_packet_bytes [r][lane]   length, -2 = no block in this lane, -1 = packet outgrew the buffer
 *     d_info         [r][lane]   vbm_packet_info of the lane's block (stream = -1: none); device or pinned host memory
 *     d_counts       [r][4]      blocks of each type (device memory: the round's kernels read it while they run)
 *   all complete on `stream` when the call's work has run (lazy = 1: the long-block batch of the call only after
 *   the next call or vbm_frontend_join, as vbm_frontend_encode_rounds_lazy; lazy = 2: nothing is tied to `stream` by
 *   the call — a consumer calls vbm_frontend_join(fe, its_stream) before it reads, so the stream that feeds PCM never
 *   waits for packets).  The buffers belong to the call until then.
 *   The encoder must have been created with max_batch >= vbm_device_round_lanes(setup, nstreams).
 * Mixing with the host-built rounds above is allowed (e.g. to drain completely before vbm_frontend_finish): the first
 * such call waits for the front end's stream and fetches the buffer fills it needs.  While only device-built rounds
 * run, vbm_frontend_write does not check the buffer fill: the caller keeps rounds and writes in balance (two rounds
 * per 1024-sample write keep every stream ahead of its input at the block-switching rates of music; the fill can be
 * watched with vbm_frontend_max_buffered, which synchronises). */
int vbm_device_round_lanes(const vbm_setup_handle *setup, int nstreams);
int vbm_frontend_encode_rounds_device(vbm_frontend *fe, int nrounds, uint8_t *d_packets, int *d_packet_bytes,
                                      vbm_packet_info *d_info, int *d_counts, int lazy, void *stream);
/* The block type every stream had ready when the newest device-built round was planned (-1: none), [nstreams] bytes,
 * copied to `out` (host) on `stream` after the round's outputs: a stream with a type here and no record in the round
 * found its type's lane region full and delivers the block in the next round. */
int vbm_frontend_round_types(vbm_frontend *fe, signed char *out, void *stream);
/* running totals of the device-built rounds: out[0..3] blocks of type 0..3, out[4] samples all streams advanced by,
 * out[5] writes the device refused because a stream's buffer was full (must stay 0: the refused samples are lost —
 * the device-side counterpart of vbm_frontend_write's VBM_EINVAL).  Synchronises the front end's stream. */
int vbm_frontend_device_stats(vbm_frontend *fe, unsigned long long *out);

/* ---- stream wrapper, host only ------------------------------------------------
 * vbm_header_packets = vorbis_analysis_headerout (reference lib/info.c:636-717): the identification,
 * comment and setup header packets of a setup, concatenated in buf with their sizes in lens[3]
 * (buf == NULL: sizes only).  vendor NULL = the string of the reference's scalar build
 * (lib/info.c:43).  They are packets 0..2 of a stream; the audio packets of
 * vbm_frontend_encode_round follow with packetno 3...
 * vbm_ogg_stream_* = libogg's ogg_stream_packetin / ogg_stream_pageout / ogg_stream_flush (libogg is
 * external to the reference; page format per the reference's doc/framing.html): packetin queues a
 * packet with the granulepos / e_o_s of its vbm_packet_info; pageout returns 1 and a pointer to a
 * complete page (valid until the next call) when one is due, 0 otherwise; flush != 0 forces out
 * what is queued (the reference application flushes after the three headers,
 * examples/encoder_example.c:150-157). */
int vbm_header_packets(const vbm_setup_handle *setup, const char *vendor, const char *const *comments,
                       int ncomments, uint8_t *buf, long cap, long *lens);
/* The comment header alone (reference vorbis_commentheader_out, lib/info.c:600-617); buf == NULL: size query. */
int vbm_comment_packet(const char *vendor, const char *const *comments, int ncomments, uint8_t *buf, long cap, long *len);
typedef struct vbm_ogg_stream vbm_ogg_stream;
int vbm_ogg_stream_create(vbm_ogg_stream **os, int serialno);
void vbm_ogg_stream_destroy(vbm_ogg_stream *os);
int vbm_ogg_stream_packetin(vbm_ogg_stream *os, const uint8_t *packet, long bytes, int e_o_s, long long granulepos);
int vbm_ogg_stream_pageout(vbm_ogg_stream *os, int flush, const uint8_t **page, long *bytes);

/* Per-stage timing of vbm_analysis_batch: HIP events are recorded between the pipeline's kernels,
 * on the stream each kernel is launched on, for the next `max_calls` calls; profile_end waits for
 * the device and returns the summed milliseconds per stage over all launches
 * (vbm_encoder_stage_count() entries, names from vbm_encoder_stage_name) and the number of calls
 * covered.  The first three stages are launched once per call, the others once per sub-batch. */
int vbm_encoder_profile_begin(vbm_encoder *enc, int max_calls);
int vbm_encoder_profile_end(vbm_encoder *enc, float *stage_ms, int *ncalls);
/* stream-blocks covered by the stage times gathered since vbm_encoder_profile_begin (for a round — vbm_analysis_round*,
 * the front end — the stages of its largest batch are the ones timed); read before vbm_encoder_profile_end */
long long vbm_encoder_profile_blocks(const vbm_encoder *enc);
int vbm_encoder_stage_count(void);
const char *vbm_encoder_stage_name(int k);

/* Host-only table builders (no device needed): the lookup tables the plans upload, for
 * integrators and for CPU-side parity checks.
 *   vbm_host_mdct_trig     n + n/4 floats  (mdct_init, lib/mdct.c:67-76)
 *   vbm_host_fft_twiddles  n floats        (drfti1,   lib/smallft.c:5576-5644; = trigcache + n) */
int vbm_host_mdct_trig(int n, float *out);
/*   vbm_host_book_lattice  {quantvals, minval, delta} of a maptype-1 codebook header, as vorbis_book_init_encode
 *                          derives them (lib/sharedbook.c:303-317: _book_maptype1_quantvals, _float32_unpack) */
int vbm_host_book_lattice(long q_min, long q_delta, long entries, int dim, int *out);
int vbm_host_fft_twiddles(int n, float *out);

/* Bench helper: time `iters` back-to-back launches of vbm_window_mdct_batch with HIP
 * events recorded on `stream` (the stream the kernel runs on).  *ms_total receives the
 * elapsed milliseconds for all iters. */
int vbm_window_mdct_time(const vbm_mdct_plan *plan, const float *d_pcm, float *d_out,
                         const uint8_t *d_wflags, long nblocks, int iters, void *stream,
                         float *ms_total);

/* ---- test instrumentation --------------------------------------------------------------------------------
 * The path is spread over several internal HIP streams; events order them.  These calls make a missing edge show
 * deterministically (tests/test_ordering_gpu.py); they change timing and scratch contents only, never results.
 *   vbm_debug_set_delay      a kernel that spins for `usec` microseconds (<= 100000) is put in front of the work at
 *                            every point of the host code whose bit is set in `mask` (points: csrc/vbm_internal.h,
 *                            enum vbm_delay_point); process-wide; usec = 0 switches it off
 *   vbm_debug_poison_workspace  fills the scratch arrays of encoder workspace w (-1: all) with `byte`, device idle
 *   vbm_debug_poison_frontend   the same for the front end's block buffers, search spectra and round lists */
int vbm_debug_set_delay(unsigned mask, int usec);
int vbm_debug_poison_workspace(vbm_encoder *enc, int w, int byte);
int vbm_debug_poison_frontend(vbm_frontend *fe, int byte);

#ifdef __cplusplus
}
#endif
#endif /* VORBIS_MI355X_H */
