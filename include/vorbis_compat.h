/* vorbis_compat.h — the reference's OWN entry points for the per-block encode path, exported by
 * libvorbis_mi355x.so with the reference's names, argument meaning, return codes and public struct
 * layouts, so that a program written against <vorbis/codec.h> + <vorbis/vorbisenc.h> compiles and
 * links against this library without source changes on the encode side:
 *
 *     vorbis_encode_init_vbr / vorbis_encode_init          include/vorbis/vorbisenc.h:59, :157
 *     vorbis_analysis_init, vorbis_block_init              include/vorbis/codec.h:185, :175
 *     vorbis_analysis_headerout                            codec.h:187
 *     vorbis_analysis_buffer, vorbis_analysis_wrote        codec.h:192-193   lib/block.c:411-553
 *     vorbis_analysis_blockout                             codec.h:194       lib/block.c:557-812
 *     vorbis_analysis                                      codec.h:195       lib/analysis.c:29-63
 *     vorbis_bitrate_addblock, vorbis_bitrate_flushpacket  codec.h:197-198   lib/bitrate.c:73-252
 *     vorbis_block_clear, vorbis_dsp_clear                 codec.h:176-177
 *     vorbis_info_*, vorbis_comment_*                      codec.h:164-172
 *
 * Layouts: vorbis_info, vorbis_dsp_state, vorbis_block, vorbis_comment have the field order of
 * include/vorbis/codec.h:27-53, :58-85, :87-119, :141-149 (applications allocate them on the stack,
 * examples/encoder_example.c:47-52).  The three pointers the reference keeps opaque — vi->codec_setup,
 * vd->backend_state, vb->internal — point to this library's own objects.
 *
 * What runs where.  Every vorbis_dsp_state is one SLOT of a device-resident pool of streams of its
 * encoder class (one vbm_encoder + vbm_frontend, include/vorbis_mi355x.h).  vorbis_analysis_wrote
 * queues the samples; the first vorbis_analysis_blockout that cannot be answered from the stream's
 * queue uploads the queued samples of EVERY stream of the pool, runs one blockout round of the device
 * front end for all of them (envelope search, block carve-out, and the whole per-block path of
 * lib/mapping0.c:738-1322 behind it: at most one block per stream) and files each stream's block and
 * packet in its queue.  The following vorbis_analysis / vorbis_bitrate_addblock /
 * vorbis_bitrate_flushpacket calls hand that packet out with the reference's bookkeeping.  A server
 * that feeds S streams and then drains them therefore costs one batched device round per block
 * generation, not S — without any change to the per-stream call sequence.  Packets, block sequence,
 * granule positions and packet numbers are those of the reference's scalar build.
 *
 * Differences from the reference, all outside the data path:
 *   - vb->pcm is NULL: the block's PCM stays on the device (lib/block.c:653-698 copies it to the arena).
 *   - Carve-ahead: a round carves the next block of every pool stream that has one, not only of the
 *     asking stream.  Blocks and packets do not depend on when they are carved, with one exception:
 *     vorbis_analysis_wrote(vd,0) fits the end-of-stream extrapolation to the samples the buffer holds
 *     at that moment (lib/block.c:497-537).  A stream that drains its blocks before declaring the end
 *     (examples/encoder_example.c) or declares it before its first blockout (test/write_read.c:95-99)
 *     is not affected; one that leaves blocks un-asked-for WHILE another stream of the pool triggers
 *     a round may find them carved already.  vorbis_mi355x_ctl(VORBIS_MI355X_CARVE_AHEAD, 0) turns
 *     carve-ahead off: a round then holds the asking stream only, exactly the reference's order.
 *   - vorbis_encode_init_vbr / vorbis_encode_init pick a SHIPPED mode pack (libvorbisenc's setup code
 *     is out of scope): OV_EIMPL for a (channels, rate, quality) without one.
 *   - No decode side (vorbis_synthesis_*).
 * Thread safety: calls on different vorbis_dsp_states may come from different threads (a pool
 * serialises them); calls on one vorbis_dsp_state from one thread at a time, as in the reference.
 */
#ifndef VORBIS_COMPAT_H
#define VORBIS_COMPAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the part of <ogg/ogg.h> the encode API names (libogg is external to the reference and not in
 * this image; skipped when a real ogg.h was included first) ---------------------------------------- */
#ifndef _OGG_H
#define _OGG_H
typedef int64_t ogg_int64_t;
typedef int32_t ogg_int32_t;
typedef uint32_t ogg_uint32_t;
typedef int16_t ogg_int16_t;
typedef uint16_t ogg_uint16_t;

typedef struct {
  long endbyte;
  int endbit;
  unsigned char *buffer;
  unsigned char *ptr;
  long storage;
} oggpack_buffer;

typedef struct {
  unsigned char *packet;
  long bytes;
  long b_o_s;
  long e_o_s;
  ogg_int64_t granulepos;
  ogg_int64_t packetno;
} ogg_packet;

/* Page framing (libogg's ogg_stream_* / ogg_page_*; format: reference doc/framing.html).  ogg_stream_state
 * is opaque here: a handle to vbm_ogg_stream (include/vorbis_mi355x.h). */
typedef struct {
  unsigned char *header;
  long header_len;
  unsigned char *body;
  long body_len;
} ogg_page;
typedef struct {
  void *impl;
} ogg_stream_state;
int ogg_stream_init(ogg_stream_state *os, int serialno);
int ogg_stream_clear(ogg_stream_state *os);
int ogg_stream_packetin(ogg_stream_state *os, ogg_packet *op);
int ogg_stream_pageout(ogg_stream_state *os, ogg_page *og);
int ogg_stream_flush(ogg_stream_state *os, ogg_page *og);
int ogg_page_eos(const ogg_page *og);
#endif /* _OGG_H */

/* ---- include/vorbis/codec.h:27-149 ----------------------------------------------------------- */
typedef struct vorbis_info {
  int version;
  int channels;
  long rate;
  long bitrate_upper;
  long bitrate_nominal;
  long bitrate_lower;
  long bitrate_window;
  void *codec_setup;                /* this library: the encoder class (mode pack + stream pools) */
} vorbis_info;

typedef struct vorbis_dsp_state {
  int analysisp;
  vorbis_info *vi;
  float **pcm;                      /* NULL: the PCM ring lives on the device */
  float **pcmret;                   /* what vorbis_analysis_buffer returned last */
  int pcm_storage;
  int pcm_current;
  int pcm_returned;
  int preextrapolate;
  int eofflag;
  long lW;
  long W;
  long nW;
  long centerW;
  ogg_int64_t granulepos;
  ogg_int64_t sequence;
  ogg_int64_t glue_bits;
  ogg_int64_t time_bits;
  ogg_int64_t floor_bits;
  ogg_int64_t res_bits;
  void *backend_state;              /* this library: the stream's slot */
} vorbis_dsp_state;

struct alloc_chain;
typedef struct vorbis_block {
  float **pcm;                      /* NULL (see above) */
  oggpack_buffer opb;               /* buffer/ptr/endbyte describe the block's packet after vorbis_analysis */
  long lW;
  long W;
  long nW;
  int pcmend;
  int mode;
  int eofflag;
  ogg_int64_t granulepos;
  ogg_int64_t sequence;
  vorbis_dsp_state *vd;
  void *localstore;
  long localtop;
  long localalloc;
  long totaluse;
  struct alloc_chain *reap;
  long glue_bits;
  long time_bits;
  long floor_bits;
  long res_bits;
  void *internal;                   /* this library: the carved block (block type, packet) */
} vorbis_block;

struct alloc_chain {
  void *ptr;
  struct alloc_chain *next;
};

typedef struct vorbis_comment {
  char **user_comments;
  int *comment_lengths;
  int comments;
  char *vendor;
} vorbis_comment;

/* ---- codec.h:164-177 --------------------------------------------------------------------------- */
void vorbis_info_init(vorbis_info *vi);
void vorbis_info_clear(vorbis_info *vi);
int vorbis_info_blocksize(vorbis_info *vi, int zo);
void vorbis_comment_init(vorbis_comment *vc);
void vorbis_comment_add(vorbis_comment *vc, const char *comment);
void vorbis_comment_add_tag(vorbis_comment *vc, const char *tag, const char *contents);
char *vorbis_comment_query(vorbis_comment *vc, const char *tag, int count);
int vorbis_comment_query_count(vorbis_comment *vc, const char *tag);
void vorbis_comment_clear(vorbis_comment *vc);
int vorbis_block_init(vorbis_dsp_state *v, vorbis_block *vb);
int vorbis_block_clear(vorbis_block *vb);
void vorbis_dsp_clear(vorbis_dsp_state *v);
double vorbis_granule_time(vorbis_dsp_state *v, ogg_int64_t granulepos);
const char *vorbis_version_string(void);

/* ---- codec.h:185-198 --------------------------------------------------------------------------- */
/* 0, or 1 on failure (lib/block.c:306-344 returns _vds_shared_init's 1); here also when no HIP device
 * is present — there is no CPU path behind these entry points. */
int vorbis_analysis_init(vorbis_dsp_state *v, vorbis_info *vi);
int vorbis_analysis_headerout(vorbis_dsp_state *v, vorbis_comment *vc, ogg_packet *op, ogg_packet *op_comm,
                              ogg_packet *op_code);
/* host buffers of `vals` floats per channel, valid until the next call on v (lib/block.c:411-436) */
float **vorbis_analysis_buffer(vorbis_dsp_state *v, int vals);
/* vals > 0: the first `vals` samples of the buffers are the stream's next samples; vals <= 0: end of stream
 * (lib/block.c:482-553).  OV_EINVAL when more was written than asked for (:540-541) or after the end. */
int vorbis_analysis_wrote(vorbis_dsp_state *v, int vals);
/* 1: vb describes the stream's next block; 0: more PCM needed (or stream over) — lib/block.c:557-812 */
int vorbis_analysis_blockout(vorbis_dsp_state *v, vorbis_block *vb);
/* 0; OV_EINVAL when op != NULL on a managed-bitrate stream (lib/analysis.c:50-53) */
int vorbis_analysis(vorbis_block *vb, ogg_packet *op);
/* 0; -1 when the previous block was added and not flushed (VBR, lib/bitrate.c:92) */
int vorbis_bitrate_addblock(vorbis_block *vb);
/* 1 and *op filled (op->packet stays valid until the next blockout on vd), 0 when nothing is parked */
int vorbis_bitrate_flushpacket(vorbis_dsp_state *vd, ogg_packet *op);

/* ---- the reference's INTERNAL plugin seam for this path -------------------------------------------------------
 * lib/backends.h:121-128 (vorbis_func_mapping), instance mapping0_exportbundle lib/mapping0.c:1500-1506, table
 * _mapping_P lib/registry.c:42-44; vorbis_analysis reaches mapping0_forward through it: lib/analysis.c:45-46
 *     ret = _mapping_P[ci->map_type[ci->mode_param[vb->mode]->mapping]]->forward(vb)
 * A libvorbis that keeps its OWN vorbis_analysis_blockout (host PCM ring, envelope detector) and its own
 * vorbis_bitrate_* plugs the device in here and nowhere else: it replaces its table entry by
 * &mapping0_exportbundle_mi355x (or calls vbm_mapping0_forward where it called mapping0_forward).
 *   in   vb->vd        a vorbis_dsp_state made by THIS library's vorbis_analysis_init (its slot carries the stream's
 *                      encoder state — aoTuV's mblock / tblock / lW_block_mode ..., lib/codec_internal.h:85-92 — on the device)
 *        vb->pcm       host pointers [channels][vb->pcmend]: the block as lib/block.c:653-698 carves it (un-windowed)
 *        vb->lW, vb->W, vb->nW, vb->pcmend = blocksizes[W]
 *        vb->internal  the caller's vorbis_block_internal, whose leading members are (lib/codec_internal.h:42-49)
 *                      { float **pcmdelay; float ampmax; int blocktype; oggpack_buffer *packetblob[15]; } = the struct
 *                      below; blocktype as lib/block.c:620-638 sets it (BLOCKTYPE_IMPULSE 0 / PADDING 1 for short
 *                      blocks, TRANSITION 0 / LONG 1 for long ones)
 *   out  the block's packet (VBR: packetblob[PACKETBLOBS/2]; managed setups: the packet the stream's reservoirs chose) in
 *        vb->opb (buffer / ptr / endbyte, valid until the stream's next call) and, when vbi->packetblob[7] is not NULL,
 *        copied into that oggpack_buffer (its buffer grown with realloc, as libogg's own writer grows it), so that the
 *        caller's vorbis_bitrate_addblock / _flushpacket find it where mapping0_forward leaves it (lib/mapping0.c:1204-1313).
 *   returns 0, OV_EINVAL for a malformed block, OV_EFAULT for a device failure.
 * The blocks of one stream must come in stream order (the reference's contract too); blocks of different streams may
 * come from different threads.  One call = a batch of one through vbm_analysis_batch; the batched forms
 * (vbm_analysis_batch / _round, include/vorbis_mi355x.h) are what a caller with many streams uses. */
typedef struct vorbis_block_internal {
  float **pcmdelay;
  float ampmax;
  int blocktype;
  oggpack_buffer *packetblob[15];
} vorbis_block_internal;
typedef struct vorbis_func_mapping {
  void (*pack)(vorbis_info *, void *, oggpack_buffer *);
  void *(*unpack)(vorbis_info *, oggpack_buffer *);
  void (*free_info)(void *);
  int (*forward)(struct vorbis_block *vb);
  int (*inverse)(struct vorbis_block *vb, void *);
} vorbis_func_mapping;
int vbm_mapping0_forward(vorbis_block *vb);
extern const vorbis_func_mapping mapping0_exportbundle_mi355x;   /* forward = vbm_mapping0_forward; the header / decode members are NULL */

/* ---- include/vorbis/vorbisenc.h:59, :157 ---------------------------------------------------------- */
/* The three-step form of the same setup (reference include/vorbis/vorbisenc.h:96-219, lib/vorbisenc.c:977-1260), as
 * oggenc-style programs use it: vorbis_encode_setup_vbr / _managed choose the mode pack, vorbis_encode_ctl may look at it,
 * vorbis_encode_setup_init seals it (a vorbis_info whose setup is not sealed is refused by vorbis_analysis_init).  The
 * setup is a SHIPPED pack, so vorbis_encode_ctl answers the read requests from it and accepts only those changes that
 * change nothing:
 *   OV_ECTL_RATEMANAGE2_GET  filled from the pack (reservoir, bias, damping as vorbis_encode_setup_managed leaves them)
 *   OV_ECTL_RATEMANAGE2_SET  NULL on a VBR setup (what oggenc -q does: "no management"): 0; the values the setup already
 *                            has: 0; anything else OV_EIMPL
 *   OV_ECTL_LOWPASS_GET / _IBLOCK_GET / _COUPLING_GET   the pack's lowpass (kHz), 0., 1
 *   OV_ECTL_LOWPASS_SET / _IBLOCK_SET / _COUPLING_SET   the value the setup already has: 0; anything else OV_EIMPL
 *   the deprecated OV_ECTL_RATEMANAGE_* requests       OV_EIMPL
 * and, like the reference, refuses every change after vorbis_encode_setup_init with OV_EINVAL. */
struct ovectl_ratemanage2_arg {
    int management_active;
    long bitrate_limit_min_kbps;
    long bitrate_limit_max_kbps;
    long bitrate_limit_reservoir_bits;
    double bitrate_limit_reservoir_bias;
    long bitrate_average_kbps;
    double bitrate_average_damping;
};
#define OV_ECTL_RATEMANAGE_GET   0x10
#define OV_ECTL_RATEMANAGE_SET   0x11
#define OV_ECTL_RATEMANAGE_AVG   0x12
#define OV_ECTL_RATEMANAGE_HARD  0x13
#define OV_ECTL_RATEMANAGE2_GET  0x14
#define OV_ECTL_RATEMANAGE2_SET  0x15
#define OV_ECTL_LOWPASS_GET      0x20
#define OV_ECTL_LOWPASS_SET      0x21
#define OV_ECTL_IBLOCK_GET       0x30
#define OV_ECTL_IBLOCK_SET       0x31
#define OV_ECTL_COUPLING_GET     0x40
#define OV_ECTL_COUPLING_SET     0x41
int vorbis_encode_setup_vbr(vorbis_info *vi, long channels, long rate, float quality);
int vorbis_encode_setup_managed(vorbis_info *vi, long channels, long rate, long max_bitrate, long nominal_bitrate,
                                long min_bitrate);
int vorbis_encode_setup_init(vorbis_info *vi);
int vorbis_encode_ctl(vorbis_info *vi, int number, void *arg);
/* the comment header alone; op->packet is malloc'ed and belongs to the caller (lib/info.c:600-617) */
int vorbis_commentheader_out(vorbis_comment *vc, ogg_packet *op);
int vorbis_encode_init_vbr(vorbis_info *vi, long channels, long rate, float base_quality);
int vorbis_encode_init(vorbis_info *vi, long channels, long rate, long max_bitrate, long nominal_bitrate,
                       long min_bitrate);

/* ---- codec.h:221-238 ---------------------------------------------------------------------------- */
#define OV_FALSE      -1
#define OV_EOF        -2
#define OV_HOLE       -3
#define OV_EREAD      -128
#define OV_EFAULT     -129
#define OV_EIMPL      -130
#define OV_EINVAL     -131
#define OV_ENOTVORBIS -132
#define OV_EBADHEADER -133
#define OV_EVERSION   -134
#define OV_ENOTAUDIO  -135
#define OV_EBADPACKET -136
#define OV_EBADLINK   -137
#define OV_ENOSEEK    -138

/* ---- knobs of this library (no reference counterpart) ------------------------------------------- */
#define VORBIS_MI355X_POOL_STREAMS 1   /* arg int*: slots per device pool, read when a class makes a pool
                                          (default 64, environment VORBIS_MI355X_POOL_STREAMS) */
#define VORBIS_MI355X_CARVE_AHEAD  2   /* arg int*: 1 (default) a round carves for every pool stream, 0 only
                                          for the asking one */
#define VORBIS_MI355X_DATA_DIR     3   /* arg const char*: directory of common.vpk / mode_*.vpk (default: data/
                                          beside the library, environment VORBIS_MI355X_DATA) */
#define VORBIS_MI355X_ROUNDS       4   /* arg long long* (out): device rounds run so far, all pools */
#define VORBIS_MI355X_TIMES        6   /* arg double[8] (out): seconds of host time, summed over all threads, spent in 0 the copy of
                                          vorbis_analysis_wrote, 1 staging copies, 2 uploads (H2D, append kernels, waits), 3 device
                                          rounds, 4 packet compaction + D2H, 5 filing packets per stream */
#define VORBIS_MI355X_DEFER_BLOCKS 5   /* arg int*: 0 (default) vorbis_analysis_blockout hands out every block the stream has, like the
                                          reference; 1 throughput mode: one block per write (two every third write) unless the
                                          stream's buffer is half full or the stream is ending — a stream inside a run of short
                                          blocks then no longer forces up to eight extra device rounds per write on its whole
                                          pool.  Same packets, later delivery.  Environment VORBIS_MI355X_DEFER_BLOCKS. */
int vorbis_mi355x_ctl(int request, void *arg);

#ifdef __cplusplus
}
#endif
#endif /* VORBIS_COMPAT_H */
