// Device-side view of one homogeneous batch: `nsb` stream-blocks of the same block_mode
// (0 impulse, 1 padding, 2 transition, 3 long — reference lib/mapping0.c:768-775), each with
// `ch` channels.  The serial stages of the encode path run one LANE per channel-block
// (psy, floor) or per stream-block (couple/quantise, residue, bit packing): 64 independent
// blocks per wavefront, all walking the same table-driven loops in lockstep.  To keep those
// accesses coalesced every per-bin array is stored bin-major ("transposed"):
//     X[bin][L]   with L = channel-block count rounded up to 64,
// so lane l reads X[bin*L + l] — 256 contiguous bytes per wave-instruction.
// Block-major arrays (pcm, mdct_bm, logfft_bm) are the interface to the wave-per-block
// transform kernels; transposes go through LDS tiles (util_kernels.hip).
#pragma once
#include <stdint.h>
#include "setup.h"

struct vbm_stream_state {           // per-encoder, S streams (reference lib/codec_internal.h:85-92,
    int S, ch;                      //  vorbis_block_internal.ampmax, vorbis_look_psy_global.ampmax)
    int Lc;                         // S*ch rounded up to 64
    size_t slab_words;              // words per 64-column tile of the carried buffers: (2048+256)*64
    float *mblock;                  // tile layout [col>>6][2048 rows][64]: lW logmdct buffer (lastmdct)
    float *tblock;                  // = mblock + 2048*64 within each tile: temporal masking buffer (tempmdct)
    float *lowcomp;                 // [Lc]        lownoise_compand_level
    float *g_ampmax;                // [S]         psy_g_look->ampmax
    float *vbi_ampmax;              // [S]         vorbis_block_internal.ampmax of the stream's block
    int *lW_block_mode, *lW_no, *impadnum;  // [S]
    // bitrate_manager_state (lib/bitrate.h:25-39) of managed-bitrate setups
    long long *bm_avg_reservoir, *bm_minmax_reservoir;   // [S]
    double *bm_avgfloat;                                 // [S]
};

struct vbm_batch {
    const vbm_setup *setup;         // device copy
    vbm_stream_state st;
    int block_mode;                 // uniform for the batch
    int W;                          // block_mode >> 1
    int N, n;                       // block size, n = N/2
    int ch;
    int nsb;                        // stream-blocks in this batch (with d_nsb set: the most it may hold = launch bound)
    int ncb;                        // channel-blocks = nsb*ch
    int few;                        // the batch holds few blocks whatever nsb (= the launch bound) says: latency-bound variants
    const int *d_nsb;               // device-resident count of a round built on the device (frontend: k_fe_plan), or NULL:
                                    //   kernels take vbm_nsb(b) / vbm_ncb(b), launchers size their grids for nsb
    int L;                          // ncb rounded up to 64
    int Ls;                         // nsb rounded up to 64
    const int *stream_id;           // [nsb] stream index of every stream-block
    const uint8_t *wflags;          // [nsb] bit0 = lW, bit1 = nW (vb->lW, vb->nW)
    // block-major transform interface
    const float *pcm;               // [ncb][N] un-windowed block PCM
    float *mdct_bm;                 // [ncb][n]
    float *logfft_bm;               // [ncb][n]
    uint16_t *qf_bm;                // [ncb][n]  floor fit input: dBquant(logmask) | (logmdct+twofitatten >= logmask) << 15
    float *local_ampmax;            // [ncb]
    uint8_t *wflags_cb;             // [ncb] wflags replicated per channel-block
    // Tiled bin-major stage buffers.  Lanes are grouped in tiles of 64 (one wavefront); each tile
    // owns one contiguous slab of `slab_words` 4-byte words holding ALL of its arrays, an array
    // being [rows][64].  Element (row i, lane l) of array X lives at
    //     X[(l >> 6) * slab_words + i * 64 + (l & 63)]
    // so a wave walks 256-byte rows that are adjacent in memory (DRAM-page / TLB friendly) and
    // the row stride is the compile-time constant 64.  The pointers below already include the
    // array's offset inside the slab.
    size_t slab_words;
    size_t sb_slab_words;           // same for the stream-block-lane arrays (partwordT, workvqT)
    float *mdctT, *logmdctT, *noiseT, *toneT, *logmaskT, *epeakT;
    float *npeakT;                  // [n/partition][L]
    float *poste;                   // [L]
    float *global_ampmax;           // [Ls]
    int *postT;                     // [VIF_POSIT+2][L]  floor posts (fit, then quantised by encode)
    int *post_valid;                // [L]
    int *floor_outT;                // [VIF_POSIT+2][L]  wrapped deviations to be entropy coded
    int *iworkT;                    // [n][L]  ilogmask, then quantised residue
    int *nonzero;                   // [L]
    // stream-block lanes (leading dimension Ls)
    int *partwordT;                 // [max partvals][Ls * max bundle]
    int *workvqT;                   // [ch*n][Ls]  interleaved residue vector of res2 (lib/res0.c:781-787)
    float *m6defT;                  // [partitions][coupling steps][Ls]  aoTuV M6 temp_def per partition (-1: none)
    int couple_parallel;            // coupling steps use disjoint channels: partitions may run sliced
    int couple_parts, couple_m6parts;  // partitions below the lowpass / those in the M6 range
    int couple_fast;                // 0 general kernel; 1 lane-per-bin kernel, no coupling; 2 lane-per-bin, stereo one step
    int *vqlenT, *vqoffT;           // [stages][ch][max partvals][Ls]  bits / bit offset of every residue run
    uint64_t *vqcodeT;              // [stages][ch*n][64] per tile of `vq_slab_words` 8-byte words: code | len << 32
    size_t vq_slab_words;
    uint8_t *packetT;               // [max_packet_bytes/4 words][64] per tile: little-endian 32-bit packet words
    int *packet_bytes;              // [Ls]
    int *packet_bits;               // [Ls]  write position while the packet is assembled
    int max_packet_bytes;           // multiple of 4
    // managed bitrate (lib/mapping0.c:1097-1181, :1204): the back half runs once per packetblob
    int fit_max_posts;              // largest post count of the floors this block type's channels use (host copy)
    int mix_makes_qf;               // k_mix leaves the floor fit's input words (qf_bm) itself: no k_floor_prep pass
    int blobno;                     // blob the floor encode / couple / pack kernels work on (PACKETBLOBS/2 for VBR)
    int *postT_blob;                // [PACKETBLOBS][VIF_POSIT+2][L] in the tile slab: fits of blobs 0, 7, 14 + the interpolated ones
    int *post_valid_blob;           // [PACKETBLOBS][L]
    uint8_t *packetT_blob;          // [PACKETBLOBS] packet tiles, each laid out like packetT
    int *packet_bytes_blob;         // [PACKETBLOBS][Ls]
    int *choice;                    // [Ls] bm->choice of the block
    // Round 3: the back half of a managed block runs ALL its packetblobs per launch (blob = blockIdx.z of the floor encode /
    // render, M6 statistics and packet kernels; k_couple_fast walks the blobs in a loop of its own because blob k reads the
    // npeak rows as blobs 0..k-1 left them, lib/mapping0.c:1249-1260).  Every array the back half writes exists once per blob
    // (blob-major copies with the same inner layout); vbm_blob_select() points the single-blob names above at blob k.
    int nblobs;                     // 1 (VBR, or a host loop over the blobs), or VBM_PACKETBLOBS
    int *floor_outT_blob, *iworkT_blob;     // tile slab: blob k at + k * rows * 64 (rows: VIF_POSIT+2, blob_iwork_rows)
    int blob_iwork_rows;
    int *nonzero_blob;              // [nblobs][L]
    int *packet_bits_blob;          // [nblobs][Ls]
    int *partwordT_blob, *vqlenT_blob, *vqoffT_blob;    // stream-block slab: blob k at + k * rows * 64
    float *m6defT_blob;
    int blob_pw_rows, blob_m6_rows, blob_len_rows;
    uint64_t *vqcodeT_blob;         // blob k at + k * vq_blob_words
    size_t vq_blob_words;
    // Fused packet assembly (k_pack_fused, pack_kernels.hip): one wavefront per stream-block, codewords in LDS.  Setups with
    // one residue submap whose channels form ONE coded vector (stereo coupled res2, mono) on the lane-per-bin couple kernel.
    int noise_ring;                 // k_noisemask keeps its running sums in a 512-row ring (host-checked window reaches, configure())
    int pack_fused;                 // the batch takes that path (host decision, configure())
    int *res_bm;                    // [nsb][n * ch] quantised residue as the residue coder reads it: bin-interleaved channels
                                    //   (work[x] = in[x % ch][x / ch], lib/res0.c:781-787), written by k_couple_fast
                                    //   instead of the tiled iworkT rows when pack_fused
                                    // packetT then holds ROWS: [sb][max_packet_bytes], only the packet's own bytes written
    int pack_submaps;               // residue submaps of this block type and their partition counts (host copy)
    int pack_partvals[16];
    int pack_spp[16];               // samples per partition (residue grouping) of each submap
};

// the single-blob names of a managed batch pointed at packetblob k (host: loops over the blobs; device: vbm_blob_enter)
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
void vbm_blob_select(vbm_batch &b, int k)
{
    b.blobno = k;
    b.postT = b.postT_blob + (size_t)k * (VBM_VIF_POSIT + 2) * 64;
    b.post_valid = b.post_valid_blob + (size_t)k * b.L;
    b.floor_outT = b.floor_outT_blob + (size_t)k * (VBM_VIF_POSIT + 2) * 64;
    b.iworkT = b.iworkT_blob + (size_t)k * b.blob_iwork_rows * 64;
    b.nonzero = b.nonzero_blob + (size_t)k * b.L;
    b.packetT = b.packetT_blob + (size_t)k * b.Ls * b.max_packet_bytes;
    b.packet_bytes = b.packet_bytes_blob + (size_t)k * b.Ls;
    b.packet_bits = b.packet_bits_blob + (size_t)k * b.Ls;
    b.partwordT = b.partwordT_blob + (size_t)k * b.blob_pw_rows * 64;
    b.m6defT = b.m6defT_blob + (size_t)k * b.blob_m6_rows * 64;
    b.vqlenT = b.vqlenT_blob + (size_t)k * b.blob_len_rows * 64;
    b.vqoffT = b.vqoffT_blob + (size_t)k * b.blob_len_rows * 64;
    b.vqcodeT = b.vqcodeT_blob + (size_t)k * b.vq_blob_words;
}

#ifdef __HIPCC__
// First statement of a back-half kernel.  The kernels are templates on BLOBS: the managed instantiation takes the blob from
// blockIdx.z; the other one leaves its argument alone (a kernel argument that is written to moves, with the arrays that
// are indexed at run time, from the constant kernarg segment into registers and scratch: measured on k_res_vq and the
// couple kernel, + 17 % on the from-PCM step).
template <bool BLOBS>
__device__ __forceinline__ void vbm_blob_enter(vbm_batch &b)
{
    if (BLOBS) vbm_blob_select(b, (int)blockIdx.z);
}
__device__ __forceinline__ int vbm_nsb(const vbm_batch &b) { return b.d_nsb ? *b.d_nsb : b.nsb; }
__device__ __forceinline__ int vbm_ncb(const vbm_batch &b) { return vbm_nsb(b) * b.ch; }
// lib/scales.h:43-51
__device__ __forceinline__ float vbm_todB(float x)
{
    uint32_t i = __float_as_uint(x) & 0x7fffffffu;
    return (float)((float)i * 7.17711438e-7f - 764.6161886f);
}
// lib/scales.h:32-40
__device__ __forceinline__ float vbm_unitnorm(float x)
{
    return __uint_as_float((__float_as_uint(x) & 0x80000000u) | 0x3f800000u);
}
#endif
