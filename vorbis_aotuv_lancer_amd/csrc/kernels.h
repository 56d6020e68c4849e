// internal: launchers of the batch kernels (each returns 0, or -2 on a launch error)
#pragma once
#include <hip/hip_runtime.h>
#include "batch.h"

extern "C" {
int vbm_launch_transpose_in(const vbm_batch *b, hipStream_t st);   // mdct_bm -> mdctT
int vbm_launch_prologue(const vbm_batch *b, hipStream_t st);
int vbm_launch_noisemask(const vbm_batch *b, hipStream_t st);
int vbm_launch_tonemask(const vbm_batch *b, int total_octave_lines, hipStream_t st);   // tone_kernels.hip
int vbm_launch_mix(const vbm_batch *b, hipStream_t st);
int vbm_mix_can_make_qf(const vbm_batch *b);
int vbm_launch_reset_streams(const vbm_stream_state *st, const int *d_ids, int n, long long bm_fill, hipStream_t q);
int vbm_launch_block_state(const vbm_batch *b, hipStream_t st);   // aoTuV block-state update (after mix)
int vbm_launch_mix_managed(const vbm_batch *b, int offset_select, hipStream_t st);   // managed bitrate: 1, 2, 0
int vbm_launch_block_state_managed(const vbm_batch *b, hipStream_t st);
int vbm_launch_floor_interp(const vbm_batch *b, hipStream_t st);   // blobs between the fitted ones
int vbm_launch_bitrate_choose(const vbm_batch *b, uint8_t *d_packets, hipStream_t st);
int vbm_launch_floor_fit(const vbm_batch *b, hipStream_t st);
int vbm_launch_floor_encode(const vbm_batch *b, hipStream_t st);
int vbm_launch_couple_quantize(const vbm_batch *b, hipStream_t st);
int vbm_launch_pack(const vbm_batch *b, hipStream_t st);          // = pack_head, then pack_residue
// packets of the batch to the caller's buffers: dst [nsb][max_packet_bytes] (bytes past a packet's length zero), lengths to dst_bytes;
// either may be NULL; d_nsb as in vbm_batch (device-resident count or NULL).  Handles both packet layouts (tiles / fused rows).
int vbm_launch_packets_out(const vbm_batch *b, uint8_t *dst, int *dst_bytes, hipStream_t st);
// the quantised residue of a pack_fused batch as block-major rows [channel-block][n] (parity tests)
int vbm_launch_res_bm_rows(const vbm_batch *b, int *dst, hipStream_t st);
int vbm_launch_pack_head(const vbm_batch *b, hipStream_t st);     // header + floor bits: may run beside couple/quantise
int vbm_launch_pack_residue(const vbm_batch *b, hipStream_t st);  // after both: nonzero propagation, residue class + VQ + bits
// tiled [col>>6][rows][64] (tile stride `slab` elements) -> block-major dst[col][rows]
int vbm_launch_untranspose_f32(const float *srcT, float *dst_bm, int rows, size_t slab, int ncols, hipStream_t st);
int vbm_launch_untranspose_i32(const int *srcT, int *dst_bm, int rows, size_t slab, int ncols, hipStream_t st);
// packet rows -> one byte run per packet at 4-byte aligned offsets (off[n+1], exclusive scan of the padded lengths)
int vbm_launch_compact(const uint8_t *rows, const int *len, int n, int maxb, long long *off, uint8_t *out, hipStream_t q);
// the same with a device-resident column count (NULL: ncols), and a plain int copy of the first *d_count (or n) entries
int vbm_launch_untranspose_counted(const int *srcT, int *dst_bm, int rows, size_t slab, int ncols, const int *d_ncols, hipStream_t st);
int vbm_launch_copy_counted(int *dst, const int *src, int n, const int *d_count, hipStream_t st);
int vbm_launch_untranspose_u8(const uint8_t *srcT, uint8_t *dst_bm, int rows, size_t slab, int ncols, hipStream_t st);
}
