// internal: launchers of the stream front end kernels (frontend_kernels.hip); 0 or -2 on a launch error
#pragma once
#include <hip/hip_runtime.h>
#include "frontend.h"

extern "C" {
int vbm_fe_launch_append(const vbm_fe_state *f, const float *d_src, int vals, float pre_amplitude, hipStream_t st);
int vbm_fe_launch_append_ids(const vbm_fe_state *f, const int *d_ids, int n, const float *d_src, int vals,
                             float pre_amplitude, long stream_stride, long ch_stride, int by_slot, hipStream_t st);
int vbm_fe_launch_restart(const vbm_fe_state *f, const int *d_ids, int n, int long_n, hipStream_t st);
int vbm_fe_launch_extrapolate(const vbm_fe_state *f, const int *d_ids, int nids, int mode, int long_n, hipStream_t st);
int vbm_fe_launch_ve_range(const vbm_fe_state *f, hipStream_t st);
int vbm_fe_launch_ve_filter(const vbm_fe_state *f, const vbm_setup *d_setup, int t0, hipStream_t st);
int vbm_fe_launch_decide(const vbm_fe_state *f, const vbm_setup *d_setup, vbm_fe_decision *d_out, const uint8_t *d_hold,
                         hipStream_t st);
int vbm_fe_launch_gather(const vbm_fe_state *f, const int *d_ids, const int *d_begin, int count, int N, float *d_dst,
                         const int *d_count, hipStream_t st);
int vbm_fe_launch_round_plan(const vbm_fe_state *f, const vbm_setup *d_setup, const vbm_fe_round *r, signed char *d_type,
                             vbm_fe_decision *d_dec, int *d_packet_bytes, int lanes, hipStream_t st);
int vbm_fe_launch_shift(const vbm_fe_state *f, const vbm_fe_decision *d_dec, hipStream_t st);
}
