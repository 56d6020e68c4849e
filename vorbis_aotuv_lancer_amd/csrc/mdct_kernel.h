// internal: kernel launchers (device code lives in mdct_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" int vbm_launch_window_mdct(const float *d_pcm, float *d_out, const uint8_t *d_wflags,
                                      const float *d_trig, const float *d_win_self,
                                      const float *d_win_short, int n, int short_n,
                                      int apply_window, long nblocks, int max_workgroups,
                                      hipStream_t stream);

extern "C" int vbm_launch_window_fft_log(const float *d_pcm, float *d_logfft, float *d_local_ampmax,
                                         const uint8_t *d_wflags, const float *d_wa, const float *d_win_self,
                                         const float *d_win_short, int n, int short_n, long nblocks,
                                         hipStream_t stream);
