// internal: kernel launchers (device code lives in mdct_kernel.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// addressing of the envelope detector's 128-sample search windows inside the per-channel PCM buffers
// (frontend.h): block blk = channel * steps + t covers samples [64*j, 64*j + 128) of its channel with
// j = first[stream] + t0 + t, and exists while j < last[stream].  first == nullptr: plain block-major rows.
struct vbm_ve_gather {
    const float *pcm;       // buffer 0 of channel 0
    const int *first, *last, *parity;   // [S]
    const int *base;                    // [S] origin of the stream's samples inside its channel buffers
    int ch, steps, t0;
    long cap, plane;        // floats per channel buffer, floats per buffer set (all channels)
};

extern "C" int vbm_launch_ve_mdct(const vbm_ve_gather *g, float *d_out, const float *d_trig, const float *d_win,
                                  long nblocks, hipStream_t stream);

extern "C" int vbm_launch_window_mdct(const float *d_pcm, float *d_out, const uint8_t *d_wflags,
                                      const float *d_trig, const float *d_win_self,
                                      const float *d_win_short, int n, int short_n,
                                      int apply_window, long nblocks, int max_workgroups,
                                      const int *d_live, int live_mult,   // NULL, or device count: *d_live * live_mult blocks exist
                                      hipStream_t stream);

extern "C" int vbm_launch_window_fft_log(const float *d_pcm, float *d_logfft, float *d_local_ampmax,
                                         const uint8_t *d_wflags, const float *d_wa, const float *d_win_self,
                                         const float *d_win_short, int n, int short_n, long nblocks,
                                         const int *d_live, int live_mult, hipStream_t stream);
