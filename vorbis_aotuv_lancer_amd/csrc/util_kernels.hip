// Layout changes between the wave-per-block transform kernels (block-major rows) and the
// lane-per-block stage kernels (bin-major, batch.h): 64x64 tiles through LDS so that both the
// read and the write side move whole 256-byte rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "kernels.h"

namespace {

// src[r][c] (rows x cols, leading dimension lds) -> dst[c][r] (leading dimension ldd)
template <typename T>
__global__ void k_transpose(const T *__restrict__ src, T *__restrict__ dst, int rows, int cols, size_t lds,
                            size_t ldd)
{
    __shared__ T tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < rows && c < cols) ? src[(size_t)r * lds + c] : T(0);
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[tx][cc];
    }
}

__global__ void k_spread_flags(const uint8_t *__restrict__ wflags, uint8_t *__restrict__ wflags_cb, int nsb, int ch)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsb * ch) wflags_cb[i] = wflags[i / ch];
}

template <typename T>
int transpose(const T *src, T *dst, int rows, int cols, size_t lds, size_t ldd, hipStream_t st)
{
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(k_transpose<T>, grid, dim3(256), 0, st, src, dst, rows, cols, lds, ldd);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" int vbm_launch_spread_flags(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_spread_flags, dim3((unsigned)((b->ncb + 255) / 256)), dim3(256), 0, st, b->wflags,
                       b->wflags_cb, b->nsb, b->ch);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_transpose_in(const vbm_batch *b, hipStream_t st)
{
    // [ncb][n] block-major -> [n][L] bin-major
    int rc = transpose<float>(b->mdct_bm, b->mdctT, b->ncb, b->n, (size_t)b->n, (size_t)b->L, st);
    if (rc) return rc;
    return transpose<float>(b->logfft_bm, b->logfftT, b->ncb, b->n, (size_t)b->n, (size_t)b->L, st);
}

extern "C" int vbm_launch_untranspose_f32(const float *srcT, float *dst_bm, int rows, int L, int ncb, hipStream_t st)
{
    return transpose<float>(srcT, dst_bm, rows, ncb, (size_t)L, (size_t)rows, st);
}
extern "C" int vbm_launch_untranspose_i32(const int *srcT, int *dst_bm, int rows, int L, int ncb, hipStream_t st)
{
    return transpose<int>(srcT, dst_bm, rows, ncb, (size_t)L, (size_t)rows, st);
}
extern "C" int vbm_launch_untranspose_u8(const uint8_t *srcT, uint8_t *dst_bm, int rows, int L, int ncb, hipStream_t st)
{
    return transpose<uint8_t>(srcT, dst_bm, rows, ncb, (size_t)L, (size_t)rows, st);
}
