// Layout changes between the wave-per-block transform kernels (block-major rows) and the
// lane-per-block stage kernels (tiled bin-major, batch.h): 64x64 tiles through LDS so that both
// the read and the write side move whole rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "batch.h"
#include "kernels.h"

namespace {

// block-major src[c][r] (ncols blocks x rows, row length = rows) -> tiled dst:
//   dst[(c >> 6) * slab + r * 64 + (c & 63)]
template <typename T>
__global__ void k_to_tiled(const T *__restrict__ src, T *__restrict__ dst, int ncols, int rows, size_t slab)
{
    __shared__ T tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        tile[cc][tx] = (c < ncols && r < rows) ? src[(size_t)c * rows + r] : T(0);
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        if (r < rows && c < ncols) dst[(size_t)(c >> 6) * slab + (size_t)r * 64 + (c & 63)] = tile[tx][rr];
    }
}

// tiled src -> block-major dst[c][r]
template <typename T>
__global__ void k_from_tiled(const T *__restrict__ src, T *__restrict__ dst, int ncols, int rows, size_t slab,
                             const int *__restrict__ d_ncols)
{
    if (d_ncols) ncols = *d_ncols;                 // device-resident count: `ncols` was the launch bound
    if ((int)(blockIdx.x * 64) >= ncols) return;
    __shared__ T tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr, c = c0 + tx;
        tile[rr][tx] = (r < rows && c < ncols) ? src[(size_t)(c >> 6) * slab + (size_t)r * 64 + (c & 63)] : T(0);
    }
    __syncthreads();
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, r = r0 + tx;
        if (c < ncols && r < rows) dst[(size_t)c * rows + r] = tile[tx][cc];
    }
}

__global__ void k_copy_counted(int *__restrict__ dst, const int *__restrict__ src, int n, const int *__restrict__ d_count)
{
    if (d_count) n = *d_count;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void k_spread_flags(const uint8_t *__restrict__ wflags, uint8_t *__restrict__ wflags_cb, int nsb, int ch)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nsb * ch) wflags_cb[i] = wflags[i / ch];
}

template <typename T>
int to_tiled(const T *src, T *dst, int ncols, int rows, size_t slab, hipStream_t st)
{
    dim3 grid((unsigned)((ncols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(k_to_tiled<T>, grid, dim3(256), 0, st, src, dst, ncols, rows, slab);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T>
int from_tiled(const T *src, T *dst, int ncols, int rows, size_t slab, hipStream_t st, const int *d_ncols = nullptr)
{
    dim3 grid((unsigned)((ncols + 63) / 64), (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL(k_from_tiled<T>, grid, dim3(256), 0, st, src, dst, ncols, rows, slab, d_ncols);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// carried encoder state of the listed streams back to its initial values (vorbis_analysis_init,
// lib/block.c:306-344; _vp_global_look ampmax -9999): one lane per (listed stream, channel)
__global__ void k_reset_streams(vbm_stream_state st, const int *__restrict__ ids, int n, long long bm_fill)
{
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= n * st.ch) return;
    const int s = ids[lane / st.ch], ci = lane % st.ch;
    const int col = s * st.ch + ci;
    float *mb = st.mblock + (size_t)(col >> 6) * st.slab_words + (col & 63);
    for (int i = 0; i < 2048 + 256; i++) mb[(size_t)i * 64] = 0.f;   // mblock rows, then tblock rows
    st.lowcomp[col] = 0.f;
    if (ci == 0) {
        st.g_ampmax[s] = -9999.f;
        st.vbi_ampmax[s] = -9999.f;
        st.lW_block_mode[s] = 0;
        st.lW_no[s] = 0;
        st.impadnum[s] = 0;
        st.bm_avg_reservoir[s] = bm_fill;
        st.bm_minmax_reservoir[s] = bm_fill;
        st.bm_avgfloat[s] = (double)(VBM_PACKETBLOBS / 2);
    }
}

// Packet compaction for host consumers: rows of `maxb` bytes with `len[k]` used -> one byte run per packet at
// 4-byte-aligned offsets.  k_compact_scan: exclusive prefix sum of the padded lengths by one workgroup (the running
// total is carried from chunk to chunk); off[n] = total.
__global__ __launch_bounds__(1024) void k_compact_scan(const int *__restrict__ len, int n, long long *__restrict__ off)
{
    __shared__ long long s_part[16];
    __shared__ long long s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + threadIdx.x;
        int l = k < n ? len[k] : 0;
        if (l < 0) l = 0;
        long long v = (l + 3) & ~3, incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            long long t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        long long before = s_carry;
        for (int w = 0; w < wave; w++) before += s_part[w];
        if (k < n) off[k] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = s_carry;
}

// one wavefront per packet, 32-bit words (rows and offsets are 4-byte aligned; the pad bytes of the last word are
// whatever the row holds there)
__global__ void k_compact_copy(const uint8_t *__restrict__ rows, const int *__restrict__ len, int n, int maxb,
                               const long long *__restrict__ off, uint8_t *__restrict__ out)
{
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n) return;
    const int l = len[k];
    if (l <= 0) return;
    const uint32_t *src = (const uint32_t *)(rows + (size_t)k * maxb);
    uint32_t *dst = (uint32_t *)(out + off[k]);
    for (int w = lane; w < (l + 3) / 4; w += 64) dst[w] = src[w];
}

}  // namespace

extern "C" int vbm_launch_compact(const uint8_t *rows, const int *len, int n, int maxb, long long *off, uint8_t *out,
                                  hipStream_t q)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, q, len, n, off);
    hipLaunchKernelGGL(k_compact_copy, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, q, rows, len, n, maxb, off, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_reset_streams(const vbm_stream_state *st, const int *d_ids, int n, long long bm_fill, hipStream_t q)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_reset_streams, dim3((unsigned)((n * st->ch + 63) / 64)), dim3(64), 0, q, *st, d_ids, n, bm_fill);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_spread_flags(const vbm_batch *b, hipStream_t st)
{
    hipLaunchKernelGGL(k_spread_flags, dim3((unsigned)((b->ncb + 255) / 256)), dim3(256), 0, st, b->wflags,
                       b->wflags_cb, b->nsb, b->ch);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vbm_launch_transpose_in(const vbm_batch *b, hipStream_t st)
{
    // (the log spectrum stays block-major: its only reader, k_tonemask, takes whole rows)
    return to_tiled<float>(b->mdct_bm, b->mdctT, b->ncb, b->n, b->slab_words, st);
}

extern "C" int vbm_launch_untranspose_f32(const float *srcT, float *dst_bm, int rows, size_t slab, int ncols,
                                          hipStream_t st)
{
    return from_tiled<float>(srcT, dst_bm, ncols, rows, slab, st);
}
extern "C" int vbm_launch_untranspose_i32(const int *srcT, int *dst_bm, int rows, size_t slab, int ncols,
                                          hipStream_t st)
{
    return from_tiled<int>(srcT, dst_bm, ncols, rows, slab, st);
}
extern "C" int vbm_launch_untranspose_counted(const int *srcT, int *dst_bm, int rows, size_t slab, int ncols,
                                              const int *d_ncols, hipStream_t st)
{
    return from_tiled<int>(srcT, dst_bm, ncols, rows, slab, st, d_ncols);
}
extern "C" int vbm_launch_copy_counted(int *dst, const int *src, int n, const int *d_count, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_copy_counted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n, d_count);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vbm_launch_untranspose_u8(const uint8_t *srcT, uint8_t *dst_bm, int rows, size_t slab, int ncols,
                                         hipStream_t st)
{
    return from_tiled<uint8_t>(srcT, dst_bm, ncols, rows, slab, st);
}

// ---- timing experiments (VBM_DEBUG_STAMPS=1): device-side time stamps on the internal streams, without a profiler in
//      the way (rocprofv3 slows the host's graph launches so much that the from-PCM timeline it shows is the host's)
namespace {
__global__ void k_stamp(unsigned long long *out, int slot, int tag)
{
    out[2 * slot] = (unsigned long long)tag;
    out[2 * slot + 1] = wall_clock64();      // constant 100 MHz counter
}
unsigned long long *g_stamps = nullptr;
int g_stamp_next = 0, g_stamp_on = -1;
const int kStampMax = 8192;
}  // namespace
extern "C" void vbm_debug_stamp(hipStream_t st, int tag)
{
    if (g_stamp_on < 0) g_stamp_on = getenv("VBM_DEBUG_STAMPS") ? 1 : 0;
    if (!g_stamp_on) return;
    if (!g_stamps && hipMalloc((void **)&g_stamps, (size_t)kStampMax * 16) != hipSuccess) { g_stamp_on = 0; return; }
    if (g_stamp_next >= kStampMax) return;
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, st, g_stamps, g_stamp_next++, tag);
}
extern "C" int vbm_debug_stamps_read(unsigned long long *out, int max_pairs)
{
    if (!g_stamps) return 0;
    (void)hipDeviceSynchronize();
    const int n = g_stamp_next < max_pairs ? g_stamp_next : max_pairs;
    (void)hipMemcpy(out, g_stamps, (size_t)n * 16, hipMemcpyDeviceToHost);
    return n;
}
