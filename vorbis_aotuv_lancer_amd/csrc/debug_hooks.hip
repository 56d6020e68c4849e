// Test instrumentation of the stream / event choreography (include/vorbis_mi355x.h, "test instrumentation").
//
// The per-block path is spread over several internal HIP streams (capi_encoder.cpp, capi_frontend.cpp); what orders
// them is events, and a missing edge only shows when the timing happens to expose it.  vbm_debug_set_delay() makes
// the timing adversarial on purpose: at every marked point of the host code a kernel that spins for `usec`
// microseconds is put in front of what follows on that stream, so the work behind the point runs LATE while
// everything that is not ordered behind it runs on.  With every edge in place the packets do not change, whatever
// the mask (tests/test_ordering_gpu.py runs each point once); a consumer that is not ordered behind a delayed
// producer reads stale data deterministically.
#include <hip/hip_runtime.h>
#include <atomic>
#include "vbm_internal.h"

namespace {

std::atomic<unsigned> g_delay_mask{0};
std::atomic<int> g_delay_usec{0};

// wall_clock64: the constant 100 MHz counter.  The loop is bounded twice (time and iterations): it always exits.
__global__ void k_spin(const unsigned long long ticks, const long max_iter)
{
    const unsigned long long t0 = wall_clock64();
    for (long i = 0; i < max_iter; i++) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}

}  // namespace

void vbm_debug_delay_point(int point, hipStream_t q)
{
    const unsigned m = g_delay_mask.load(std::memory_order_relaxed);
    if (!((m >> point) & 1u)) return;
    const int usec = g_delay_usec.load(std::memory_order_relaxed);
    if (usec <= 0) return;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, q, (unsigned long long)usec * 100ull, (long)usec * 64 + 4096);
    (void)hipGetLastError();
}

extern "C" int vbm_debug_set_delay(unsigned mask, int usec)
{
    if (usec < 0 || usec > 100000) return -131;
    g_delay_usec.store(usec, std::memory_order_relaxed);
    g_delay_mask.store(usec ? mask : 0u, std::memory_order_relaxed);
    return 0;
}
