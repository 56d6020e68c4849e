// Host side of the MDCT entry points of include/vorbis_mi355x.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "vorbis_mi355x.h"
#include "mdct_kernel.h"
#include "vbm_internal.h"

thread_local std::string g_vbm_err;

int vbm_set_hip_error(hipError_t e, const char *where)
{
    g_vbm_err = std::string(where) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? VBM_ENODEV : VBM_EHIP;
}

extern "C" const char *vbm_version(void) { return "vorbis_mi355x 0.1 (gfx950)"; }

extern "C" const char *vbm_last_error(void) { return g_vbm_err.c_str(); }

extern "C" int vbm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return vbm_set_hip_error(e, "hipGetDeviceCount");
    return n;
}

struct vbm_mdct_plan {
    int n;
    int short_n;
    int has_window;
    std::vector<float> trig;   // host copy, n + n/4
    float *d_trig;
    float *d_win_n;
    float *d_win_short;
};

// Trig table of mdct_init (reference lib/mdct.c:67-76): double libm, rounded on store.
static void build_trig(std::vector<float> &T, int n)
{
    T.resize(n + n / 4);
    const int n2 = n >> 1;
    for (int i = 0; i < n / 4; i++) {
        T[i * 2]          = (float)cos((M_PI / n) * (4 * i));
        T[i * 2 + 1]      = (float)-sin((M_PI / n) * (4 * i));
        T[n2 + i * 2]     = (float)cos((M_PI / (2 * n)) * (2 * i + 1));
        T[n2 + i * 2 + 1] = (float)sin((M_PI / (2 * n)) * (2 * i + 1));
    }
    for (int i = 0; i < n / 8; i++) {
        T[n + i * 2]     = (float)(cos((M_PI / n) * (4 * i + 2)) * .5);
        T[n + i * 2 + 1] = (float)(-sin((M_PI / n) * (4 * i + 2)) * .5);
    }
}

extern "C" int vbm_mdct_plan_create(vbm_mdct_plan **out, int n, int short_n,
                                    const float *win_n, const float *win_short)
{
    if (!out) return VBM_EINVAL;
    *out = nullptr;
    if (n != 2048 && n != 256) return VBM_EIMPL;
    if (n == 2048 && win_n && (!win_short || short_n <= 0 || short_n > n || (short_n & 7))) return VBM_EINVAL;
    int ndev = vbm_device_count();
    if (ndev < 0) return ndev;
    if (ndev == 0) {
        g_vbm_err = "no HIP device: the MI355X path has no CPU fallback";
        return VBM_ENODEV;
    }
    vbm_mdct_plan *p = new vbm_mdct_plan();
    p->n = n;
    p->short_n = (n == 2048) ? short_n : n;
    p->has_window = win_n != nullptr;
    p->d_trig = p->d_win_n = p->d_win_short = nullptr;
    build_trig(p->trig, n);
    hipError_t e;
#define CK(x) do { e = (x); if (e != hipSuccess) { int rc = vbm_set_hip_error(e, #x); vbm_mdct_plan_destroy(p); return rc; } } while (0)
    CK(hipMalloc((void **)&p->d_trig, p->trig.size() * sizeof(float)));
    CK(hipMemcpy(p->d_trig, p->trig.data(), p->trig.size() * sizeof(float), hipMemcpyHostToDevice));
    if (win_n) {
        CK(hipMalloc((void **)&p->d_win_n, (n / 2) * sizeof(float)));
        CK(hipMemcpy(p->d_win_n, win_n, (n / 2) * sizeof(float), hipMemcpyHostToDevice));
        if (n == 2048) {
            CK(hipMalloc((void **)&p->d_win_short, (short_n / 2) * sizeof(float)));
            CK(hipMemcpy(p->d_win_short, win_short, (short_n / 2) * sizeof(float), hipMemcpyHostToDevice));
        }
    }
#undef CK
    *out = p;
    return VBM_OK;
}

extern "C" void vbm_mdct_plan_destroy(vbm_mdct_plan *p)
{
    if (!p) return;
    if (p->d_trig) (void)hipFree(p->d_trig);
    if (p->d_win_n) (void)hipFree(p->d_win_n);
    if (p->d_win_short) (void)hipFree(p->d_win_short);
    delete p;
}

extern "C" const float *vbm_mdct_plan_trig(const vbm_mdct_plan *p) { return p ? p->trig.data() : nullptr; }

static int check_batch_args(const vbm_mdct_plan *p, const void *in, const void *out, long nblocks)
{
    if (!p || nblocks < 0) return VBM_EINVAL;
    if (nblocks > 0 && (!in || !out)) return VBM_EINVAL;
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return VBM_EINVAL;  // 16-B vector loads/stores
    return VBM_OK;
}

extern "C" int vbm_mdct_forward_batch(const vbm_mdct_plan *p, const float *d_in, float *d_out,
                                      long nblocks, void *stream)
{
    int rc = check_batch_args(p, d_in, d_out, nblocks);
    if (rc) return rc;
    rc = vbm_launch_window_mdct(d_in, d_out, nullptr, p->d_trig, nullptr, nullptr, p->n, p->short_n,
                                0, nblocks, 0, (hipStream_t)stream);
    return rc ? VBM_EHIP : VBM_OK;
}

extern "C" int vbm_window_mdct_batch(const vbm_mdct_plan *p, const float *d_pcm, float *d_out,
                                     const uint8_t *d_wflags, long nblocks, void *stream)
{
    int rc = check_batch_args(p, d_pcm, d_out, nblocks);
    if (rc) return rc;
    if (!p->has_window) return VBM_EINVAL;
    rc = vbm_launch_window_mdct(d_pcm, d_out, d_wflags, p->d_trig, p->d_win_n, p->d_win_short, p->n,
                                p->short_n, 1, nblocks, 0, (hipStream_t)stream);
    return rc ? VBM_EHIP : VBM_OK;
}

extern "C" int vbm_window_mdct_time(const vbm_mdct_plan *p, const float *d_pcm, float *d_out,
                                    const uint8_t *d_wflags, long nblocks, int iters, void *stream,
                                    float *ms_total)
{
    if (!ms_total || iters <= 0) return VBM_EINVAL;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return vbm_set_hip_error(e, "hipEventCreate");
    if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return vbm_set_hip_error(e, "hipEventCreate"); }
    int rc = VBM_OK;
    (void)hipEventRecord(e0, (hipStream_t)stream);
    for (int i = 0; i < iters && rc == VBM_OK; i++)
        rc = vbm_window_mdct_batch(p, d_pcm, d_out, d_wflags, nblocks, stream);
    (void)hipEventRecord(e1, (hipStream_t)stream);
    e = hipEventSynchronize(e1);
    if (e != hipSuccess) rc = vbm_set_hip_error(e, "hipEventSynchronize");
    else (void)hipEventElapsedTime(ms_total, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
