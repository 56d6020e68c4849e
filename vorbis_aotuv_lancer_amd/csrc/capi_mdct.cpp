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
    std::vector<float> fftwa;  // FFTPACK twiddles, n floats
    float *d_trig;
    float *d_fftwa;
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

// Twiddles of drfti1 (reference lib/smallft.c:5576-5644): factor n with trial order {4,2,3,5},
// move a factor 2 to the front, then for every factor but the last store cos/sin(fi*argld)
// with arg formed in float and libm evaluated in double.
static void build_fft_twiddles(std::vector<float> &wa, int n)
{
    wa.assign(n, 0.f);
    std::vector<int> fac;
    {
        static const int tries[4] = {4, 2, 3, 5};
        int nl = n, j = 0, ntry = 0;
        while (nl != 1) {
            ntry = (j < 4) ? tries[j] : ntry + 2;
            j++;
            while (nl % ntry == 0) {
                nl /= ntry;
                if (ntry == 2 && !fac.empty()) fac.insert(fac.begin(), 2);
                else fac.push_back(ntry);
            }
        }
    }
    const float tpi = 6.28318530717958648f;
    const float argh = tpi / n;
    int is = 0, l1 = 1;
    for (size_t k1 = 0; k1 + 1 < fac.size(); k1++) {
        int ip = fac[k1], ld = 0, l2 = l1 * ip, ido = n / l2;
        for (int j = 0; j < ip - 1; j++) {
            ld += l1;
            int i = is;
            float argld = (float)ld * argh;
            float fi = 0.f;
            for (int ii = 2; ii < ido; ii += 2) {
                fi += 1.f;
                float arg = fi * argld;
                wa[i++] = (float)cos((double)arg);   // C semantics: double libm on a float argument
                wa[i++] = (float)sin((double)arg);   // (C++ would otherwise pick the float overload)
            }
            is += ido;
        }
        l1 = l2;
    }
}

extern "C" int vbm_host_mdct_trig(int n, float *out)
{
    if (!out || n < 64 || (n & (n - 1))) return VBM_EINVAL;
    std::vector<float> t;
    build_trig(t, n);
    memcpy(out, t.data(), t.size() * sizeof(float));
    return VBM_OK;
}

extern "C" int vbm_host_fft_twiddles(int n, float *out)
{
    if (!out || n < 4 || (n & (n - 1))) return VBM_EINVAL;
    std::vector<float> t;
    build_fft_twiddles(t, n);
    memcpy(out, t.data(), t.size() * sizeof(float));
    return VBM_OK;
}

extern "C" int vbm_mdct_plan_create(vbm_mdct_plan **out, int n, int short_n,
                                    const float *win_n, const float *win_short)
{
    if (!out) return VBM_EINVAL;
    *out = nullptr;
    if (n != 4096 && n != 2048 && n != 1024 && n != 512 && n != 256) return VBM_EIMPL;
    // a plan with a second (short) window serves long blocks, whose halves take either shape
    const bool is_long = win_n && win_short && short_n > 0 && short_n < n;
    if (win_short && win_n && (short_n <= 0 || short_n > n || (short_n & 7))) return VBM_EINVAL;
    int ndev = vbm_device_count();
    if (ndev < 0) return ndev;
    if (ndev == 0) {
        g_vbm_err = "no HIP device: the MI355X path has no CPU fallback";
        return VBM_ENODEV;
    }
    vbm_mdct_plan *p = new vbm_mdct_plan();
    p->n = n;
    p->short_n = is_long ? short_n : n;
    p->has_window = win_n != nullptr;
    p->d_trig = p->d_win_n = p->d_win_short = p->d_fftwa = nullptr;
    build_trig(p->trig, n);
    build_fft_twiddles(p->fftwa, n);
    hipError_t e;
#define CK(x) do { e = (x); if (e != hipSuccess) { int rc = vbm_set_hip_error(e, #x); vbm_mdct_plan_destroy(p); return rc; } } while (0)
    CK(hipMalloc((void **)&p->d_trig, p->trig.size() * sizeof(float)));
    CK(hipMemcpy(p->d_trig, p->trig.data(), p->trig.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&p->d_fftwa, p->fftwa.size() * sizeof(float)));
    CK(hipMemcpy(p->d_fftwa, p->fftwa.data(), p->fftwa.size() * sizeof(float), hipMemcpyHostToDevice));
    if (win_n) {
        CK(hipMalloc((void **)&p->d_win_n, (n / 2) * sizeof(float)));
        CK(hipMemcpy(p->d_win_n, win_n, (n / 2) * sizeof(float), hipMemcpyHostToDevice));
        if (is_long) {
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
    if (p->d_fftwa) (void)hipFree(p->d_fftwa);
    if (p->d_win_n) (void)hipFree(p->d_win_n);
    if (p->d_win_short) (void)hipFree(p->d_win_short);
    delete p;
}

extern "C" const float *vbm_mdct_plan_trig(const vbm_mdct_plan *p) { return p ? p->trig.data() : nullptr; }
extern "C" const float *vbm_mdct_plan_fft_twiddles(const vbm_mdct_plan *p) { return p ? p->fftwa.data() : nullptr; }

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
                                0, nblocks, 0, nullptr, 0, (hipStream_t)stream);
    return rc ? VBM_EHIP : VBM_OK;
}

extern "C" int vbm_window_mdct_batch(const vbm_mdct_plan *p, const float *d_pcm, float *d_out,
                                     const uint8_t *d_wflags, long nblocks, void *stream)
{
    int rc = check_batch_args(p, d_pcm, d_out, nblocks);
    if (rc) return rc;
    if (!p->has_window) return VBM_EINVAL;
    rc = vbm_launch_window_mdct(d_pcm, d_out, p->d_win_short ? d_wflags : nullptr, p->d_trig, p->d_win_n, p->d_win_short,
                                p->n, p->short_n, 1, nblocks, 0, nullptr, 0, (hipStream_t)stream);
    return rc ? VBM_EHIP : VBM_OK;
}

extern "C" int vbm_window_fft_log_batch(const vbm_mdct_plan *p, const float *d_pcm, float *d_logfft,
                                        float *d_local_ampmax, const uint8_t *d_wflags, long nblocks,
                                        void *stream)
{
    int rc = check_batch_args(p, d_pcm, d_logfft, nblocks);
    if (rc) return rc;
    if (!p->has_window || (nblocks > 0 && !d_local_ampmax)) return VBM_EINVAL;
    rc = vbm_launch_window_fft_log(d_pcm, d_logfft, d_local_ampmax, p->d_win_short ? d_wflags : nullptr, p->d_fftwa,
                                   p->d_win_n, p->d_win_short, p->n, p->short_n, nblocks, nullptr, 0, (hipStream_t)stream);
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
