// Host side of the encoder setup: reads the mode pack + common tables (VPK), derives the
// per-class lookup tables the reference builds in vorbis_analysis_init(), and lays everything
// out in one arena that is uploaded to the device unchanged (pointers rebased).
//
// Look derivations restated here (product code; shares nothing with the test checker):
//   _vp_psy_init        reference lib/psy.c:352-507
//   setup_tone_curves   lib/psy.c:171-350
//   floor1_look         lib/floor1.c:183-258
//   res0_look           lib/res0.c:255-313
//   vorbis_book_init_encode / _make_words / _book_maptype1_quantvals  lib/sharedbook.c:85-317
// C promotion rules matter for bit-exact tables; every mixed float/double expression of the
// source is spelled out with explicit casts (C++ would otherwise pick float overloads of
// atan/log/exp).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "setup_host.h"
#include "vpk.h"
#include "vbm_internal.h"

namespace {

struct Arena {
    std::vector<unsigned char> bytes;
    template <typename T>
    size_t put(const T *p, size_t count)
    {
        size_t off = (bytes.size() + 15) & ~(size_t)15;
        bytes.resize(off + count * sizeof(T));
        if (count) memcpy(bytes.data() + off, p, count * sizeof(T));
        return off;
    }
    template <typename T>
    size_t put(const std::vector<T> &v) { return put(v.data(), v.size()); }
};

template <typename T>
const T *as_off(size_t off) { return reinterpret_cast<const T *>(off + 1); }  // +1: offset 0 must not look NULL

template <typename T>
void rebase(const T *&p, const unsigned char *base)
{
    if (p) p = reinterpret_cast<const T *>(base + (reinterpret_cast<size_t>(p) - 1));
}

// ---- C-semantics helpers for lib/scales.h:69-87 --------------------------------------
inline double toBARK_long(long n)
{
    float a = .00074f * (float)n;
    float b = (float)(n * n) * 1.85e-8f;
    float c = 1e-4f * (float)n;
    return (double)13.1f * atan((double)a) + (double)2.24f * atan((double)b) + (double)c;
}
inline double toOC_d(double n) { return log(n) * (double)1.442695f - (double)5.965784f; }
inline double fromOC_d(double o) { return exp((o + (double)5.965784f) * (double).693147f); }

struct Pack {
    vpk_file f;
    bool ok = false;
    explicit Pack(const char *path) { ok = vpk_open(&f, path) == 0; }
    ~Pack() { if (ok) vpk_close(&f); }
    template <typename T>
    const T *get(const std::string &name, int dtype, size_t *n = nullptr) const
    {
        size_t nn = 0;
        const void *p = vpk_get(&f, name.c_str(), dtype, &nn);
        if (!p) throw std::string("mode pack entry missing or mistyped: ") + name;
        if (n) *n = nn;
        return (const T *)p;
    }
    const int *i32(const std::string &n, size_t *c = nullptr) const { return get<int>(n, VPK_I32, c); }
    const float *f32(const std::string &n, size_t *c = nullptr) const { return get<float>(n, VPK_F32, c); }
    const long long *i64(const std::string &n, size_t *c = nullptr) const { return get<long long>(n, VPK_I64, c); }
    const double *f64(const std::string &n, size_t *c = nullptr) const { return get<double>(n, VPK_F64, c); }
    bool has(const std::string &n) const { size_t nn = 0; return vpk_get(&f, n.c_str(), VPK_I32, &nn) != nullptr; }
};

// ---- tone curves (lib/psy.c:171-350) ---------------------------------------------------
void tone_curves(const float *ATH, const float *tonemasks /*[17][6][56]*/, std::vector<float> &ret,
                 const float *curveatt_dB, float binHz, int n, float center_boost, float center_decay_rate)
{
    const int EM = VBM_EHMER_MAX, PB = VBM_P_BANDS, PL = VBM_P_LEVELS;
    ret.assign((size_t)PB * PL * (EM + 2), 0.f);
    std::vector<float> workc((size_t)PB * PL * EM, 0.f), brute(n);
    auto W = [&](int i, int j) { return &workc[((size_t)i * PL + j) * EM]; };
    auto R = [&](int i, int m) { return &ret[((size_t)i * PL + m) * (EM + 2)]; };
    float ath[VBM_EHMER_MAX], athc[VBM_P_LEVELS][VBM_EHMER_MAX];

    for (int i = 0; i < PB; i++) {
        int ath_offset = i * 4;
        for (int j = 0; j < EM; j++) {
            float mn = 999.f;
            for (int k = 0; k < 4; k++) {
                float v = (j + k + ath_offset < VBM_MAX_ATH) ? ATH[j + k + ath_offset] : ATH[VBM_MAX_ATH - 1];
                if (mn > v) mn = v;
            }
            ath[j] = mn;
        }
        for (int j = 0; j < 6; j++) memcpy(W(i, j + 2), tonemasks + ((size_t)i * 6 + j) * EM, EM * sizeof(float));
        memcpy(W(i, 0), tonemasks + ((size_t)i * 6) * EM, EM * sizeof(float));
        memcpy(W(i, 1), tonemasks + ((size_t)i * 6) * EM, EM * sizeof(float));

        for (int j = 0; j < PL; j++)
            for (int k = 0; k < EM; k++) {
                float adj = center_boost + (float)abs(VBM_EHMER_OFFSET - k) * center_decay_rate;
                if ((double)adj < 0. && center_boost > 0) adj = 0.f;
                if ((double)adj > 0. && center_boost < 0) adj = 0.f;
                W(i, j)[k] += adj;
            }

        for (int j = 0; j < PL; j++) {
            // attenuate_curve(c, float att): the double expression is narrowed to float at the call
            float att1 = (float)((double)curveatt_dB[i] + 100. - (double)((j < 2 ? 2 : j)) * 10. - 30.);
            for (int k = 0; k < EM; k++) W(i, j)[k] += att1;
            memcpy(athc[j], ath, sizeof(ath));
            float att2 = (float)(+100. - (double)((float)j * 10.f) - 30.);
            for (int k = 0; k < EM; k++) athc[j][k] += att2;
            for (int k = 0; k < EM; k++)
                if (W(i, j)[k] > athc[j][k]) athc[j][k] = W(i, j)[k];
        }
        for (int j = 1; j < PL; j++) {
            for (int k = 0; k < EM; k++)
                if (athc[j - 1][k] < athc[j][k]) athc[j][k] = athc[j - 1][k];
            for (int k = 0; k < EM; k++)
                if (athc[j][k] < W(i, j)[k]) W(i, j)[k] = athc[j][k];
        }
    }

    auto paint = [&](int k, int m, int ocbase) {
        // render curve k (level m) into bins, octave offsets relative to band `ocbase`
        int l = 0;
        for (int j = 0; j < EM; j++) {
            int lo_bin = (int)(fromOC_d(j * .125 + ocbase * .5 - 2.0625) / (double)binHz);
            int hi_bin = (int)(fromOC_d(j * .125 + ocbase * .5 - 1.9375) / (double)binHz + 1);
            if (lo_bin < 0) lo_bin = 0;
            if (lo_bin > n) lo_bin = n;
            if (lo_bin < l) l = lo_bin;
            if (hi_bin < 0) hi_bin = 0;
            if (hi_bin > n) hi_bin = n;
            for (; l < hi_bin && l < n; l++)
                if (brute[l] > W(k, m)[j]) brute[l] = W(k, m)[j];
        }
        for (; l < n; l++)
            if (brute[l] > W(k, m)[EM - 1]) brute[l] = W(k, m)[EM - 1];
    };

    for (int i = 0; i < PB; i++) {
        int bin = (int)floor(fromOC_d(i * .5) / (double)binHz);
        int lo_curve = (int)ceil(toOC_d((double)((float)bin * binHz + 1)) * 2);
        int hi_curve = (int)floor(toOC_d((double)((float)(bin + 1) * binHz)) * 2);
        if (lo_curve > i) lo_curve = i;
        if (lo_curve < 0) lo_curve = 0;
        if (hi_curve >= PB) hi_curve = PB - 1;

        for (int m = 0; m < PL; m++) {
            for (int j = 0; j < n; j++) brute[j] = 999.f;
            for (int k = lo_curve; k <= hi_curve; k++) paint(k, m, k);
            if (i + 1 < PB) paint(i + 1, m, i);   // "valid up to next half octave": curve i+1 at band i's offsets

            float *r = R(i, m);
            for (int j = 0; j < EM; j++) {
                int b = (int)(fromOC_d(j * .125 + i * .5 - 2.) / (double)binHz);
                r[j + 2] = (b < 0 || b >= n) ? -999.f : brute[b];
            }
            int j;
            for (j = 0; j < VBM_EHMER_OFFSET; j++)
                if (r[j + 2] > -200.f) break;
            r[0] = (float)j;
            for (j = EM - 1; j > VBM_EHMER_OFFSET + 1; j--)
                if (r[j + 2] > -200.f) break;
            r[1] = (float)j;
        }
    }
}

struct PsyTables {
    std::vector<float> tonecurves, ath, ntfix, noiseoffset[VBM_P_NOISECURVES];
    std::vector<int> octave, bark_lo, bark_hi, group_start, seg_p0, seg_p1, group_tab;
};

// lib/psy.c:352-507
void psy_look(vbm_psy &p, PsyTables &t, const Pack &common, int eighth_octave_lines, int n, long rate)
{
    const float *ATH = common.f32("ATH");
    const float *tonemasks = common.f32("tonemasks");
    const float *ntfix_offset = common.f32("ntfix_offset");
    const int *aot_i = common.i32("aotuv_preset/ints");
    const float *aot_f = common.f32("aotuv_preset/tonecomp_thres");

    p.n = n;
    p.rate = rate;
    p.eighth_octave_lines = eighth_octave_lines;
    long shiftoc = (long)rint(log((double)((float)eighth_octave_lines * 8.f)) / log((double)2.f)) - 1;
    p.shiftoc = (int)shiftoc;
    // toOC(.25f*rate*.5/n)*(1<<(shiftoc+1)) - eighth_octave_lines  -> long (truncation)
    {
        double a = (double)(.25f * (float)rate) * .5 / (double)n;
        p.firstoc = (int)(long)(toOC_d(a) * (double)(1 << (shiftoc + 1)) - (double)eighth_octave_lines);
        double b = (double)(((float)n + .25f) * (float)rate) * .5 / (double)n;
        long maxoc = (long)(toOC_d(b) * (double)(1 << (shiftoc + 1)) + (double).5f);
        p.total_octave_lines = (int)(maxoc - p.firstoc + 1);
    }
    p.n25p = n / 4;
    p.n33p = n / 3;
    p.n75p = p.n25p * 3;

    int select = -1;
    for (int i = 0; i < 4; i++) p.m3n[i] = 0;
    auto m3 = [&](const char *name) { const int *m = common.i32(name); for (int i = 0; i < 3; i++) p.m3n[i] = m[i]; };
    if (rate < 26000) {
        p.m_val = 0;
    } else if (rate < 38000) {
        p.m_val = (float).93;
        if (n == 128) { select = 0; m3("m3n32"); } else if (n == 256) { select = 1; m3("m3n32x2"); }
        else if (n == 1024) select = 2; else if (n == 2048) select = 3;
    } else if (rate > 46000) {
        p.m_val = (float)1.205;
        if (n == 128) { select = 4; m3("m3n48"); } else if (n == 256) { select = 5; m3("m3n48x2"); }
        else if (n == 1024) select = 6; else if (n == 2048) select = 7;
    } else {
        p.m_val = 1.f;
        if (n == 128) { select = 8; m3("m3n44"); } else if (n == 256) { select = 9; m3("m3n44x2"); }
        else if (n == 1024) select = 10; else if (n == 2048) select = 11;
    }
    if (select < 0) {
        p.tonecomp_endp = 0; p.tonecomp_thres = .25f; p.min_nn_lp = 0; p.tonefix_end = 0;
    } else {
        p.tonecomp_endp = aot_i[select * 3]; p.tonecomp_thres = aot_f[select];
        p.min_nn_lp = aot_i[select * 3 + 1]; p.tonefix_end = aot_i[select * 3 + 2];
    }

    // ATH curve
    t.ath.assign(n, 0.f);
    long j = 0;
    for (long i = 0; i < VBM_MAX_ATH - 1; i++) {
        int endpos = (int)rint(fromOC_d((double)(i + 1) * .125 - 2.) * 2 * n / (double)rate);
        float base = ATH[i];
        if (j < endpos) {
            float delta = (ATH[i + 1] - base) / (float)(endpos - j);
            for (; j < endpos && j < n; j++) {
                t.ath[j] = (float)((double)base + 100.);
                base += delta;
            }
        }
    }
    {
        float cs = t.ath[j - 1];
        float ds = t.ath[j - 1] - t.ath[j - 2];
        for (long i = j; i < n; i++, cs += ds) t.ath[i] = cs;
    }

    // bark-scale noise windows
    t.bark_lo.resize(n);
    t.bark_hi.resize(n);
    long lo = -99, hi = 1;
    for (long i = 0; i < n; i++) {
        float bark = (float)toBARK_long(rate / (2 * n) * i);
        for (; lo + p.noisewindowlomin < i && toBARK_long(rate / (2 * n) * lo) < (double)(bark - p.noisewindowlo); lo++)
            ;
        for (; hi <= n && (hi < i + p.noisewindowhimin ||
                           toBARK_long(rate / (2 * n) * hi) < (double)(bark + p.noisewindowhi));
             hi++)
            ;
        // the reference packs ((lo-1)<<16)+(hi-1) and reads back b>>16 / b&0xffff
        long packed = ((lo - 1) << 16) + (hi - 1);
        t.bark_lo[i] = (int)(packed >> 16);
        t.bark_hi[i] = (int)(packed & 0xffff);
    }

    t.octave.resize(n);
    for (long i = 0; i < n; i++) {
        double a = (double)((float)i + .25f) * .5 * (double)rate / (double)n;
        t.octave[i] = (int)(long)(toOC_d(a) * (double)(1 << (shiftoc + 1)) + (double).5f);
    }

    // seed_loop groups bins that share an octave line; max_seeds walks seed lines and bins together
    // under table-only conditions.  Both walks are recorded here.
    for (long i = 0; i < n; i++)
        if (i == 0 || t.octave[i] != t.octave[i - 1]) t.group_start.push_back((int)i);
    p.ngroups = (int)t.group_start.size();
    t.group_start.push_back(n);
    t.seg_p0.assign(n, -1);
    t.seg_p1.assign(n, -1);
    {
        const long linesper = eighth_octave_lines;
        long linpos = 0, pos = t.octave[0] - p.firstoc - (linesper >> 1);
        while (linpos + 1 < n) {
            const long p0 = pos, before = linpos;
            long end = ((t.octave[linpos] + t.octave[linpos + 1]) >> 1) - p.firstoc;
            while (pos + 1 <= end) pos++;
            end = pos + p.firstoc;
            for (; linpos < n && t.octave[linpos] <= end; linpos++) {
                t.seg_p0[linpos] = (int)p0;
                t.seg_p1[linpos] = (int)pos;
            }
            if (linpos == before || p0 < 0 || pos >= p.total_octave_lines)
                throw std::string("psy_look: max_seeds walk leaves the seed array");
        }
    }

    tone_curves(ATH, tonemasks, t.tonecurves, p.toneatt, (float)((double)rate * .5 / (double)n), n,
                p.tone_centerboost, p.tone_decay);

    for (int c = 0; c < VBM_P_NOISECURVES; c++) t.noiseoffset[c].assign(n, 0.f);
    t.ntfix.assign(n, 0.f);
    for (long i = 0; i < n; i++) {
        float halfoc = (float)(toOC_d(((double)i + .5) * (double)rate / (2. * (double)n)) * 2.);
        if (halfoc < 0) halfoc = 0;
        if (halfoc >= VBM_P_BANDS - 1) halfoc = VBM_P_BANDS - 1;
        int inthalfoc = (int)halfoc;
        float del = halfoc - (float)inthalfoc;
        for (int c = 0; c < VBM_P_NOISECURVES; c++)
            // a*(1.-del) + b*del : the first product is double, the second is a FLOAT product
            t.noiseoffset[c][i] = (float)((double)p.noiseoff[c][inthalfoc] * (1. - (double)del) +
                                          (double)(p.noiseoff[c][inthalfoc + 1] * del));
        t.ntfix[i] = (float)((double)ntfix_offset[inthalfoc] * (1. - (double)del) +
                             (double)(ntfix_offset[inthalfoc + 1] * del));
    }
}

// lib/floor1.c:183-258
void floor_look(vbm_floor &f)
{
    int n = 0;
    f.n = f.postlist[1];
    for (int i = 0; i < f.partitions; i++) n += f.class_dim[f.partitionclass[i]];
    n += 2;
    f.posts = n;
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return f.postlist[a] < f.postlist[b]; });
    for (int i = 0; i < n; i++) f.forward_index[i] = order[i];
    for (int i = 0; i < n; i++) f.reverse_index[f.forward_index[i]] = i;
    for (int i = 0; i < n; i++) f.sorted_index[i] = f.postlist[f.forward_index[i]];
    static const int qq[5] = {0, 256, 128, 86, 64};
    f.quant_q = qq[f.mult];
    for (int i = 0; i < n - 2; i++) {
        int lo = 0, hi = 1, lx = 0, hx = f.n, cur = f.postlist[i + 2];
        for (int j = 0; j < i + 2; j++) {
            int x = f.postlist[j];
            if (x > lx && x < cur) { lo = j; lx = x; }
            if (x < hx && x > cur) { hi = j; hx = x; }
        }
        f.loneighbor[i] = lo;
        f.hineighbor[i] = hi;
    }
}

int ilog(uint32_t v) { int r = 0; while (v) { r++; v >>= 1; } return r; }

// lib/sharedbook.c:85-169 (sparsecount == 0)
std::vector<uint32_t> make_words(const signed char *l, int n)
{
    std::vector<uint32_t> r(n, 0);
    uint32_t marker[33];
    memset(marker, 0, sizeof(marker));
    for (int i = 0; i < n; i++) {
        int length = l[i];
        if (length <= 0) continue;
        uint32_t entry = marker[length];
        if (length < 32 && (entry >> length)) throw std::string("overpopulated codebook tree");
        r[i] = entry;
        for (int j = length; j > 0; j--) {
            if (marker[j] & 1) {
                if (j == 1) marker[1]++;
                else marker[j] = marker[j - 1] << 1;
                break;
            }
            marker[j]++;
        }
        for (int j = length + 1; j < 33; j++) {
            if ((marker[j] >> 1) == entry) {
                entry = marker[j];
                marker[j] = marker[j - 1] << 1;
            } else
                break;
        }
    }
    for (int i = 0; i < n; i++) {   // bit-reverse for the LSb-first packer
        uint32_t temp = 0;
        for (int j = 0; j < l[i]; j++) {
            temp <<= 1;
            temp |= (r[i] >> j) & 1;
        }
        r[i] = temp;
    }
    return r;
}

// greatest v with v^dim <= entries (lib/sharedbook.c:174-209, integer verification loop)
int quantvals1(long entries, int dim)
{
    if (entries < 1) return 0;
    long vals = (long)floor(pow((double)(float)entries, (double)(1.f / (float)dim)));
    if (vals < 1) vals = 1;
    for (;;) {
        long acc = 1, acc1 = 1;
        int i;
        for (i = 0; i < dim; i++) {
            if (entries / vals < acc) break;
            acc *= vals;
            if (0x7fffffffffffffffL / (vals + 1) < acc1) acc1 = 0x7fffffffffffffffL;
            else acc1 *= vals + 1;
        }
        if (i >= dim && acc <= entries && acc1 > entries) return (int)vals;
        if (i < dim || acc > entries) vals--;
        else vals++;
    }
}

float float32_unpack(long val)   // lib/sharedbook.c:66-80
{
    double mant = (double)(val & 0x1fffff);
    bool sign = (val & 0x80000000L) != 0;
    long e = (val & 0x7fe00000L) >> 21;
    if (sign) mant = -mant;
    e = e - 20 - 768;
    if (e > 63) e = 63;
    if (e < -63) e = -63;
    return (float)ldexp(mant, (int)e);
}

}  // namespace

// encode-side lattice of a maptype-1 book as vorbis_book_init_encode derives it (lib/sharedbook.c:303-317):
// out = {quantvals, minval, delta}.  Exposed for the CPU parity test on the reference's own self-test books.
extern "C" int vbm_host_book_lattice(long q_min, long q_delta, long entries, int dim, int *out)
{
    if (!out || dim < 1 || entries < 1) return -131;
    out[0] = quantvals1(entries, dim);
    out[1] = (int)rint((double)float32_unpack(q_min));
    out[2] = (int)rint((double)float32_unpack(q_delta));
    return 0;
}

struct vbm_setup_host {
    Arena arena;
    vbm_setup s;                    // pointer fields hold (offset+1) until rebased
    std::vector<vbm_book> books;    // same
    size_t books_off = 0;
    vbm_setup host;                 // rebased onto arena.bytes
    std::vector<vbm_book> host_books;
    // device
    unsigned char *d_arena = nullptr;
    vbm_setup *d_setup = nullptr;
    vbm_setup dev_view;             // host-resident struct whose pointers are DEVICE addresses
};

static void rebase_setup(vbm_setup &s, const unsigned char *base)
{
    for (int i = 0; i < 4; i++) {
        vbm_psy &p = s.psy[i];
        rebase(p.tonecurves, base);
        for (int c = 0; c < VBM_P_NOISECURVES; c++) rebase(p.noiseoffset[c], base);
        rebase(p.ath, base);
        rebase(p.octave, base);
        rebase(p.bark_lo, base);
        rebase(p.bark_hi, base);
        rebase(p.group_start, base);
        rebase(p.group_tab, base);
        rebase(p.seg_p0, base);
        rebase(p.seg_p1, base);
        rebase(p.ntfix_noiseoffset, base);
    }
    rebase(s.book, base);
    rebase(s.freq_bfn128, base);
    rebase(s.freq_bfn256, base);
    rebase(s.fromdB, base);
    for (int i = 0; i < 2; i++) {
        rebase(s.window[i], base);
        rebase(s.mdct_trig[i], base);
        rebase(s.fft_wa[i], base);
    }
    rebase(s.ve.mdct_win, base);
    rebase(s.ve.mdct_trig, base);
}

static void rebase_book(vbm_book &b, const unsigned char *base)
{
    rebase(b.lengthlist, base);
    rebase(b.codelist, base);
    rebase(b.used_index, base);
    rebase(b.used_point, base);
    rebase(b.used_pack, base);
    rebase(b.used_norm, base);
}

extern "C" int vbm_host_mdct_trig(int n, float *out);
extern "C" int vbm_host_fft_twiddles(int n, float *out);

vbm_setup_host *vbm_setup_host_load(const char *common_path, const char *mode_path, std::string &err)
{
    try {
        Pack common(common_path), mode(mode_path);
        if (!common.ok) throw std::string("cannot open ") + common_path;
        if (!mode.ok) throw std::string("cannot open ") + mode_path;
        vbm_setup_host *H = new vbm_setup_host();
        vbm_setup &s = H->s;
        memset(&s, 0, sizeof(s));
        Arena &A = H->arena;

        s.channels = *mode.i32("info/channels");
        s.rate = (long)*mode.get<int64_t>("info/rate", VPK_I64);
        const int *bs = mode.i32("info/blocksizes");
        const int *cnt = mode.i32("info/counts");
        const int *lp = mode.i32("info/block_lowpassr");
        s.blocksizes[0] = bs[0]; s.blocksizes[1] = bs[1];
        s.modes = cnt[0]; s.maps = cnt[1]; s.floors = cnt[2]; s.residues = cnt[3]; s.books = cnt[4]; s.psys = cnt[5];
        s.block_lowpassr[0] = lp[0]; s.block_lowpassr[1] = lp[1];
        s.pre_amplitude = *mode.f32("hi/pre_amplitude");
        s.modebits = ilog((uint32_t)(s.modes - 1));
        if (s.channels > VBM_MAXCH || s.floors > 4 || s.residues > 4 || s.psys > 4 || s.modes > 2)
            throw std::string("setup exceeds static limits");
        for (int b = 0; b < 2; b++)
            if (s.blocksizes[b] != 256 && s.blocksizes[b] != 512 && s.blocksizes[b] != 1024 && s.blocksizes[b] != 2048 &&
                s.blocksizes[b] != 4096)
                throw std::string("block sizes 256/512/1024/2048/4096 are implemented");

        for (int i = 0; i < s.maps; i++) {
            vbm_map &m = s.map[i];
            std::string pre = "map/" + std::to_string(i) + "/";
            size_t n;
            m.submaps = *mode.i32(pre + "submaps");
            const int *p = mode.i32(pre + "chmuxlist", &n);
            for (size_t j = 0; j < n && j < VBM_MAXCH; j++) m.chmuxlist[j] = p[j];
            p = mode.i32(pre + "floorsubmap"); memcpy(m.floorsubmap, p, sizeof(m.floorsubmap));
            p = mode.i32(pre + "residuesubmap"); memcpy(m.residuesubmap, p, sizeof(m.residuesubmap));
            m.coupling_steps = *mode.i32(pre + "coupling_steps");
            if (m.coupling_steps > 16) throw std::string("too many coupling steps");
            p = mode.i32(pre + "coupling_mag", &n);
            for (size_t j = 0; j < n && j < 16; j++) m.coupling_mag[j] = p[j];
            p = mode.i32(pre + "coupling_ang", &n);
            for (size_t j = 0; j < n && j < 16; j++) m.coupling_ang[j] = p[j];
        }

        // ---- books
        H->books.resize(s.books);
        for (int i = 0; i < s.books; i++) {
            vbm_book &b = H->books[i];
            std::string pre = "book/" + std::to_string(i) + "/";
            const int64_t *h = mode.get<int64_t>(pre + "head", VPK_I64);
            b.dim = (int)h[0];
            b.entries = (int)h[1];
            if (b.dim > VBM_MAX_BOOK_DIM) throw std::string("codebook dimension > 8");
            const signed char *ll = mode.get<signed char>(pre + "lengthlist", VPK_I8);
            b.quantvals = quantvals1(b.entries, b.dim);
            b.minval = (int)rint((double)float32_unpack((long)h[3]));
            b.delta = (int)rint((double)float32_unpack((long)h[4]));
            std::vector<uint32_t> words = make_words(ll, b.entries);
            b.lengthlist = as_off<signed char>(A.put(ll, b.entries));
            b.codelist = as_off<uint32_t>(A.put(words));
            // used entries with their lattice points: the odometer of lib/res0.c:362-368
            std::vector<int> uidx, upt;
            if ((int)h[2] == 1) {
                int e[VBM_MAX_BOOK_DIM] = {0, 0, 0, 0, 0, 0, 0, 0};
                int maxval = b.minval + b.delta * (b.quantvals - 1);
                for (int k = 0; k < b.entries; k++) {
                    if (ll[k] > 0) {
                        uidx.push_back(k);
                        for (int d = 0; d < b.dim; d++) upt.push_back(e[d]);
                    }
                    if (k + 1 >= b.entries) break;
                    int j = 0;
                    while (j < b.dim && e[j] >= maxval) e[j++] = 0;
                    if (j >= b.dim) break;   // lattice exhausted before `entries` (not for vq/-built books)
                    if (e[j] >= 0) e[j] += b.delta;
                    e[j] = -e[j];
                }
            }
            b.used = (int)uidx.size();
            b.used_index = as_off<int>(A.put(uidx));
            b.used_point = as_off<int>(A.put(upt));
            b.used_pack = nullptr;
            b.used_norm = nullptr;
            if (b.used) {
                std::vector<short> pk((size_t)b.used * 8, 0);
                std::vector<int> nrm(b.used, 0);
                bool fits = true;
                for (int u = 0; u < b.used; u++)
                    for (int d = 0; d < b.dim; d++) {
                        const int v = upt[(size_t)u * b.dim + d];
                        if (v < -32768 || v > 32767) fits = false;
                        pk[(size_t)u * 8 + d] = (short)v;
                        nrm[u] += v * v;
                    }
                if (fits) {
                    b.used_pack = as_off<short>(A.put(pk));
                    b.used_norm = as_off<int>(A.put(nrm));
                }
            }
        }
        H->books_off = 0;

        // ---- floors
        for (int i = 0; i < s.floors; i++) {
            vbm_floor &f = s.floor[i];
            std::string pre = "floor/" + std::to_string(i) + "/";
            f.partitions = *mode.i32(pre + "partitions");
            memcpy(f.partitionclass, mode.i32(pre + "partitionclass"), sizeof(f.partitionclass));
            memcpy(f.class_dim, mode.i32(pre + "class_dim"), sizeof(f.class_dim));
            memcpy(f.class_subs, mode.i32(pre + "class_subs"), sizeof(f.class_subs));
            memcpy(f.class_book, mode.i32(pre + "class_book"), sizeof(f.class_book));
            memcpy(f.class_subbook, mode.i32(pre + "class_subbook"), sizeof(f.class_subbook));
            f.mult = *mode.i32(pre + "mult");
            memcpy(f.postlist, mode.i32(pre + "postlist"), sizeof(f.postlist));
            const float *q = mode.f32(pre + "fit");
            f.maxover = q[0]; f.maxunder = q[1]; f.maxerr = q[2]; f.twofitweight = q[3]; f.twofitatten = q[4];
            f.info_n = *mode.i32(pre + "n");
            floor_look(f);
        }

        // ---- residues (res0_look, lib/res0.c:255-313)
        for (int i = 0; i < s.residues; i++) {
            vbm_residue &r = s.residue[i];
            std::string pre = "residue/" + std::to_string(i) + "/";
            const int *h = mode.i32(pre + "head");
            r.type = h[0]; r.begin = h[1]; r.end = h[2]; r.grouping = h[3]; r.partitions = h[4]; r.groupbook = h[6];
            memcpy(r.secondstages, mode.i32(pre + "secondstages"), sizeof(r.secondstages));
            memcpy(r.classmetric1, mode.i32(pre + "classmetric1"), sizeof(r.classmetric1));
            memcpy(r.classmetric2, mode.i32(pre + "classmetric2"), sizeof(r.classmetric2));
            const int *booklist = mode.i32(pre + "booklist");
            int acc = 0, maxstage = 0;
            for (int j = 0; j < 64; j++)
                for (int k = 0; k < 8; k++) r.partbook[j][k] = -1;
            for (int j = 0; j < r.partitions; j++) {
                int stages = ilog((uint32_t)r.secondstages[j]);
                if (stages > maxstage) maxstage = stages;
                for (int k = 0; k < stages; k++)
                    if (r.secondstages[j] & (1 << k)) r.partbook[j][k] = booklist[acc++];
            }
            r.stages = maxstage;
            r.phrase_dim = H->books[r.groupbook].dim;
        }

        // ---- global psy
        {
            memcpy(s.coupling_pointlimit, mode.i32("psy_g/coupling_pointlimit"), sizeof(s.coupling_pointlimit));
            memcpy(s.coupling_prepointamp, mode.i32("psy_g/coupling_prepointamp"), sizeof(s.coupling_prepointamp));
            memcpy(s.coupling_postpointamp, mode.i32("psy_g/coupling_postpointamp"), sizeof(s.coupling_postpointamp));
            memcpy(s.sliding_lowpass, mode.i32("psy_g/sliding_lowpass"), sizeof(s.sliding_lowpass));
            s.ampmax_att_per_sec = mode.f32("psy_g/floats")[2];
            s.managed = mode.has("info/managed") ? *mode.i32("info/managed") : 0;
            s.hi_lowpass_khz = *mode.f64("hi/lowpass_kHz");
            if (s.managed) {   // lib/vorbisenc.c:890-901
                const long long *r = mode.i64("bi/rates");
                const double *d = mode.f64("bi/floats");
                s.bi_avg_rate = r[0]; s.bi_min_rate = r[1]; s.bi_max_rate = r[2]; s.bi_reservoir_bits = r[3];
                s.bi_reservoir_bias = d[0];
                s.bi_slew_damp = d[1];
            }
        }
        int eighth = *mode.i32("psy_g/eighth_octave_lines");

        // ---- psys
        for (int i = 0; i < s.psys; i++) {
            vbm_psy &p = s.psy[i];
            std::string pre = "psy/" + std::to_string(i) + "/";
            const int *a = mode.i32(pre + "ints");
            const float *q = mode.f32(pre + "floats");
            p.blockflag = a[0]; p.noisemaskp = a[1]; p.noisewindowlomin = a[2]; p.noisewindowhimin = a[3];
            p.noisewindowfixed = a[4]; p.normal_p = a[5]; p.normal_start = a[6]; p.normal_partition = a[7];
            p.ath_adjatt = q[0]; p.ath_maxatt = q[1]; p.tone_centerboost = q[2]; p.tone_decay = q[3];
            p.tone_abs_limit = q[4]; p.noisemaxsupp = q[5]; p.noisewindowlo = q[6]; p.noisewindowhi = q[7];
            p.flacint = q[8]; p.max_curve_dB = q[9];
            memcpy(p.tone_masteratt, mode.f32(pre + "tone_masteratt"), sizeof(p.tone_masteratt));
            memcpy(p.toneatt, mode.f32(pre + "toneatt"), sizeof(p.toneatt));
            memcpy(p.noiseoff, mode.f32(pre + "noiseoff"), sizeof(p.noiseoff));
            memcpy(p.noisecompand, mode.f32(pre + "noisecompand"), sizeof(p.noisecompand));
            memcpy(p.noisecompand_high, mode.f32(pre + "noisecompand_high"), sizeof(p.noisecompand_high));
            p.normal_thresh = *mode.get<double>(pre + "normal_thresh", VPK_F64);
            PsyTables t;
            psy_look(p, t, common, eighth, s.blocksizes[p.blockflag] / 2, s.rate);
            p.tonecurves = as_off<float>(A.put(t.tonecurves));
            for (int c = 0; c < VBM_P_NOISECURVES; c++) p.noiseoffset[c] = as_off<float>(A.put(t.noiseoffset[c]));
            p.ath = as_off<float>(A.put(t.ath));
            p.octave = as_off<int>(A.put(t.octave));
            p.bark_lo = as_off<int>(A.put(t.bark_lo));
            p.bark_hi = as_off<int>(A.put(t.bark_hi));
            p.ntfix_noiseoffset = as_off<float>(A.put(t.ntfix));
            // one 16-byte record per run of equal octave[]: first bin, end bin, ath[last] (bits), octave[last]
            for (int g = 0; g < p.ngroups; g++) {
                const int last = t.group_start[g + 1] - 1;
                int bits;
                memcpy(&bits, &t.ath[last], 4);
                t.group_tab.push_back(t.group_start[g]);
                t.group_tab.push_back(t.group_start[g + 1]);
                t.group_tab.push_back(bits);
                t.group_tab.push_back(t.octave[last]);
            }
            p.group_tab = as_off<int>(A.put(t.group_tab));
            p.group_start = as_off<int>(A.put(t.group_start));
            p.seg_p0 = as_off<int>(A.put(t.seg_p0));
            p.seg_p1 = as_off<int>(A.put(t.seg_p1));
            {   // loop limits of bark_noise_hybridmp: table-only conditions, evaluated once here
                const int n = p.n, fixed = p.noisewindowfixed;
                int k = 0;
                for (; k < n; k++) {
                    int lo = t.bark_lo[k], hi = t.bark_hi[k];
                    if (lo >= 0 || -lo >= n) break;
                    if (hi >= n) break;
                }
                p.hy_i1 = k;
                for (; k < n; k++) {
                    int lo = t.bark_lo[k], hi = t.bark_hi[k];
                    if (lo < 0 || lo >= n) break;
                    if (hi >= n) break;
                }
                p.hy_i2 = k;
                p.hy_f1 = p.hy_f2 = 0;
                if (fixed > 0) {
                    for (k = 0; k < n; k++) {
                        int hi = k + fixed / 2, lo = hi - fixed;
                        if (hi >= n) break;
                        if (lo >= 0) break;
                    }
                    p.hy_f1 = k;
                    for (; k < n; k++) {
                        int hi = k + fixed / 2, lo = hi - fixed;
                        if (hi >= n) break;
                        if (lo < 0) break;
                    }
                    p.hy_f2 = k;
                }
                // mirrored window edges read the sums at -lo: every row below this bound is kept (psy_kernels.hip)
                int rb = 1;
                for (k = 0; k < p.hy_i1; k++)
                    if (-t.bark_lo[k] + 1 > rb) rb = -t.bark_lo[k] + 1;
                for (k = 0; k < p.hy_f1; k++) {
                    int lo = k + fixed / 2 - fixed;
                    if (-lo + 1 > rb) rb = -lo + 1;
                }
                p.hy_rb = (rb + 15) & ~15;
                if (p.hy_rb > n) p.hy_rb = n;
                // ring form of the noise mask kernel (noise_kernels.hip, NM_RING = 512 rows, NM_LAG = 4, chunks of 64 bins):
                // bin i of chunk c = i / 64 is solved in iteration c + 4, when the sums are final through chunk c + 3 and
                // the addends of chunk min(c + 5, last) have overwritten chunk min(c + 5, last) - 8.  Every row a bin reads
                // (both edges of the variable window, of the fixed window, mirrored or not; bins past the loop limits keep
                // the window of the last bin inside them) has to lie between the two.
                {
                    const int nch = n / 64;
                    bool ok = (n == 1024) && p.hy_i2 > 0 && (fixed <= 0 || p.hy_f2 > 0);
                    auto row_ok = [&](int i, int row) {
                        const int c = i >> 6, rc = row >> 6;
                        const int done = (c + 3 < nch - 1) ? c + 3 : nch - 1;
                        const int dead = ((c + 5 < nch - 1) ? c + 5 : nch - 1) - 8;
                        return row >= 0 && row < n && rc <= done && rc > dead;
                    };
                    for (k = 0; k < n && ok; k++) {
                        const int iw = (k < p.hy_i2) ? k : p.hy_i2 - 1;
                        const int lo = t.bark_lo[iw], hi = t.bark_hi[iw];
                        ok = ok && row_ok(k, hi) && row_ok(k, iw < p.hy_i1 ? -lo : lo);
                        if (fixed > 0) {
                            const int fw = (k < p.hy_f2) ? k : p.hy_f2 - 1;
                            const int fhi = fw + fixed / 2, flo = fhi - fixed;
                            ok = ok && row_ok(k, fhi) && row_ok(k, fw < p.hy_f1 ? -flo : flo);
                        }
                    }
                    p.hy_ring = ok ? 1 : 0;
                }
            }
        }

        // ---- static tables
        memcpy(s.stereo_threshholds, common.get<double>("stereo_threshholds", VPK_F64), sizeof(s.stereo_threshholds));
        memcpy(s.stereo_threshholds_X, common.get<double>("stereo_threshholds_X", VPK_F64), sizeof(s.stereo_threshholds_X));
        memcpy(s.stn_compand, common.i32("stn_compand"), sizeof(s.stn_compand));
        s.freq_bfn128 = as_off<int>(A.put(common.i32("freq_bfn128"), 128));
        s.freq_bfn256 = as_off<int>(A.put(common.i32("freq_bfn256"), 256));
        s.fromdB = as_off<float>(A.put(common.f32("FLOOR1_fromdB_LOOKUP"), 256));
        for (int b = 0; b < 2; b++) {
            int N = s.blocksizes[b];
            s.window[b] = as_off<float>(A.put(common.f32("window/" + std::to_string(N)), N / 2));
            std::vector<float> trig(N + N / 4), wa(N);
            vbm_host_mdct_trig(N, trig.data());
            vbm_host_fft_twiddles(N, wa.data());
            s.mdct_trig[b] = as_off<float>(A.put(trig));
            s.fft_wa[b] = as_off<float>(A.put(wa));
        }
        // ---- envelope detector look (lib/envelope.c:42-87)
        {
            vbm_envelope &ve = s.ve;
            memcpy(ve.preecho_thresh, mode.f32("psy_g/preecho_thresh"), sizeof(ve.preecho_thresh));
            memcpy(ve.postecho_thresh, mode.f32("psy_g/postecho_thresh"), sizeof(ve.postecho_thresh));
            ve.stretch_penalty = mode.f32("psy_g/floats")[0];
            ve.minenergy = mode.f32("psy_g/floats")[1];
            const int *bb = common.i32("envelope/band_begin"), *be = common.i32("envelope/band_end");
            for (int j = 0; j < VBM_VE_BANDS; j++) {
                ve.band_begin[j] = bb[j];
                ve.band_end[j] = be[j];
                if (be[j] > VBM_VE_MAXBAND || bb[j] + be[j] > 32) throw std::string("envelope band out of range");
                float total = 0.f;
                for (int i = 0; i < VBM_VE_MAXBAND; i++) ve.band_window[j][i] = 0.f;
                for (int i = 0; i < be[j]; i++) {
                    ve.band_window[j][i] = (float)sin(((double)i + .5) / (double)be[j] * M_PI);
                    total += ve.band_window[j][i];
                }
                ve.band_total[j] = (float)(1. / (double)total);
            }
            std::vector<float> win(128), trig(128 + 32);
            for (int i = 0; i < 128; i++) {
                float t = (float)sin((double)i / (128 - 1.) * M_PI);
                win[i] = t * t;
            }
            vbm_host_mdct_trig(128, trig.data());
            ve.mdct_win = as_off<float>(A.put(win));
            ve.mdct_trig = as_off<float>(A.put(trig));
        }
        // the book array itself also lives in the arena (device view is rebased separately)
        H->books_off = A.put(H->books);
        s.book = as_off<vbm_book>(H->books_off);

        // host view
        H->host = s;
        rebase_setup(H->host, A.bytes.data());
        H->host_books = H->books;
        for (auto &b : H->host_books) rebase_book(b, A.bytes.data());
        H->host.book = H->host_books.data();
        return H;
    } catch (const std::string &e) {
        err = e;
        return nullptr;
    }
}

const vbm_setup *vbm_setup_host_view(const vbm_setup_host *H) { return &H->host; }

int vbm_setup_host_upload(vbm_setup_host *H)
{
    if (H->d_setup) return 0;
    hipError_t e;
    size_t sz = H->arena.bytes.size();
    if ((e = hipMalloc((void **)&H->d_arena, sz)) != hipSuccess) return vbm_set_hip_error(e, "hipMalloc(setup arena)");
    // device copy of the arena, with the embedded book structs rebased to device addresses
    std::vector<unsigned char> img = H->arena.bytes;
    vbm_book *bk = reinterpret_cast<vbm_book *>(img.data() + H->books_off);
    for (int i = 0; i < H->s.books; i++) rebase_book(bk[i], H->d_arena);
    if ((e = hipMemcpy(H->d_arena, img.data(), sz, hipMemcpyHostToDevice)) != hipSuccess)
        return vbm_set_hip_error(e, "hipMemcpy(setup arena)");
    vbm_setup dev = H->s;
    rebase_setup(dev, H->d_arena);
    H->dev_view = dev;
    if ((e = hipMalloc((void **)&H->d_setup, sizeof(vbm_setup))) != hipSuccess)
        return vbm_set_hip_error(e, "hipMalloc(setup)");
    if ((e = hipMemcpy(H->d_setup, &dev, sizeof(dev), hipMemcpyHostToDevice)) != hipSuccess)
        return vbm_set_hip_error(e, "hipMemcpy(setup)");
    return 0;
}

const vbm_setup *vbm_setup_device(const vbm_setup_host *H) { return H->d_setup; }
const vbm_setup *vbm_setup_device_ptrs(const vbm_setup_host *H) { return &H->dev_view; }

void vbm_setup_host_free(vbm_setup_host *H)
{
    if (!H) return;
    if (H->d_arena) (void)hipFree(H->d_arena);
    if (H->d_setup) (void)hipFree(H->d_setup);
    delete H;
}
