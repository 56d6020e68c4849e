/* ORACLE — test infrastructure only.
 *
 * Loads a mode pack (tools/make_modepack.py) + data/common.vpk and derives every lookup the
 * reference builds in vorbis_analysis_init():
 *   _vds_shared_init      lib/block.c:181-303   (mdct/fft looks, window ids, books, psy, floor, residue)
 *   _vp_psy_init          lib/psy.c:352-507     (ath, bark windows, octave map, noise offsets)
 *   setup_tone_curves     lib/psy.c:171-350
 *   floor1_look           lib/floor1.c:183-258
 *   res0_look             lib/res0.c:255-313
 *   _ve_envelope_init     lib/envelope.c:42-87
 * All libm calls are the host's, in the same float/double mix as the source expressions.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"
#include "vpk.h"

/* lib/scales.h:69-87 — macros, so argument types drive the promotions exactly as in C */
#define toBARK(n) (13.1f * atan(.00074f * (n)) + 2.24f * atan((n) * (n)*1.85e-8f) + 1e-4f * (n))
#define toOC(n) (log(n) * 1.442695f - 5.965784f)
#define fromOC(o) (exp(((o) + 5.965784f) * .693147f))

static const void *need(const vpk_file *f, const char *name, int dtype, size_t *n)
{
    const void *p = vpk_get(f, name, dtype, n);
    if (!p) {
        fprintf(stderr, "oracle: mode pack entry '%s' missing or of wrong type\n", name);
        abort();
    }
    return p;
}

static const int *geti(const vpk_file *f, const char *fmt, int idx, const char *leaf, size_t *n)
{
    char name[128];
    snprintf(name, sizeof(name), fmt, idx, leaf);
    return (const int *)need(f, name, VPK_I32, n);
}

static const float *getf(const vpk_file *f, const char *fmt, int idx, const char *leaf, size_t *n)
{
    char name[128];
    snprintf(name, sizeof(name), fmt, idx, leaf);
    return (const float *)need(f, name, VPK_F32, n);
}

/* ---- tone curves, lib/psy.c:171-350 ------------------------------------------------- */
static void min_curve(float *c, const float *c2)
{
    int i;
    for (i = 0; i < ORC_EHMER_MAX; i++)
        if (c2[i] < c[i]) c[i] = c2[i];
}
static void max_curve(float *c, const float *c2)
{
    int i;
    for (i = 0; i < ORC_EHMER_MAX; i++)
        if (c2[i] > c[i]) c[i] = c2[i];
}
static void attenuate_curve(float *c, float att)
{
    int i;
    for (i = 0; i < ORC_EHMER_MAX; i++) c[i] += att;
}

static void setup_tone_curves(const orc_common *cm, float (*ret)[ORC_P_LEVELS][ORC_EHMER_MAX + 2],
                              const float curveatt_dB[ORC_P_BANDS], float binHz, int n, float center_boost,
                              float center_decay_rate)
{
    int i, j, k, m;
    float ath[ORC_EHMER_MAX];
    static float workc[ORC_P_BANDS][ORC_P_LEVELS][ORC_EHMER_MAX];
    float athc[ORC_P_LEVELS][ORC_EHMER_MAX];
    float *brute_buffer = (float *)malloc(n * sizeof(*brute_buffer));
    const float *ATH = cm->ATH;
    const float(*tonemasks)[6][ORC_EHMER_MAX] = (const float(*)[6][ORC_EHMER_MAX])cm->tonemasks;

    memset(workc, 0, sizeof(workc));

    for (i = 0; i < ORC_P_BANDS; i++) {
        int ath_offset = i * 4;
        for (j = 0; j < ORC_EHMER_MAX; j++) {
            float min = 999.;
            for (k = 0; k < 4; k++)
                if (j + k + ath_offset < ORC_MAX_ATH) {
                    if (min > ATH[j + k + ath_offset]) min = ATH[j + k + ath_offset];
                } else {
                    if (min > ATH[ORC_MAX_ATH - 1]) min = ATH[ORC_MAX_ATH - 1];
                }
            ath[j] = min;
        }

        for (j = 0; j < 6; j++) memcpy(workc[i][j + 2], tonemasks[i][j], ORC_EHMER_MAX * sizeof(float));
        memcpy(workc[i][0], tonemasks[i][0], ORC_EHMER_MAX * sizeof(float));
        memcpy(workc[i][1], tonemasks[i][0], ORC_EHMER_MAX * sizeof(float));

        for (j = 0; j < ORC_P_LEVELS; j++) {
            for (k = 0; k < ORC_EHMER_MAX; k++) {
                float adj = center_boost + abs(ORC_EHMER_OFFSET - k) * center_decay_rate;
                if (adj < 0. && center_boost > 0) adj = 0.;
                if (adj > 0. && center_boost < 0) adj = 0.;
                workc[i][j][k] += adj;
            }
        }

        for (j = 0; j < ORC_P_LEVELS; j++) {
            attenuate_curve(workc[i][j], curveatt_dB[i] + 100. - (j < 2 ? 2 : j) * 10. - 30.);
            memcpy(athc[j], ath, ORC_EHMER_MAX * sizeof(**athc));
            attenuate_curve(athc[j], +100. - j * 10.f - 30.);
            max_curve(athc[j], workc[i][j]);
        }

        for (j = 1; j < ORC_P_LEVELS; j++) {
            min_curve(athc[j], athc[j - 1]);
            min_curve(workc[i][j], athc[j]);
        }
    }

    for (i = 0; i < ORC_P_BANDS; i++) {
        int hi_curve, lo_curve, bin;

        bin = floor(fromOC(i * .5) / binHz);
        lo_curve = ceil(toOC(bin * binHz + 1) * 2);
        hi_curve = floor(toOC((bin + 1) * binHz) * 2);
        if (lo_curve > i) lo_curve = i;
        if (lo_curve < 0) lo_curve = 0;
        if (hi_curve >= ORC_P_BANDS) hi_curve = ORC_P_BANDS - 1;

        for (m = 0; m < ORC_P_LEVELS; m++) {
            for (j = 0; j < n; j++) brute_buffer[j] = 999.;

            for (k = lo_curve; k <= hi_curve; k++) {
                int l = 0;
                for (j = 0; j < ORC_EHMER_MAX; j++) {
                    int lo_bin = fromOC(j * .125 + k * .5 - 2.0625) / binHz;
                    int hi_bin = fromOC(j * .125 + k * .5 - 1.9375) / binHz + 1;
                    if (lo_bin < 0) lo_bin = 0;
                    if (lo_bin > n) lo_bin = n;
                    if (lo_bin < l) l = lo_bin;
                    if (hi_bin < 0) hi_bin = 0;
                    if (hi_bin > n) hi_bin = n;
                    for (; l < hi_bin && l < n; l++)
                        if (brute_buffer[l] > workc[k][m][j]) brute_buffer[l] = workc[k][m][j];
                }
                for (; l < n; l++)
                    if (brute_buffer[l] > workc[k][m][ORC_EHMER_MAX - 1])
                        brute_buffer[l] = workc[k][m][ORC_EHMER_MAX - 1];
            }

            if (i + 1 < ORC_P_BANDS) {
                int l = 0;
                k = i + 1;
                for (j = 0; j < ORC_EHMER_MAX; j++) {
                    int lo_bin = fromOC(j * .125 + i * .5 - 2.0625) / binHz;
                    int hi_bin = fromOC(j * .125 + i * .5 - 1.9375) / binHz + 1;
                    if (lo_bin < 0) lo_bin = 0;
                    if (lo_bin > n) lo_bin = n;
                    if (lo_bin < l) l = lo_bin;
                    if (hi_bin < 0) hi_bin = 0;
                    if (hi_bin > n) hi_bin = n;
                    for (; l < hi_bin && l < n; l++)
                        if (brute_buffer[l] > workc[k][m][j]) brute_buffer[l] = workc[k][m][j];
                }
                for (; l < n; l++)
                    if (brute_buffer[l] > workc[k][m][ORC_EHMER_MAX - 1])
                        brute_buffer[l] = workc[k][m][ORC_EHMER_MAX - 1];
            }

            for (j = 0; j < ORC_EHMER_MAX; j++) {
                int bin2 = fromOC(j * .125 + i * .5 - 2.) / binHz;
                if (bin2 < 0) {
                    ret[i][m][j + 2] = -999.;
                } else {
                    if (bin2 >= n) ret[i][m][j + 2] = -999.;
                    else ret[i][m][j + 2] = brute_buffer[bin2];
                }
            }

            for (j = 0; j < ORC_EHMER_OFFSET; j++)
                if (ret[i][m][j + 2] > -200.f) break;
            ret[i][m][0] = j;

            for (j = ORC_EHMER_MAX - 1; j > ORC_EHMER_OFFSET + 1; j--)
                if (ret[i][m][j + 2] > -200.f) break;
            ret[i][m][1] = j;
        }
    }
    free(brute_buffer);
}

/* lib/psy.c:352-507 */
static void psy_look_init(const orc_common *cm, orc_psy *p, const orc_psyg *gi, int n, long rate)
{
    long i, j, lo = -99, hi = 1;
    long maxoc, select = -1;
    const float *ATH = cm->ATH;

    p->eighth_octave_lines = gi->eighth_octave_lines;
    p->shiftoc = rint(log(gi->eighth_octave_lines * 8.f) / log(2.f)) - 1;

    p->firstoc = toOC(.25f * rate * .5 / n) * (1 << (p->shiftoc + 1)) - gi->eighth_octave_lines;
    maxoc = toOC((n + .25f) * rate * .5 / n) * (1 << (p->shiftoc + 1)) + .5f;
    p->total_octave_lines = maxoc - p->firstoc + 1;
    p->ath = (float *)malloc(n * sizeof(*p->ath));
    p->octave = (long *)malloc(n * sizeof(*p->octave));
    p->bark = (long *)malloc(n * sizeof(*p->bark));
    p->n = n;
    p->rate = rate;

    p->n25p = n / 4;
    p->n33p = n / 3;
    p->n75p = p->n25p * 3;
    p->nn25pt = p->normal_partition / 4;
    p->nn50pt = p->nn25pt + p->nn25pt;
    p->nn75pt = p->nn25pt * 3;

    for (i = 0; i < 4; i++) p->m3n[i] = 0;
    if (rate < 26000) {
        p->m_val = 0;
        select = -1;
    } else if (rate < 38000) {
        p->m_val = .93;
        if (n == 128) { select = 0; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n32[i]; }
        else if (n == 256) { select = 1; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n32x2[i]; }
        else if (n == 1024) select = 2;
        else if (n == 2048) select = 3;
    } else if (rate > 46000) {
        p->m_val = 1.205;
        if (n == 128) { select = 4; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n48[i]; }
        else if (n == 256) { select = 5; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n48x2[i]; }
        else if (n == 1024) select = 6;
        else if (n == 2048) select = 7;
    } else {
        p->m_val = 1.;
        if (n == 128) { select = 8; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n44[i]; }
        else if (n == 256) { select = 9; for (i = 0; i < 3; i++) p->m3n[i] = cm->m3n44x2[i]; }
        else if (n == 1024) select = 10;
        else if (n == 2048) select = 11;
    }

    if (select < 0) {
        p->tonecomp_endp = 0;
        p->tonecomp_thres = .25;
        p->min_nn_lp = 0;
        p->tonefix_end = 0;
    } else {
        p->tonecomp_endp = cm->aotuv_ints[select * 3 + 0];
        p->tonecomp_thres = cm->aotuv_thres[select];
        p->min_nn_lp = cm->aotuv_ints[select * 3 + 1];
        p->tonefix_end = cm->aotuv_ints[select * 3 + 2];
    }

    for (i = 0, j = 0; i < ORC_MAX_ATH - 1; i++) {
        int endpos = rint(fromOC((i + 1) * .125 - 2.) * 2 * n / rate);
        float base = ATH[i];
        if (j < endpos) {
            float delta = (ATH[i + 1] - base) / (endpos - j);
            for (; j < endpos && j < n; j++) {
                p->ath[j] = base + 100.;
                base += delta;
            }
        }
    }
    {
        float cs = p->ath[j - 1];
        float ds = p->ath[j - 1] - p->ath[j - 2];
        for (i = j; i < n; i++, cs += ds) p->ath[i] = cs;
    }

    for (i = 0; i < n; i++) {
        float bark = toBARK(rate / (2 * n) * i);

        for (; lo + p->noisewindowlomin < i && toBARK(rate / (2 * n) * lo) < (bark - p->noisewindowlo); lo++)
            ;
        for (; hi <= n && (hi < i + p->noisewindowhimin || toBARK(rate / (2 * n) * hi) < (bark + p->noisewindowhi));
             hi++)
            ;
        p->bark[i] = ((lo - 1) << 16) + (hi - 1);
    }

    for (i = 0; i < n; i++) p->octave[i] = toOC((i + .25f) * .5 * rate / n) * (1 << (p->shiftoc + 1)) + .5f;

    setup_tone_curves(cm, p->tonecurves, p->toneatt, rate * .5 / n, n, p->tone_centerboost, p->tone_decay);

    for (i = 0; i < ORC_P_NOISECURVES; i++) p->noiseoffset[i] = (float *)malloc(n * sizeof(float));
    p->ntfix_noiseoffset = (float *)malloc(n * sizeof(float));

    for (i = 0; i < n; i++) {
        float halfoc = toOC((i + .5) * rate / (2. * n)) * 2.;
        int inthalfoc;
        float del;

        if (halfoc < 0) halfoc = 0;
        if (halfoc >= ORC_P_BANDS - 1) halfoc = ORC_P_BANDS - 1;
        inthalfoc = (int)halfoc;
        del = halfoc - inthalfoc;

        for (j = 0; j < ORC_P_NOISECURVES; j++)
            p->noiseoffset[j][i] = p->noiseoff[j][inthalfoc] * (1. - del) + p->noiseoff[j][inthalfoc + 1] * del;

        p->ntfix_noiseoffset[i] = cm->ntfix_offset[inthalfoc] * (1. - del) + cm->ntfix_offset[inthalfoc + 1] * del;
    }
}

/* lib/floor1.c:183-258 */
static int icomp_ptr(const void *a, const void *b) { return (**(int **)a > **(int **)b) - (**(int **)a < **(int **)b); }

static void floor_look_init(orc_floor *look)
{
    int *sortpointer[ORC_VIF_POSIT + 2];
    int i, j, n = 0;
    look->n = look->postlist[1];
    for (i = 0; i < look->partitions; i++) n += look->class_dim[look->partitionclass[i]];
    n += 2;
    look->posts = n;
    for (i = 0; i < n; i++) sortpointer[i] = look->postlist + i;
    qsort(sortpointer, n, sizeof(*sortpointer), icomp_ptr);
    for (i = 0; i < n; i++) look->forward_index[i] = sortpointer[i] - look->postlist;
    for (i = 0; i < n; i++) look->reverse_index[look->forward_index[i]] = i;
    for (i = 0; i < n; i++) look->sorted_index[i] = look->postlist[look->forward_index[i]];
    switch (look->mult) {
    case 1: look->quant_q = 256; break;
    case 2: look->quant_q = 128; break;
    case 3: look->quant_q = 86; break;
    case 4: look->quant_q = 64; break;
    }
    for (i = 0; i < n - 2; i++) {
        int lo = 0, hi = 1, lx = 0, hx = look->n;
        int currentx = look->postlist[i + 2];
        for (j = 0; j < i + 2; j++) {
            int x = look->postlist[j];
            if (x > lx && x < currentx) { lo = j; lx = x; }
            if (x < hx && x > currentx) { hi = j; hx = x; }
        }
        look->loneighbor[i] = lo;
        look->hineighbor[i] = hi;
    }
}

/* lib/res0.c:255-313 (encode-relevant part) */
static void residue_look_init(orc_residue *r, const orc_book *books)
{
    int j, k, acc = 0, dim, maxstage = 0;
    r->parts = r->partitions;
    r->phrasebook = books + r->groupbook;
    dim = r->phrasebook->dim;
    memset(r->partbooks, 0, sizeof(r->partbooks));
    for (j = 0; j < r->parts; j++) {
        int stages = orc_ilog(r->secondstages[j]);
        if (stages) {
            if (stages > maxstage) maxstage = stages;
            for (k = 0; k < stages; k++)
                if (r->secondstages[j] & (1 << k)) r->partbooks[j][k] = books + r->booklist[acc++];
        }
    }
    r->partvals = 1;
    for (j = 0; j < dim; j++) r->partvals *= r->parts;
    r->stages = maxstage;
}

/* lib/envelope.c:42-87 */
static void envelope_look_init(orc_setup *s)
{
    int i, j, n = 128;
    s->ve_minenergy = s->psy_g.preecho_minenergy;
    s->ve_mdct_win = (float *)calloc(n, sizeof(float));
    orc_mdct_init(&s->ve_mdct, n);
    for (i = 0; i < n; i++) {
        float t = sin(i / (n - 1.) * M_PI);
        s->ve_mdct_win[i] = t * t;
    }
    for (j = 0; j < ORC_VE_BANDS; j++) {
        orc_ve_band *b = &s->ve_band[j];
        b->begin = s->c.ve_band_begin[j];
        b->end = s->c.ve_band_end[j];
        n = b->end;
        b->window = (float *)malloc(n * sizeof(float));
        b->total = 0.f;
        for (i = 0; i < n; i++) {
            b->window[i] = sin((i + .5) / n * M_PI);
            b->total += b->window[i];
        }
        b->total = 1. / b->total;
    }
}

static int winno(long blocksize) { return orc_ilog((uint32_t)blocksize) - 7; }

orc_setup *orc_setup_load(const char *common_vpk, const char *mode_vpk)
{
    orc_setup *s = (orc_setup *)calloc(1, sizeof(*s));
    vpk_file *fc = (vpk_file *)calloc(1, sizeof(vpk_file));
    vpk_file *fm = (vpk_file *)calloc(1, sizeof(vpk_file));
    size_t n;
    int i, j;
    char name[128];
    if (vpk_open(fc, common_vpk) || vpk_open(fm, mode_vpk)) {
        fprintf(stderr, "oracle: cannot open %s / %s\n", common_vpk, mode_vpk);
        free(fc); free(fm); free(s);
        return NULL;
    }
    s->packs[0] = fc;
    s->packs[1] = fm;

    /* common tables */
    for (i = 0; i < 8; i++) {
        snprintf(name, sizeof(name), "window/%d", 64 << i);
        s->c.window[i] = (const float *)need(fc, name, VPK_F32, &n);
    }
    s->c.ATH = (const float *)need(fc, "ATH", VPK_F32, &n);
    s->c.tonemasks = (const float *)need(fc, "tonemasks", VPK_F32, &n);
    s->c.stereo_threshholds = (const double *)need(fc, "stereo_threshholds", VPK_F64, &n);
    s->c.stereo_threshholds_X = (const double *)need(fc, "stereo_threshholds_X", VPK_F64, &n);
    s->c.m3n32 = (const int *)need(fc, "m3n32", VPK_I32, &n);
    s->c.m3n44 = (const int *)need(fc, "m3n44", VPK_I32, &n);
    s->c.m3n48 = (const int *)need(fc, "m3n48", VPK_I32, &n);
    s->c.m3n32x2 = (const int *)need(fc, "m3n32x2", VPK_I32, &n);
    s->c.m3n44x2 = (const int *)need(fc, "m3n44x2", VPK_I32, &n);
    s->c.m3n48x2 = (const int *)need(fc, "m3n48x2", VPK_I32, &n);
    s->c.freq_bfn128 = (const int *)need(fc, "freq_bfn128", VPK_I32, &n);
    s->c.freq_bfn256 = (const int *)need(fc, "freq_bfn256", VPK_I32, &n);
    s->c.stn_compand = (const int *)need(fc, "stn_compand", VPK_I32, &n);
    s->c.ntfix_offset = (const float *)need(fc, "ntfix_offset", VPK_F32, &n);
    s->c.aotuv_ints = (const int *)need(fc, "aotuv_preset/ints", VPK_I32, &n);
    s->c.aotuv_thres = (const float *)need(fc, "aotuv_preset/tonecomp_thres", VPK_F32, &n);
    s->c.fromdB = (const float *)need(fc, "FLOOR1_fromdB_LOOKUP", VPK_F32, &n);
    s->c.ve_band_begin = (const int *)need(fc, "envelope/band_begin", VPK_I32, &n);
    s->c.ve_band_end = (const int *)need(fc, "envelope/band_end", VPK_I32, &n);

    /* codec_setup_info */
    s->channels = *(const int *)need(fm, "info/channels", VPK_I32, &n);
    s->rate = (long)*(const int64_t *)need(fm, "info/rate", VPK_I64, &n);
    {
        const int *bs = (const int *)need(fm, "info/blocksizes", VPK_I32, &n);
        const int *cnt = (const int *)need(fm, "info/counts", VPK_I32, &n);
        const int *lp = (const int *)need(fm, "info/block_lowpassr", VPK_I32, &n);
        s->blocksizes[0] = bs[0];
        s->blocksizes[1] = bs[1];
        s->modes = cnt[0]; s->maps = cnt[1]; s->floors = cnt[2];
        s->residues = cnt[3]; s->books = cnt[4]; s->psys = cnt[5];
        s->block_lowpassr[0] = lp[0];
        s->block_lowpassr[1] = lp[1];
    }
    s->pre_amplitude = *(const float *)need(fm, "hi/pre_amplitude", VPK_F32, &n);
    s->managed = *(const int *)need(fm, "info/managed", VPK_I32, &n);
    {
        const int64_t *br = (const int64_t *)need(fm, "info/bitrates", VPK_I64, &n);   /* upper, nominal, lower */
        s->bitrate_upper = (long)br[0];
        s->bitrate_nominal = (long)br[1];
        s->bitrate_lower = (long)br[2];
    }
    if (s->managed) { /* lib/vorbisenc.c:890-901 */
        const int64_t *r = (const int64_t *)need(fm, "bi/rates", VPK_I64, &n);
        const double *d = (const double *)need(fm, "bi/floats", VPK_F64, &n);
        s->bi_avg_rate = (long)r[0]; s->bi_min_rate = (long)r[1]; s->bi_max_rate = (long)r[2];
        s->bi_reservoir_bits = (long)r[3];
        s->bi_reservoir_bias = d[0];
        s->bi_slew_damp = d[1];
    }
    if (s->channels > ORC_MAXCH || s->floors > 4 || s->residues > 4 || s->psys > 4 || s->modes > 2) {
        fprintf(stderr, "oracle: setup exceeds static limits\n");
        abort();
    }
    for (i = 0; i < s->modes; i++) {
        const int *m = geti(fm, "mode/%d%s", i, "", &n);
        s->mode_blockflag[i] = m[0];
        s->mode_mapping[i] = m[3];
    }
    for (i = 0; i < s->maps; i++) {
        orc_map *m = &s->map[i];
        const int *p;
        m->submaps = *geti(fm, "map/%d/%s", i, "submaps", &n);
        p = geti(fm, "map/%d/%s", i, "chmuxlist", &n);
        for (j = 0; j < (int)n; j++) m->chmuxlist[j] = p[j];
        p = geti(fm, "map/%d/%s", i, "floorsubmap", &n);
        for (j = 0; j < 16; j++) m->floorsubmap[j] = p[j];
        p = geti(fm, "map/%d/%s", i, "residuesubmap", &n);
        for (j = 0; j < 16; j++) m->residuesubmap[j] = p[j];
        m->coupling_steps = *geti(fm, "map/%d/%s", i, "coupling_steps", &n);
        p = geti(fm, "map/%d/%s", i, "coupling_mag", &n);
        for (j = 0; j < (int)n; j++) m->coupling_mag[j] = p[j];
        p = geti(fm, "map/%d/%s", i, "coupling_ang", &n);
        for (j = 0; j < (int)n; j++) m->coupling_ang[j] = p[j];
    }
    /* books */
    s->book = (orc_book *)calloc(s->books, sizeof(orc_book));
    for (i = 0; i < s->books; i++) {
        orc_book *b = &s->book[i];
        const int64_t *h;
        snprintf(name, sizeof(name), "book/%d/head", i);
        h = (const int64_t *)need(fm, name, VPK_I64, &n);
        b->dim = (int)h[0]; b->entries = (int)h[1]; b->maptype = (int)h[2];
        b->q_min = (long)h[3]; b->q_delta = (long)h[4]; b->q_quant = (int)h[5]; b->q_sequencep = (int)h[6];
        b->nquant = (int)h[7];
        snprintf(name, sizeof(name), "book/%d/lengthlist", i);
        b->lengthlist = (const signed char *)need(fm, name, VPK_I8, &n);
        snprintf(name, sizeof(name), "book/%d/quantlist", i);
        b->quantlist = (const int *)vpk_get(fm, name, VPK_I32, &n);
        orc_book_init_encode(b);
    }
    for (i = 0; i < s->floors; i++) {
        orc_floor *f = &s->floor[i];
        const int *p;
        const float *q;
        f->partitions = *geti(fm, "floor/%d/%s", i, "partitions", &n);
        p = geti(fm, "floor/%d/%s", i, "partitionclass", &n); memcpy(f->partitionclass, p, sizeof(f->partitionclass));
        p = geti(fm, "floor/%d/%s", i, "class_dim", &n); memcpy(f->class_dim, p, sizeof(f->class_dim));
        p = geti(fm, "floor/%d/%s", i, "class_subs", &n); memcpy(f->class_subs, p, sizeof(f->class_subs));
        p = geti(fm, "floor/%d/%s", i, "class_book", &n); memcpy(f->class_book, p, sizeof(f->class_book));
        p = geti(fm, "floor/%d/%s", i, "class_subbook", &n); memcpy(f->class_subbook, p, sizeof(f->class_subbook));
        f->mult = *geti(fm, "floor/%d/%s", i, "mult", &n);
        p = geti(fm, "floor/%d/%s", i, "postlist", &n); memcpy(f->postlist, p, sizeof(f->postlist));
        q = getf(fm, "floor/%d/%s", i, "fit", &n);
        f->maxover = q[0]; f->maxunder = q[1]; f->maxerr = q[2]; f->twofitweight = q[3]; f->twofitatten = q[4];
        f->info_n = *geti(fm, "floor/%d/%s", i, "n", &n);
        floor_look_init(f);
    }
    for (i = 0; i < s->residues; i++) {
        orc_residue *r = &s->residue[i];
        const int *h = geti(fm, "residue/%d/%s", i, "head", &n);
        const int *p;
        r->type = h[0]; r->begin = h[1]; r->end = h[2]; r->grouping = h[3]; r->partitions = h[4];
        r->partvals_info = h[5]; r->groupbook = h[6];
        p = geti(fm, "residue/%d/%s", i, "secondstages", &n); memcpy(r->secondstages, p, sizeof(r->secondstages));
        p = geti(fm, "residue/%d/%s", i, "booklist", &n); memcpy(r->booklist, p, sizeof(r->booklist));
        p = geti(fm, "residue/%d/%s", i, "classmetric1", &n); memcpy(r->classmetric1, p, sizeof(r->classmetric1));
        p = geti(fm, "residue/%d/%s", i, "classmetric2", &n); memcpy(r->classmetric2, p, sizeof(r->classmetric2));
        residue_look_init(r, s->book);
    }
    {
        orc_psyg *g = &s->psy_g;
        const int *p;
        const float *q;
        g->eighth_octave_lines = *(const int *)need(fm, "psy_g/eighth_octave_lines", VPK_I32, &n);
        q = (const float *)need(fm, "psy_g/preecho_thresh", VPK_F32, &n); memcpy(g->preecho_thresh, q, sizeof(g->preecho_thresh));
        q = (const float *)need(fm, "psy_g/postecho_thresh", VPK_F32, &n); memcpy(g->postecho_thresh, q, sizeof(g->postecho_thresh));
        q = (const float *)need(fm, "psy_g/floats", VPK_F32, &n);
        g->stretch_penalty = q[0]; g->preecho_minenergy = q[1]; g->ampmax_att_per_sec = q[2];
        p = (const int *)need(fm, "psy_g/coupling_pkHz", VPK_I32, &n); memcpy(g->coupling_pkHz, p, sizeof(g->coupling_pkHz));
        p = (const int *)need(fm, "psy_g/coupling_pointlimit", VPK_I32, &n); memcpy(g->coupling_pointlimit, p, sizeof(g->coupling_pointlimit));
        p = (const int *)need(fm, "psy_g/coupling_prepointamp", VPK_I32, &n); memcpy(g->coupling_prepointamp, p, sizeof(g->coupling_prepointamp));
        p = (const int *)need(fm, "psy_g/coupling_postpointamp", VPK_I32, &n); memcpy(g->coupling_postpointamp, p, sizeof(g->coupling_postpointamp));
        p = (const int *)need(fm, "psy_g/sliding_lowpass", VPK_I32, &n); memcpy(g->sliding_lowpass, p, sizeof(g->sliding_lowpass));
    }
    for (i = 0; i < s->psys; i++) {
        orc_psy *p = &s->psy[i];
        const int *a = geti(fm, "psy/%d/%s", i, "ints", &n);
        const float *q = getf(fm, "psy/%d/%s", i, "floats", &n);
        const float *t;
        p->blockflag = a[0]; p->noisemaskp = a[1]; p->noisewindowlomin = a[2]; p->noisewindowhimin = a[3];
        p->noisewindowfixed = a[4]; p->normal_p = a[5]; p->normal_start = a[6]; p->normal_partition = a[7];
        p->ath_adjatt = q[0]; p->ath_maxatt = q[1]; p->tone_centerboost = q[2]; p->tone_decay = q[3];
        p->tone_abs_limit = q[4]; p->noisemaxsupp = q[5]; p->noisewindowlo = q[6]; p->noisewindowhi = q[7];
        p->flacint = q[8]; p->max_curve_dB = q[9];
        t = getf(fm, "psy/%d/%s", i, "tone_masteratt", &n); memcpy(p->tone_masteratt, t, sizeof(p->tone_masteratt));
        t = getf(fm, "psy/%d/%s", i, "toneatt", &n); memcpy(p->toneatt, t, sizeof(p->toneatt));
        t = getf(fm, "psy/%d/%s", i, "noiseoff", &n); memcpy(p->noiseoff, t, sizeof(p->noiseoff));
        t = getf(fm, "psy/%d/%s", i, "noisecompand", &n); memcpy(p->noisecompand, t, sizeof(p->noisecompand));
        t = getf(fm, "psy/%d/%s", i, "noisecompand_high", &n); memcpy(p->noisecompand_high, t, sizeof(p->noisecompand_high));
        snprintf(name, sizeof(name), "psy/%d/normal_thresh", i);
        p->normal_thresh = *(const double *)need(fm, name, VPK_F64, &n);
        psy_look_init(&s->c, p, &s->psy_g, (int)(s->blocksizes[p->blockflag] / 2), s->rate);
    }

    s->modebits = orc_ilog((uint32_t)(s->modes - 1));
    orc_mdct_init(&s->mdct[0], (int)s->blocksizes[0]);
    orc_mdct_init(&s->mdct[1], (int)s->blocksizes[1]);
    orc_drft_init(&s->fft[0], (int)s->blocksizes[0]);
    orc_drft_init(&s->fft[1], (int)s->blocksizes[1]);
    s->window[0] = winno(s->blocksizes[0]);
    s->window[1] = winno(s->blocksizes[1]);
    envelope_look_init(s);
    return s;
}

void orc_setup_free(orc_setup *s)
{
    int i, j;
    if (!s) return;
    for (i = 0; i < s->books; i++) free(s->book[i].codelist);
    free(s->book);
    for (i = 0; i < s->psys; i++) {
        orc_psy *p = &s->psy[i];
        free(p->ath); free(p->octave); free(p->bark); free(p->ntfix_noiseoffset);
        for (j = 0; j < ORC_P_NOISECURVES; j++) free(p->noiseoffset[j]);
    }
    orc_mdct_clear(&s->mdct[0]); orc_mdct_clear(&s->mdct[1]);
    orc_drft_clear(&s->fft[0]); orc_drft_clear(&s->fft[1]);
    orc_mdct_clear(&s->ve_mdct);
    free(s->ve_mdct_win);
    for (j = 0; j < ORC_VE_BANDS; j++) free(s->ve_band[j].window);
    for (i = 0; i < 2; i++)
        if (s->packs[i]) { vpk_close((vpk_file *)s->packs[i]); free(s->packs[i]); }
    free(s);
}

/* name-based access to the derived tables, for comparing the product's host tables with the
 * oracle's (tests/test_setup_tables.py).  kind: 'f' float32, 'i' int32, 'q' int64, 'u' uint32, 'b' int8 */
int orc_setup_table(const orc_setup *s, const char *name, const void **data, long *count, char *kind)
{
    int idx = -1;
    char leaf[64] = {0};
    static int sc[16];
#define RET(ptr, n, k) do { *data = (ptr); *count = (long)(n); *kind = (k); return 0; } while (0)
    if (sscanf(name, "psy/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->psys) {
        const orc_psy *p = &s->psy[idx];
        if (!strcmp(leaf, "ath")) RET(p->ath, p->n, 'f');
        if (!strcmp(leaf, "octave")) RET(p->octave, p->n, 'q');
        if (!strcmp(leaf, "bark")) RET(p->bark, p->n, 'q');
        if (!strcmp(leaf, "tonecurves")) RET(p->tonecurves, ORC_P_BANDS * ORC_P_LEVELS * (ORC_EHMER_MAX + 2), 'f');
        if (!strcmp(leaf, "noiseoffset0")) RET(p->noiseoffset[0], p->n, 'f');
        if (!strcmp(leaf, "noiseoffset1")) RET(p->noiseoffset[1], p->n, 'f');
        if (!strcmp(leaf, "noiseoffset2")) RET(p->noiseoffset[2], p->n, 'f');
        if (!strcmp(leaf, "ntfix_noiseoffset")) RET(p->ntfix_noiseoffset, p->n, 'f');
        if (!strcmp(leaf, "scalars")) {
            union { float f; int i; } u;
            int v[16] = {p->n, (int)p->firstoc, (int)p->shiftoc, p->eighth_octave_lines, p->total_octave_lines,
                         p->m3n[0], p->m3n[1], p->m3n[2], p->tonecomp_endp, p->min_nn_lp, p->tonefix_end,
                         p->n25p, p->n33p, p->n75p, 0, 0};
            memcpy(sc, v, sizeof(v));
            u.f = p->m_val; sc[14] = u.i;
            u.f = p->tonecomp_thres; sc[15] = u.i;
            RET(sc, 16, 'i');
        }
    }
    if (sscanf(name, "floor/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->floors) {
        const orc_floor *f = &s->floor[idx];
        if (!strcmp(leaf, "sorted_index")) RET(f->sorted_index, f->posts, 'i');
        if (!strcmp(leaf, "forward_index")) RET(f->forward_index, f->posts, 'i');
        if (!strcmp(leaf, "reverse_index")) RET(f->reverse_index, f->posts, 'i');
        if (!strcmp(leaf, "loneighbor")) RET(f->loneighbor, f->posts - 2, 'i');
        if (!strcmp(leaf, "hineighbor")) RET(f->hineighbor, f->posts - 2, 'i');
        if (!strcmp(leaf, "scalars")) {
            sc[0] = f->posts; sc[1] = f->n; sc[2] = f->quant_q; sc[3] = f->info_n;
            RET(sc, 4, 'i');
        }
    }
    if (sscanf(name, "book/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->books) {
        const orc_book *b = &s->book[idx];
        if (!strcmp(leaf, "codelist")) RET(b->codelist, b->entries, 'u');
        if (!strcmp(leaf, "lengthlist")) RET(b->lengthlist, b->entries, 'b');
        if (!strcmp(leaf, "scalars")) {
            sc[0] = b->dim; sc[1] = b->entries; sc[2] = b->quantvals; sc[3] = b->minval; sc[4] = b->delta; sc[5] = 0;
            RET(sc, 6, 'i');
        }
    }
    if (sscanf(name, "residue/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->residues) {
        const orc_residue *r = &s->residue[idx];
        if (!strcmp(leaf, "partbook")) {
            static int pb[64 * 8];
            int j, k;
            for (j = 0; j < 64; j++)
                for (k = 0; k < 8; k++) pb[j * 8 + k] = r->partbooks[j][k] ? (int)(r->partbooks[j][k] - s->book) : -1;
            RET(pb, 64 * 8, 'i');
        }
        if (!strcmp(leaf, "scalars")) {
            sc[0] = r->type; sc[1] = (int)r->begin; sc[2] = (int)r->end; sc[3] = r->grouping; sc[4] = r->partitions;
            sc[5] = r->groupbook; sc[6] = r->stages; sc[7] = r->phrasebook->dim;
            RET(sc, 8, 'i');
        }
    }
    if (!strcmp(name, "info")) {
        int v[12] = {s->channels, (int)s->rate, (int)s->blocksizes[0], (int)s->blocksizes[1], s->modes, s->maps,
                     s->floors, s->residues, s->books, s->psys, s->block_lowpassr[0], s->block_lowpassr[1]};
        memcpy(sc, v, sizeof(v));
        RET(sc, 12, 'i');
    }
#undef RET
    return -1;
}
