/* ORACLE — test infrastructure only.
 *
 * Psychoacoustic passes of the reference's scalar path (aoTuV b6.03 tuning):
 *   _postnoise_detection            lib/psy.c:619-648
 *   seed_curve / seed_loop          lib/psy.c:652-771
 *   seed_chase / max_seeds          lib/psy.c:773-1085
 *   bark_noise_hybridmp             lib/psy.c:3480-3638
 *   ntfix                           lib/psy.c:3645-3768
 *   _vp_noisemask                   lib/psy.c:3770-4074
 *   _vp_tonemask                    lib/psy.c:4076-4142
 *   set_m3p / _vp_offset_and_mix    lib/psy.c:4148-4502
 *   flag_lossless .. noise_normalize  lib/psy.c:4584-4854
 *   _vp_couple_quantize_normalize   lib/psy.c:4858-5142
 *   lb_loudnoise_fix                lib/psy.c:5152-5180
 * Float/double promotions follow the C source expression by expression (constants without
 * an f suffix are double there and here).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

#define NEGINF -9999.f
#define existe(x, y) (x < -y || x >= y)
#define refer_phase(a, b) ((a > 0. && b < 0.) || (b > 0. && a < 0.))

/* ---- M2 pre-detection ------------------------------------------------------------- */
float orc_postnoise_detection(const float *pcm, int nn, int mode, int lw_mode)
{
    int i;
    int sn = nn >> 2;
    int mn = sn + sn;
    int en = sn + (nn >> 1);
    float ret = -1.0;
    double upt = 0, unt = 0;

    if (mode != 2) return ret;
    if (lw_mode != 0) return ret;
    if (nn < 2048) return ret;

    for (i = sn; i < mn; i++) upt += fabs(*(pcm + i));
    for (i = mn; i < en; i++) unt += fabs(*(pcm + i));
    if (unt / sn > 0.01) return ret;

    upt *= upt;
    unt *= unt;
    unt *= 15;

    if (upt > unt) {
        ret = upt - unt;
        if (ret < 0.1) ret = -1.0;
    }
    return ret;
}

/* ---- tone masking ------------------------------------------------------------------ */
static void seed_curve(float *seed, const float (*curves)[ORC_EHMER_MAX + 2], const float amp, const int oc,
                       const int n, const int linesper, const float dBoffset)
{
    int i, post1;
    int seedptr;
    const float *posts, *curve;

    int choice = (int)((amp + dBoffset - 30.) * .1f);
    choice = ORC_MAX(choice, 0);
    choice = ORC_MIN(choice, ORC_P_LEVELS - 1);
    posts = curves[choice];
    curve = posts + 2;
    post1 = (int)posts[1];
    seedptr = oc + (posts[0] - ORC_EHMER_OFFSET) * linesper - (linesper >> 1);

    for (i = posts[0]; i < post1; i++) {
        if (seedptr > 0) {
            float lin = amp + curve[i];
            if (seed[seedptr] < lin) seed[seedptr] = lin;
        }
        seedptr += linesper;
        if (seedptr >= n) break;
    }
}

static void seed_loop(const orc_psy *p, const float *f, const float *flr, float *seed, const float specmax)
{
    long n = p->n, i;
    float dBoffset = p->max_curve_dB - specmax;

    for (i = 0; i < n; i++) {
        float max = f[i];
        long oc = p->octave[i];
        while (i + 1 < n && p->octave[i + 1] == oc) {
            i++;
            if (f[i] > max) max = f[i];
        }

        if (max + 6.f > flr[i]) {
            oc = oc >> p->shiftoc;
            if (oc >= ORC_P_BANDS) oc = ORC_P_BANDS - 1;
            if (oc < 0) oc = 0;
            seed_curve(seed, p->tonecurves[oc], max, p->octave[i] - p->firstoc, p->total_octave_lines,
                       p->eighth_octave_lines, dBoffset);
        }
    }
}

static void seed_chase(float *seeds, const int linesper, const long n)
{
    long *posstack = (long *)malloc(n * sizeof(*posstack));
    float *ampstack = (float *)malloc(n * sizeof(*ampstack));
    long stack = 0;
    long pos = 0;
    long i;

    for (i = 0; i < n; i++) {
        if (stack < 2) {
            posstack[stack] = i;
            ampstack[stack++] = seeds[i];
        } else {
            while (1) {
                if (seeds[i] < ampstack[stack - 1]) {
                    posstack[stack] = i;
                    ampstack[stack++] = seeds[i];
                    break;
                } else {
                    if (i < posstack[stack - 1] + linesper) {
                        if (stack > 1 && ampstack[stack - 1] <= ampstack[stack - 2] &&
                            i < posstack[stack - 2] + linesper) {
                            stack--;
                            continue;
                        }
                    }
                    posstack[stack] = i;
                    ampstack[stack++] = seeds[i];
                    break;
                }
            }
        }
    }

    for (i = 0; i < stack; i++) {
        long endpos;
        if (i < stack - 1 && ampstack[i + 1] > ampstack[i]) {
            endpos = posstack[i + 1];
        } else {
            endpos = posstack[i] + linesper + 1;
        }
        if (endpos > n) endpos = n;
        for (; pos < endpos; pos++) seeds[pos] = ampstack[i];
    }
    free(posstack);
    free(ampstack);
}

static void max_seeds(const orc_psy *p, float *seed, float *flr)
{
    long n = p->total_octave_lines;
    int linesper = p->eighth_octave_lines;
    long linpos = 0;
    long pos;

    seed_chase(seed, linesper, n);

    pos = p->octave[0] - p->firstoc - (linesper >> 1);

    while (linpos + 1 < p->n) {
        float minV = seed[pos];
        long end = ((p->octave[linpos] + p->octave[linpos + 1]) >> 1) - p->firstoc;
        if (minV > p->tone_abs_limit) minV = p->tone_abs_limit;
        while (pos + 1 <= end) {
            pos++;
            if ((seed[pos] > NEGINF && seed[pos] < minV) || minV == NEGINF) minV = seed[pos];
        }

        end = pos + p->firstoc;
        for (; linpos < p->n && p->octave[linpos] <= end; linpos++)
            if (flr[linpos] < minV) flr[linpos] = minV;
    }

    {
        float minV = seed[p->total_octave_lines - 1];
        for (; linpos < p->n; linpos++)
            if (flr[linpos] < minV) flr[linpos] = minV;
    }
}

void orc_tonemask(const orc_psy *p, const float *logfft, float *logmask, float global_specmax,
                  float local_specmax)
{
    int i, n = p->n;
    float *seed = (float *)malloc(sizeof(*seed) * p->total_octave_lines);
    float att = local_specmax + p->ath_adjatt;
    for (i = 0; i < p->total_octave_lines; i++) seed[i] = NEGINF;

    if (att < p->ath_maxatt) att = p->ath_maxatt;

    for (i = 0; i < n; i++) logmask[i] = p->ath[i] + att;

    seed_loop(p, logfft, logmask, seed, global_specmax);
    max_seeds(p, seed, logmask);
    free(seed);
}

/* ---- noise masking ----------------------------------------------------------------- */
static void bark_noise_hybridmp(int n, const long *b, const float *f, float *noise, const float offset,
                                const int fixed)
{
    float *N = (float *)malloc(n * sizeof(*N));
    float *X = (float *)malloc(n * sizeof(*N));
    float *XX = (float *)malloc(n * sizeof(*N));
    float *Y = (float *)malloc(n * sizeof(*N));
    float *XY = (float *)malloc(n * sizeof(*N));

    float tN, tX, tXX, tY, tXY;
    int i;

    int lo, hi;
    float R = 0.f;
    float A = 0.f;
    float B = 0.f;
    float D = 1.f;
    float w, x, y;

    tN = tX = tXX = tY = tXY = 0.f;

    y = f[0] + offset;
    if (y < 1.f) y = 1.f;

    w = y * y * .5;

    tN += w;
    tX += w;
    tY += w * y;

    N[0] = tN;
    X[0] = tX;
    XX[0] = tXX;
    Y[0] = tY;
    XY[0] = tXY;

    for (i = 1, x = 1.f; i < n; i++, x += 1.f) {
        y = f[i] + offset;
        if (y < 1.f) y = 1.f;

        w = y * y;

        tN += w;
        tX += w * x;
        tXX += w * x * x;
        tY += w * y;
        tXY += w * x * y;

        N[i] = tN;
        X[i] = tX;
        XX[i] = tXX;
        Y[i] = tY;
        XY[i] = tXY;
    }

    for (i = 0, x = 0.f; i < n; i++, x += 1.f) {
        lo = b[i] >> 16;
        hi = b[i] & 0xffff;
        if (lo >= 0 || -lo >= n) break;
        if (hi >= n) break;

        tN = N[hi] + N[-lo];
        tX = X[hi] - X[-lo];
        tXX = XX[hi] + XX[-lo];
        tY = Y[hi] + Y[-lo];
        tXY = XY[hi] - XY[-lo];

        A = tY * tXX - tX * tXY;
        B = tN * tXY - tX * tY;
        D = tN * tXX - tX * tX;
        R = (A + x * B) / D;
        if (R < 0.f) R = 0.f;

        noise[i] = R - offset;
    }

    for (; i < n; i++, x += 1.f) {
        lo = b[i] >> 16;
        hi = b[i] & 0xffff;
        if (lo < 0 || lo >= n) break;
        if (hi >= n) break;

        tN = N[hi] - N[lo];
        tX = X[hi] - X[lo];
        tXX = XX[hi] - XX[lo];
        tY = Y[hi] - Y[lo];
        tXY = XY[hi] - XY[lo];

        A = tY * tXX - tX * tXY;
        B = tN * tXY - tX * tY;
        D = tN * tXX - tX * tX;
        R = (A + x * B) / D;
        if (R < 0.f) R = 0.f;

        noise[i] = R - offset;
    }

    for (; i < n; i++, x += 1.f) {
        R = (A + x * B) / D;
        if (R < 0.f) R = 0.f;

        noise[i] = R - offset;
    }

    if (fixed <= 0) goto done;

    for (i = 0, x = 0.f; i < n; i++, x += 1.f) {
        hi = i + fixed / 2;
        lo = hi - fixed;
        if (hi >= n) break;
        if (lo >= 0) break;

        tN = N[hi] + N[-lo];
        tX = X[hi] - X[-lo];
        tXX = XX[hi] + XX[-lo];
        tY = Y[hi] + Y[-lo];
        tXY = XY[hi] - XY[-lo];

        A = tY * tXX - tX * tXY;
        B = tN * tXY - tX * tY;
        D = tN * tXX - tX * tX;
        R = (A + x * B) / D;

        if (R - offset < noise[i]) noise[i] = R - offset;
    }
    for (; i < n; i++, x += 1.f) {
        hi = i + fixed / 2;
        lo = hi - fixed;
        if (hi >= n) break;
        if (lo < 0) break;

        tN = N[hi] - N[lo];
        tX = X[hi] - X[lo];
        tXX = XX[hi] - XX[lo];
        tY = Y[hi] - Y[lo];
        tXY = XY[hi] - XY[lo];

        A = tY * tXX - tX * tXY;
        B = tN * tXY - tX * tY;
        D = tN * tXX - tX * tX;
        R = (A + x * B) / D;

        if (R - offset < noise[i]) noise[i] = R - offset;
    }
    for (; i < n; i++, x += 1.f) {
        R = (A + x * B) / D;
        if (R - offset < noise[i]) noise[i] = R - offset;
    }
done:
    free(N); free(X); free(XX); free(Y); free(XY);
}

/* aoTuV M7 */
static void ntfix(const orc_psy *p, const float *spectral, float *noise, int block_mode)
{
    int i, j, k;
    int n = p->n;
    int nx = p->tonefix_end;
    float temp[256], inmod[256];
    float limit = fabs(p->noiseoffset[1][0]);

    if (!nx) return;

    memset(temp, 0, 256 * sizeof(*temp));
    memset(inmod, 0, 256 * sizeof(*inmod));

    if (block_mode <= 1) {
        const int freq_upc = 3;
        const int freq_unc = 4;
        int nxplus = nx + freq_unc;

        float tolerance = 9.f;
        float strength = .6f;
        if (n == 256) tolerance = 15.f;
        if (nxplus > n) {
            nx = n;
            nxplus = n - freq_unc;
        }

        for (i = 0; i < nxplus; i++) {
            if (spectral[i] < -70) inmod[i] = -70 + (spectral[i] + 70) * .1;
            else inmod[i] = spectral[i];
        }
        for (i = freq_unc; i < nx; i++) {
            if ((spectral[i] > spectral[i - 1]) && (spectral[i] > spectral[i + 1])) {
                int ps = i - 1;
                int pe = i + 1;
                int upper = i - freq_upc;
                int under = i + freq_unc;
                for (j = ps; j > upper; j--) {
                    if (spectral[j + 1] < spectral[j]) break;
                    ps = j;
                }
                for (j = pe; j < under; j++) {
                    if (spectral[j - 1] < spectral[j]) break;
                    pe = j;
                }
                {
                    float ss = inmod[i] - inmod[ps];
                    ss = ORC_MAX(ss, inmod[i] - inmod[pe]);
                    if (ss > tolerance) {
                        if (spectral[i] > noise[i]) {
                            ss -= tolerance;
                            ss *= strength;
                        }
                        for (j = ps; j <= pe; j++) {
                            temp[j] = ORC_MAX(ss, temp[j]);
                            if (temp[j] < 0) temp[j] = 0;
                        }
                    }
                }
                i = pe;
            }
        }
        for (i = freq_unc - 1; i < nx; i++) {
            float test = ORC_MIN(p->ntfix_noiseoffset[i], p->noiseoffset[1][i] + limit);
            if (temp[i] > test) temp[i] = test;
            noise[i] -= temp[i];
        }

    } else if (block_mode == 2) {
        for (i = 0, k = 0; i < nx; i += 8, k++) {
            double na = 0;
            for (j = 0; j < 8; j++) na += noise[i + j];
            na /= 8;
            temp[k] = na;
        }
        nx /= 8;
        for (i = 3; i < nx; i++) {
            if ((temp[i] > temp[i - 1]) && (temp[i] > temp[i + 1])) {
                int a = 0;
                int b = 0;
                float thres = 0;

                if (temp[i - 1] > temp[i - 2]) {
                    thres = temp[i - 2];
                    a = i - 3;
                } else {
                    thres = temp[i - 1];
                    a = i - 2;
                }
                b = i + 3;
                thres = temp[i] - thres;
                if (thres > 2.) {
                    int eightimes = i * 8;
                    float test = ORC_MIN(p->ntfix_noiseoffset[eightimes], p->noiseoffset[1][eightimes] + limit);
                    thres = ORC_MIN(thres - 2, test);
                    a *= 8;
                    b *= 8;
                    for (j = a; j <= b; j++) noise[j] -= thres;
                }
            }
        }
    }
}

void orc_noisemask(const orc_setup *s, const orc_psy *p, float noise_compand_level, const float *logmdct,
                   const float *lastmdct, float *epeak, float *npeak, float *logmask, float poste,
                   int block_mode)
{
    int i, j, k, n = p->n;
    int partition = (p->normal_p ? p->normal_partition : 16);
    const int *stn_compand = s->c.stn_compand;
    float *work = (float *)malloc(n * sizeof(*work));

    bark_noise_hybridmp(n, p->bark, logmdct, logmask, 140., -1);

    for (i = 0; i < n; i++) work[i] = logmdct[i] - logmask[i];

    bark_noise_hybridmp(n, p->bark, work, logmask, 0., p->noisewindowfixed);

    for (i = 0; i < n; i++) work[i] = logmdct[i] - work[i];

    ntfix(p, logmdct, work, block_mode);

    /* noise compand & aoTuV M5 extension & pre-store tone peak */
    i = 0;
    if (noise_compand_level > 0) {
        int thter = p->n33p;
        for (; i < thter; i++) {
            int dB = logmask[i] + .5;
            if (dB >= ORC_NOISE_COMPAND_LEVELS) dB = ORC_NOISE_COMPAND_LEVELS - 1;
            if (dB < 0) dB = 0;
            epeak[i] = work[i] + stn_compand[dB];
            logmask[i] = work[i] + p->noisecompand[dB] -
                         ((p->noisecompand[dB] - p->noisecompand_high[dB]) * noise_compand_level);
        }
    }
    for (; i < n; i++) {
        int dB = logmask[i] + .5;
        if (dB >= ORC_NOISE_COMPAND_LEVELS) dB = ORC_NOISE_COMPAND_LEVELS - 1;
        if (dB < 0) dB = 0;
        epeak[i] = work[i] + stn_compand[dB];
        logmask[i] = work[i] + p->noisecompand[dB];
    }

    for (i = 0, k = 0; i < n; i += partition, k++) npeak[k] = 0.f;

    /* reduction of post-echo (postprocessing of aoTuV M2) */
    if (poste > 0) {
        for (i = 0, k = 0; i < p->min_nn_lp; i += partition, k++) {
            float temp = ORC_MIN(ORC_MIN(poste, 30.f), p->noiseoffset[1][i] + 30.f);
            if (temp <= 0) continue;
            npeak[k] = -1.f;
            for (j = 0; j < partition; j++) logmask[i + j] -= temp;
        }
    }

    /* M8 */
    for (k = 0, i = 0; i < p->min_nn_lp; i += partition, k++) {
        const float nt = 4;
        float o = p->noiseoffset[1][i + partition - 1] + 6;
        float me = 0;
        float avge = 0;

        if (o <= 0) continue;
        if (npeak[k] < -0.5) continue;

        for (j = 0; j < partition; j++) {
            float temp = logmdct[i + j] - logmask[i + j];
            if (me < temp) me = temp;
            avge += logmdct[i + j];
        }
        if (avge < (-95 * partition)) continue;

        if (me < nt) {
            npeak[k] = (ORC_MIN(o, nt - me)) / nt;
        }
    }

    /* M9 */
    {
        i = 0;
        if (block_mode > 1) {
            for (; i < p->tonecomp_endp; i++) {
                float temp = logmdct[i] - epeak[i];
                epeak[i] = 0.f;
                if (temp >= 12.f) {
                    float mi = logmdct[i] - lastmdct[i];
                    if (mi >= 1) epeak[i] = mi;
                }
            }
        }
        memset(epeak + i, 0, sizeof(epeak[0]) * (n - i));
    }
    free(work);
}

/* ---- offset and mix ------------------------------------------------------------------ */
typedef struct {
    int sw;
    int mdctbuf_flag;
    float noise_rate;
    float noise_rate_low;
    float noise_center;
    float tone_rate;
} local_mod3_psy;

typedef struct {
    int start;
    int end;
    int lp_pos;
    int end_block;
    float thres;
} local_mod4_psy;

static void set_m3p(const orc_setup *s, local_mod3_psy *mp, const int lW_no, const int impadnum, const int n,
                    const int hs_rate, const float toneatt, const float *logmdct, const float *lastmdct,
                    float *tempmdct, const int block_mode, const int lW_block_mode, const int bit_managed,
                    const int offset_select)
{
    int i, j, count;
    float freqbuf, cell;
    const int *freq_bfn128 = s->c.freq_bfn128;
    const int *freq_bfn256 = s->c.freq_bfn256;

    if (!hs_rate) {
        mp->sw = 0;
        mp->mdctbuf_flag = 0;
        return;
    }

    if (!bit_managed || offset_select == 2) {
        mp->mdctbuf_flag = 1;
    } else {
        mp->mdctbuf_flag = 0;
        if (offset_select == 0) {
            mp->sw = 0;
            return;
        }
    }

    if (block_mode) {
        mp->sw = 0;
        return;
    }

    switch (n) {
    case 128:
        if (toneatt < 3) count = 2;
        else count = 3;

        if (!lW_block_mode) {
            if (lW_no < 8) {
                mp->noise_rate = 0.7 - (float)(lW_no - 1) / 17;
                mp->noise_center = (float)(lW_no * count);
                mp->tone_rate = 8 - lW_no;
            } else {
                mp->noise_rate = 0.3;
                mp->noise_center = 25;
                mp->tone_rate = 0;
                if ((lW_no * count) < 24) mp->noise_center = lW_no * count;
            }
            if (mp->mdctbuf_flag == 1) {
                for (i = 0; i < n; i++) tempmdct[i] -= 5;
            }
        } else {
            mp->noise_rate = 0.7;
            mp->noise_center = 0;
            mp->tone_rate = 8.;
            if (mp->mdctbuf_flag == 1) {
                for (i = 0; i < n; i++) tempmdct[i] = lastmdct[i] - 5;
            }
        }
        mp->noise_rate_low = 0;
        mp->sw = 1;
        if (impadnum) mp->noise_rate *= (impadnum * 0.125);
        for (i = 0; i < n; i++) {
            cell = 75 / (float)freq_bfn128[i];
            for (j = 1; j < freq_bfn128[i]; j++) {
                freqbuf = logmdct[i] - (cell * j);
                if ((tempmdct[i + j] < freqbuf) && (mp->mdctbuf_flag == 1))
                    tempmdct[i + j] += (5. / (float)freq_bfn128[i + j]);
            }
        }
        break;

    case 256:
        if (!lW_block_mode) {
            count = 6;
            if (lW_no < 4) {
                mp->noise_rate = 0.4 - (float)(lW_no - 1) / 11;
                mp->noise_center = (float)(lW_no * count + 12);
                mp->tone_rate = 8 - lW_no * 2;
            } else {
                mp->noise_rate = 0.2;
                mp->noise_center = 30;
                mp->tone_rate = 0;
            }
            if (mp->mdctbuf_flag == 1) {
                for (i = 0; i < n; i++) tempmdct[i] -= 10;
            }
        } else {
            mp->noise_rate = 0.6;
            mp->noise_center = 12;
            mp->tone_rate = 8.;
            if (mp->mdctbuf_flag == 1) {
                for (i = 0; i < n; i++) tempmdct[i] = lastmdct[i] - 10;
            }
        }
        mp->noise_rate_low = 0;
        mp->sw = 1;
        if (impadnum) mp->noise_rate *= (impadnum * 0.0625);
        for (i = 0; i < n; i++) {
            cell = 75 / (float)freq_bfn256[i];
            for (j = 1; j < freq_bfn256[i]; j++) {
                freqbuf = logmdct[i] - (cell * j);
                if ((tempmdct[i + j] < freqbuf) && (mp->mdctbuf_flag == 1))
                    tempmdct[i + j] += (10. / (float)freq_bfn256[i + j]);
            }
        }
        break;

    default:
        mp->sw = 0;
        break;
    }

    if (bit_managed && (offset_select == 0) && mp->sw) mp->noise_rate *= 0.2;
}

void orc_offset_and_mix(const orc_setup *s, const orc_psy *p, const float *noise, const float *tone,
                        int offset_select, int bit_managed, float *logmask, float *mdct, float *logmdct,
                        float *lastmdct, float *tempmdct, float low_compand, float *npeak, int end_block,
                        int block_mode, int nW_modenumber, int lW_block_mode, int lW_no, int impadnum)
{
    int i, j, k, n = p->n;
    int hsrate = ((p->rate < 26000) ? 0 : 1);
    int partition = (p->normal_p ? p->normal_partition : 16);
    float m1_de, m1_coeffi;
    float toneatt = p->tone_masteratt[offset_select];

    local_mod3_psy mp3;
    local_mod4_psy mp4;

    memset(&mp3, 0, sizeof(mp3));

    mp4.start = p->normal_start;
    mp4.end = p->tonecomp_endp;
    mp4.thres = p->tonecomp_thres;
    mp4.lp_pos = 9999;
    mp4.end_block = end_block;

    if (low_compand < 0 || toneatt < 25.) low_compand = 0;
    else low_compand *= (toneatt - 25.);

    set_m3p(s, &mp3, lW_no, impadnum, n, hsrate, toneatt, logmdct, lastmdct, tempmdct, block_mode, lW_block_mode,
            bit_managed, offset_select);

    mp4.end_block += p->normal_partition;
    if (mp4.end_block > n) mp4.end_block = n;
    if (!hsrate) {
        mp4.end = mp4.end_block;
    } else {
        if (p->normal_thresh > 1.) {
            mp4.start = 9999;
        } else {
            if (mp4.end > mp4.end_block) mp4.lp_pos = mp4.end;
            else mp4.lp_pos = mp4.end_block;
        }
    }

    for (i = 0; i < n; i++) {
        float val = noise[i] + p->noiseoffset[offset_select][i];
        float tval = tone[i] + toneatt;
        if (i <= mp4.start) tval -= low_compand;
        if (val > p->noisemaxsupp) val = p->noisemaxsupp;

        /* M3 */
        if (mp3.sw) {
            if (val > tval) {
                if ((val > lastmdct[i]) && (logmdct[i] > (tempmdct[i] + mp3.noise_center))) {
                    int toneac = 0;
                    float valmask = 0;
                    float rate_mod;
                    float mainth;

                    if (mp3.mdctbuf_flag == 1) tempmdct[i] = logmdct[i];
                    if (logmdct[i] > lastmdct[i]) {
                        rate_mod = mp3.noise_rate;
                    } else {
                        rate_mod = mp3.noise_rate_low;
                    }
                    if (!impadnum && (i < p->tonecomp_endp) && ((val - lastmdct[i]) > 20.f)) {
                        float dBsub = (logmdct[i] - lastmdct[i]);
                        if (dBsub > 25.f) {
                            toneac = 1;
                            if (tval > -100.f && ((logmdct[i] - tval) < 48.f)) {
                                float tr_cur = mp3.tone_rate;
                                if (dBsub < 35.f) tr_cur *= ((35.f - dBsub) * .1f);
                                tval -= tr_cur;
                                if (tval < -100.f) tval = -100.f;
                                if ((logmdct[i] - tval) > 48.f) tval = logmdct[i] - 48.f;
                            }
                        }
                    }
                    if (i > p->m3n[0]) {
                        mainth = 30.f;
                    } else if (i > p->m3n[1]) {
                        mainth = 20.f;
                    } else if (i > p->m3n[2]) {
                        mainth = 10.f;
                        rate_mod *= .5f;
                    } else {
                        mainth = 10.f;
                        rate_mod *= .3f;
                    }
                    if ((val - tval) > mainth) valmask = ((val - tval - mainth) * .1f + mainth) * rate_mod;
                    else valmask = (val - tval) * rate_mod;

                    if ((val - valmask) > lastmdct[i]) val -= valmask;
                    else val = lastmdct[i];

                    if (toneac) {
                        float temp = val - ORC_MAX(lastmdct[i], -140);
                        if (temp > 20.f) val -= (temp - 20.f) * .2f;
                    }

                    if (toneac == 1) npeak[i / partition] = -1.f;
                    else if (npeak[i / partition] > 0) npeak[i / partition] = 0;
                }
            }
        }

        /* M4 */
        if (val > tval) {
            logmask[i] = val;
        } else if ((i > mp4.start) && (i < mp4.end)) {
            if (logmdct[i] < tval) {
                if (logmdct[i] < val) {
                    tval -= (tval - val) * mp4.thres;
                } else {
                    tval = logmdct[i];
                }
            }
            logmask[i] = tval;
        } else
            logmask[i] = tval;

        /* M1 */
        if (offset_select == 1) {
            m1_coeffi = -17.2;
            val = val - logmdct[i];

            if (val > m1_coeffi) {
                m1_de = 1.0 - ((val - m1_coeffi) * 0.005 * p->m_val);
                if (m1_de < 0) m1_de = 0.0001;
            } else
                m1_de = 1.0 - ((val - m1_coeffi) * 0.0003 * p->m_val);

            mdct[i] *= m1_de;
        }
    }

    /* M3 SET lastmdct */
    if (mp3.mdctbuf_flag == 1) {
        const int mag = 8;
        switch (block_mode) {
        case 0:
        case 1:
            if (nW_modenumber) {
                for (i = 0, k = 0; i < n; i++, k += mag) {
                    for (j = 0; j < mag; j++) {
                        lastmdct[k + j] = logmdct[i];
                    }
                }
            } else {
                memcpy(lastmdct, logmdct, n * sizeof(*lastmdct));
            }
            break;

        case 2:
            if (!nW_modenumber) {
                int nsh = n >> 3;
                for (i = 0; i < nsh; i++) {
                    int ni = i * mag;
                    lastmdct[i] = logmdct[ni];
                    for (j = 1; j < mag; j++) {
                        if (lastmdct[i] > logmdct[ni + j]) {
                            lastmdct[i] = logmdct[ni + j];
                        }
                    }
                }
            } else {
                memcpy(lastmdct, logmdct, n * sizeof(*lastmdct));
            }
            break;

        case 3:
            memcpy(lastmdct, logmdct, n * sizeof(*lastmdct));
            break;

        default:
            break;
        }
    }
}

/* ---- couple / quantise / normalise --------------------------------------------------- */
static void flag_lossless(int limit, float prepoint, float postpoint, float prepoint_r, float postpoint_r,
                          float *res, float *mdct, float *enpeak, float *floor, int *flag, int i, int jn)
{
    int j, ps = 0;
    int pointlimit = limit - i;
    float point1, point2, bakp1, r;
    float ps1 = 0.f, ps2 = 0.f;

    if (pointlimit > 0) {
        point1 = prepoint;
        point2 = prepoint_r;
        if ((pointlimit - jn) <= 0) {
            ps1 = (postpoint - prepoint) / jn;
            ps2 = (postpoint_r - prepoint_r) / jn;
            ps = 1;
        }
    } else {
        point1 = postpoint;
        point2 = postpoint_r;
    }
    for (j = 0; j < jn; j++) {
        if (ps == 1) {
            point1 += ps1;
            point2 += ps2;
        }
        bakp1 = point1;

        res[j] = mdct[j] / floor[j];
        r = fabs(res[j]);
        point1 -= enpeak[j];
        if (point1 < prepoint) point1 = prepoint;
        if (r < point1) {
            if (r < point2) {
                flag[j] = 0;
            } else
                flag[j] = -1;
        } else {
            flag[j] = 1;
        }
        point1 = bakp1;
    }
}

static void lossless_coupling(int *Mag, int *Ang)
{
    int A = *Mag;
    int B = *Ang;

    if (abs(A) > abs(B)) {
        *Ang = (A > 0 ? A - B : B - A);
    } else {
        *Ang = (B > 0 ? A - B : B - A);
        *Mag = B;
    }
    if (*Ang >= abs(*Mag) * 2) {
        *Ang = -*Ang;
        *Mag = -*Mag;
    }
}

static void lossless_couplingf(float *Mag, float *Ang)
{
    float A = *Mag;
    float B = *Ang;

    if (fabs(A) > fabs(B)) {
        *Ang = (A > 0 ? A - B : B - A);
    } else {
        *Ang = (B > 0 ? A - B : B - A);
        *Mag = B;
    }
    if (*Ang >= fabs(*Mag) * 2) {
        *Ang = -*Ang;
        *Mag = -*Mag;
    }
}

static float min_indemnity_dipole_hypot(const float a, const float b, const float threv)
{
    const float thnor = 0.94;
    float a2 = fabs(a * thnor);
    float b2 = fabs(b * thnor);

    if (a > 0.) {
        if (b > 0.) return (a2 + b2);
        if (a > -b) return (a2 - b2 * threv);
        return -(b2 - a2 * threv);
    }
    if (b < 0.) return -(a2 + b2);
    if (-a > b) return -(a2 - b2 * threv);
    return (b2 - a2 * threv);
}

static void ssort(const int range, int bthresh, float **sort)
{
    int i, j;
    int large;
    float *temp;

    if (range < bthresh) bthresh = range;
    for (i = 0; i < bthresh; i++) {
        large = i;
        for (j = i + 1; j < range; j++) {
            if (*sort[large] < *sort[j]) large = j;
        }
        temp = sort[i];
        sort[i] = sort[large];
        sort[large] = temp;
    }
}

static float noise_normalize(const orc_psy *p, const int limit, float *r, float *q, const float *f, float *res,
                             const int *flags, float acc, const float nepeak, const int i, const int n,
                             int *out)
{
    float *sort[64];
    int j, k, count = 0;
    int start = (p->normal_p ? p->normal_start - i : n);
    if ((start > n) || (nepeak < -0.5)) start = n;

    acc = 0.f;

    j = 0;
    if (!flags) {
        for (; j < start; j++) {
            out[j] = rint(res[j]);
        }
    } else {
        for (; j < start; j++) {
            if (flags[j] != 1) {
                float ve = sqrt(q[j] / f[j]);
                if (r[j] < 0) {
                    out[j] = -rint(ve);
                    res[j] = -ve;
                } else {
                    out[j] = rint(ve);
                    res[j] = ve;
                }
            }
        }
    }

    if (flags) {
        for (; j < n; j++) {
            float ve;
            if (flags[j] != 1) {
                ve = q[j] / f[j];
            } else {
                continue;
            }
            if (ve < .25f && j >= limit - i) {
                acc += ve;
                sort[count++] = q + j;
                if (r[j] < 0) {
                    res[j] = -sqrt(ve);
                } else {
                    res[j] = sqrt(ve);
                }
            } else {
                ve = sqrt(ve);
                if (r[j] < 0) {
                    out[j] = -rint(ve);
                    res[j] = -ve;
                } else {
                    out[j] = rint(ve);
                    res[j] = ve;
                }
                q[j] = out[j] * out[j] * f[j];
            }
        }
    } else {
        for (; j < n; j++) {
            float ve = res[j] * res[j];
            if (ve < .25f) {
                acc += ve;
                sort[count++] = q + j;
            } else {
                out[j] = rint(res[j]);
                q[j] = out[j] * out[j] * f[j];
            }
        }
    }

    acc += acc * nepeak * nepeak;

    if (count) {
        int iacc = ((int)acc) + 1;
        if (iacc > n) iacc = n;
        ssort(count, iacc, sort);

        for (k = 0; k < count; k++) {
            int e = sort[k] - q;
            if (acc >= p->normal_thresh) {
                out[e] = orc_unitnorm(r[e]);
                acc -= 1.f;
                q[e] = f[e];
            } else {
                out[e] = 0;
                q[e] = 0.f;
            }
        }
    }

    return acc;
}

void orc_couple_quantize_normalize(const orc_setup *s, int blobno, const orc_psy *p, const orc_map *vi,
                                   float **mdct, float **enpeak, float **nepeak, int **iwork, int *nonzero,
                                   int sliding_lowpass, int ch, int lowpassr)
{
    const orc_psyg *g = &s->psy_g;
    const float *FROMDB = s->c.fromdB;
    int i, pi;
    int n = p->n;
    int partition = (p->normal_p ? p->normal_partition : 16);
    int limit = g->coupling_pointlimit[p->blockflag][blobno];
    float prepoint = s->c.stereo_threshholds[g->coupling_prepointamp[blobno]];
    float postpoint = s->c.stereo_threshholds[g->coupling_postpointamp[blobno]];
    float prepoint_x = s->c.stereo_threshholds_X[g->coupling_prepointamp[blobno]];
    float postpoint_x = s->c.stereo_threshholds_X[g->coupling_postpointamp[blobno]];
    float prae;

    float *raw[ORC_MAXCH], *quant[ORC_MAXCH], *floor[ORC_MAXCH], *res[ORC_MAXCH];
    int *flag[ORC_MAXCH];
    int nz[ORC_MAXCH];
    float acc[ORC_MAXCH + 16];
    float side_resdef[16];

    raw[0] = (float *)malloc(ch * partition * sizeof(float));
    quant[0] = (float *)malloc(ch * partition * sizeof(float));
    floor[0] = (float *)malloc(ch * partition * sizeof(float));
    res[0] = (float *)malloc(ch * partition * sizeof(float));
    flag[0] = (int *)malloc(ch * partition * sizeof(int));

    for (i = 1; i < ch; i++) {
        raw[i] = &raw[0][partition * i];
        quant[i] = &quant[0][partition * i];
        floor[i] = &floor[0][partition * i];
        res[i] = &res[0][partition * i];
        flag[i] = &flag[0][partition * i];
    }

    for (i = 0; i < ch + vi->coupling_steps; i++) acc[i] = 0.f;

    if (prepoint_x < prepoint) prepoint_x = prepoint;
    if (postpoint_x < prepoint) postpoint_x = prepoint;

    for (i = 0; i < vi->coupling_steps; i++) side_resdef[i] = -1.f;

    if (vi->coupling_steps == 1) prae = 0.34;
    else prae = 0.825;

    for (i = 0, pi = 0; i < lowpassr; i += partition, pi++) {
        int k, j, jn = partition > n - i ? n - i : partition;
        int step, track = 0;

        memcpy(nz, nonzero, sizeof(*nz) * ch);

        memset(flag[0], 0, ch * partition * sizeof(**flag));
        for (k = 0; k < ch; k++) {
            int *iout = &iwork[k][i];
            if (nz[k]) {
                for (j = 0; j < jn; j++) floor[k][j] = FROMDB[iout[j]];

                flag_lossless(limit, prepoint, postpoint, prepoint_x, postpoint_x, res[k], &mdct[k][i],
                              &enpeak[k][i], floor[k], flag[k], i, jn);

                for (j = 0; j < jn; j++) {
                    quant[k][j] = raw[k][j] = mdct[k][i + j] * mdct[k][i + j];
                    if (mdct[k][i + j] < 0.f) raw[k][j] *= -1.f;
                    floor[k][j] *= floor[k][j];
                }

                acc[track] = noise_normalize(p, limit, raw[k], quant[k], floor[k], res[k], NULL, acc[track],
                                             nepeak[k][pi], i, jn, iout);
            } else {
                for (j = 0; j < jn; j++) {
                    floor[k][j] = 1e-10f;
                    raw[k][j] = 0.f;
                    quant[k][j] = 0.f;
                    res[k][j] = 0.f;
                    flag[k][j] = 0;
                    iout[j] = 0;
                }
                acc[track] = 0.f;
            }
            track++;
        }

        /* coupling */
        for (step = 0; step < vi->coupling_steps; step++) {
            int Mi = vi->coupling_mag[step];
            int Ai = vi->coupling_ang[step];
            int *iM = &iwork[Mi][i];
            int *iA = &iwork[Ai][i];
            float *reM = raw[Mi];
            float *reA = raw[Ai];
            float *qeM = quant[Mi];
            float *qeA = quant[Ai];
            float *floorM = floor[Mi];
            float *floorA = floor[Ai];
            float *resM = res[Mi];
            float *resA = res[Ai];
            int *fM = flag[Mi];
            int *fA = flag[Ai];
            int pointflag = 0;

            if (nz[Mi] || nz[Ai]) {
                nz[Mi] = nz[Ai] = 1;

                /* M6 */
                if (p->tonefix_end > i) {
                    int rp = 0;
                    int pp = 0;
                    int ap;
                    float residue_def = 0;

                    for (j = 0; j < jn; j++) {
                        if (existe(resM[j], 0.5) || existe(resA[j], 0.5)) {
                            if (refer_phase(reM[j], reA[j])) {
                                rp++;
                            } else
                                pp++;
                            residue_def += fabs(fabs(resM[j]) - fabs(resA[j]));
                        }
                    }
                    ap = rp + pp;

                    if (ap != 0) {
                        float temp_def = residue_def = residue_def / ap;
                        if (side_resdef[step] > 0) residue_def = temp_def * 0.5 + side_resdef[step] * 0.5;
                        side_resdef[step] = temp_def;
                        if (residue_def > 1.f) {
                            for (j = 0; j < jn; j++) {
                                if (fM[j] == -1 || fA[j] == -1) fM[j] = 1;
                            }
                        }
                        if ((float)rp / ap >= prae) {
                            for (j = 0; j < jn; j++) {
                                if ((fM[j] == -1 || fA[j] == -1) && refer_phase(reM[j], reA[j])) fM[j] = 1;
                            }
                        }
                    } else
                        side_resdef[step] = -1.f;
                }

                for (j = 0; j < jn; j++) {
                    if (j < sliding_lowpass - i) {
                        if (fM[j] == 1 || fA[j] == 1) {
                            /* lossless coupling */
                            reM[j] = fabs(reM[j]) + fabs(reA[j]);
                            qeM[j] = qeM[j] + qeA[j];
                            fM[j] = fA[j] = 1;

                            lossless_couplingf(&resM[j], &resA[j]);
                            lossless_coupling(&iM[j], &iA[j]);
                        } else {
                            /* lossy (point) coupling */
                            float hpL;
                            float hpH;
                            if (vi->coupling_steps == 1 || step == 3) {
                                hpL = .18f;
                                hpH = .12f;
                            } else {
                                hpL = .18f;
                                hpH = .04f;
                            }
                            if (j < limit - i) {
                                reM[j] = min_indemnity_dipole_hypot(reM[j], reA[j], hpL);
                            } else {
                                reM[j] = min_indemnity_dipole_hypot(reM[j], reA[j], hpH);
                            }

                            qeM[j] = fabs(reM[j]);
                            reA[j] = qeA[j] = 0.f;
                            fA[j] = 1;
                            iA[j] = 0;
                            resA[j] = 0;

                            if ((nepeak[Mi][pi] < -0.5) || (nepeak[Ai][pi] < -0.5)) {
                                nepeak[Mi][pi] = -1;
                            } else {
                                nepeak[Mi][pi] = ORC_MIN(nepeak[Mi][pi], nepeak[Ai][pi]);
                            }

                            pointflag |= 1;
                        }
                    }
                    floorM[j] = floorA[j] = floorM[j] + floorA[j];
                }
                if (pointflag)
                    acc[track] = noise_normalize(p, limit, raw[Mi], quant[Mi], floor[Mi], res[Mi], flag[Mi],
                                                 acc[track], nepeak[Mi][pi], i, jn, iM);
                track++;
            }
        }
    }

    if (lowpassr < n) {
        int j, block = n - lowpassr;
        for (j = 0; j < ch; j++) memset(iwork[j] + lowpassr, 0, sizeof(**iwork) * block);
    }

    for (i = 0; i < vi->coupling_steps; i++) {
        if (nonzero[vi->coupling_mag[i]] || nonzero[vi->coupling_ang[i]]) {
            nonzero[vi->coupling_mag[i]] = 1;
            nonzero[vi->coupling_ang[i]] = 1;
        }
    }
    free(raw[0]); free(quant[0]); free(floor[0]); free(res[0]); free(flag[0]);
}

/* aoTuV M5 */
float orc_lb_loudnoise_fix(const orc_psy *p, float noise_compand_level, const float *logmdct, int block_mode,
                           int lW_block_mode)
{
    int i, n = p->n, nq1 = p->n25p, nq3 = p->n75p;
    double hi_th = 0;

    if (p->m_val < 0.5) return (-1);
    if (p->normal_thresh > .45) return (-1);

    if (!((block_mode == 2 && lW_block_mode == 3) || (block_mode == 3 && lW_block_mode == 2)))
        return (noise_compand_level);

    for (i = nq1; i < nq3; i++) {
        if (logmdct[i] > -130) hi_th += logmdct[i];
        else hi_th += -130;
    }
    hi_th /= n;

    if (hi_th > -40.) noise_compand_level = -1;
    else if (hi_th < -50.) noise_compand_level = 1.;
    else noise_compand_level = 1. - ((hi_th + 50) / 10);

    return (noise_compand_level);
}
