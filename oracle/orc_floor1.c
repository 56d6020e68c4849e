/* ORACLE — test infrastructure only.
 *
 * Floor 1 fit and encode, restating the reference's scalar path:
 *   render_point / render_line0     lib/floor1.c:260-274, 397-424
 *   vorbis_dBquant                  lib/floor1.c:285-299 (scalar: (int)(x*7.3142857f+1023.5f))
 *   accumulate_fit / fit_line       lib/floor1.c:427-535
 *   inspect_error / post_Y          lib/floor1.c:537-595
 *   floor1_fit                      lib/floor1.c:597-750
 *   floor1_encode                   lib/floor1.c:774-974
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

typedef struct {
    int x0, x1;
    int xa, ya, x2a, y2a, xya, an;
    int xb, yb, x2b, y2b, xyb, bn;
} lsfit_acc;

static int render_point(int x0, int x1, int y0, int y1, int x)
{
    y0 &= 0x7fff;
    y1 &= 0x7fff;
    {
        int dy = y1 - y0;
        int adx = x1 - x0;
        int ady = abs(dy);
        int err = ady * (x - x0);
        int off = err / adx;
        if (dy < 0) return (y0 - off);
        return (y0 + off);
    }
}

static int dBquant(const float *x)
{
    int i = *x * 7.3142857f + 1023.5f;
    if (i > 1023) return (1023);
    if (i < 0) return (0);
    return i;
}

static void render_line0(int n, int x0, int x1, int y0, int y1, int *d)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = abs(dy);
    int base = dy / adx;
    int sy = (dy < 0 ? base - 1 : base + 1);
    int x = x0;
    int y = y0;
    int err = 0;

    ady -= abs(base * adx);

    if (n > x1) n = x1;

    if (x < n) d[x] = y;

    while (++x < n) {
        err = err + ady;
        if (err >= adx) {
            err -= adx;
            y += sy;
        } else {
            y += base;
        }
        d[x] = y;
    }
}

static int accumulate_fit(const float *flr, const float *mdct, int x0, int x1, lsfit_acc *a, int n,
                          const orc_floor *info)
{
    long i;
    int xa = 0, ya = 0, x2a = 0, y2a = 0, xya = 0, na = 0, xb = 0, yb = 0, x2b = 0, y2b = 0, xyb = 0, nb = 0;

    memset(a, 0, sizeof(*a));
    a->x0 = x0;
    a->x1 = x1;
    if (x1 >= n) x1 = n - 1;

    for (i = x0; i <= x1; i++) {
        int quantized = dBquant(flr + i);
        if (quantized) {
            if (mdct[i] + info->twofitatten >= flr[i]) {
                xa += i;
                ya += quantized;
                x2a += i * i;
                y2a += quantized * quantized;
                xya += i * quantized;
                na++;
            } else {
                xb += i;
                yb += quantized;
                x2b += i * i;
                y2b += quantized * quantized;
                xyb += i * quantized;
                nb++;
            }
        }
    }

    a->xa = xa; a->ya = ya; a->x2a = x2a; a->y2a = y2a; a->xya = xya; a->an = na;
    a->xb = xb; a->yb = yb; a->x2b = x2b; a->y2b = y2b; a->xyb = xyb; a->bn = nb;
    return (na);
}

static int fit_line(lsfit_acc *a, int fits, int *y0, int *y1, const orc_floor *info)
{
    double xb = 0, yb = 0, x2b = 0, y2b = 0, xyb = 0, bn = 0;
    int i;
    int x0 = a[0].x0;
    int x1 = a[fits - 1].x1;

    for (i = 0; i < fits; i++) {
        double weight = (a[i].bn + a[i].an) * info->twofitweight / (a[i].an + 1) + 1.;

        xb += a[i].xb + a[i].xa * weight;
        yb += a[i].yb + a[i].ya * weight;
        x2b += a[i].x2b + a[i].x2a * weight;
        y2b += a[i].y2b + a[i].y2a * weight;
        xyb += a[i].xyb + a[i].xya * weight;
        bn += a[i].bn + a[i].an * weight;
    }

    if (*y0 >= 0) {
        xb += x0;
        yb += *y0;
        x2b += x0 * x0;
        y2b += *y0 * *y0;
        xyb += *y0 * x0;
        bn++;
    }

    if (*y1 >= 0) {
        xb += x1;
        yb += *y1;
        x2b += x1 * x1;
        y2b += *y1 * *y1;
        xyb += *y1 * x1;
        bn++;
    }

    {
        double denom = (bn * x2b - xb * xb);

        if (denom > 0.) {
            double aa = (yb * x2b - xyb * xb) / denom;
            double bb = (bn * xyb - xb * yb) / denom;
            *y0 = rint(aa + bb * x0);
            *y1 = rint(aa + bb * x1);

            if (*y0 > 1023) *y0 = 1023;
            if (*y1 > 1023) *y1 = 1023;
            if (*y0 < 0) *y0 = 0;
            if (*y1 < 0) *y1 = 0;

            return 0;
        } else {
            *y0 = 0;
            *y1 = 0;
            return 1;
        }
    }
}

static int inspect_error(int x0, int x1, int y0, int y1, const float *mask, const float *mdct,
                         const orc_floor *info)
{
    int dy = y1 - y0;
    int adx = x1 - x0;
    int ady = abs(dy);
    int base = dy / adx;
    int sy = (dy < 0 ? base - 1 : base + 1);
    int x = x0;
    int y = y0;
    int err = 0;
    int val = dBquant(mask + x);
    int mse = 0;
    int n = 0;

    ady -= abs(base * adx);

    mse = (y - val);
    mse *= mse;
    n++;
    if (mdct[x] + info->twofitatten >= mask[x]) {
        if (y + info->maxover < val) return (1);
        if (y - info->maxunder > val) return (1);
    }

    while (++x < x1) {
        err = err + ady;
        if (err >= adx) {
            err -= adx;
            y += sy;
        } else {
            y += base;
        }

        val = dBquant(mask + x);
        mse += ((y - val) * (y - val));
        n++;
        if (mdct[x] + info->twofitatten >= mask[x]) {
            if (val) {
                if (y + info->maxover < val) return (1);
                if (y - info->maxunder > val) return (1);
            }
        }
    }

    if (info->maxover * info->maxover / n > info->maxerr) return (0);
    if (info->maxunder * info->maxunder / n > info->maxerr) return (0);
    if (mse / n > info->maxerr) return (1);
    return (0);
}

static int post_Y(int *A, int *B, int pos)
{
    if (A[pos] < 0) return B[pos];
    if (B[pos] < 0) return A[pos];
    return (A[pos] + B[pos]) >> 1;
}

/* lib/floor1.c:752-771: 16.16 fixed-point blend of two fits; 0 (NULL) unless both exist */
int orc_floor1_interpolate_fit(const orc_floor *look, const int *A, const int *B, int del, int *output)
{
    long i;
    long posts = look->posts;
    if (!A || !B) return 0;
    for (i = 0; i < posts; i++) {
        output[i] = ((65536 - del) * (A[i] & 0x7fff) + del * (B[i] & 0x7fff) + 32768) >> 16;
        if (A[i] & 0x8000 && B[i] & 0x8000) output[i] |= 0x8000;
    }
    return 1;
}

int orc_floor1_fit(const orc_floor *look, const float *logmdct, const float *logmask, int *output)
{
    long i, j;
    const orc_floor *info = look;
    long n = look->n; /* = postlist[1] (lib/floor1.c:194, :602); info->n only feeds offset_and_mix */
    long posts = look->posts;
    long nonzero = 0;
    lsfit_acc fits[ORC_VIF_POSIT + 1];
    int fit_valueA[ORC_VIF_POSIT + 2];
    int fit_valueB[ORC_VIF_POSIT + 2];
    int loneighbor[ORC_VIF_POSIT + 2];
    int hineighbor[ORC_VIF_POSIT + 2];
    int memo[ORC_VIF_POSIT + 2];

    for (i = 0; i < posts; i++) fit_valueA[i] = -200;
    for (i = 0; i < posts; i++) fit_valueB[i] = -200;
    for (i = 0; i < posts; i++) loneighbor[i] = 0;
    for (i = 0; i < posts; i++) hineighbor[i] = 1;
    for (i = 0; i < posts; i++) memo[i] = -1;

    if (posts == 0) {
        nonzero += accumulate_fit(logmask, logmdct, 0, n, fits, n, info);
    } else {
        for (i = 0; i < posts - 1; i++)
            nonzero += accumulate_fit(logmask, logmdct, look->sorted_index[i], look->sorted_index[i + 1],
                                      fits + i, n, info);
    }

    if (!nonzero) return 0;

    {
        int y0 = -200;
        int y1 = -200;
        fit_line(fits, posts - 1, &y0, &y1, info);

        fit_valueA[0] = y0;
        fit_valueB[0] = y0;
        fit_valueB[1] = y1;
        fit_valueA[1] = y1;

        for (i = 2; i < posts; i++) {
            int sortpos = look->reverse_index[i];
            int ln = loneighbor[sortpos];
            int hn = hineighbor[sortpos];

            if (memo[ln] != hn) {
                int lsortpos = look->reverse_index[ln];
                int hsortpos = look->reverse_index[hn];
                memo[ln] = hn;

                {
                    int lx = info->postlist[ln];
                    int hx = info->postlist[hn];
                    int ly = post_Y(fit_valueA, fit_valueB, ln);
                    int hy = post_Y(fit_valueA, fit_valueB, hn);

                    if (ly == -1 || hy == -1) {
                        fprintf(stderr, "oracle: floor1_fit hit the reference's exit(1) condition\n");
                        exit(1);
                    }

                    if (inspect_error(lx, hx, ly, hy, logmask, logmdct, info)) {
                        int ly0 = -200;
                        int ly1 = -200;
                        int hy0 = -200;
                        int hy1 = -200;
                        int ret0 = fit_line(fits + lsortpos, sortpos - lsortpos, &ly0, &ly1, info);
                        int ret1 = fit_line(fits + sortpos, hsortpos - sortpos, &hy0, &hy1, info);

                        if (ret0) {
                            ly0 = ly;
                            ly1 = hy0;
                        }
                        if (ret1) {
                            hy0 = ly1;
                            hy1 = hy;
                        }

                        if (ret0 && ret1) {
                            fit_valueA[i] = -200;
                            fit_valueB[i] = -200;
                        } else {
                            fit_valueB[ln] = ly0;
                            if (ln == 0) fit_valueA[ln] = ly0;
                            fit_valueA[i] = ly1;
                            fit_valueB[i] = hy0;
                            fit_valueA[hn] = hy1;
                            if (hn == 1) fit_valueB[hn] = hy1;

                            if (ly1 >= 0 || hy0 >= 0) {
                                for (j = sortpos - 1; j >= 0; j--)
                                    if (hineighbor[j] == hn) hineighbor[j] = i;
                                    else break;
                                for (j = sortpos + 1; j < posts; j++)
                                    if (loneighbor[j] == ln) loneighbor[j] = i;
                                    else break;
                            }
                        }
                    } else {
                        fit_valueA[i] = -200;
                        fit_valueB[i] = -200;
                    }
                }
            }
        }

        output[0] = post_Y(fit_valueA, fit_valueB, 0);
        output[1] = post_Y(fit_valueA, fit_valueB, 1);

        for (i = 2; i < posts; i++) {
            int ln = look->loneighbor[i - 2];
            int hn = look->hineighbor[i - 2];
            int x0 = info->postlist[ln];
            int x1 = info->postlist[hn];
            int y0 = output[ln];
            int y1 = output[hn];

            int predicted = render_point(x0, x1, y0, y1, info->postlist[i]);
            int vx = post_Y(fit_valueA, fit_valueB, i);

            if (vx >= 0 && predicted != vx) {
                output[i] = vx;
            } else {
                output[i] = predicted | 0x8000;
            }
        }
    }
    return 1;
}

int orc_floor1_encode(const orc_setup *s, orc_bits *opb, const orc_floor *look, int *post, int *ilogmask,
                      int n_half)
{
    long i, j;
    const orc_floor *info = look;
    long posts = look->posts;
    int out[ORC_VIF_POSIT + 2];
    const orc_book *books = s->book;

    if (post) {
        for (i = 0; i < posts; i++) {
            int val = post[i] & 0x7fff;
            switch (info->mult) {
            case 1: val >>= 2; break;
            case 2: val >>= 3; break;
            case 3: val /= 12; break;
            case 4: val >>= 4; break;
            }
            post[i] = val | (post[i] & 0x8000);
        }

        out[0] = post[0];
        out[1] = post[1];

        for (i = 2; i < posts; i++) {
            int ln = look->loneighbor[i - 2];
            int hn = look->hineighbor[i - 2];
            int x0 = info->postlist[ln];
            int x1 = info->postlist[hn];
            int y0 = post[ln];
            int y1 = post[hn];

            int predicted = render_point(x0, x1, y0, y1, info->postlist[i]);

            if ((post[i] & 0x8000) || (predicted == post[i])) {
                post[i] = predicted | 0x8000;
                out[i] = 0;
            } else {
                int headroom = (look->quant_q - predicted < predicted ? look->quant_q - predicted : predicted);
                int val = post[i] - predicted;

                if (val < 0)
                    if (val < -headroom) val = headroom - val - 1;
                    else val = -1 - (val << 1);
                else if (val >= headroom) val = val + headroom;
                else val <<= 1;

                out[i] = val;
                post[ln] &= 0x7fff;
                post[hn] &= 0x7fff;
            }
        }

        orc_bits_write(opb, 1, 1);

        orc_bits_write(opb, out[0], orc_ilog(look->quant_q - 1));
        orc_bits_write(opb, out[1], orc_ilog(look->quant_q - 1));

        for (i = 0, j = 2; i < info->partitions; i++) {
            int class = info->partitionclass[i];
            int cdim = info->class_dim[class];
            int csubbits = info->class_subs[class];
            int csub = 1 << csubbits;
            int bookas[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int cval = 0;
            int cshift = 0;
            int k, l;

            if (csubbits) {
                int maxval[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (k = 0; k < csub; k++) {
                    int booknum = info->class_subbook[class][k];
                    if (booknum < 0) {
                        maxval[k] = 1;
                    } else {
                        maxval[k] = books[info->class_subbook[class][k]].entries;
                    }
                }
                for (k = 0; k < cdim; k++) {
                    for (l = 0; l < csub; l++) {
                        int val = out[j + k];
                        if (val < maxval[l]) {
                            bookas[k] = l;
                            break;
                        }
                    }
                    cval |= bookas[k] << cshift;
                    cshift += csubbits;
                }
                orc_book_encode(books + info->class_book[class], cval, opb);
            }

            for (k = 0; k < cdim; k++) {
                int book = info->class_subbook[class][bookas[k]];
                if (book >= 0) {
                    if (out[j + k] < (books + book)->entries) orc_book_encode(books + book, out[j + k], opb);
                }
            }
            j += cdim;
        }

        {
            int hx = 0;
            int lx = 0;
            int ly = post[0] * info->mult;
            int n = n_half;

            for (j = 1; j < look->posts; j++) {
                int current = look->forward_index[j];
                int hy = post[current] & 0x7fff;
                if (hy == post[current]) {
                    hy *= info->mult;
                    hx = info->postlist[current];

                    render_line0(n, lx, hx, ly, hy, ilogmask);

                    lx = hx;
                    ly = hy;
                }
            }
            for (j = hx; j < n_half; j++) ilogmask[j] = ly;
            return (1);
        }
    } else {
        orc_bits_write(opb, 0, 1);
        memset(ilogmask, 0, n_half * sizeof(*ilogmask));
        return (0);
    }
}
