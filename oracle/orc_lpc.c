/* ORACLE — test infrastructure only.
 * LPC analysis / prediction used to extrapolate stream start and end:
 *   vorbis_lpc_from_data  lib/lpc.c:60-130 (Levinson-Durbin in double)
 *   vorbis_lpc_predict    lib/lpc.c:132-159 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

float orc_lpc_from_data(float *data, float *lpci, int n, int m)
{
    double *aut = (double *)malloc(sizeof(*aut) * (m + 1));
    double *lpc = (double *)malloc(sizeof(*lpc) * (m));
    double error;
    double epsilon;
    int i, j;

    j = m + 1;
    while (j--) {
        double d = 0;
        for (i = j; i < n; i++) d += (double)data[i] * data[i - j];
        aut[j] = d;
    }

    error = aut[0] * (1. + 1e-10);
    epsilon = 1e-9 * aut[0] + 1e-10;

    for (i = 0; i < m; i++) {
        double r = -aut[i + 1];

        if (error < epsilon) {
            memset(lpc + i, 0, (m - i) * sizeof(*lpc));
            goto done;
        }

        for (j = 0; j < i; j++) r -= lpc[j] * aut[i - j];
        r /= error;

        lpc[i] = r;
        for (j = 0; j < i / 2; j++) {
            double tmp = lpc[j];
            lpc[j] += r * lpc[i - 1 - j];
            lpc[i - 1 - j] += r * tmp;
        }
        if (i & 1) lpc[j] += lpc[j] * r;

        error *= 1. - r * r;
    }

done:
    {
        double g = .99;
        double damp = g;
        for (j = 0; j < m; j++) {
            lpc[j] *= damp;
            damp *= g;
        }
    }

    for (j = 0; j < m; j++) lpci[j] = (float)lpc[j];
    free(aut);
    free(lpc);
    return error;
}

void orc_lpc_predict(float *coeff, float *prime, int m, float *data, long n)
{
    long i, j, o, p;
    float y;
    float *work = (float *)malloc(sizeof(*work) * (m + n));

    if (!prime)
        for (i = 0; i < m; i++) work[i] = 0.f;
    else
        for (i = 0; i < m; i++) work[i] = prime[i];

    for (i = 0; i < n; i++) {
        y = 0;
        o = i;
        p = m;
        for (j = 0; j < m; j++) y -= work[o++] * coeff[--p];
        data[i] = work[o] = y;
    }
    free(work);
}
