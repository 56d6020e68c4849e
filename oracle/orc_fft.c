/* ORACLE — test infrastructure only.
 *
 * Real forward FFT as the reference's scalar path runs it (FFTPACK, radix 4 and 2 only —
 * block sizes are powers of two):
 *   drfti1   lib/smallft.c:5576-5644   factorisation {4,2,3,5} with the 2 moved first,
 *                                       twiddles = (float)cos/sin((float)arg) via double libm
 *   dradf2   lib/smallft.c:5652-5705
 *   dradf4   lib/smallft.c:5707-5807
 *   drftf1   lib/smallft.c:6111-6170   factors applied last to first, ping-pong c <-> ch
 * Output layout: FFTPACK half-complex [r0, r1,i1, ..., r(n/2-1),i(n/2-1), r(n/2)].
 *
 * Written with the FFTPACK array views CC(i,k,j) / CH(i,j,k) instead of the running
 * offsets t0..t6 of the source; every arithmetic expression keeps its shape.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static void fft_factor_and_twiddle(int n, float *wa, int *ifac)
{
    static const int ntryh[4] = {4, 2, 3, 5};
    const float tpi = 6.28318530717958648f;
    int ntry = 0, j = -1, nl = n, nf = 0;
    int i, k1, l1, l2, ld, ii, ip, is, ido, ipm;
    float arg, argh, argld, fi;

    for (;;) {
        j++;
        if (j < 4) ntry = ntryh[j];
        else ntry += 2;
        while (nl % ntry == 0) {
            nf++;
            ifac[nf + 1] = ntry;
            nl /= ntry;
            if (ntry == 2 && nf != 1) {
                for (i = 1; i < nf; i++) {
                    int ib = nf - i + 1;
                    ifac[ib + 1] = ifac[ib];
                }
                ifac[2] = 2;
            }
            if (nl == 1) goto factored;
        }
    }
factored:
    ifac[0] = n;
    ifac[1] = nf;
    argh = tpi / n;
    is = 0;
    l1 = 1;
    if (nf - 1 == 0) return;
    for (k1 = 0; k1 < nf - 1; k1++) {
        ip = ifac[k1 + 2];
        ld = 0;
        l2 = l1 * ip;
        ido = n / l2;
        ipm = ip - 1;
        for (j = 0; j < ipm; j++) {
            ld += l1;
            i = is;
            argld = (float)ld * argh;
            fi = 0.f;
            for (ii = 2; ii < ido; ii += 2) {
                fi += 1.f;
                arg = fi * argld;
                wa[i++] = (float)cos(arg);
                wa[i++] = (float)sin(arg);
            }
            is += ido;
        }
        l1 = l2;
    }
}

#define CC(i, k, j) cc[(i) + ido * ((k) + l1 * (j))]

static void radf2(int ido, int l1, const float *cc, float *ch, const float *wa1)
{
#define CH(i, j, k) ch[(i) + ido * ((j) + 2 * (k))]
    int i, k;
    for (k = 0; k < l1; k++) {
        CH(0, 0, k) = CC(0, k, 0) + CC(0, k, 1);
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 1);
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (k = 0; k < l1; k++) {
            for (i = 2; i < ido; i += 2) {
                int ic = ido - i;
                float tr2 = wa1[i - 2] * CC(i - 1, k, 1) + wa1[i - 1] * CC(i, k, 1);
                float ti2 = wa1[i - 2] * CC(i, k, 1) - wa1[i - 1] * CC(i - 1, k, 1);
                CH(i, 0, k) = CC(i, k, 0) + ti2;
                CH(ic, 1, k) = ti2 - CC(i, k, 0);
                CH(i - 1, 0, k) = CC(i - 1, k, 0) + tr2;
                CH(ic - 1, 1, k) = CC(i - 1, k, 0) - tr2;
            }
        }
        if (ido % 2 == 1) return;
    }
    for (k = 0; k < l1; k++) {
        CH(0, 1, k) = -CC(ido - 1, k, 1);
        CH(ido - 1, 0, k) = CC(ido - 1, k, 0);
    }
#undef CH
}

static void radf4(int ido, int l1, const float *cc, float *ch, const float *wa1, const float *wa2,
                  const float *wa3)
{
#define CH(i, j, k) ch[(i) + ido * ((j) + 4 * (k))]
    const float hsqt2 = .70710678118654752f;
    int i, k;
    for (k = 0; k < l1; k++) {
        float tr1 = CC(0, k, 1) + CC(0, k, 3);
        float tr2 = CC(0, k, 0) + CC(0, k, 2);
        CH(0, 0, k) = tr1 + tr2;
        CH(ido - 1, 3, k) = tr2 - tr1;
        CH(ido - 1, 1, k) = CC(0, k, 0) - CC(0, k, 2);
        CH(0, 2, k) = CC(0, k, 3) - CC(0, k, 1);
    }
    if (ido < 2) return;
    if (ido > 2) {
        for (k = 0; k < l1; k++) {
            for (i = 2; i < ido; i += 2) {
                int ic = ido - i;
                float cr2 = wa1[i - 2] * CC(i - 1, k, 1) + wa1[i - 1] * CC(i, k, 1);
                float ci2 = wa1[i - 2] * CC(i, k, 1) - wa1[i - 1] * CC(i - 1, k, 1);
                float cr3 = wa2[i - 2] * CC(i - 1, k, 2) + wa2[i - 1] * CC(i, k, 2);
                float ci3 = wa2[i - 2] * CC(i, k, 2) - wa2[i - 1] * CC(i - 1, k, 2);
                float cr4 = wa3[i - 2] * CC(i - 1, k, 3) + wa3[i - 1] * CC(i, k, 3);
                float ci4 = wa3[i - 2] * CC(i, k, 3) - wa3[i - 1] * CC(i - 1, k, 3);
                float tr1 = cr2 + cr4, tr4 = cr4 - cr2;
                float ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                float ti2 = CC(i, k, 0) + ci3, ti3 = CC(i, k, 0) - ci3;
                float tr2 = CC(i - 1, k, 0) + cr3, tr3 = CC(i - 1, k, 0) - cr3;
                CH(i - 1, 0, k) = tr1 + tr2;
                CH(i, 0, k) = ti1 + ti2;
                CH(ic - 1, 1, k) = tr3 - ti4;
                CH(ic, 1, k) = tr4 - ti3;
                CH(i - 1, 2, k) = ti4 + tr3;
                CH(i, 2, k) = tr4 + ti3;
                CH(ic - 1, 3, k) = tr2 - tr1;
                CH(ic, 3, k) = ti1 - ti2;
            }
        }
        if (ido & 1) return;
    }
    for (k = 0; k < l1; k++) {
        float ti1 = -hsqt2 * (CC(ido - 1, k, 1) + CC(ido - 1, k, 3));
        float tr1 = hsqt2 * (CC(ido - 1, k, 1) - CC(ido - 1, k, 3));
        CH(ido - 1, 0, k) = tr1 + CC(ido - 1, k, 0);
        CH(ido - 1, 2, k) = CC(ido - 1, k, 0) - tr1;
        CH(0, 1, k) = ti1 - CC(ido - 1, k, 2);
        CH(0, 3, k) = ti1 + CC(ido - 1, k, 2);
    }
#undef CH
}
#undef CC

void orc_drft_init(orc_drft *l, int n)
{
    l->n = n;
    l->trigcache = (float *)calloc((size_t)3 * n, sizeof(float));
    l->splitcache = (int *)calloc(32, sizeof(int));
    if (n == 1) return;
    fft_factor_and_twiddle(n, l->trigcache + n, l->splitcache); /* fdrffti: drfti1(n, wsave+n, ifac) */
}

void orc_drft_clear(orc_drft *l)
{
    free(l->trigcache);
    free(l->splitcache);
    memset(l, 0, sizeof(*l));
}

void orc_drft_forward(const orc_drft *l, float *c)
{
    int n = l->n;
    float *ch = l->trigcache;          /* scratch */
    const float *wa = l->trigcache + n; /* drftf1(n, data, trigcache, trigcache+n, splitcache) */
    const int *ifac = l->splitcache;
    int nf, na, l2, iw, k1, i;
    if (n == 1) return;
    nf = ifac[1];
    na = 1;
    l2 = n;
    iw = n;
    for (k1 = 0; k1 < nf; k1++) {
        int kh = nf - k1;
        int ip = ifac[kh + 1];
        int l1 = l2 / ip;
        int ido = n / l2;
        iw -= (ip - 1) * ido;
        na = 1 - na;
        if (ip == 4) {
            int ix2 = iw + ido, ix3 = ix2 + ido;
            if (na != 0) radf4(ido, l1, ch, c, wa + iw - 1, wa + ix2 - 1, wa + ix3 - 1);
            else radf4(ido, l1, c, ch, wa + iw - 1, wa + ix2 - 1, wa + ix3 - 1);
        } else { /* ip == 2 */
            if (na != 0) radf2(ido, l1, ch, c, wa + iw - 1);
            else radf2(ido, l1, c, ch, wa + iw - 1);
        }
        l2 = l1;
    }
    if (na == 1) return;
    for (i = 0; i < n; i++) c[i] = ch[i];
}
