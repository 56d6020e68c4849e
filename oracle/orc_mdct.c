/* ORACLE — test infrastructure only (never linked into or called by the product).
 *
 * CPU restatement of the reference's scalar forward MDCT:
 *   lookup init      /root/reference lib/mdct.c:54-92   (mdct_init)
 *   forward          lib/mdct.c:1799-1869               (mdct_forward, !__SSE__ branch)
 *   butterflies      lib/mdct.c:1105-1135, 854-894, 1032-1079, 602-658, 495-528, 432-452
 *   bit reverse      lib/mdct.c:1228-1272
 *
 * Every float expression keeps the reference's shape (two rounded products,
 * one rounded sum; no FMA — compile with -ffp-contract=off).  The structure is
 * ours: one generic radix-2 stage routine instead of the unrolled first/generic
 * pair, complex-pair indexing instead of pointer walks.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define K_PI3_8 .38268343236508977175F   /* lib/mdct.h:44-46 */
#define K_PI2_8 .70710678118654752441F
#define K_PI1_8 .92387953251128675613F

void orc_mdct_init(orc_mdct *m, int n)
{
    int i, j;
    int n2 = n >> 1;
    m->n = n;
    /* lib/mdct.c:60: rint(log((float)n)/log(2.f)) — evaluated in double */
    m->log2n = (int)rint(log((float)n) / log(2.f));
    m->trig = (float *)malloc(sizeof(float) * (n + n / 4));
    m->bitrev = (int *)malloc(sizeof(int) * (n / 4));

    /* lib/mdct.c:67-76: double-precision libm, rounded to float on store */
    for (i = 0; i < n / 4; i++) {
        m->trig[i * 2]          = (float)cos((M_PI / n) * (4 * i));
        m->trig[i * 2 + 1]      = (float)-sin((M_PI / n) * (4 * i));
        m->trig[n2 + i * 2]     = (float)cos((M_PI / (2 * n)) * (2 * i + 1));
        m->trig[n2 + i * 2 + 1] = (float)sin((M_PI / (2 * n)) * (2 * i + 1));
    }
    for (i = 0; i < n / 8; i++) {
        m->trig[n + i * 2]     = (float)(cos((M_PI / n) * (4 * i + 2)) * .5);
        m->trig[n + i * 2 + 1] = (float)(-sin((M_PI / n) * (4 * i + 2)) * .5);
    }
    /* lib/mdct.c:80-91 */
    {
        int mask = (1 << (m->log2n - 1)) - 1;
        int msb = 1 << (m->log2n - 2);
        for (i = 0; i < n / 8; i++) {
            int acc = 0;
            for (j = 0; msb >> j; j++)
                if ((msb >> j) & i) acc |= 1 << j;
            m->bitrev[i * 2] = ((~acc) & mask) - 1;
            m->bitrev[i * 2 + 1] = acc;
        }
    }
    m->scale = 4.f / n;   /* lib/mdct.c:92 */
}

void orc_mdct_clear(orc_mdct *m)
{
    free(m->trig);
    free(m->bitrev);
    memset(m, 0, sizeof(*m));
}

/* One radix-2 stage over a block of `points` floats (points/2 complex values).
 * Upper-half element u and lower-half element l (u = l + points/4 complex):
 *   u' = u + l ;  l' = (u - l) rotated by the twiddle at T[step*t], t counted
 * from the top of each half downwards.  lib/mdct.c:854-894 is this with
 * step = 4, lib/mdct.c:1032-1079 with step = trigint. */
static void stage(const float *T, float *x, int points, int step)
{
    int half = points >> 2; /* complex elements per half */
    int t;
    for (t = 0; t < half; t++) {
        float *lo = x + 2 * (half - 1 - t);
        float *up = lo + (points >> 1);
        const float *w = T + (long)step * t;
        float r0 = up[0] - lo[0];
        float r1 = up[1] - lo[1];
        up[0] += lo[0];
        up[1] += lo[1];
        lo[0] = r1 * w[1] + r0 * w[0];
        lo[1] = r1 * w[0] - r0 * w[1];
    }
}

/* lib/mdct.c:432-452 */
static void bfly8(float *x)
{
    float a = x[6] + x[2], b = x[6] - x[2];
    float c = x[4] + x[0], d = x[4] - x[0];
    float e = x[5] - x[1], f = x[7] - x[3];
    float g = x[5] + x[1], h = x[7] + x[3];
    x[6] = a + c;
    x[4] = a - c;
    x[0] = b + e;
    x[2] = b - e;
    x[3] = f + d;
    x[1] = f - d;
    x[7] = h + g;
    x[5] = h - g;
}

/* lib/mdct.c:495-528 */
static void bfly16(float *x)
{
    float r0, r1;
    r0 = x[1] - x[9];  r1 = x[0] - x[8];
    x[8] += x[0];  x[9] += x[1];
    x[0] = (r0 + r1) * K_PI2_8;
    x[1] = (r0 - r1) * K_PI2_8;

    r0 = x[3] - x[11]; r1 = x[10] - x[2];
    x[10] += x[2]; x[11] += x[3];
    x[2] = r0;  x[3] = r1;

    r0 = x[12] - x[4]; r1 = x[13] - x[5];
    x[12] += x[4]; x[13] += x[5];
    x[4] = (r0 - r1) * K_PI2_8;
    x[5] = (r0 + r1) * K_PI2_8;

    r0 = x[14] - x[6]; r1 = x[15] - x[7];
    x[14] += x[6]; x[15] += x[7];
    x[6] = r0;  x[7] = r1;

    bfly8(x);
    bfly8(x + 8);
}

/* lib/mdct.c:602-658 */
static void bfly32(float *x)
{
    float r0, r1;
    r0 = x[30] - x[14]; r1 = x[31] - x[15];
    x[30] += x[14]; x[31] += x[15];
    x[14] = r0;  x[15] = r1;

    r0 = x[28] - x[12]; r1 = x[29] - x[13];
    x[28] += x[12]; x[29] += x[13];
    x[12] = r0 * K_PI1_8 - r1 * K_PI3_8;
    x[13] = r0 * K_PI3_8 + r1 * K_PI1_8;

    r0 = x[26] - x[10]; r1 = x[27] - x[11];
    x[26] += x[10]; x[27] += x[11];
    x[10] = (r0 - r1) * K_PI2_8;
    x[11] = (r0 + r1) * K_PI2_8;

    r0 = x[24] - x[8];  r1 = x[25] - x[9];
    x[24] += x[8];  x[25] += x[9];
    x[8] = r0 * K_PI3_8 - r1 * K_PI1_8;
    x[9] = r1 * K_PI3_8 + r0 * K_PI1_8;

    r0 = x[22] - x[6];  r1 = x[7] - x[23];
    x[22] += x[6];  x[23] += x[7];
    x[6] = r1;  x[7] = r0;

    r0 = x[4] - x[20];  r1 = x[5] - x[21];
    x[20] += x[4];  x[21] += x[5];
    x[4] = r1 * K_PI1_8 + r0 * K_PI3_8;
    x[5] = r1 * K_PI3_8 - r0 * K_PI1_8;

    r0 = x[2] - x[18];  r1 = x[3] - x[19];
    x[18] += x[2];  x[19] += x[3];
    x[2] = (r1 + r0) * K_PI2_8;
    x[3] = (r1 - r0) * K_PI2_8;

    r0 = x[0] - x[16];  r1 = x[1] - x[17];
    x[16] += x[0];  x[17] += x[1];
    x[0] = r1 * K_PI3_8 + r0 * K_PI1_8;
    x[1] = r1 * K_PI1_8 - r0 * K_PI3_8;

    bfly16(x);
    bfly16(x + 16);
}

/* lib/mdct.c:1105-1135 */
void orc_mdct_butterflies(const orc_mdct *m, float *x, int points)
{
    int stages = m->log2n - 5;
    int i, j;
    if (--stages > 0) stage(m->trig, x, points, 4);
    for (i = 1; --stages > 0; i++)
        for (j = 0; j < (1 << i); j++)
            stage(m->trig, x + (points >> i) * j, points >> i, 4 << i);
    for (j = 0; j < points; j += 32) bfly32(x + j);
}

/* lib/mdct.c:1228-1272.  Reads the upper half of w (x = w + n/2), writes w[0..n/2). */
void orc_mdct_bitreverse(const orc_mdct *m, float *w)
{
    int n = m->n;
    const float *x = w + (n >> 1);
    const float *T = m->trig + n;
    const int *bit = m->bitrev;
    int u, subs = n >> 3; /* n/8 gathers, each yields one low and one high output pair */
    int n4c = n >> 2;     /* complex outputs */
    for (u = 0; u < subs; u++) {
        const float *x0 = x + bit[2 * u];
        const float *x1 = x + bit[2 * u + 1];
        float r0 = x0[1] - x1[1];
        float r1 = x0[0] + x1[0];
        float r2 = r1 * T[2 * u] + r0 * T[2 * u + 1];
        float r3 = r1 * T[2 * u + 1] - r0 * T[2 * u];
        float h0 = (x0[1] + x1[1]) * .5f;
        float h1 = (x0[0] - x1[0]) * .5f;
        float *lo = w + 2 * u;
        float *hi = w + 2 * (n4c - 1 - u);
        lo[0] = h0 + r2;
        hi[0] = h0 - r2;
        lo[1] = h1 + r3;
        hi[1] = r3 - h1;
    }
}

void orc_mdct_forward(const orc_mdct *m, const float *in, float *out)
{
    int n = m->n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
    float *w = (float *)malloc(sizeof(float) * n);
    float *w2 = w + n2;
    int p, pairs = n4; /* n2/2 complex values */

    /* fold + pre-twiddle, lib/mdct.c:1819-1851; pair p writes w2[2p], w2[2p+1] */
    for (p = 0; p < pairs; p++) {
        const float *T = m->trig + n2 - 2 * (p + 1);
        float r0, r1;
        if (2 * p < n8) {
            const float *x0 = in + n2 + n4 - 4 * (p + 1);
            const float *x1 = in + n2 + n4 + 1 + 4 * p;
            r0 = x0[2] + x1[0];
            r1 = x0[0] + x1[2];
        } else if (2 * p < n2 - n8) {
            const float *x0 = in + n2 + n4 - 4 * (p + 1);
            const float *x1 = in + 1 + 4 * (p - n8 / 2);
            r0 = x0[2] - x1[0];
            r1 = x0[0] - x1[2];
        } else {
            const float *x0 = in + n - 4 * (p - (n2 - n8) / 2 + 1);
            const float *x1 = in + 1 + 4 * (p - n8 / 2);
            r0 = -x0[2] - x1[0];
            r1 = -x0[0] - x1[2];
        }
        w2[2 * p]     = r1 * T[1] + r0 * T[0];
        w2[2 * p + 1] = r1 * T[0] - r0 * T[1];
    }

    orc_mdct_butterflies(m, w2, n2);
    orc_mdct_bitreverse(m, w);

    /* post-twiddle + scale, lib/mdct.c:1859-1868 */
    {
        const float *T = m->trig + n2;
        int i;
        for (i = 0; i < n4; i++) {
            out[i]          = (w[2 * i] * T[2 * i] + w[2 * i + 1] * T[2 * i + 1]) * m->scale;
            out[n2 - 1 - i] = (w[2 * i] * T[2 * i + 1] - w[2 * i + 1] * T[2 * i]) * m->scale;
        }
    }
    free(w);
}
