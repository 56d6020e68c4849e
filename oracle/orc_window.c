/* ORACLE — test infrastructure only.
 * Window application, restating lib/window.c:2137-2147 + scalar loop :2247-2258. */
#include "oracle.h"

void orc_apply_window(float *d, long n, const float *win_l, long ln, const float *win_r, long rn)
{
    long leftbegin = n / 4 - ln / 4;
    long leftend = leftbegin + ln / 2;
    long rightbegin = n / 2 + n / 4 - rn / 4;
    long rightend = rightbegin + rn / 2;
    long i, p;
    for (i = 0; i < leftbegin; i++) d[i] = 0.f;
    for (p = 0; i < leftend; i++, p++) d[i] *= win_l[p];
    for (i = rightbegin, p = rn / 2 - 1; i < rightend; i++, p--) d[i] *= win_r[p];
    for (; i < n; i++) d[i] = 0.f;
}
