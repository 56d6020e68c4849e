/* ORACLE — test infrastructure only.
 *
 * Transient detector that drives long/short block switching, restating the scalar path of
 *   _ve_amp              lib/envelope.c:101-562
 *   _ve_envelope_search  lib/envelope.c:569-681
 *   _ve_envelope_mark    lib/envelope.c:683-707
 *   _ve_envelope_shift   lib/envelope.c:709-728
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

static int ve_amp(const orc_setup *s, orc_stream *v, const float *data, orc_ve_filter *filters)
{
    const orc_psyg *gi = &s->psy_g;
    const orc_ve_band *bands = s->ve_band;
    long n = 128;
    int ret = 0;
    long i, j;
    float decay;

    float minV = s->ve_minenergy;
    float vec[128];

    int stretch = ORC_MAX(ORC_VE_MINSTRETCH, v->ve_stretch / 2);
    float penalty = gi->stretch_penalty - (v->ve_stretch / 2 - ORC_VE_MINSTRETCH);
    if (penalty < 0.f) penalty = 0.f;
    if (penalty > gi->stretch_penalty) penalty = gi->stretch_penalty;

    for (i = 0; i < n; i++) vec[i] = data[i] * s->ve_mdct_win[i];
    orc_mdct_forward(&s->ve_mdct, vec, vec);

    /* near-DC spreading function */
    {
        float temp = vec[0] * vec[0] + .7 * vec[1] * vec[1] + .2 * vec[2] * vec[2];
        int ptr = filters->nearptr;

        if (ptr == 0) {
            decay = filters->nearDC_acc = filters->nearDC_partialacc + temp;
            filters->nearDC_partialacc = temp;
        } else {
            decay = filters->nearDC_acc += temp;
            filters->nearDC_partialacc += temp;
        }
        filters->nearDC_acc -= filters->nearDC[ptr];
        filters->nearDC[ptr] = temp;

        decay *= (1. / (ORC_VE_NEARDC + 1));
        filters->nearptr++;
        if (filters->nearptr >= ORC_VE_NEARDC) filters->nearptr = 0;
        decay = orc_todB(&decay) * .5 - 15.f;
    }

    /* spreading, limiting, spectrum smoothing */
    for (i = 0; i < n / 2; i += 2) {
        float val = vec[i] * vec[i] + vec[i + 1] * vec[i + 1];
        val = orc_todB(&val) * .5f;
        if (val < decay) val = decay;
        if (val < minV) val = minV;
        vec[i >> 1] = val;
        decay -= 8.;
    }

    /* preecho / postecho triggering by band */
    for (j = 0; j < ORC_VE_BANDS; j++) {
        float acc = 0.;
        float valmax, valmin;

        for (i = 0; i < bands[j].end; i++) acc += vec[i + bands[j].begin] * bands[j].window[i];
        acc *= bands[j].total;

        {
            int p, this = filters[j].ampptr;
            float postmax, postmin, premax = -99999.f, premin = 99999.f;

            p = this;
            p--;
            if (p < 0) p += ORC_VE_AMP;
            postmax = ORC_MAX(acc, filters[j].ampbuf[p]);
            postmin = ORC_MIN(acc, filters[j].ampbuf[p]);

            for (i = 0; i < stretch; i++) {
                p--;
                if (p < 0) p += ORC_VE_AMP;
                premax = ORC_MAX(premax, filters[j].ampbuf[p]);
                premin = ORC_MIN(premin, filters[j].ampbuf[p]);
            }

            valmin = postmin - premin;
            valmax = postmax - premax;

            filters[j].ampbuf[this] = acc;
            filters[j].ampptr++;
            if (filters[j].ampptr >= ORC_VE_AMP) filters[j].ampptr = 0;
        }

        if (valmax > gi->preecho_thresh[j] + penalty) {
            ret |= 1;
            ret |= 4;
        }
        if (valmin < gi->postecho_thresh[j] - penalty) ret |= 2;
    }
    return (ret);
}

long orc_ve_envelope_search(orc_stream *v)
{
    const orc_setup *s = v->s;
    long i, j;
    const int searchstep = 64;
    int first = v->ve_current / searchstep;
    int last = v->pcm_current / searchstep - ORC_VE_WIN;
    if (first < 0) first = 0;

    if (last + ORC_VE_WIN + ORC_VE_POST > v->ve_storage) {
        long old = v->ve_storage;
        v->ve_storage = last + ORC_VE_WIN + ORC_VE_POST;
        v->ve_mark = (int *)realloc(v->ve_mark, v->ve_storage * sizeof(*v->ve_mark));
        memset(v->ve_mark + old, 0, (v->ve_storage - old) * sizeof(*v->ve_mark)); /* realloc'd tail is
            written before it is read in the reference; zeroing keeps valgrind quiet */
    }

    for (j = first; j < last; j++) {
        int ret = 0;

        v->ve_stretch++;
        if (v->ve_stretch > ORC_VE_MAXSTRETCH * 2) v->ve_stretch = ORC_VE_MAXSTRETCH * 2;

        for (i = 0; i < s->channels; i++) {
            float *pcm = v->pcm[i] + searchstep * (j);
            ret |= ve_amp(s, v, pcm, v->ve_filter + i * ORC_VE_BANDS);
        }

        v->ve_mark[j + ORC_VE_POST] = 0;
        if (ret & 1) {
            v->ve_mark[j] = 1;
            v->ve_mark[j + 1] = 1;
        }

        if (ret & 2) {
            v->ve_mark[j] = 1;
            if (j > 0) v->ve_mark[j - 1] = 1;
        }

        if (ret & 4) v->ve_stretch = -1;
    }

    v->ve_current = last * searchstep;

    {
        long centerW = v->centerW;
        long testW = centerW + s->blocksizes[v->W] / 4 + s->blocksizes[1] / 2 + s->blocksizes[0] / 4;
        j = v->ve_cursor;

        while (j < v->ve_current - (searchstep)) {
            if (j >= testW) return (1);

            v->ve_cursor = j;

            if (v->ve_mark[j / searchstep]) {
                if (j > centerW) {
                    v->ve_curmark = j;
                    if (j >= testW) return (1);
                    return (0);
                }
            }
            j += searchstep;
        }
    }

    return (-1);
}

int orc_ve_envelope_mark(orc_stream *v)
{
    const orc_setup *s = v->s;
    const int searchstep = 64;
    long centerW = v->centerW;
    long beginW = centerW - s->blocksizes[v->W] / 4;
    long endW = centerW + s->blocksizes[v->W] / 4;
    if (v->W) {
        beginW -= s->blocksizes[v->lW] / 4;
        endW += s->blocksizes[v->nW] / 4;
    } else {
        beginW -= s->blocksizes[0] / 4;
        endW += s->blocksizes[0] / 4;
    }

    if (v->ve_curmark >= beginW && v->ve_curmark < endW) return (1);
    {
        long first = beginW / searchstep;
        long last = endW / searchstep;
        long i;
        for (i = first; i < last; i++)
            if (v->ve_mark[i]) return (1);
    }
    return (0);
}

void orc_ve_envelope_shift(orc_stream *v, long shift)
{
    const int searchstep = 64;
    int smallsize = v->ve_current / searchstep + ORC_VE_POST;
    int smallshift = shift / searchstep;

    memmove(v->ve_mark, v->ve_mark + smallshift, (smallsize - smallshift) * sizeof(*v->ve_mark));

    v->ve_current -= shift;
    if (v->ve_curmark >= 0) v->ve_curmark -= shift;
    v->ve_cursor -= shift;
}
