/* ORACLE — test infrastructure only.
 *
 * Per-stream state machine of the encoder, restating
 *   vorbis_analysis_init      lib/block.c:306-331 (+ _vds_shared_init :181-303 storage part)
 *   vorbis_block_init         lib/block.c:84-107
 *   vorbis_analysis_buffer    lib/block.c:411-436
 *   _preextrapolate_helper    lib/block.c:438-477
 *   vorbis_analysis_wrote     lib/block.c:482-553
 *   vorbis_analysis_blockout  lib/block.c:557-812
 *   _vp_ampmax_decay          lib/psy.c:4504-4515
 *   vorbis_analysis           lib/analysis.c:29-63
 *   vorbis_bitrate_addblock / flushpacket (VBR: pass-through)  lib/bitrate.c:88-96, 229-252
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "oracle.h"
#include "orc_internal.h"

orc_stream *orc_stream_new(const orc_setup *s)
{
    orc_stream *v = (orc_stream *)calloc(1, sizeof(*v));
    int i;
    v->s = s;
    v->pcm_storage = (int)s->blocksizes[1];
    for (i = 0; i < s->channels; i++) v->pcm[i] = (float *)calloc(v->pcm_storage, sizeof(float));
    v->lW = 0;
    v->W = 0;
    v->centerW = s->blocksizes[1] / 2;
    v->pcm_current = (int)v->centerW;
    v->g_ampmax = -9999.;
    v->mblock = (float *)calloc(2048 * s->channels, sizeof(float));
    v->tblock = (float *)calloc(256 * s->channels, sizeof(float));
    v->lownoise_compand_level = (float *)calloc(s->channels, sizeof(float));
    v->impadnum = 0;
    /* _ve_envelope_init: storage 128, cursor = blocksizes[1]/2, everything else zero */
    v->ve_filter = (orc_ve_filter *)calloc(ORC_VE_BANDS * s->channels, sizeof(orc_ve_filter));
    v->ve_storage = 128;
    v->ve_mark = (int *)calloc(v->ve_storage, sizeof(int));
    v->ve_cursor = s->blocksizes[1] / 2;
    v->sequence = 3;
    /* vorbis_bitrate_init, lib/bitrate.c:28-56 */
    if (s->managed && s->bi_reservoir_bits > 0) {
        long ratesamples = s->rate;
        int halfsamples = (int)(s->blocksizes[0] >> 1);
        v->bm_short_per_long = s->blocksizes[1] / s->blocksizes[0];
        v->bm_managed = 1;
        v->bm_avg_bitsper = rint(1. * s->bi_avg_rate * halfsamples / ratesamples);
        v->bm_min_bitsper = rint(1. * s->bi_min_rate * halfsamples / ratesamples);
        v->bm_max_bitsper = rint(1. * s->bi_max_rate * halfsamples / ratesamples);
        v->bm_avgfloat = ORC_PACKETBLOBS / 2;
        {
            long desired_fill = s->bi_reservoir_bits * s->bi_reservoir_bias;
            v->bm_minmax_reservoir = desired_fill;
            v->bm_avg_reservoir = desired_fill;
        }
    }
    return v;
}

void orc_stream_free(orc_stream *v)
{
    int i;
    if (!v) return;
    for (i = 0; i < v->s->channels; i++) free(v->pcm[i]);
    free(v->mblock); free(v->tblock); free(v->lownoise_compand_level);
    free(v->ve_filter); free(v->ve_mark);
    free(v);
}

orc_block *orc_block_new(const orc_setup *s)
{
    orc_block *b = (orc_block *)calloc(1, sizeof(*b));
    int i, n = (int)s->blocksizes[1];
    b->ampmax = -9999;
    for (i = 0; i < s->channels; i++) {
        b->pcmbuf[i] = (float *)calloc(n, sizeof(float));
        b->cap_windowed[i] = (float *)calloc(n, sizeof(float));
        b->cap_gmdct_raw[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_gmdct[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_logfft[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_logmdct[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_noise[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_tone[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_logmask[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_epeak[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_npeak[i] = (float *)calloc(n / 2, sizeof(float));
        b->cap_ilogmask[i] = (int *)calloc(n / 2, sizeof(int));
        b->cap_residue[i] = (int *)calloc(n / 2, sizeof(int));
    }
    orc_bits_init(&b->opb);
    for (i = 0; i < ORC_PACKETBLOBS; i++) orc_bits_init(&b->blob[i]);
    b->choice = ORC_PACKETBLOBS / 2;
    return b;
}

void orc_block_free(orc_block *b)
{
    int i;
    if (!b) return;
    for (i = 0; i < ORC_MAXCH; i++) {
        free(b->pcmbuf[i]); free(b->cap_windowed[i]); free(b->cap_gmdct_raw[i]); free(b->cap_gmdct[i]);
        free(b->cap_logfft[i]); free(b->cap_logmdct[i]); free(b->cap_noise[i]); free(b->cap_tone[i]);
        free(b->cap_logmask[i]); free(b->cap_epeak[i]); free(b->cap_npeak[i]); free(b->cap_ilogmask[i]);
        free(b->cap_residue[i]);
    }
    orc_bits_clear(&b->opb);
    for (i = 0; i < ORC_PACKETBLOBS; i++) orc_bits_clear(&b->blob[i]);
    free(b);
}

float **orc_analysis_buffer(orc_stream *v, int vals, float **ret)
{
    int i;
    const orc_setup *s = v->s;
    if (v->pcm_current + vals >= v->pcm_storage) {
        v->pcm_storage = v->pcm_current + vals * 2;
        for (i = 0; i < s->channels; i++)
            v->pcm[i] = (float *)realloc(v->pcm[i], v->pcm_storage * sizeof(float));
    }
    for (i = 0; i < s->channels; i++) ret[i] = v->pcm[i] + v->pcm_current;
    return ret;
}

static void preextrapolate(orc_stream *v)
{
    int i;
    int order = 16;
    float lpc[16];
    float *work = (float *)malloc(v->pcm_current * sizeof(*work));
    long j;
    v->preextrapolate = 1;

    if (v->pcm_current - v->centerW > order * 2) {
        for (i = 0; i < v->s->channels; i++) {
            for (j = 0; j < v->pcm_current; j++) work[j] = v->pcm[i][v->pcm_current - j - 1];

            orc_lpc_from_data(work, lpc, v->pcm_current - v->centerW, order);
            orc_lpc_predict(lpc, work + v->pcm_current - v->centerW - order, order,
                            work + v->pcm_current - v->centerW, v->centerW);

            for (j = 0; j < v->pcm_current; j++) v->pcm[i][v->pcm_current - j - 1] = work[j];
        }
    }
    free(work);
}

int orc_analysis_wrote(orc_stream *v, int vals)
{
    const orc_setup *s = v->s;
    {
        int i, j;
        for (i = 0; i < vals; i++)
            for (j = 0; j < s->channels; j++) v->pcm[j][v->pcm_current + i] *= s->pre_amplitude;
    }

    if (vals <= 0) {
        int order = 32;
        int i;
        float lpc[32];
        float *dummy[ORC_MAXCH];

        if (!v->preextrapolate) preextrapolate(v);

        orc_analysis_buffer(v, (int)s->blocksizes[1] * 3, dummy);
        v->eofflag = v->pcm_current;
        v->pcm_current += s->blocksizes[1] * 3;

        for (i = 0; i < s->channels; i++) {
            if (v->eofflag > order * 2) {
                long n;
                n = v->eofflag;
                if (n > s->blocksizes[1]) n = s->blocksizes[1];
                orc_lpc_from_data(v->pcm[i] + v->eofflag - n, lpc, n, order);
                orc_lpc_predict(lpc, v->pcm[i] + v->eofflag - order, order, v->pcm[i] + v->eofflag,
                                v->pcm_current - v->eofflag);
            } else {
                memset(v->pcm[i] + v->eofflag, 0, (v->pcm_current - v->eofflag) * sizeof(*v->pcm[i]));
            }
        }
    } else {
        if (v->pcm_current + vals > v->pcm_storage) return (-131);

        v->pcm_current += vals;

        if (!v->preextrapolate && v->pcm_current - v->centerW > s->blocksizes[1]) preextrapolate(v);
    }
    return (0);
}

static float ampmax_decay(float amp, const orc_stream *v)
{
    const orc_setup *s = v->s;
    int n = s->blocksizes[v->W] / 2;
    float secs = (float)n / s->rate;

    amp += secs * s->psy_g.ampmax_att_per_sec;
    if (amp < -9999) amp = -9999;
    return (amp);
}

int orc_analysis_blockout(orc_stream *v, orc_block *vb)
{
    int i;
    const orc_setup *s = v->s;
    long beginW = v->centerW - s->blocksizes[v->W] / 2, centerNext;

    if (!v->preextrapolate) return (0);
    if (v->eofflag == -1) return (0);

    {
        long bp = orc_ve_envelope_search(v);
        if (bp == -1) {
            if (v->eofflag == 0) return (0);
            v->nW = 0;
        } else {
            if (s->blocksizes[0] == s->blocksizes[1]) v->nW = 0;
            else v->nW = bp;
        }
    }

    centerNext = v->centerW + s->blocksizes[v->W] / 4 + s->blocksizes[v->nW] / 4;

    {
        long blockbound = centerNext + s->blocksizes[v->nW] / 2;
        if (v->pcm_current < blockbound) return (0);
    }

    vb->lW = v->lW;
    vb->W = v->W;
    vb->nW = v->nW;

    if (v->W) {
        if (!v->lW || !v->nW) vb->blocktype = 0; /* BLOCKTYPE_TRANSITION */
        else vb->blocktype = 1;                  /* BLOCKTYPE_LONG */
    } else {
        if (orc_ve_envelope_mark(v)) vb->blocktype = 0; /* BLOCKTYPE_IMPULSE */
        else vb->blocktype = 1;                         /* BLOCKTYPE_PADDING */
    }

    vb->sequence = v->sequence++;
    vb->granulepos = v->granulepos;
    vb->pcmend = s->blocksizes[v->W];
    vb->eofflag = 0;

    if (vb->ampmax > v->g_ampmax) v->g_ampmax = vb->ampmax;
    v->g_ampmax = ampmax_decay(v->g_ampmax, v);
    vb->ampmax = v->g_ampmax;

    for (i = 0; i < s->channels; i++)
        memcpy(vb->pcmbuf[i], v->pcm[i] + beginW, vb->pcmend * sizeof(float));

    if (v->eofflag) {
        if (v->centerW >= v->eofflag) {
            v->eofflag = -1;
            vb->eofflag = 1;
            return (1);
        }
    }

    {
        int new_centerNext = s->blocksizes[1] / 2;
        int movementW = centerNext - new_centerNext;

        if (movementW > 0) {
            orc_ve_envelope_shift(v, movementW);
            v->pcm_current -= movementW;

            for (i = 0; i < s->channels; i++)
                memmove(v->pcm[i], v->pcm[i] + movementW, v->pcm_current * sizeof(*v->pcm[i]));

            v->lW = v->W;
            v->W = v->nW;
            v->centerW = new_centerNext;

            if (v->eofflag) {
                v->eofflag -= movementW;
                if (v->eofflag <= 0) v->eofflag = -1;
                if (v->centerW >= v->eofflag) {
                    v->granulepos += movementW - (v->centerW - v->eofflag);
                } else {
                    v->granulepos += movementW;
                }
            } else {
                v->granulepos += movementW;
            }
        }
    }
    return (1);
}

/* vorbis_bitrate_addblock, managed branch (lib/bitrate.c:98-226): pick one of the PACKETBLOBS packets of
 * this block from the running reservoirs, truncate or zero-pad it, update the reservoirs */
static void bitrate_addblock(orc_stream *v, orc_block *vb)
{
    const orc_setup *s = v->s;
    int choice = rint(v->bm_avgfloat);
    long this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
    long min_target_bits = (vb->W ? v->bm_min_bitsper * v->bm_short_per_long : v->bm_min_bitsper);
    long max_target_bits = (vb->W ? v->bm_max_bitsper * v->bm_short_per_long : v->bm_max_bitsper);
    int samples = (int)(s->blocksizes[vb->W] >> 1);
    long desired_fill = s->bi_reservoir_bits * s->bi_reservoir_bias;

    if (v->bm_avg_bitsper > 0) {
        double slew = 0.;
        long avg_target_bits = (vb->W ? v->bm_avg_bitsper * v->bm_short_per_long : v->bm_avg_bitsper);
        double slewlimit = 15. / s->bi_slew_damp;

        if (v->bm_avg_reservoir + (this_bits - avg_target_bits) > desired_fill) {
            while (choice > 0 && this_bits > avg_target_bits &&
                   v->bm_avg_reservoir + (this_bits - avg_target_bits) > desired_fill) {
                choice--;
                this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
            }
        } else if (v->bm_avg_reservoir + (this_bits - avg_target_bits) < desired_fill) {
            while (choice + 1 < ORC_PACKETBLOBS && this_bits < avg_target_bits &&
                   v->bm_avg_reservoir + (this_bits - avg_target_bits) < desired_fill) {
                choice++;
                this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
            }
        }

        slew = rint(choice - v->bm_avgfloat) / samples * s->rate;
        if (slew < -slewlimit) slew = -slewlimit;
        if (slew > slewlimit) slew = slewlimit;
        choice = rint(v->bm_avgfloat += slew / s->rate * samples);
        this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
    }

    if (v->bm_min_bitsper > 0) {
        if (this_bits < min_target_bits) {
            while (v->bm_minmax_reservoir - (min_target_bits - this_bits) < 0) {
                choice++;
                if (choice >= ORC_PACKETBLOBS) break;
                this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
            }
        }
    }

    if (v->bm_max_bitsper > 0) {
        if (this_bits > max_target_bits) {
            while (v->bm_minmax_reservoir + (this_bits - max_target_bits) > s->bi_reservoir_bits) {
                choice--;
                if (choice < 0) break;
                this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
            }
        }
    }

    if (choice < 0) {
        long maxsize = (max_target_bits + (s->bi_reservoir_bits - v->bm_minmax_reservoir)) / 8;
        vb->choice = choice = 0;
        if (orc_bits_bytes(&vb->blob[choice]) > maxsize) {
            orc_bits_writetrunc(&vb->blob[choice], maxsize * 8);
            this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
        }
    } else {
        long minsize = (min_target_bits - v->bm_minmax_reservoir + 7) / 8;
        if (choice >= ORC_PACKETBLOBS) choice = ORC_PACKETBLOBS - 1;
        vb->choice = choice;
        minsize -= orc_bits_bytes(&vb->blob[choice]);
        while (minsize-- > 0) orc_bits_write(&vb->blob[choice], 0, 8);
        this_bits = orc_bits_bytes(&vb->blob[choice]) * 8;
    }

    if (v->bm_min_bitsper > 0 || v->bm_max_bitsper > 0) {
        if (max_target_bits > 0 && this_bits > max_target_bits) {
            v->bm_minmax_reservoir += (this_bits - max_target_bits);
        } else if (min_target_bits > 0 && this_bits < min_target_bits) {
            v->bm_minmax_reservoir += (this_bits - min_target_bits);
        } else {
            if (v->bm_minmax_reservoir > desired_fill) {
                if (max_target_bits > 0) {
                    v->bm_minmax_reservoir += (this_bits - max_target_bits);
                    if (v->bm_minmax_reservoir < desired_fill) v->bm_minmax_reservoir = desired_fill;
                } else {
                    v->bm_minmax_reservoir = desired_fill;
                }
            } else {
                if (min_target_bits > 0) {
                    v->bm_minmax_reservoir += (this_bits - min_target_bits);
                    if (v->bm_minmax_reservoir > desired_fill) v->bm_minmax_reservoir = desired_fill;
                } else {
                    v->bm_minmax_reservoir = desired_fill;
                }
            }
        }
    }

    if (v->bm_avg_bitsper > 0) {
        long avg_target_bits = (vb->W ? v->bm_avg_bitsper * v->bm_short_per_long : v->bm_avg_bitsper);
        v->bm_avg_reservoir += this_bits - avg_target_bits;
    }
}

/* vorbis_analysis(vb, NULL) + vorbis_bitrate_addblock + vorbis_bitrate_flushpacket (lib/analysis.c:30-62,
 * lib/bitrate.c:73-252): the packet of the block is vb->opb afterwards in both modes */
int orc_analysis(orc_stream *v, orc_block *vb)
{
    int i, ret;
    orc_bits_reset(&vb->opb);
    if (v->bm_managed)
        for (i = 0; i < ORC_PACKETBLOBS; i++) orc_bits_reset(&vb->blob[i]);
    ret = orc_mapping0_forward(v, vb);
    if (ret) return ret;
    vb->choice = ORC_PACKETBLOBS / 2;
    if (v->bm_managed) {
        const orc_bits *c;
        long nb;
        for (i = 0; i < ORC_PACKETBLOBS; i++) vb->blob_bytes[i] = (int)orc_bits_bytes(&vb->blob[i]);
        bitrate_addblock(v, vb);
        c = &vb->blob[vb->choice];
        nb = orc_bits_bytes(c);
        for (i = 0; i < nb; i++) orc_bits_write(&vb->opb, c->buf[i], 8);
    }
    return 0;
}

int orc_block_choice(const orc_block *vb, int *blob_bytes)
{
    if (blob_bytes) memcpy(blob_bytes, vb->blob_bytes, sizeof(vb->blob_bytes));
    return vb->choice;
}

const unsigned char *orc_block_blob(const orc_block *vb, int k, long *bytes)
{
    *bytes = orc_bits_bytes(&vb->blob[k]);
    return vb->blob[k].buf;
}

void orc_stream_bitrate_state(const orc_stream *v, int64_t *out, double *avgfloat)
{
    out[0] = v->bm_avg_reservoir;
    out[1] = v->bm_minmax_reservoir;
    *avgfloat = v->bm_avgfloat;
}

const unsigned char *orc_block_packet(const orc_block *vb, long *bytes)
{
    *bytes = orc_bits_bytes(&vb->opb);
    return vb->opb.buf;
}

/* ---- survey probe: signal + driver of SURVEY.md Appendix B ----------------------------- */
static float probe_rnd(unsigned *lcg)          /* state is the caller's: the probe runs one stream per thread */
{
    *lcg = *lcg * 1664525u + 1013904223u;
    return ((*lcg >> 8) & 0xffff) / 32768.f - 1.f;
}

long orc_encode_probe(const orc_setup *s, int secs, const char *out_path, double *seconds_spent)
{
    orc_stream *v = orc_stream_new(s);
    orc_block *vb = orc_block_new(s);
    FILE *f = out_path ? fopen(out_path, "wb") : NULL;
    long rate = s->rate, total = rate * secs, pos = 0, npk = 0;
    int ch = s->channels, i, c;
    struct timespec t0, t1;
    float *b[ORC_MAXCH];
    unsigned probe_lcg = 12345u;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (1) {
        if (pos >= total) {
            orc_analysis_wrote(v, 0);
        } else {
            int n = 1024;
            orc_analysis_buffer(v, 1024, b);
            for (i = 0; i < n; i++) {
                double t = (double)(pos + i) / rate;
                int burst = (((pos + i) / (rate / 3)) % 4 == 3) && ((pos + i) % (rate / 3)) < 200;
                for (c = 0; c < ch; c++) {
                    float smp = 0.3f * sin(2 * M_PI * 440.0 * (c + 1) * t) + 0.2f * sin(2 * M_PI * 3000.0 * t + c) +
                                0.05f * probe_rnd(&probe_lcg);
                    if (burst) smp += 0.6f * probe_rnd(&probe_lcg);
                    b[c][i] = smp;
                }
            }
            orc_analysis_wrote(v, n);
            pos += n;
        }
        while (orc_analysis_blockout(v, vb) == 1) {
            long bytes;
            const unsigned char *pkt;
            orc_analysis(v, vb);
            pkt = orc_block_packet(vb, &bytes);
            if (f) {
                int len = (int)bytes;
                fwrite(&len, 4, 1, f);
                fwrite(pkt, 1, bytes, f);
            }
            npk++;
        }
        if (pos >= total) break;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_spent) *seconds_spent = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    if (f) fclose(f);
    orc_block_free(vb);
    orc_stream_free(v);
    return npk;
}

/* the probe driver's PCM on its own: out[c * nsamples + i], nsamples a multiple of 1024 (the driver
 * writes whole 1024-sample chunks) */
void orc_probe_signal(int ch, long rate, long nsamples, float *out)
{
    long i;
    int c;
    unsigned probe_lcg = 12345u;
    for (i = 0; i < nsamples; i++) {
        double t = (double)i / rate;
        int burst = ((i / (rate / 3)) % 4 == 3) && (i % (rate / 3)) < 200;
        for (c = 0; c < ch; c++) {
            float smp = 0.3f * sin(2 * M_PI * 440.0 * (c + 1) * t) + 0.2f * sin(2 * M_PI * 3000.0 * t + c) +
                        0.05f * probe_rnd(&probe_lcg);
            if (burst) smp += 0.6f * probe_rnd(&probe_lcg);
            out[c * nsamples + i] = smp;
        }
    }
}

/* ---- accessors for the Python test harness (tests/orc.py) -------------------------------- */
void orc_stream_set_capture(orc_stream *v, int on) { v->capture = on; }

void orc_block_info(const orc_block *vb, int *out)
{
    out[0] = vb->lW; out[1] = vb->W; out[2] = vb->nW; out[3] = vb->blocktype; out[4] = vb->pcmend;
    out[5] = vb->cap_block_mode; out[6] = vb->eofflag;
}

void orc_block_info64(const orc_block *vb, int64_t *out)
{
    out[0] = vb->granulepos; out[1] = vb->sequence;
}

const void *orc_block_cap(const orc_block *vb, const char *name, int ch)
{
    if (!strcmp(name, "pcm")) return vb->pcmbuf[ch];
    if (!strcmp(name, "windowed")) return vb->cap_windowed[ch];
    if (!strcmp(name, "mdct_raw")) return vb->cap_gmdct_raw[ch];
    if (!strcmp(name, "mdct")) return vb->cap_gmdct[ch];
    if (!strcmp(name, "logfft")) return vb->cap_logfft[ch];
    if (!strcmp(name, "logmdct")) return vb->cap_logmdct[ch];
    if (!strcmp(name, "noise")) return vb->cap_noise[ch];
    if (!strcmp(name, "tone")) return vb->cap_tone[ch];
    if (!strcmp(name, "logmask")) return vb->cap_logmask[ch];
    if (!strcmp(name, "epeak")) return vb->cap_epeak[ch];
    if (!strcmp(name, "npeak")) return vb->cap_npeak[ch];
    if (!strcmp(name, "ilogmask")) return vb->cap_ilogmask[ch];
    if (!strcmp(name, "residue")) return vb->cap_residue[ch];
    if (!strcmp(name, "post")) return vb->cap_post[ch];
    if (!strcmp(name, "post_valid")) return &vb->cap_post_valid[ch];
    if (!strcmp(name, "nonzero")) return &vb->cap_nonzero[ch];
    if (!strcmp(name, "local_ampmax")) return &vb->cap_local_ampmax[ch];
    if (!strcmp(name, "global_ampmax")) return &vb->cap_global_ampmax;
    return 0;
}
