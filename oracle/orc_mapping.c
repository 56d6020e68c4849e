/* ORACLE — test infrastructure only.
 *
 * Per-block pipeline orchestration, restating mapping0_forward (lib/mapping0.c:738-1322,
 * scalar branches; VBR = packetblob PACKETBLOBS/2 only, managed bitrate = all PACKETBLOBS with the
 * two extra mask fits :1097-1168 and the interpolated fits :1169-1181):
 *   loop A :783-902   post-noise flag, window, MDCT, FFT, logfft + ampmax
 *   loop B :908-1182  logmdct, loudnoise fix, noise mask, tone mask, offset+mix, floor fit
 *   loop C :1204-1313 packet header bits, floor encode, couple/quantise, residue class+forward,
 *                     aoTuV block-state update
 * Every intermediate vector of the last block is kept in orc_block.cap_* when the stream's
 * `capture` flag is set, so kernel parity tests can compare stage by stage.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

/* lib/mapping0.c:795, 847-888: FFT in place, log-power spectrum into pcm[0..n/2), returns the
 * block's local_ampmax (already clamped to <= 0) */
float orc_fft_logpower(const orc_drft *fft, float *pcm, int n)
{
    float scale = 4.f / n;
    float scale_dB = orc_todB(&scale) + .345;
    float *logfft = pcm;
    float local_ampmax;
    int j;
    orc_drft_forward(fft, pcm);
    logfft[0] = scale_dB + orc_todB(pcm) + .345;
    local_ampmax = logfft[0];
    for (j = 1; j < n - 1; j += 2) {
        float temp = pcm[j] * pcm[j] + pcm[j + 1] * pcm[j + 1];
        temp = logfft[(j + 1) >> 1] = scale_dB + .5f * orc_todB(&temp) + .345;
        if (temp > local_ampmax) local_ampmax = temp;
    }
    if (local_ampmax > 0.f) local_ampmax = 0.f;
    return local_ampmax;
}

int orc_mapping0_forward(orc_stream *v, orc_block *vb)
{
    const orc_setup *s = v->s;
    int ch = s->channels;
    const orc_floor *vif = &s->floor[vb->W];
    int n = vb->pcmend;
    int i, j, k;

    int nonzero[ORC_MAXCH];
    float poste[ORC_MAXCH];
    float *gmdct[ORC_MAXCH], *epeak[ORC_MAXCH], *npeak[ORC_MAXCH];
    int *iwork[ORC_MAXCH];
    int floor_posts[ORC_MAXCH][ORC_PACKETBLOBS][ORC_VIF_POSIT + 2];
    int floor_valid[ORC_MAXCH][ORC_PACKETBLOBS]; /* 0 = the reference's NULL post vector (:930-931) */
    const int managed = v->bm_managed;
    const int MID = ORC_PACKETBLOBS / 2;

    float global_ampmax = vb->ampmax;
    float local_ampmax[ORC_MAXCH];
    int blocktype = vb->blocktype;
    int modenumber = vb->W;
    int block_mode;
    int lowpass_residue;
    const orc_map *info = &s->map[modenumber];
    const orc_psy *psy_look = &s->psy[blocktype + (vb->W ? 2 : 0)];
    int partition = (psy_look->normal_p ? psy_look->normal_partition : 16);
    const float *win_l, *win_r;
    long ln, rn;

    vb->mode = modenumber;

    block_mode = blocktype;
    block_mode |= (modenumber << 1);
    vb->cap_block_mode = block_mode;

    if (modenumber) lowpass_residue = s->block_lowpassr[1];
    else lowpass_residue = s->block_lowpassr[0];
    if (lowpass_residue % psy_look->normal_partition)
        lowpass_residue = (lowpass_residue / psy_look->normal_partition + 1) * psy_look->normal_partition;

    /* _vorbis_apply_window(pcm, b->window, ci->blocksizes, lW, W, nW): lib/window.c:2139-2151 */
    {
        int lW = (vb->W ? vb->lW : 0), nW = (vb->W ? vb->nW : 0);
        win_l = s->c.window[s->window[lW]];
        win_r = s->c.window[s->window[nW]];
        ln = s->blocksizes[lW];
        rn = s->blocksizes[nW];
    }

    for (i = 0; i < ch; i++) {
        float *pcm = vb->pcmbuf[i];
        float *logfft = pcm;

        iwork[i] = (int *)malloc(n / 2 * sizeof(int));
        gmdct[i] = (float *)malloc(n / 2 * sizeof(float));
        epeak[i] = (float *)malloc(n / 2 * sizeof(float));
        npeak[i] = (float *)malloc((n / 2 / partition + 1) * sizeof(float));

        poste[i] = orc_postnoise_detection(pcm, n, block_mode, v->lW_block_mode);

        orc_apply_window(pcm, n, win_l, ln, win_r, rn);
        if (v->capture) memcpy(vb->cap_windowed[i], pcm, n * sizeof(float));

        orc_mdct_forward(&s->mdct[vb->W], pcm, gmdct[i]);
        if (v->capture) memcpy(vb->cap_gmdct_raw[i], gmdct[i], n / 2 * sizeof(float));

        local_ampmax[i] = orc_fft_logpower(&s->fft[vb->W], pcm, n);
        if (local_ampmax[i] > global_ampmax) global_ampmax = local_ampmax[i];
        if (v->capture) memcpy(vb->cap_logfft[i], logfft, n / 2 * sizeof(float));
    }

    {
        float *noise = (float *)malloc(n / 2 * sizeof(*noise));
        float *tone = (float *)malloc(n / 2 * sizeof(*tone));

        for (i = 0; i < ch; i++) {
            int submap = info->chmuxlist[i];
            float *mdct = gmdct[i];
            float *logfft = vb->pcmbuf[i];
            float *logmdct = logfft + n / 2;
            float *logmask = logfft;
            float *enpeak = epeak[i];
            float *nepeak = npeak[i];
            float *lastmdct = v->mblock + i * 2048;
            float *tempmdct = v->tblock + i * 256;
            float *lowcomp = v->lownoise_compand_level + i;

            for (j = 0; j < n / 2; j++) logmdct[j] = orc_todB(mdct + j) + .345;
            if (v->capture) memcpy(vb->cap_logmdct[i], logmdct, n / 2 * sizeof(float));

            *lowcomp = orc_lb_loudnoise_fix(psy_look, *lowcomp, logmdct, block_mode, v->lW_block_mode);

            orc_noisemask(s, psy_look, *lowcomp, logmdct, lastmdct, enpeak, nepeak, noise, poste[i], block_mode);
            if (v->capture) memcpy(vb->cap_noise[i], noise, n / 2 * sizeof(float));

            orc_tonemask(psy_look, logfft, tone, global_ampmax, local_ampmax[i]);
            if (v->capture) memcpy(vb->cap_tone[i], tone, n / 2 * sizeof(float));

            memset(floor_valid[i], 0, sizeof(floor_valid[i]));
            orc_offset_and_mix(s, psy_look, noise, tone, 1, managed, logmask, mdct, logmdct, lastmdct, tempmdct,
                               *lowcomp, nepeak, vif->info_n, block_mode, vb->nW, v->lW_block_mode, v->lW_no,
                               v->impadnum);
            if (v->capture) {
                memcpy(vb->cap_logmask[i], logmask, n / 2 * sizeof(float));
                memcpy(vb->cap_gmdct[i], mdct, n / 2 * sizeof(float));
            }

            floor_valid[i][MID] =
                orc_floor1_fit(&s->floor[info->floorsubmap[submap]], logmdct, logmask, floor_posts[i][MID]);
            if (v->capture) {
                vb->cap_post_valid[i] = floor_valid[i][MID];
                memcpy(vb->cap_post[i], floor_posts[i][MID], sizeof(floor_posts[i][MID]));
            }

            /* lib/mapping0.c:1097-1181: two more fits (hi/lo rate) and the interpolated ones between */
            if (managed && floor_valid[i][MID]) {
                const orc_floor *fl = &s->floor[info->floorsubmap[submap]];
                orc_offset_and_mix(s, psy_look, noise, tone, 2, managed, logmask, mdct, logmdct, lastmdct, tempmdct,
                                   *lowcomp, nepeak, vif->info_n, block_mode, vb->nW, v->lW_block_mode, v->lW_no,
                                   v->impadnum);
                floor_valid[i][ORC_PACKETBLOBS - 1] =
                    orc_floor1_fit(fl, logmdct, logmask, floor_posts[i][ORC_PACKETBLOBS - 1]);

                orc_offset_and_mix(s, psy_look, noise, tone, 0, managed, logmask, mdct, logmdct, lastmdct, tempmdct,
                                   *lowcomp, nepeak, vif->info_n, block_mode, vb->nW, v->lW_block_mode, v->lW_no,
                                   v->impadnum);
                floor_valid[i][0] = orc_floor1_fit(fl, logmdct, logmask, floor_posts[i][0]);

                for (k = 1; k < MID; k++)
                    floor_valid[i][k] = orc_floor1_interpolate_fit(
                        fl, floor_valid[i][0] ? floor_posts[i][0] : NULL, floor_posts[i][MID], k * 65536 / MID,
                        floor_posts[i][k]);
                for (k = MID + 1; k < ORC_PACKETBLOBS - 1; k++)
                    floor_valid[i][k] = orc_floor1_interpolate_fit(
                        fl, floor_posts[i][MID],
                        floor_valid[i][ORC_PACKETBLOBS - 1] ? floor_posts[i][ORC_PACKETBLOBS - 1] : NULL,
                        (k - MID) * 65536 / MID, floor_posts[i][k]);
            }
        }
        free(noise);
        free(tone);
    }
    vb->ampmax = global_ampmax;
    vb->cap_global_ampmax = global_ampmax;
    for (i = 0; i < ch; i++) vb->cap_local_ampmax[i] = local_ampmax[i];

    {
        int *couple_bundle[ORC_MAXCH];
        int zerobundle[ORC_MAXCH];

        /* once for VBR, PACKETBLOBS times for managed bitrate (lib/mapping0.c:1204-1206); note that the
         * aoTuV block-state update at the end sits INSIDE this loop in the reference (:1296-1304) */
        for (k = (managed ? 0 : MID); k <= (managed ? ORC_PACKETBLOBS - 1 : MID); k++) {
        orc_bits *opb = managed ? &vb->blob[k] : &vb->opb;
        const int capture = v->capture && k == MID;

        orc_bits_write(opb, 0, 1);
        orc_bits_write(opb, modenumber, s->modebits);
        if (vb->W) {
            orc_bits_write(opb, vb->lW, 1);
            orc_bits_write(opb, vb->nW, 1);
        }

        for (i = 0; i < ch; i++) {
            int submap = info->chmuxlist[i];
            int *ilogmask = iwork[i];
            nonzero[i] = orc_floor1_encode(s, opb, &s->floor[info->floorsubmap[submap]],
                                           floor_valid[i][k] ? floor_posts[i][k] : NULL, ilogmask, n / 2);
            if (capture) memcpy(vb->cap_ilogmask[i], ilogmask, n / 2 * sizeof(int));
        }

        orc_couple_quantize_normalize(s, k, psy_look, info, gmdct, epeak, npeak, iwork, nonzero,
                                      s->psy_g.sliding_lowpass[vb->W][k], ch, lowpass_residue);
        if (capture)
            for (i = 0; i < ch; i++) {
                memcpy(vb->cap_residue[i], iwork[i], n / 2 * sizeof(int));
                memcpy(vb->cap_epeak[i], epeak[i], n / 2 * sizeof(float));
                memcpy(vb->cap_npeak[i], npeak[i], (n / 2 / partition) * sizeof(float));
                vb->cap_nonzero[i] = nonzero[i];
            }

        for (i = 0; i < info->submaps; i++) {
            int ch_in_bundle = 0;
            int resnum = info->residuesubmap[i];
            const orc_residue *r = &s->residue[resnum];
            long *partword[ORC_MAXCH];
            int partvals = (int)((r->end - r->begin) / r->grouping);
            int got;

            for (j = 0; j < ch; j++) {
                if (info->chmuxlist[j] == i) {
                    zerobundle[ch_in_bundle] = 0;
                    if (nonzero[j]) zerobundle[ch_in_bundle] = 1;
                    couple_bundle[ch_in_bundle++] = iwork[j];
                }
            }
            for (j = 0; j < ch_in_bundle; j++) partword[j] = (long *)calloc(partvals + 1, sizeof(long));

            got = orc_res_class(r, couple_bundle, zerobundle, ch_in_bundle, partword);

            ch_in_bundle = 0;
            for (j = 0; j < ch; j++)
                if (info->chmuxlist[j] == i) couple_bundle[ch_in_bundle++] = iwork[j];

            if (got) orc_res_forward(opb, r, couple_bundle, zerobundle, ch_in_bundle, partword, n / 2);
            for (j = 0; j < ch_in_bundle; j++) free(partword[j]);
        }

        if (block_mode >= 2) v->impadnum = 0;
        if ((!v->lW_block_mode) && (block_mode == 1)) v->impadnum = 1;
        else if (v->impadnum && v->impadnum < 8) v->impadnum++;
        if (v->lW_block_mode == block_mode) v->lW_no++;
        else v->lW_no = 1;
        v->lW_block_mode = block_mode;
        }
    }

    for (i = 0; i < ch; i++) {
        free(iwork[i]); free(gmdct[i]); free(epeak[i]); free(npeak[i]);
    }
    return (0);
}
