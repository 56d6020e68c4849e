/* ORACLE — test infrastructure only.
 *
 * Plain-C, scalar, single-threaded restatement of the reference encoder's hot path
 * (spvkgn/vorbis-aotuv-lancer = libvorbis 1.3.7 + aoTuV b6.03, SCALAR C path, i.e. the
 * `#else` branches of every `#ifdef __SSE__`).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this library; the product
 * (vorbis_aotuv_lancer_amd/) never does.
 *
 * Parity pin: oracle/README.md.  The reference itself is NOT buildable in this image under
 * the build rules (every translation unit needs libogg's <ogg/ogg.h>, which is absent and
 * may not be stood in for), so there is no oracle/_ref.  The restatement is pinned
 * end-to-end against packet dumps of the reference's scalar build recorded by the survey
 * stage (md5 0b15c75f… and 4e93ce63…), committed under tests/golden/.
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off)
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PACKETBLOBS 15
#define ORC_P_BANDS 17
#define ORC_P_LEVELS 8
#define ORC_P_NOISECURVES 3
#define ORC_NOISE_COMPAND_LEVELS 40
#define ORC_EHMER_MAX 56
#define ORC_EHMER_OFFSET 16
#define ORC_MAX_ATH 88
#define ORC_VIF_POSIT 63
#define ORC_VE_BANDS 12
#define ORC_VE_PRE 16
#define ORC_VE_WIN 4
#define ORC_VE_POST 2
#define ORC_VE_AMP (ORC_VE_PRE + ORC_VE_POST - 1)
#define ORC_VE_NEARDC 15
#define ORC_VE_MINSTRETCH 2
#define ORC_VE_MAXSTRETCH 12
#define ORC_MAXCH 8

/* ---- bit packer (libogg oggpack_* LSb-first semantics; doc/02-bitpacking.tex) -------- */
typedef struct {
    unsigned char *buf;
    long storage;
    long endbyte;
    int endbit;
} orc_bits;

void orc_bits_init(orc_bits *b);
void orc_bits_reset(orc_bits *b);
void orc_bits_clear(orc_bits *b);
void orc_bits_write(orc_bits *b, unsigned long value, int bits);
long orc_bits_bytes(const orc_bits *b);
void orc_bits_writetrunc(orc_bits *b, long bits);

/* ---- MDCT (lib/mdct.c, lib/mdct.h:55-73) ------------------------------------------- */
typedef struct {
    int n;
    int log2n;
    float *trig;  /* n + n/4 floats */
    int *bitrev;  /* n/4 ints */
    float scale;
} orc_mdct;

void orc_mdct_init(orc_mdct *m, int n);
void orc_mdct_clear(orc_mdct *m);
void orc_mdct_forward(const orc_mdct *m, const float *in, float *out);
void orc_mdct_butterflies(const orc_mdct *m, float *x, int points);
void orc_mdct_bitreverse(const orc_mdct *m, float *w);

/* ---- real FFT (lib/smallft.c scalar FFTPACK path) ------------------------------------ */
typedef struct {
    int n;
    float *trigcache; /* 3n floats: [0,n) scratch, [n,3n) twiddles */
    int *splitcache;  /* 32 ints: n, nf, factors */
} orc_drft;

void orc_drft_init(orc_drft *l, int n);
void orc_drft_clear(orc_drft *l);
void orc_drft_forward(const orc_drft *l, float *data);
float orc_fft_logpower(const orc_drft *fft, float *pcm, int n); /* lib/mapping0.c:795, 847-888 */

/* ---- window (lib/window.c:2137-2261) ------------------------------------------------- */
void orc_apply_window(float *d, long n, const float *win_l, long ln, const float *win_r, long rn);

/* ---- codebooks (lib/codebook.h:33-79, lib/sharedbook.c:85-317) ----------------------- */
typedef struct {
    int dim, entries, maptype, q_quant, q_sequencep;
    long q_min, q_delta;
    const signed char *lengthlist;
    const int *quantlist;
    int nquant;
    uint32_t *codelist; /* bit-reversed Huffman words */
    int quantvals, minval, delta;
} orc_book;

int orc_ilog(uint32_t v);
int orc_book_encode(const orc_book *b, int a, orc_bits *opb);
int orc_book_besterror(const orc_book *book, int *a); /* lib/res0.c:316-378 */

/* ---- floor 1 -------------------------------------------------------------------------- */
typedef struct {
    /* vorbis_info_floor1, lib/backends.h:57-83 */
    int partitions;
    int partitionclass[31];
    int class_dim[16], class_subs[16], class_book[16], class_subbook[16][8];
    int mult;
    int postlist[ORC_VIF_POSIT + 2];
    float maxover, maxunder, maxerr, twofitweight, twofitatten;
    int info_n;
    /* vorbis_look_floor1, lib/codec_internal.h:138-156 */
    int sorted_index[ORC_VIF_POSIT + 2], forward_index[ORC_VIF_POSIT + 2], reverse_index[ORC_VIF_POSIT + 2];
    int hineighbor[ORC_VIF_POSIT], loneighbor[ORC_VIF_POSIT];
    int posts, n, quant_q;
} orc_floor;

/* ---- residue ---------------------------------------------------------------------------- */
typedef struct {
    int type;
    long begin, end;
    int grouping, partitions, partvals_info, groupbook;
    int secondstages[64], booklist[512], classmetric1[64], classmetric2[64];
    /* look, lib/res0.c:255-313 */
    int parts, stages, partvals;
    const orc_book *phrasebook;
    const orc_book *partbooks[64][8];
} orc_residue;

/* ---- psychoacoustics -------------------------------------------------------------------- */
typedef struct {
    /* vorbis_info_psy, lib/psy.h:38-65 */
    int blockflag;
    float ath_adjatt, ath_maxatt;
    float tone_masteratt[ORC_P_NOISECURVES];
    float tone_centerboost, tone_decay, tone_abs_limit;
    float toneatt[ORC_P_BANDS];
    int noisemaskp;
    float noisemaxsupp, noisewindowlo, noisewindowhi;
    int noisewindowlomin, noisewindowhimin, noisewindowfixed;
    float noiseoff[ORC_P_NOISECURVES][ORC_P_BANDS];
    float noisecompand[ORC_NOISE_COMPAND_LEVELS], noisecompand_high[ORC_NOISE_COMPAND_LEVELS];
    float flacint, max_curve_dB;
    int normal_p, normal_start, normal_partition;
    double normal_thresh;
    /* vorbis_look_psy, lib/psy.h:96-151 */
    int n;
    float tonecurves[ORC_P_BANDS][ORC_P_LEVELS][ORC_EHMER_MAX + 2];
    float *noiseoffset[ORC_P_NOISECURVES];
    float *ath;
    long *octave;
    long *bark;
    long firstoc, shiftoc;
    int eighth_octave_lines, total_octave_lines;
    long rate;
    int m3n[4];
    float m_val;
    int tonecomp_endp;
    float tonecomp_thres;
    float *ntfix_noiseoffset;
    int min_nn_lp, tonefix_end;
    int n25p, n33p, n75p, nn75pt, nn50pt, nn25pt;
} orc_psy;

typedef struct {
    /* vorbis_info_psy_global, lib/psy.h:67-86 */
    int eighth_octave_lines;
    float preecho_thresh[ORC_VE_BANDS], postecho_thresh[ORC_VE_BANDS];
    float stretch_penalty, preecho_minenergy, ampmax_att_per_sec;
    int coupling_pkHz[ORC_PACKETBLOBS];
    int coupling_pointlimit[2][ORC_PACKETBLOBS];
    int coupling_prepointamp[ORC_PACKETBLOBS], coupling_postpointamp[ORC_PACKETBLOBS];
    int sliding_lowpass[2][ORC_PACKETBLOBS];
} orc_psyg;

typedef struct {
    int submaps;
    int chmuxlist[256];
    int floorsubmap[16], residuesubmap[16];
    int coupling_steps;
    int coupling_mag[256], coupling_ang[256];
} orc_map;

/* static tables (data/common.vpk) */
typedef struct {
    const float *window[8];  /* 64 .. 8192, rising halves */
    const float *ATH;        /* 88 */
    const float *tonemasks;  /* [17][6][56] */
    const double *stereo_threshholds, *stereo_threshholds_X;
    const int *m3n32, *m3n44, *m3n48, *m3n32x2, *m3n44x2, *m3n48x2;
    const int *freq_bfn128, *freq_bfn256, *stn_compand;
    const float *ntfix_offset;
    const int *aotuv_ints;     /* [12][3]: tonecomp_endp, min_nn_lp, tonefix_end */
    const float *aotuv_thres;  /* [12] */
    const float *fromdB;       /* 256 */
    const int *ve_band_begin, *ve_band_end;
} orc_common;

/* envelope detector lookup, lib/envelope.h:51-70 (immutable part) */
typedef struct {
    int begin, end;
    float *window;
    float total;
} orc_ve_band;

typedef struct orc_setup {
    void *packs[2]; /* vpk images keeping table memory alive */
    orc_common c;
    int channels;
    long rate;
    long blocksizes[2];
    int modes, maps, floors, residues, books, psys;
    int mode_blockflag[2], mode_mapping[2];
    orc_map map[2];
    orc_floor floor[4];
    orc_residue residue[4];
    orc_book *book;
    orc_psy psy[4];
    orc_psyg psy_g;
    int block_lowpassr[2];
    float pre_amplitude;
    int modebits;
    /* bitrate_manager_info (lib/bitrate.h:41-50); managed = hi.managed (lib/vorbisenc.c:1037) */
    int managed;
    long bi_avg_rate, bi_min_rate, bi_max_rate, bi_reservoir_bits;
    double bi_reservoir_bias, bi_slew_damp;
    long bitrate_upper, bitrate_nominal, bitrate_lower; /* vorbis_info fields set by lib/vorbisenc.c:876-884 */
    /* looks built once (lib/block.c:181-303) */
    orc_mdct mdct[2];
    orc_drft fft[2];
    int window[2]; /* index into c.window */
    /* envelope */
    orc_mdct ve_mdct;
    float *ve_mdct_win;
    orc_ve_band ve_band[ORC_VE_BANDS];
    float ve_minenergy;
} orc_setup;

orc_setup *orc_setup_load(const char *common_vpk, const char *mode_vpk);
void orc_setup_free(orc_setup *s);
int orc_setup_table(const orc_setup *s, const char *name, const void **data, long *count, char *kind);

/* ---- stage functions (same argument meaning as the reference externs, lib/psy.h:188-242) */
float orc_postnoise_detection(const float *pcm, int nn, int mode, int lw_mode);
float orc_lb_loudnoise_fix(const orc_psy *p, float noise_compand_level, const float *logmdct,
                           int block_mode, int lW_block_mode);
void orc_noisemask(const orc_setup *s, const orc_psy *p, float noise_compand_level, const float *logmdct,
                   const float *lastmdct, float *epeak, float *npeak, float *logmask, float poste,
                   int block_mode);
void orc_tonemask(const orc_psy *p, const float *logfft, float *logmask, float global_specmax,
                  float local_specmax);
void orc_offset_and_mix(const orc_setup *s, const orc_psy *p, const float *noise, const float *tone,
                        int offset_select, int bit_managed, float *logmask, float *mdct, float *logmdct,
                        float *lastmdct, float *tempmdct, float low_compand, float *npeak, int end_block,
                        int block_mode, int nW_modenumber, int lW_block_mode, int lW_no, int impadnum);
void orc_couple_quantize_normalize(const orc_setup *s, int blobno, const orc_psy *p, const orc_map *vi,
                                   float **mdct, float **enpeak, float **nepeak, int **iwork, int *nonzero,
                                   int sliding_lowpass, int ch, int lowpassr);
/* floor1_fit: returns 1 and fills post[] if a floor was fit, 0 for "no floor" (NULL in the reference) */
int orc_floor1_fit(const orc_floor *look, const float *logmdct, const float *logmask, int *post);
int orc_floor1_interpolate_fit(const orc_floor *look, const int *A, const int *B, int del, int *output);
int orc_floor1_encode(const orc_setup *s, orc_bits *opb, const orc_floor *look, int *post /* may be NULL */,
                      int *ilogmask, int n_half);
/* residue: partword arrays are caller allocated [ch][partvals] */
int orc_res_class(const orc_residue *r, int **in, int *nonzero, int ch, long **partword);
int orc_res_forward(orc_bits *opb, const orc_residue *r, int **in, int *nonzero, int ch, long **partword,
                    int n_half);

/* ---- stream / block state (lib/block.c, lib/envelope.c, lib/mapping0.c) ----------------- */
typedef struct {
    float ampbuf[ORC_VE_AMP];
    int ampptr;
    float nearDC[ORC_VE_NEARDC];
    float nearDC_acc, nearDC_partialacc;
    int nearptr;
} orc_ve_filter;

typedef struct orc_block {
    /* vorbis_block public part */
    int lW, W, nW;
    int pcmend;
    int mode;
    int eofflag;
    int64_t granulepos, sequence;
    /* vorbis_block_internal */
    float ampmax;
    int blocktype;
    float *pcmbuf[ORC_MAXCH]; /* n floats per channel (window of the block) */
    orc_bits opb;                        /* the packet handed out (VBR: blob PACKETBLOBS/2 is written here directly) */
    orc_bits blob[ORC_PACKETBLOBS];      /* managed mode: vbi->packetblob[] */
    int blob_bytes[ORC_PACKETBLOBS];     /* sizes before the bitrate manager truncates / pads the chosen one */
    int choice;                          /* bm->choice of this block */
    /* captured intermediates of the last orc_analysis() (stage goldens for kernel tests) */
    float *cap_windowed[ORC_MAXCH], *cap_gmdct_raw[ORC_MAXCH], *cap_gmdct[ORC_MAXCH], *cap_logfft[ORC_MAXCH],
        *cap_logmdct[ORC_MAXCH], *cap_noise[ORC_MAXCH], *cap_tone[ORC_MAXCH], *cap_logmask[ORC_MAXCH],
        *cap_epeak[ORC_MAXCH], *cap_npeak[ORC_MAXCH];
    int *cap_ilogmask[ORC_MAXCH], *cap_residue[ORC_MAXCH];
    int cap_post[ORC_MAXCH][ORC_VIF_POSIT + 2];
    int cap_post_valid[ORC_MAXCH];
    int cap_nonzero[ORC_MAXCH];
    float cap_local_ampmax[ORC_MAXCH], cap_global_ampmax;
    int cap_block_mode;
} orc_block;

typedef struct orc_stream {
    const orc_setup *s;
    /* vorbis_dsp_state */
    float *pcm[ORC_MAXCH];
    int pcm_storage, pcm_current;
    int preextrapolate, eofflag;
    long lW, W, nW, centerW;
    int64_t granulepos, sequence;
    /* psy global look */
    float g_ampmax;
    /* aoTuV block state, lib/codec_internal.h:85-92 */
    float *lownoise_compand_level, *mblock, *tblock;
    int lW_block_mode, lW_no, impadnum;
    /* envelope state */
    orc_ve_filter *ve_filter;
    int ve_stretch;
    int *ve_mark;
    long ve_storage, ve_current, ve_curmark, ve_cursor;
    int capture; /* keep stage intermediates in blocks */
    /* bitrate_manager_state, lib/bitrate.h:25-39 */
    int bm_managed;
    long bm_avg_reservoir, bm_minmax_reservoir, bm_avg_bitsper, bm_min_bitsper, bm_max_bitsper, bm_short_per_long;
    double bm_avgfloat;
} orc_stream;

orc_stream *orc_stream_new(const orc_setup *s);
void orc_stream_free(orc_stream *v);
orc_block *orc_block_new(const orc_setup *s);
void orc_block_free(orc_block *b);
float **orc_analysis_buffer(orc_stream *v, int vals, float **ret);
int orc_analysis_wrote(orc_stream *v, int vals);
int orc_analysis_blockout(orc_stream *v, orc_block *vb); /* 1 = block ready */
int orc_analysis(orc_stream *v, orc_block *vb);          /* vorbis_analysis + VBR bitrate hand-off */
const unsigned char *orc_block_packet(const orc_block *vb, long *bytes);
/* managed mode: choice made by the bitrate manager and the 15 blob sizes it chose from */
int orc_block_choice(const orc_block *vb, int *blob_bytes /* [ORC_PACKETBLOBS] or NULL */);
const unsigned char *orc_block_blob(const orc_block *vb, int k, long *bytes);
void orc_stream_bitrate_state(const orc_stream *v, int64_t *out /* avg_reservoir, minmax_reservoir */, double *avgfloat);
void orc_block_info64(const orc_block *vb, int64_t *out); /* granulepos, sequence */

/* the three header packets (vorbis_analysis_headerout, lib/info.c:500-717) packed from this setup: returns the
 * total length, writes the packets back to back into buf (if cap suffices) and their lengths into lens[3] */
long orc_header_packets(const orc_setup *s, const char *vendor, const char *const *comments, int ncomments,
                        unsigned char *buf, long cap, long *lens);

/* survey probe signal + driver: encodes `secs` seconds of the synthetic
 * signal and writes [int32 len][bytes] records to `out_path`; returns the packet count. */
long orc_encode_probe(const orc_setup *s, int secs, const char *out_path, double *seconds_spent);
void orc_probe_signal(int ch, long rate, long nsamples, float *out);

#ifdef __cplusplus
}
#endif
#endif
