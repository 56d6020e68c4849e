// Encoder setup shared by the host code and the kernels: plain-old-data views of everything
// the reference keeps in codec_setup_info + the looks built by vorbis_analysis_init()
// (reference lib/codec_internal.h:101-136, lib/psy.h:96-151, lib/block.c:181-303).
// Pointers inside these structs are HOST pointers in the host copy and DEVICE pointers in the
// device copy (one arena upload, pointers rebased).
#pragma once
#include <stdint.h>

#define VBM_PACKETBLOBS 15
#define VBM_P_BANDS 17
#define VBM_P_LEVELS 8
#define VBM_P_NOISECURVES 3
#define VBM_NOISE_COMPAND_LEVELS 40
#define VBM_EHMER_MAX 56
#define VBM_EHMER_OFFSET 16
#define VBM_MAX_ATH 88
#define VBM_VIF_POSIT 63
#define VBM_MAXCH 8
#define VBM_MAX_BOOK_DIM 8

struct vbm_psy {                 // vorbis_info_psy + vorbis_look_psy
    int blockflag;
    float ath_adjatt, ath_maxatt;
    float tone_masteratt[VBM_P_NOISECURVES];
    float tone_centerboost, tone_decay, tone_abs_limit;
    float toneatt[VBM_P_BANDS];
    int noisemaskp;
    float noisemaxsupp, noisewindowlo, noisewindowhi;
    int noisewindowlomin, noisewindowhimin, noisewindowfixed;
    float noiseoff[VBM_P_NOISECURVES][VBM_P_BANDS];
    float noisecompand[VBM_NOISE_COMPAND_LEVELS], noisecompand_high[VBM_NOISE_COMPAND_LEVELS];
    float flacint, max_curve_dB;
    int normal_p, normal_start, normal_partition;
    double normal_thresh;
    // look
    int n;
    long rate;
    int firstoc, shiftoc, eighth_octave_lines, total_octave_lines;
    int m3n[4];
    float m_val;
    int tonecomp_endp;
    float tonecomp_thres;
    int min_nn_lp, tonefix_end;
    int n25p, n33p, n75p;
    int hy_i1, hy_i2;                // bark_noise_hybridmp phase limits, variable window (lib/psy.c:3543, :3565)
    int hy_rb;                       // rows [0, hy_rb) of the running sums are read directly (mirrored window edges): multiple of 16
    int hy_f1, hy_f2;                // same for the fixed window of noisewindowfixed (:3595, :3614); 0 if unused
    int hy_ring;                     // 1: every window of every bin fits the 512-row ring schedule of k_noisemask<.., RING> (setup_host.cpp)
    const float *tonecurves;         // [P_BANDS][P_LEVELS][EHMER_MAX+2]
    const float *noiseoffset[VBM_P_NOISECURVES];  // n each
    const float *ath;                // n
    const int *octave;               // n
    const int *bark_lo, *bark_hi;    // n each: lib/psy.c:471 packs ((lo-1)<<16)+(hi-1); kept unpacked-as-read
    const float *ntfix_noiseoffset;  // n
    // table-only loop structure of _vp_tonemask, unrolled by the host so the device can slice it:
    int ngroups;                     // runs of equal octave[] (the inner while of seed_loop, lib/psy.c:737-743)
    const int *group_start;          // ngroups+1 bins, last = n
    const int *group_tab;            // ngroups x {first bin, end bin, ath[last] as bits, octave[last]}: 16-byte records (k_tonemask)
    const int *seg_p0, *seg_p1;      // n each: seed lines [p0..p1] whose minimum max_seeds applies to the bin
                                     //   (lib/psy.c:1045-1076); p0 = -1 for the bins of the final tail (:1078-1084)
};

struct vbm_floor {               // vorbis_info_floor1 + vorbis_look_floor1
    int partitions;
    int partitionclass[31];
    int class_dim[16], class_subs[16], class_book[16], class_subbook[16][8];
    int mult;
    int postlist[VBM_VIF_POSIT + 2];
    float maxover, maxunder, maxerr, twofitweight, twofitatten;
    int info_n;
    int sorted_index[VBM_VIF_POSIT + 2], forward_index[VBM_VIF_POSIT + 2], reverse_index[VBM_VIF_POSIT + 2];
    int hineighbor[VBM_VIF_POSIT], loneighbor[VBM_VIF_POSIT];
    int posts, n, quant_q;
};

struct vbm_book {                // static_codebook + encode side of codebook
    int dim, entries;
    int quantvals, minval, delta;
    const signed char *lengthlist;   // entries
    const uint32_t *codelist;        // entries (bit-reversed words)
    // compact list of the entries that have a codeword, in ascending entry order, with their
    // lattice points — the exhaustive search of lib/res0.c:343-370 visits exactly these
    int used;
    const int *used_index;           // used
    const int *used_point;           // used x dim
    // the same points as 8 x int16 per entry (zero-padded past dim; 16 bytes, 16-byte aligned) and their
    // squared norms, for the packed-dot-product search; NULL if a coordinate does not fit 16 bits
    const short *used_pack;          // used x 8
    const int *used_norm;            // used
};

struct vbm_residue {             // vorbis_info_residue0 + look
    int type;
    int begin, end;
    int grouping, partitions, groupbook;
    int secondstages[64];
    int classmetric1[64], classmetric2[64];
    int stages, phrase_dim;
    int partbook[64][8];             // book index or -1
};

struct vbm_map {
    int submaps;
    int chmuxlist[VBM_MAXCH];
    int floorsubmap[16], residuesubmap[16];
    int coupling_steps;
    int coupling_mag[16], coupling_ang[16];
};

// envelope detector look (reference lib/envelope.h:27-76, _ve_envelope_init lib/envelope.c:42-87)
#define VBM_VE_BANDS 12
#define VBM_VE_PRE 16
#define VBM_VE_WIN 4
#define VBM_VE_POST 2
#define VBM_VE_AMP (VBM_VE_PRE + VBM_VE_POST - 1)
#define VBM_VE_NEARDC 15
#define VBM_VE_MINSTRETCH 2
#define VBM_VE_MAXSTRETCH 12
#define VBM_VE_MAXBAND 8
struct vbm_envelope {
    float preecho_thresh[VBM_VE_BANDS], postecho_thresh[VBM_VE_BANDS];
    float stretch_penalty, minenergy;
    int band_begin[VBM_VE_BANDS], band_end[VBM_VE_BANDS];
    float band_window[VBM_VE_BANDS][VBM_VE_MAXBAND], band_total[VBM_VE_BANDS];
    const float *mdct_win;           // 128: sin^2 window of the 128-point search MDCT
    const float *mdct_trig;          // 128 + 32
};

struct vbm_setup {
    int channels;
    long rate;
    int blocksizes[2];
    int modes, maps, floors, residues, books, psys;
    int modebits;
    int block_lowpassr[2];
    float pre_amplitude;
    vbm_map map[2];
    vbm_floor floor[4];
    vbm_residue residue[4];
    vbm_psy psy[4];
    const vbm_book *book;            // books
    // vorbis_info_psy_global (the parts the per-block path reads)
    int coupling_pointlimit[2][VBM_PACKETBLOBS];
    int coupling_prepointamp[VBM_PACKETBLOBS], coupling_postpointamp[VBM_PACKETBLOBS];
    int sliding_lowpass[2][VBM_PACKETBLOBS];
    float ampmax_att_per_sec;
    // bitrate_manager_info (reference lib/bitrate.h:41-50); managed = hi.managed (lib/vorbisenc.c:1037)
    int managed;
    long long bi_avg_rate, bi_min_rate, bi_max_rate, bi_reservoir_bits;
    double bi_reservoir_bias, bi_slew_damp;
    double hi_lowpass_khz;           // highlevel_encode_setup.lowpass_kHz (what OV_ECTL_LOWPASS_GET reports)
    // static tables
    double stereo_threshholds[9], stereo_threshholds_X[9];
    int stn_compand[VBM_NOISE_COMPAND_LEVELS];
    const int *freq_bfn128, *freq_bfn256;
    const float *fromdB;             // 256
    const float *window[2];          // rising half-windows of blocksizes[0], [1]
    const float *mdct_trig[2];       // n + n/4
    const float *fft_wa[2];          // n
    vbm_envelope ve;
};
