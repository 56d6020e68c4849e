/* ORACLE — test infrastructure only.
 *
 * Plain-C, scalar, single-threaded restatement of the reference encoder's hot path
 * (spvkgn/vorbis-aotuv-lancer = libvorbis 1.3.7 + aoTuV b6.03, SCALAR C path, i.e. the
 * `#else` branches of every `#ifdef __SSE__`).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this library; the product
 * (vorbis_aotuv_lancer_amd/) never does.
 *
 * Parity pin: see oracle/README.md.  The reference itself is NOT buildable in this
 * image under the build rules (every translation unit needs libogg's <ogg/ogg.h>,
 * which is absent and may not be stood in for), so no oracle/_ref exists.  The
 * restatement is pinned end-to-end against packet dumps of the reference's scalar
 * build that the survey stage recorded (md5 0b15c75f…, 4e93ce63…),
 * committed under tests/golden/.
 *
 * Build: make -C oracle   (gcc -O2 -fno-fast-math -ffp-contract=off)
 */
#ifndef ORACLE_H
#define ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MDCT (lib/mdct.c, lib/mdct.h:55-73) ------------------------------------ */
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

/* ---- window (lib/window.c:2137-2261) ------------------------------------------
 * win_l / win_r: rising half-windows (ln/2 and rn/2 floats) of the previous / next
 * block size; n = this block's size.  For short blocks pass ln = rn = n. */
void orc_apply_window(float *d, long n, const float *win_l, long ln, const float *win_r, long rn);

#ifdef __cplusplus
}
#endif
#endif
