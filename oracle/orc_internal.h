/* ORACLE — test infrastructure only.  Internal prototypes shared by the oracle's sources. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "oracle.h"

void orc_book_init_encode(orc_book *c);

/* lib/scales.h:43-51 */
static inline float orc_todB(const float *x)
{
    union { uint32_t i; float f; } ix;
    ix.f = *x;
    ix.i = ix.i & 0x7fffffff;
    return (float)(ix.i * 7.17711438e-7f - 764.6161886f);
}

/* lib/scales.h:32-40 */
static inline float orc_unitnorm(float x)
{
    union { uint32_t i; float f; } ix;
    ix.f = x;
    ix.i = (ix.i & 0x80000000U) | (0x3f800000U);
    return ix.f;
}

#define ORC_MIN(x, y) ((x) > (y) ? (y) : (x)) /* lib/os.h:80-86 */
#define ORC_MAX(x, y) ((x) < (y) ? (y) : (x))

/* envelope (orc_envelope.c) */
long orc_ve_envelope_search(orc_stream *v);
int orc_ve_envelope_mark(orc_stream *v);
void orc_ve_envelope_shift(orc_stream *v, long shift);

/* lpc (orc_lpc.c) */
float orc_lpc_from_data(float *data, float *lpci, int n, int m);
void orc_lpc_predict(float *coeff, float *prime, int m, float *data, long n);

/* mapping0_forward (orc_mapping.c) */
int orc_mapping0_forward(orc_stream *v, orc_block *vb);

#endif
