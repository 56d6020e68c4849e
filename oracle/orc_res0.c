/* ORACLE — test infrastructure only.
 *
 * Residue classification and cascaded lattice-VQ encode (backends 0/1/2), restating
 *   _01class      lib/res0.c:406-468     _2class      lib/res0.c:473-526
 *   _encodepart   lib/res0.c:384-404     _01forward   lib/res0.c:528-640
 *   res1_class/forward  lib/res0.c:715-745   res2_class/forward  lib/res0.c:759-799
 * (training hooks omitted: TRAIN_RES is not defined in a normal build).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "orc_internal.h"

static void class01(const orc_residue *info, int **in, int ch, long **partword)
{
    long i, j, k;
    int samples_per_partition = info->grouping;
    int possible_partitions = info->partitions;
    int n = info->end - info->begin;
    int partvals = n / samples_per_partition;
    float scale = 100. / samples_per_partition;

    for (i = 0; i < ch; i++) memset(partword[i], 0, n / samples_per_partition * sizeof(*partword[i]));

    for (i = 0; i < partvals; i++) {
        int offset = i * samples_per_partition + info->begin;
        for (j = 0; j < ch; j++) {
            int max = 0;
            int ent = 0;
            for (k = 0; k < samples_per_partition; k++) {
                if (abs(in[j][offset + k]) > max) max = abs(in[j][offset + k]);
                ent += abs(in[j][offset + k]);
            }
            ent *= scale;

            for (k = 0; k < possible_partitions - 1; k++)
                if (max <= info->classmetric1[k] && (info->classmetric2[k] < 0 || ent < info->classmetric2[k]))
                    break;

            partword[j][i] = k;
        }
    }
}

static void class2(const orc_residue *info, int **in, int ch, long **partword)
{
    long i, j, k, l;
    int samples_per_partition = info->grouping;
    int possible_partitions = info->partitions;
    int n = info->end - info->begin;
    int partvals = n / samples_per_partition;

    memset(partword[0], 0, partvals * sizeof(*partword[0]));

    for (i = 0, l = info->begin / ch; i < partvals; i++) {
        int magmax = 0;
        int angmax = 0;
        for (j = 0; j < samples_per_partition; j += ch) {
            if (abs(in[0][l]) > magmax) magmax = abs(in[0][l]);
            for (k = 1; k < ch; k++)
                if (abs(in[k][l]) > angmax) angmax = abs(in[k][l]);
            l++;
        }

        for (j = 0; j < possible_partitions - 1; j++)
            if (magmax <= info->classmetric1[j] && angmax <= info->classmetric2[j]) break;

        partword[0][i] = j;
    }
}

static int encodepart(orc_bits *opb, int *vec, int n, const orc_book *book)
{
    int i, bits = 0;
    int dim = book->dim;
    int step = n / dim;

    for (i = 0; i < step; i++) {
        int entry = orc_book_besterror(book, vec + i * dim);
        bits += orc_book_encode(book, entry, opb);
    }
    return (bits);
}

static int forward01(orc_bits *opb, const orc_residue *look, int **in, int ch, long **partword)
{
    long i, j, k, s;
    const orc_residue *info = look;
    int samples_per_partition = info->grouping;
    int possible_partitions = info->partitions;
    int partitions_per_word = look->phrasebook->dim;
    int n = info->end - info->begin;
    int partvals = n / samples_per_partition;

    for (s = 0; s < look->stages; s++) {
        for (i = 0; i < partvals;) {
            if (s == 0) {
                for (j = 0; j < ch; j++) {
                    long val = partword[j][i];
                    for (k = 1; k < partitions_per_word; k++) {
                        val *= possible_partitions;
                        if (i + k < partvals) val += partword[j][i + k];
                    }
                    if (val < look->phrasebook->entries) orc_book_encode(look->phrasebook, (int)val, opb);
                }
            }

            for (k = 0; k < partitions_per_word && i < partvals; k++, i++) {
                long offset = i * samples_per_partition + info->begin;

                for (j = 0; j < ch; j++) {
                    if (info->secondstages[partword[j][i]] & (1 << s)) {
                        const orc_book *statebook = look->partbooks[partword[j][i]][s];
                        if (statebook) encodepart(opb, in[j] + offset, samples_per_partition, statebook);
                    }
                }
            }
        }
    }
    return (0);
}

/* class: returns 0 when nothing is to be coded (the reference returns a NULL partword) */
int orc_res_class(const orc_residue *r, int **in, int *nonzero, int ch, long **partword)
{
    int i, used = 0;
    if (r->type == 2) {
        for (i = 0; i < ch; i++)
            if (nonzero[i]) used++;
        if (!used) return 0;
        class2(r, in, ch, partword);
        return 1;
    }
    /* type 0/1: compacts the in[] pointer array in place, as res1_class does (lib/res0.c:738-740) */
    for (i = 0; i < ch; i++)
        if (nonzero[i]) in[used++] = in[i];
    if (!used) return 0;
    class01(r, in, used, partword);
    return used;
}

int orc_res_forward(orc_bits *opb, const orc_residue *r, int **in, int *nonzero, int ch, long **partword,
                    int n_half)
{
    int i, used = 0;
    if (r->type == 2) {
        long j, k, n = n_half;
        int *work = (int *)malloc(ch * n * sizeof(*work));
        int ret = 0;
        for (i = 0; i < ch; i++) {
            int *pcm = in[i];
            if (nonzero[i]) used++;
            for (j = 0, k = i; j < n; j++, k += ch) work[k] = pcm[j];
        }
        if (used) ret = forward01(opb, r, &work, 1, partword);
        free(work);
        return ret;
    }
    for (i = 0; i < ch; i++)
        if (nonzero[i]) in[used++] = in[i];
    if (used) return forward01(opb, r, in, used, partword);
    return 0;
}
