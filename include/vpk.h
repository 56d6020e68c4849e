/* VPK1 reader — flat container of named numeric arrays (format: tools/vpk.py).
 *
 * The mode pack (every immutable table one (channels, rate, quality) class of the
 * batched encoder needs — SURVEY.md §7 step 0) is delivered in this container.
 * Header-only, C99/C++; no allocation besides the file image itself.
 */
#ifndef VPK_H
#define VPK_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VPK_F32 = 0, VPK_F64, VPK_I32, VPK_I64, VPK_U8, VPK_I8, VPK_I16, VPK_U16, VPK_U32 };

typedef struct {
    const char *name; /* not NUL terminated */
    int name_len;
    int dtype;
    int ndim;
    uint32_t shape[4];
    uint64_t nbytes;
    const void *data;
} vpk_entry;

typedef struct {
    unsigned char *image;
    size_t size;
    int count;
    vpk_entry *entries;
} vpk_file;

static inline size_t vpk_elems(const vpk_entry *e)
{
    size_t n = 1;
    for (int i = 0; i < e->ndim; i++) n *= e->shape[i];
    return n;
}

/* returns 0 on success */
static inline int vpk_open(vpk_file *f, const char *path)
{
    memset(f, 0, sizeof(*f));
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 8) { fclose(fp); return -2; }
    /* 8-byte aligned image so that f64/i64 payloads are naturally aligned */
    if (posix_memalign((void **)&f->image, 64, (size_t)sz + 8)) { fclose(fp); return -3; }
    if (fread(f->image, 1, (size_t)sz, fp) != (size_t)sz) { fclose(fp); free(f->image); return -4; }
    fclose(fp);
    f->size = (size_t)sz;
    if (memcmp(f->image, "VPK1", 4)) { free(f->image); f->image = 0; return -5; }
    uint32_t count;
    memcpy(&count, f->image + 4, 4);
    f->count = (int)count;
    f->entries = (vpk_entry *)calloc(count ? count : 1, sizeof(vpk_entry));
    size_t pos = 8;
    for (uint32_t i = 0; i < count; i++) {
        vpk_entry *e = &f->entries[i];
        uint16_t nl;
        if (pos + 2 > f->size) goto bad;
        memcpy(&nl, f->image + pos, 2); pos += 2;
        e->name = (const char *)f->image + pos; e->name_len = nl; pos += nl;
        if (pos + 2 > f->size) goto bad;
        e->dtype = f->image[pos]; e->ndim = f->image[pos + 1]; pos += 2;
        if (e->ndim > 4 || e->dtype > VPK_U32) goto bad;
        for (int k = 0; k < e->ndim; k++) { memcpy(&e->shape[k], f->image + pos, 4); pos += 4; }
        memcpy(&e->nbytes, f->image + pos, 8); pos += 8;
        pos += (8 - (pos & 7)) & 7;
        if (pos + e->nbytes > f->size) goto bad;
        e->data = f->image + pos;
        pos += e->nbytes;
    }
    return 0;
bad:
    free(f->entries); free(f->image);
    memset(f, 0, sizeof(*f));
    return -6;
}

static inline void vpk_close(vpk_file *f)
{
    free(f->entries);
    free(f->image);
    memset(f, 0, sizeof(*f));
}

static inline const vpk_entry *vpk_find(const vpk_file *f, const char *name)
{
    int nl = (int)strlen(name);
    for (int i = 0; i < f->count; i++)
        if (f->entries[i].name_len == nl && !memcmp(f->entries[i].name, name, (size_t)nl))
            return &f->entries[i];
    return 0;
}

/* typed lookup; returns NULL if missing or of another dtype; *n receives element count */
static inline const void *vpk_get(const vpk_file *f, const char *name, int dtype, size_t *n)
{
    const vpk_entry *e = vpk_find(f, name);
    if (!e || e->dtype != dtype) return 0;
    if (n) *n = vpk_elems(e);
    return e->data;
}

#ifdef __cplusplus
}
#endif
#endif /* VPK_H */
