// C ABI: encoder setup objects and table introspection (include/vorbis_mi355x.h)
#include <string.h>
#include <string>
#include "vorbis_mi355x.h"
#include "setup_host.h"
#include "vbm_internal.h"

struct vbm_setup_handle {
    vbm_setup_host *H;
    std::string mode_path;   // the header packets are packed from the mode pack itself (capi_stream.cpp)
};

extern "C" int vbm_setup_create(vbm_setup_handle **out, const char *common_vpk, const char *mode_vpk)
{
    if (!out || !common_vpk || !mode_vpk) return VBM_EINVAL;
    *out = nullptr;
    std::string err;
    vbm_setup_host *H = vbm_setup_host_load(common_vpk, mode_vpk, err);
    if (!H) {
        g_vbm_err = err;
        return VBM_EFAULT;
    }
    *out = new vbm_setup_handle{H, mode_vpk};
    return VBM_OK;
}

extern "C" void vbm_setup_destroy(vbm_setup_handle *h)
{
    if (!h) return;
    vbm_setup_host_free(h->H);
    delete h;
}

vbm_setup_host *vbm_setup_handle_host(vbm_setup_handle *h) { return h ? h->H : nullptr; }
const char *vbm_setup_handle_mode_path(const vbm_setup_handle *h) { return h ? h->mode_path.c_str() : ""; }

// name -> (pointer, element count, element kind 'f' float / 'i' int32 / 'd' double / 'b' int8 / 'u' uint32)
extern "C" int vbm_setup_table(const vbm_setup_handle *h, const char *name, const void **data, long *count,
                               char *kind)
{
    if (!h || !name || !data || !count || !kind) return VBM_EINVAL;
    const vbm_setup *s = vbm_setup_host_view(h->H);
    int idx = -1;
    char leaf[64] = {0};
#define RET(ptr, n, k) do { *data = (ptr); *count = (long)(n); *kind = (k); return VBM_OK; } while (0)
    if (sscanf(name, "psy/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->psys) {
        const vbm_psy &p = s->psy[idx];
        if (!strcmp(leaf, "ath")) RET(p.ath, p.n, 'f');
        if (!strcmp(leaf, "octave")) RET(p.octave, p.n, 'i');
        if (!strcmp(leaf, "bark_lo")) RET(p.bark_lo, p.n, 'i');
        if (!strcmp(leaf, "bark_hi")) RET(p.bark_hi, p.n, 'i');
        if (!strcmp(leaf, "tonecurves")) RET(p.tonecurves, VBM_P_BANDS * VBM_P_LEVELS * (VBM_EHMER_MAX + 2), 'f');
        if (!strcmp(leaf, "noiseoffset0")) RET(p.noiseoffset[0], p.n, 'f');
        if (!strcmp(leaf, "noiseoffset1")) RET(p.noiseoffset[1], p.n, 'f');
        if (!strcmp(leaf, "noiseoffset2")) RET(p.noiseoffset[2], p.n, 'f');
        if (!strcmp(leaf, "ntfix_noiseoffset")) RET(p.ntfix_noiseoffset, p.n, 'f');
        if (!strcmp(leaf, "scalars")) {
            static thread_local int sc[16];
            int v[16] = {p.n, p.firstoc, p.shiftoc, p.eighth_octave_lines, p.total_octave_lines, p.m3n[0], p.m3n[1],
                         p.m3n[2], p.tonecomp_endp, p.min_nn_lp, p.tonefix_end, p.n25p, p.n33p, p.n75p, 0, 0};
            memcpy(sc, v, sizeof(sc));
            union { float f; int i; } u;
            u.f = p.m_val; sc[14] = u.i;
            u.f = p.tonecomp_thres; sc[15] = u.i;
            RET(sc, 16, 'i');
        }
    }
    if (sscanf(name, "floor/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->floors) {
        const vbm_floor &f = s->floor[idx];
        if (!strcmp(leaf, "sorted_index")) RET(f.sorted_index, f.posts, 'i');
        if (!strcmp(leaf, "forward_index")) RET(f.forward_index, f.posts, 'i');
        if (!strcmp(leaf, "reverse_index")) RET(f.reverse_index, f.posts, 'i');
        if (!strcmp(leaf, "loneighbor")) RET(f.loneighbor, f.posts - 2, 'i');
        if (!strcmp(leaf, "hineighbor")) RET(f.hineighbor, f.posts - 2, 'i');
        if (!strcmp(leaf, "scalars")) {
            static thread_local int sc[4];
            sc[0] = f.posts; sc[1] = f.n; sc[2] = f.quant_q; sc[3] = f.info_n;
            RET(sc, 4, 'i');
        }
    }
    if (sscanf(name, "book/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->books) {
        const vbm_book &b = s->book[idx];
        if (!strcmp(leaf, "codelist")) RET(b.codelist, b.entries, 'u');
        if (!strcmp(leaf, "lengthlist")) RET(b.lengthlist, b.entries, 'b');
        if (!strcmp(leaf, "used_index")) RET(b.used_index, b.used, 'i');
        if (!strcmp(leaf, "used_point")) RET(b.used_point, b.used * b.dim, 'i');
        if (!strcmp(leaf, "scalars")) {
            static thread_local int sc[6];
            sc[0] = b.dim; sc[1] = b.entries; sc[2] = b.quantvals; sc[3] = b.minval; sc[4] = b.delta; sc[5] = b.used;
            RET(sc, 6, 'i');
        }
    }
    if (sscanf(name, "residue/%d/%63s", &idx, leaf) == 2 && idx >= 0 && idx < s->residues) {
        const vbm_residue &r = s->residue[idx];
        if (!strcmp(leaf, "partbook")) RET(&r.partbook[0][0], 64 * 8, 'i');
        if (!strcmp(leaf, "scalars")) {
            static thread_local int sc[8];
            sc[0] = r.type; sc[1] = r.begin; sc[2] = r.end; sc[3] = r.grouping; sc[4] = r.partitions;
            sc[5] = r.groupbook; sc[6] = r.stages; sc[7] = r.phrase_dim;
            RET(sc, 8, 'i');
        }
    }
    if (!strcmp(name, "info")) {
        static thread_local int sc[12];
        int v[12] = {s->channels, (int)s->rate, s->blocksizes[0], s->blocksizes[1], s->modes, s->maps, s->floors,
                     s->residues, s->books, s->psys, s->block_lowpassr[0], s->block_lowpassr[1]};
        memcpy(sc, v, sizeof(sc));
        RET(sc, 12, 'i');
    }
    if (!strcmp(name, "bitrate")) {   // managed flag + bitrate_manager_info: rates (nominal, lower, upper), reservoir, bias, damping
        static thread_local double sc[7];
        sc[0] = s->managed; sc[1] = (double)s->bi_avg_rate; sc[2] = (double)s->bi_min_rate; sc[3] = (double)s->bi_max_rate;
        sc[4] = (double)s->bi_reservoir_bits; sc[5] = s->bi_reservoir_bias; sc[6] = s->bi_slew_damp;
        RET(sc, 7, 'd');
    }
    if (!strcmp(name, "lowpass_kHz")) {
        static thread_local double lp[1];
        lp[0] = s->hi_lowpass_khz;
        RET(lp, 1, 'd');
    }
#undef RET
    g_vbm_err = std::string("unknown setup table: ") + name;
    return VBM_EINVAL;
}
