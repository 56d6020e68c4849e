#!/usr/bin/env python3
"""Build a "mode pack": the complete encoder setup for one (channels, rate, quality) class.

This restates, in Python, what libvorbisenc does once per stream class on the host —
    vorbis_encode_setup_vbr      reference lib/vorbisenc.c:952-975
    get_setup_template           lib/vorbisenc.c:674-713
    vorbis_encode_setup_setting  lib/vorbisenc.c:907-950
    vorbis_encode_setup_init     lib/vorbisenc.c:722-905   (and the helpers at :195-655)
— reading the tuning tables (lib/modes/*.h) and codebooks (lib/books/**.h) as TEXT through
tools/cinit.py.  SURVEY.md marks libvorbisenc itself out of scope as code: only its OUTPUT
(codec_setup_info) is needed by the hot path, and this script emits exactly that as a VPK
pack that both the product and the oracle load.  Runs in the authoring container only.

C float semantics are reproduced explicitly: every value the reference stores into a
`float` field goes through np.float32(), double expressions stay Python floats.

    python tools/make_modepack.py 2 44100 0.5   -> data/mode_2ch_44100_q0.5.vpk
"""
import glob
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cinit  # noqa: E402
from vpk import write_vpk  # noqa: E402

REF = os.environ.get("VORBIS_REFERENCE", "/root/reference")
DATA = os.path.join(HERE, "..", "vorbis_aotuv_lancer_amd", "data")

PACKETBLOBS = 15
P_BANDS = 17
P_NOISECURVES = 3
NOISE_COMPAND_LEVELS = 40
VE_BANDS = 12

# order of setup_list[] in lib/vorbisenc.c:167-193
SETUP_LIST = [
    "ve_setup_44_51", "ve_setup_48_stereo", "ve_setup_44_stereo", "ve_setup_48_uncoupled",
    "ve_setup_44_uncoupled", "ve_setup_32_stereo", "ve_setup_32_uncoupled", "ve_setup_22_stereo",
    "ve_setup_22_uncoupled", "ve_setup_16_stereo", "ve_setup_16_uncoupled", "ve_setup_11_stereo",
    "ve_setup_11_uncoupled", "ve_setup_8_stereo", "ve_setup_8_uncoupled", "ve_setup_X_stereo",
    "ve_setup_X_uncoupled", "ve_setup_XX_stereo", "ve_setup_XX_uncoupled",
]

# field order of ve_setup_data_template, lib/vorbisenc.c:101-152
TEMPLATE_FIELDS = [
    "mappings", "rate_mapping", "quality_mapping", "pre_amp", "coupling_restriction",
    "samplerate_min_restriction", "samplerate_max_restriction", "blocksize_short", "blocksize_long",
    "psy_tone_masteratt", "psy_tone_0dB", "psy_tone_dBsuppress", "psy_tone_adj_impulse",
    "psy_tone_adj_long", "psy_tone_adj_other", "psy_noiseguards", "psy_noise_bias_impulse",
    "psy_noise_bias_padding", "psy_noise_bias_trans", "psy_noise_bias_long", "psy_noise_dBsuppress",
    "psy_noise_compand", "psy_noise_compand_short_mapping", "psy_noise_compand_long_mapping",
    "psy_noise_normal_start", "psy_noise_normal_partition", "psy_noise_normal_thresh",
    "psy_ath_float", "psy_ath_abs", "psy_lowpass", "global_params", "global_mapping", "stereo_modes",
    "floor_books", "floor_params", "floor_mappings", "floor_mapping_list", "maps",
]

_decl_cache = None


def decls():
    global _decl_cache
    if _decl_cache is None:
        d = {}
        files = sorted(glob.glob(os.path.join(REF, "lib", "modes", "*.h"))) + \
            sorted(glob.glob(os.path.join(REF, "lib", "books", "*", "*.h")))
        for f in files:
            d.update(cinit.parse_file(f))
        # the two static conventions of lib/vorbisenc.c:154-162
        d.update(cinit.parse_file(os.path.join(REF, "lib", "vorbisenc.c")))
        _decl_cache = d
    return _decl_cache


def val(name):
    if isinstance(name, str):
        return decls()[name.lstrip("&")].value
    return name


def f32(x):
    return np.float32(x)


def f32lit(x):
    """value of a literal stored into a C float"""
    return cinit.to_f32(x)


def pad(lst, n, fill=0):
    lst = list(lst) if isinstance(lst, (list, tuple)) else [lst]
    assert len(lst) <= n, (lst, n)
    return lst + [fill] * (n - len(lst))


class Template:
    def __init__(self, name):
        self.name = name
        v = val(name)
        assert len(v) == len(TEMPLATE_FIELDS), (name, len(v))
        for k, x in zip(TEMPLATE_FIELDS, v):
            setattr(self, k, x)


def get_setup_template(ch, srate, req, q_or_bitrate=0):
    """lib/vorbisenc.c:674-713.  req: python float (double); q_or_bitrate 1 = req is a bitrate
    (divided by the channel count and looked up in rate_mapping)."""
    if q_or_bitrate:
        req /= ch
    for name in SETUP_LIST:
        if name not in decls():
            continue
        t = Template(name)
        if t.coupling_restriction not in (-1, ch):
            continue
        if not (t.samplerate_min_restriction <= srate <= t.samplerate_max_restriction):
            continue
        mp_ = t.rate_mapping if q_or_bitrate else t.quality_mapping
        if mp_ in ("NULL", "0", 0, None):
            continue
        m = [float(x) for x in val(mp_)]
        mappings = t.mappings
        if req < m[0] or req > m[mappings]:
            continue
        j = 0
        while j < mappings:
            if m[j] <= req < m[j + 1]:
                break
            j += 1
        if j == mappings:
            base = j - .001
        else:
            low = f32(m[j])
            high = f32(m[j + 1])
            # float del=(req-low)/(high-low): (double - float)/(float - float)
            dl = f32((req - float(low)) / float(f32(high - low)))
            base = j + float(dl)
        return t, base
    raise ValueError("no setup template matches")


def lerp_i(tab, is_, ds, field=None):
    """x[is]*(1.-ds)+x[is+1]*ds in double"""
    a, b = tab[is_], tab[is_ + 1]
    return float(a) * (1. - ds) + float(b) * ds


# ---- codebooks -------------------------------------------------------------------
class Books:
    def __init__(self):
        self.names = []   # index -> reference identifier

    def dup_or_new(self, name):
        """book_dup_or_new, lib/vorbisenc.c:474-480 (pointer identity == name identity)"""
        if name in self.names:
            return self.names.index(name)
        self.names.append(name)
        return len(self.names) - 1

    def append(self, name):
        self.names.append(name)
        return len(self.names) - 1


def book_arrays(name):
    """static_codebook: lib/codebook.h:33-55 {dim, entries, lengthlist, maptype, q_min, q_delta,
    q_quant, q_sequencep, quantlist, allocdflag}"""
    v = val(name)
    dim, entries, ll, maptype, q_min, q_delta, q_quant, q_seq = v[:8]
    quantlist = v[8] if len(v) > 8 else 0
    lengths = np.array(val(ll), dtype=np.int8)
    assert lengths.shape == (entries,), (name, lengths.shape, entries)
    if isinstance(quantlist, str) and quantlist not in ("NULL", "0"):
        ql = np.array(val(quantlist), dtype=np.int32)
    else:
        ql = np.zeros(0, np.int32)
    hdr = np.array([dim, entries, maptype, q_min, q_delta, q_quant, q_seq, len(ql)], dtype=np.int64)
    return hdr, lengths, ql


# ---- the setup proper ------------------------------------------------------------
def build(ch, rate, quality, managed=None):
    """managed = (max_bitrate, nominal_bitrate, min_bitrate) selects vorbis_encode_setup_managed
    (lib/vorbisenc.c:997-1047) instead of the VBR setup"""
    out = {}
    if managed is None:
        # vorbis_encode_setup_vbr, lib/vorbisenc.c:952-975: float quality += .0000001 (double add, float store)
        q = f32(float(f32(quality)) + .0000001)
        if q >= 1.:
            q = f32(.9999)
        req = float(q)
        t, base = get_setup_template(ch, rate, req)
    else:
        max_bitrate, nominal_bitrate, min_bitrate = [int(x) for x in managed]
        tnominal = float(nominal_bitrate)
        if nominal_bitrate <= 0:                      # :1013-1027 (long arithmetic: doubles stored into a long)
            if max_bitrate > 0:
                nominal_bitrate = int((max_bitrate + min_bitrate) * .5) if min_bitrate > 0 else int(max_bitrate * .875)
            elif min_bitrate > 0:
                nominal_bitrate = min_bitrate
            else:
                raise ValueError("OV_EINVAL")
        t, base = get_setup_template(ch, rate, float(nominal_bitrate), 1)
    is_ = int(base)
    ds = base - is_

    # ---- vorbis_encode_setup_setting, lib/vorbisenc.c:907-950
    hi = {}
    hi["impulse_block_p"] = 1
    hi["noise_normalize_p"] = 1
    hi["stereo_point_setting"] = base
    hi["lowpass_kHz"] = lerp_i([float(x) for x in val(t.psy_lowpass)], is_, ds)
    pre = [f32lit(x) for x in val(t.pre_amp)]
    hi["pre_amplitude"] = f32(float(pre[is_]) * (1. - ds) + float(pre[is_ + 1]) * ds)
    hi["ath_floating_dB"] = lerp_i(val(t.psy_ath_float), is_, ds)
    hi["ath_absolute_dB"] = lerp_i(val(t.psy_ath_abs), is_, ds)
    hi["amplitude_track_dBpersec"] = -6.
    hi["trigger_setting"] = base
    hi["impulse_noisetune"] = 0.   # vorbis_info_init callocs codec_setup_info
    hi["managed"] = 0
    hi["coupling_p"] = 1
    hi["bitrate_min"] = hi["bitrate_max"] = 0
    hi["bitrate_av"] = 0.
    if managed is not None:                           # :1035-1044
        hi["managed"] = 1
        hi["bitrate_min"] = min_bitrate
        hi["bitrate_max"] = max_bitrate
        hi["bitrate_av"] = int(tnominal)              # long bitrate_av
        hi["bitrate_av_damp"] = 1.5
        hi["bitrate_reservoir"] = nominal_bitrate * 2
        hi["bitrate_reservoir_bias"] = .1

    # ---- vorbis_encode_setup_init, lib/vorbisenc.c:722-905
    i0 = 0 if hi["impulse_block_p"] else 1
    if hi["ath_floating_dB"] > -80:
        hi["ath_floating_dB"] = -80.
    if hi["ath_floating_dB"] < -200:
        hi["ath_floating_dB"] = -200.
    if hi["amplitude_track_dBpersec"] > 0.:
        hi["amplitude_track_dBpersec"] = 0.
    if hi["amplitude_track_dBpersec"] < -99999.:
        hi["amplitude_track_dBpersec"] = -99999.

    blocksizes = [val(t.blocksize_short)[is_], val(t.blocksize_long)[is_]]
    singleblock = blocksizes[0] == blocksizes[1]

    books = Books()
    floors = []
    # vorbis_encode_floor_setup, lib/vorbisenc.c:195-231
    floor_tab = val(t.floor_params)
    floor_books_tab = val(t.floor_books)
    for fm in range(t.floor_mappings):
        x = val(val(t.floor_mapping_list)[fm])
        src = floor_tab[x[is_]]
        (partitions, partitionclass, class_dim, class_subs, class_book, class_subbook,
         mult, postlist, maxover, maxunder, maxerr, twofitweight, twofitatten, n) = src
        f = {
            "partitions": partitions,
            "partitionclass": pad(partitionclass, 31),
            "class_dim": pad(class_dim, 16),
            "class_subs": pad(class_subs, 16),
            "class_book": pad(class_book, 16),
            "class_subbook": [pad(r, 8) for r in pad([list(r) for r in class_subbook], 16, [])],
            "mult": mult,
            "postlist": pad(postlist, 65),
            "maxover": f32lit(maxover), "maxunder": f32lit(maxunder), "maxerr": f32lit(maxerr),
            "twofitweight": f32lit(twofitweight), "twofitatten": f32lit(twofitatten),
            "n": n,
        }
        f["class_subbook"] = [pad(r, 8) for r in f["class_subbook"]]
        maxclass = max([-1] + f["partitionclass"][:partitions])
        maxbook = -1
        nb = len(books.names)
        for i in range(maxclass + 1):
            if f["class_book"][i] > maxbook:
                maxbook = f["class_book"][i]
            f["class_book"][i] += nb
            for k in range(1 << f["class_subs"][i]):
                if f["class_subbook"][i][k] > maxbook:
                    maxbook = f["class_subbook"][i][k]
                if f["class_subbook"][i][k] >= 0:
                    f["class_subbook"][i][k] += nb
        blist = val(floor_books_tab[x[is_]])
        for i in range(maxbook + 1):
            books.append(blist[i].lstrip("&"))   # floor books are never de-duplicated (:226-227)
        floors.append(f)

    # vorbis_encode_global_psych_setup, lib/vorbisenc.c:233-255  (s = trigger_setting)
    s = hi["trigger_setting"]
    gis = int(s)
    gds = s - gis
    gx = [float(v) for v in val(t.global_mapping)]
    gtab = val(t.global_params)
    gsrc = gtab[int(gx[gis])]
    # vorbis_info_psy_global, lib/psy.h:67-86
    (eighth, pre_t, post_t, stretch_pen, pre_min, ampmax_att, c_pkHz, c_plimit, c_pre, c_post, slide) = gsrc
    g = {
        "eighth_octave_lines": eighth,
        "preecho_thresh": [f32lit(v) for v in pad(pre_t, VE_BANDS, 0.)],
        "postecho_thresh": [f32lit(v) for v in pad(post_t, VE_BANDS, 0.)],
        "stretch_penalty": f32lit(stretch_pen),
        "preecho_minenergy": f32lit(pre_min),
        "ampmax_att_per_sec": f32lit(ampmax_att),
        "coupling_pkHz": [int(v) for v in pad(c_pkHz, PACKETBLOBS)],
        "coupling_pointlimit": [[int(v) for v in pad(r, PACKETBLOBS)] for r in c_plimit],
        "coupling_prepointamp": [int(v) for v in pad(c_pre, PACKETBLOBS)],
        "coupling_postpointamp": [int(v) for v in pad(c_post, PACKETBLOBS)],
        "sliding_lowpass": [[int(v) for v in pad(r, PACKETBLOBS)] for r in slide],
    }
    gds2 = gx[gis] * (1. - gds) + gx[gis + 1] * gds
    gis2 = int(gds2)
    gds2 -= gis2
    if gds2 == 0 and gis2 > 0:
        gis2 -= 1
        gds2 = 1.
    for i in range(4):
        a = [f32lit(v) for v in pad(gtab[gis2][1], VE_BANDS, 0.)]
        b = [f32lit(v) for v in pad(gtab[gis2 + 1][1], VE_BANDS, 0.)]
        g["preecho_thresh"][i] = f32(float(a[i]) * (1. - gds2) + float(b[i]) * gds2)
        a = [f32lit(v) for v in pad(gtab[gis2][2], VE_BANDS, 0.)]
        b = [f32lit(v) for v in pad(gtab[gis2 + 1][2], VE_BANDS, 0.)]
        g["postecho_thresh"][i] = f32(float(a[i]) * (1. - gds2) + float(b[i]) * gds2)
    g["ampmax_att_per_sec"] = f32(hi["amplitude_track_dBpersec"])

    # vorbis_encode_global_stereo, lib/vorbisenc.c:257-307
    fs = f32(hi["stereo_point_setting"])
    sis = int(fs)
    sds = float(fs) - sis
    stereo = val(t.stereo_modes) if isinstance(t.stereo_modes, str) else None
    if stereo:
        p0, p1 = stereo[sis], stereo[sis + 1]
        g["coupling_prepointamp"] = [int(v) for v in pad(p0[0], PACKETBLOBS)]
        g["coupling_postpointamp"] = [int(v) for v in pad(p0[1], PACKETBLOBS)]
        mid = PACKETBLOBS // 2
        k0 = [f32lit(v) for v in pad(p0[2], PACKETBLOBS, 0.)]
        k1 = [f32lit(v) for v in pad(p1[2], PACKETBLOBS, 0.)]
        l0 = [f32lit(v) for v in pad(p0[3], PACKETBLOBS, 0.)]
        l1 = [f32lit(v) for v in pad(p1[3], PACKETBLOBS, 0.)]
        for i in range(PACKETBLOBS):
            src = i if hi["managed"] else mid         # managed: every blob its own thresholds (:270-283)
            kHz = f32(float(k0[src]) * (1. - sds) + float(k1[src]) * sds)
            g["coupling_pointlimit"][0][i] = int(float(kHz) * 1000. / rate * blocksizes[0])
            g["coupling_pointlimit"][1][i] = int(float(kHz) * 1000. / rate * blocksizes[1])
            g["coupling_pkHz"][i] = int(kHz)
            kHz = f32(float(l0[src]) * (1. - sds) + float(l1[src]) * sds)
            g["sliding_lowpass"][0][i] = int(float(kHz) * 1000. / rate * blocksizes[0])
            g["sliding_lowpass"][1][i] = int(float(kHz) * 1000. / rate * blocksizes[1])
    else:
        for i in range(PACKETBLOBS):
            g["sliding_lowpass"][0][i] = blocksizes[0]
            g["sliding_lowpass"][1][i] = blocksizes[1]

    # psy templates, lib/vorbisenc.c:309-472
    tmpl = val("_psy_info_template")
    (blockflag, ath_adjatt, ath_maxatt, tone_masteratt, tone_centerboost, tone_decay, tone_abs_limit,
     toneatt, noisemaskp, noisemaxsupp, noisewindowlo, noisewindowhi, noisewindowlomin,
     noisewindowhimin, noisewindowfixed, noiseoff, noisecompand, noisecompand_high, flacint,
     max_curve_dB, normal_p, normal_start, normal_partition, normal_thresh) = tmpl

    def new_psy():
        return {
            "blockflag": blockflag,
            "ath_adjatt": f32lit(ath_adjatt), "ath_maxatt": f32lit(ath_maxatt),
            "tone_masteratt": [f32lit(v) for v in pad(tone_masteratt, P_NOISECURVES, 0.)],
            "tone_centerboost": f32lit(tone_centerboost), "tone_decay": f32lit(tone_decay),
            "tone_abs_limit": f32lit(tone_abs_limit),
            "toneatt": [f32lit(v) for v in pad(toneatt, P_BANDS, 0.)],
            "noisemaskp": noisemaskp, "noisemaxsupp": f32lit(noisemaxsupp),
            "noisewindowlo": f32lit(noisewindowlo), "noisewindowhi": f32lit(noisewindowhi),
            "noisewindowlomin": noisewindowlomin, "noisewindowhimin": noisewindowhimin,
            "noisewindowfixed": noisewindowfixed,
            "noiseoff": [[f32lit(v) for v in pad(r, P_BANDS, 0.)] for r in noiseoff],
            "noisecompand": [f32lit(v) for v in pad(noisecompand, NOISE_COMPAND_LEVELS, 0.)],
            "noisecompand_high": [f32lit(v) for v in pad(noisecompand_high, NOISE_COMPAND_LEVELS, 0.)],
            "flacint": f32lit(flacint), "max_curve_dB": f32lit(max_curve_dB),
            "normal_p": normal_p, "normal_start": normal_start, "normal_partition": normal_partition,
            "normal_thresh": float(normal_thresh),
        }

    npsy = 2 if singleblock else 4
    psys = [None] * npsy

    def psyset(block, which):   # vorbis_encode_psyset_setup :309-337
        p = new_psy()
        p["blockflag"] = block >> 1
        if hi["noise_normalize_p"]:
            p["normal_p"] = 1
            p["normal_start"] = val(t.psy_noise_normal_start[which])[is_]
            p["normal_partition"] = val(t.psy_noise_normal_partition[which])[is_]
            p["normal_thresh"] = float(val(t.psy_noise_normal_thresh)[is_])
        psys[block] = p

    psyset(0, 0)
    psyset(1, 0)
    if not singleblock:
        psyset(2, 1)
        psyset(3, 1)

    def tonemask(block, adj):   # vorbis_encode_tonemask_setup :339-362 (s == base for every block)
        p = psys[block]
        att = val(t.psy_tone_masteratt)
        a, b = att[is_], att[is_ + 1]
        for k in range(3):
            p["tone_masteratt"][k] = f32(float(a[0][k]) * (1. - ds) + float(b[0][k]) * ds)
        p["tone_centerboost"] = f32(float(f32lit(a[1])) * (1. - ds) + float(f32lit(b[1])) * ds)
        p["tone_decay"] = f32(float(f32lit(a[2])) * (1. - ds) + float(f32lit(b[2])) * ds)
        mx = val(t.psy_tone_0dB)
        p["max_curve_dB"] = f32(float(mx[is_]) * (1. - ds) + float(mx[is_ + 1]) * ds)
        tab = val(adj)
        ra = pad(tab[is_][0] if isinstance(tab[is_][0], list) else tab[is_], P_BANDS)
        rb = pad(tab[is_ + 1][0] if isinstance(tab[is_ + 1][0], list) else tab[is_ + 1], P_BANDS)
        for i in range(P_BANDS):
            p["toneatt"][i] = f32(float(ra[i]) * (1. - ds) + float(rb[i]) * ds)

    tonemask(0, t.psy_tone_adj_impulse)
    tonemask(1, t.psy_tone_adj_other)
    if not singleblock:
        tonemask(2, t.psy_tone_adj_other)
        tonemask(3, t.psy_tone_adj_long)

    def compand(block, xmap):   # vorbis_encode_compand_setup :365-422
        p = psys[block]
        x = [float(v) for v in val(xmap)]
        tab = val(t.psy_noise_compand)
        cis, cds = is_, ds
        hcm_stop = min(5, t.mappings)
        p["flacint"] = f32(cds)
        cds = x[cis] * (1. - cds) + x[cis + 1] * cds
        cis = int(cds)
        cds -= cis
        if cds == 0 and cis > 0:
            cis -= 1
            cds = 1.
        ishcm = cis
        dshcm = cds + .3
        if dshcm > 1.0:
            ishcm += 1
            dshcm = dshcm - 1
        if x[hcm_stop] < (float(ishcm) + dshcm):
            ishcm = int(x[hcm_stop])
            dshcm = x[hcm_stop] - ishcm
            if (float(ishcm) + dshcm) < (float(cis) + cds):
                ishcm = cis
                dshcm = cds
        if dshcm == 0 and ishcm > 0:
            ishcm -= 1
            dshcm = 1.

        def row(r):
            r = tab[r]
            r = r[0] if (len(r) == 1 and isinstance(r[0], list)) else r
            return pad(r, NOISE_COMPAND_LEVELS)
        ra, rb = row(cis), row(cis + 1)
        for i in range(NOISE_COMPAND_LEVELS):
            p["noisecompand"][i] = f32(float(ra[i]) * (1. - cds) + float(rb[i]) * cds)
        ra, rb = row(ishcm), row(ishcm + 1)
        for i in range(NOISE_COMPAND_LEVELS):
            p["noisecompand_high"][i] = f32(float(ra[i]) * (1. - dshcm) + float(rb[i]) * dshcm)

    compand(0, t.psy_noise_compand_short_mapping)
    compand(1, t.psy_noise_compand_short_mapping)
    if not singleblock:
        compand(2, t.psy_noise_compand_long_mapping)
        compand(3, t.psy_noise_compand_long_mapping)

    sup = val(t.psy_tone_dBsuppress)
    for b in range(npsy):   # vorbis_encode_peak_setup :424-435
        psys[b]["tone_abs_limit"] = f32(float(sup[is_]) * (1. - ds) + float(sup[is_ + 1]) * ds)

    def noisebias(block, tabname, userbias):   # vorbis_encode_noisebias_setup :437-470
        p = psys[block]
        nsup = val(t.psy_noise_dBsuppress)
        p["noisemaxsupp"] = f32(float(nsup[is_]) * (1. - ds) + float(nsup[is_ + 1]) * ds)
        guard = val(t.psy_noiseguards)
        p["noisewindowlomin"], p["noisewindowhimin"], p["noisewindowfixed"] = guard[block]
        tab = val(tabname)

        def rows(r):
            r = tab[r]
            r = r[0] if (len(r) == 1 and isinstance(r[0][0], list)) else r
            return [pad(x, 17) for x in r]
        ra, rb = rows(is_), rows(is_ + 1)
        for j in range(P_NOISECURVES):
            for i in range(P_BANDS):
                p["noiseoff"][j][i] = f32(float(ra[j][i]) * (1. - ds) + float(rb[j][i]) * ds)
        for j in range(P_NOISECURVES):
            mn = f32(float(p["noiseoff"][j][0]) + 6)   # float min = noiseoff + 6 (int) -> float add
            for i in range(P_BANDS):
                p["noiseoff"][j][i] = f32(float(p["noiseoff"][j][i]) + userbias)
                if p["noiseoff"][j][i] < mn:
                    p["noiseoff"][j][i] = mn

    noisebias(0, t.psy_noise_bias_impulse, hi["impulse_noisetune"] if i0 == 0 else 0.)
    noisebias(1, t.psy_noise_bias_padding, 0.)
    if not singleblock:
        noisebias(2, t.psy_noise_bias_trans, 0.)
        noisebias(3, t.psy_noise_bias_long, 0.)

    for b in range(npsy):   # vorbis_encode_ath_setup :472-479
        psys[b]["ath_adjatt"] = f32(hi["ath_floating_dB"])
        psys[b]["ath_maxatt"] = f32(hi["ath_absolute_dB"])

    # vorbis_encode_map_n_res_setup :627-655 and vorbis_encode_residue_setup :493-625
    mt = val(t.maps)[is_]
    map_tab = val(mt[0])
    res_tab = val(mt[1])
    modes_tab = val("_mode_template")
    nmodes = 1 if singleblock else 2
    maps, modes = [], []
    residues = {}
    block_lowpassr = [0, 0]
    for i in range(nmodes):
        modes.append(list(modes_tab[i]))
        (submaps, chmux, floorsub, ressub, csteps, cmag, cang) = (map_tab[i] + [0, [0], [0]])[:7] \
            if len(map_tab[i]) >= 4 else None
        m = {"submaps": submaps, "chmuxlist": pad(chmux, 256)[:max(ch, 1)],
             "floorsubmap": pad(floorsub, 16), "residuesubmap": pad(ressub, 16),
             "coupling_steps": csteps, "coupling_mag": pad(cmag, 256)[:max(csteps, 1)],
             "coupling_ang": pad(cang, 256)[:max(csteps, 1)]}
        maps.append(m)
    # residue setup runs inside the mode loop AFTER map_param[i] was stored; the channel count
    # search (:588-597) walks ci->maps, which at that point holds maps 0..i
    for i in range(nmodes):
        m = maps[i]
        for j in range(m["submaps"]):
            number = m["residuesubmap"][j]
            rt = res_tab[number]
            res_type, limit_type, grouping, resname, book_aux, book_aux_m, books_base, books_base_m = rt
            rsrc = val(resname)
            begin, end, grp, partitions, partvals, groupbook = rsrc[:6]
            r = {"type": res_type, "begin": begin, "end": end, "grouping": grouping,
                 "partitions": partitions, "partvals": partvals, "groupbook": groupbook,
                 "secondstages": pad(rsrc[6], 64), "booklist": pad(rsrc[7], 512),
                 "classmetric1": pad(rsrc[8], 64), "classmetric2": pad(rsrc[9], 64)}
            if hi["managed"]:                        # :513-530
                books_base, book_aux = books_base_m, book_aux_m
            bb = val(books_base)[0]
            bb = [pad(row if isinstance(row, list) else [row], 4) for row in bb]
            bb = bb + [[0, 0, 0, 0]] * (12 - len(bb))
            for pi in range(partitions):
                for k in range(4):
                    if bb[pi][k]:
                        r["secondstages"][pi] |= (1 << k)
            r["groupbook"] = books.dup_or_new(book_aux.lstrip("&"))
            bl = 0
            for pi in range(partitions):
                for k in range(4):
                    if bb[pi][k]:
                        r["booklist"][bl] = books.dup_or_new(bb[pi][k].lstrip("&"))
                        bl += 1
            # lowpass / pointlimit, :548-623
            block = i
            freq = hi["lowpass_kHz"] * 1000.
            f = floors[block]
            nyq = rate / 2.
            blocksize = blocksizes[block] >> 1
            if freq > nyq:
                freq = nyq
            f["n"] = int(freq / nyq * blocksize)
            if limit_type == 1:
                freq = g["coupling_pkHz"][PACKETBLOBS - 1 if hi["managed"] else PACKETBLOBS // 2] * 1000.
                if freq > nyq:
                    freq = nyq
            elif limit_type == 2:
                freq = 250.
            if res_type == 2:
                chs = 0
                for mi in maps[:i + 1]:
                    if chs:
                        break
                    for jj in range(mi["submaps"]):
                        if chs:
                            break
                        if mi["residuesubmap"][jj] == number:
                            for k in range(ch):
                                if mi["chmuxlist"][k] == jj:
                                    chs += 1
                r["end"] = int((freq / nyq * blocksize * chs) / r["grouping"] + .9) * r["grouping"]
                if r["end"] > blocksize * chs:
                    r["end"] = blocksize * chs // r["grouping"] * r["grouping"]
                if freq != 250.:
                    block_lowpassr[block] = r["end"] // chs
            else:
                r["end"] = int((freq / nyq * blocksize) / r["grouping"] + .9) * r["grouping"]
                if r["end"] > blocksize:
                    r["end"] = blocksize // r["grouping"] * r["grouping"]
                if freq != 250.:
                    block_lowpassr[block] = r["end"]
            if r["end"] == 0:
                r["end"] = r["grouping"]
            residues[number] = r
    nres = max(residues) + 1

    # ---- emit ---------------------------------------------------------------------
    I32 = np.int32
    F32 = np.float32
    out["info/channels"] = np.array([ch], I32)
    out["info/rate"] = np.array([rate], np.int64)
    out["info/quality"] = np.array([float(f32(quality))], np.float64)
    out["info/base_setting"] = np.array([base], np.float64)
    # vi->bitrate_upper / _nominal / _lower as the identification header carries them (lib/info.c:518-520):
    # VBR leaves hi->bitrate_max/_min at 0 (lib/vorbisenc.c:883-884); nominal = setting_to_approx_bitrate
    # (lib/vorbisenc.c:659-672), a double stored into a long
    rm = t.rate_mapping
    if rm in ("NULL", "0", 0, None):
        nominal = -1
    else:
        r = [float(x) for x in val(rm)]
        nominal = int((r[is_] * (1. - ds) + r[is_ + 1] * ds) * ch)
    if hi["bitrate_av"] > 0:                          # :877-884
        nominal = int(hi["bitrate_av"])
    out["info/bitrates"] = np.array([hi["bitrate_max"], nominal, hi["bitrate_min"]], np.int64)
    if hi["managed"]:                                 # bitrate_manager_info, :890-901
        out["bi/rates"] = np.array([hi["bitrate_av"], hi["bitrate_min"], hi["bitrate_max"],
                                    hi["bitrate_reservoir"]], np.int64)
        out["bi/floats"] = np.array([hi["bitrate_reservoir_bias"], hi["bitrate_av_damp"]], np.float64)
    out["info/template"] = np.frombuffer(t.name.encode(), dtype=np.uint8)
    out["info/blocksizes"] = np.array(blocksizes, I32)
    out["info/counts"] = np.array([nmodes, nmodes, len(floors), nres, len(books.names), npsy], I32)
    out["info/block_lowpassr"] = np.array(block_lowpassr, I32)
    out["info/managed"] = np.array([hi["managed"]], I32)
    out["hi/pre_amplitude"] = np.array([hi["pre_amplitude"]], F32)
    out["hi/lowpass_kHz"] = np.array([hi["lowpass_kHz"]], np.float64)
    for i, m in enumerate(modes):
        out[f"mode/{i}"] = np.array(m, I32)
    for i, m in enumerate(maps):
        out[f"map/{i}/submaps"] = np.array([m["submaps"]], I32)
        out[f"map/{i}/chmuxlist"] = np.array(m["chmuxlist"], I32)
        out[f"map/{i}/floorsubmap"] = np.array(m["floorsubmap"], I32)
        out[f"map/{i}/residuesubmap"] = np.array(m["residuesubmap"], I32)
        out[f"map/{i}/coupling_steps"] = np.array([m["coupling_steps"]], I32)
        out[f"map/{i}/coupling_mag"] = np.array(m["coupling_mag"], I32)
        out[f"map/{i}/coupling_ang"] = np.array(m["coupling_ang"], I32)
    for i, f in enumerate(floors):
        out[f"floor/{i}/partitions"] = np.array([f["partitions"]], I32)
        out[f"floor/{i}/partitionclass"] = np.array(f["partitionclass"], I32)
        out[f"floor/{i}/class_dim"] = np.array(f["class_dim"], I32)
        out[f"floor/{i}/class_subs"] = np.array(f["class_subs"], I32)
        out[f"floor/{i}/class_book"] = np.array(f["class_book"], I32)
        out[f"floor/{i}/class_subbook"] = np.array(f["class_subbook"], I32)
        out[f"floor/{i}/mult"] = np.array([f["mult"]], I32)
        out[f"floor/{i}/postlist"] = np.array(f["postlist"], I32)
        out[f"floor/{i}/fit"] = np.array([f["maxover"], f["maxunder"], f["maxerr"],
                                          f["twofitweight"], f["twofitatten"]], F32)
        out[f"floor/{i}/n"] = np.array([f["n"]], I32)
    for i in range(nres):
        r = residues[i]
        out[f"residue/{i}/head"] = np.array([r["type"], r["begin"], r["end"], r["grouping"],
                                             r["partitions"], r["partvals"], r["groupbook"]], I32)
        out[f"residue/{i}/secondstages"] = np.array(r["secondstages"], I32)
        out[f"residue/{i}/booklist"] = np.array(r["booklist"], I32)
        out[f"residue/{i}/classmetric1"] = np.array(r["classmetric1"], I32)
        out[f"residue/{i}/classmetric2"] = np.array(r["classmetric2"], I32)
    for i, name in enumerate(books.names):
        hdr, lengths, ql = book_arrays(name)
        out[f"book/{i}/head"] = hdr
        out[f"book/{i}/lengthlist"] = lengths
        out[f"book/{i}/quantlist"] = ql
        out[f"book/{i}/name"] = np.frombuffer(name.encode(), dtype=np.uint8)
    for i, p in enumerate(psys):
        out[f"psy/{i}/ints"] = np.array([p["blockflag"], p["noisemaskp"], p["noisewindowlomin"],
                                         p["noisewindowhimin"], p["noisewindowfixed"], p["normal_p"],
                                         p["normal_start"], p["normal_partition"]], I32)
        out[f"psy/{i}/floats"] = np.array([p["ath_adjatt"], p["ath_maxatt"], p["tone_centerboost"],
                                           p["tone_decay"], p["tone_abs_limit"], p["noisemaxsupp"],
                                           p["noisewindowlo"], p["noisewindowhi"], p["flacint"],
                                           p["max_curve_dB"]], F32)
        out[f"psy/{i}/tone_masteratt"] = np.array(p["tone_masteratt"], F32)
        out[f"psy/{i}/toneatt"] = np.array(p["toneatt"], F32)
        out[f"psy/{i}/noiseoff"] = np.array(p["noiseoff"], F32)
        out[f"psy/{i}/noisecompand"] = np.array(p["noisecompand"], F32)
        out[f"psy/{i}/noisecompand_high"] = np.array(p["noisecompand_high"], F32)
        out[f"psy/{i}/normal_thresh"] = np.array([p["normal_thresh"]], np.float64)
    out["psy_g/eighth_octave_lines"] = np.array([g["eighth_octave_lines"]], I32)
    out["psy_g/preecho_thresh"] = np.array(g["preecho_thresh"], F32)
    out["psy_g/postecho_thresh"] = np.array(g["postecho_thresh"], F32)
    out["psy_g/floats"] = np.array([g["stretch_penalty"], g["preecho_minenergy"], g["ampmax_att_per_sec"]], F32)
    out["psy_g/coupling_pkHz"] = np.array(g["coupling_pkHz"], I32)
    out["psy_g/coupling_pointlimit"] = np.array(g["coupling_pointlimit"], I32)
    out["psy_g/coupling_prepointamp"] = np.array(g["coupling_prepointamp"], I32)
    out["psy_g/coupling_postpointamp"] = np.array(g["coupling_postpointamp"], I32)
    out["psy_g/sliding_lowpass"] = np.array(g["sliding_lowpass"], I32)
    return out


def pack_name(ch, rate, quality, managed=None):
    if managed is not None:
        mx, nom, mn = managed
        return f"mode_{ch}ch_{rate}_b{nom}" + (f"_max{mx}" if mx > 0 else "") + (f"_min{mn}" if mn > 0 else "") + ".vpk"
    return f"mode_{ch}ch_{rate}_q{quality:g}.vpk"


if __name__ == "__main__":
    # make_modepack.py ch rate quality            (vorbis_encode_init_vbr)
    # make_modepack.py ch rate b<nominal> [max [min]]   (vorbis_encode_init: managed bitrate, bits/s)
    ch, rate = int(sys.argv[1]), int(sys.argv[2])
    managed, quality = None, None
    if sys.argv[3].startswith("b"):
        managed = (int(sys.argv[4]) if len(sys.argv) > 4 else -1, int(sys.argv[3][1:]),
                   int(sys.argv[5]) if len(sys.argv) > 5 else -1)
    else:
        quality = float(sys.argv[3])
    arrays = build(ch, rate, quality, managed)
    os.makedirs(DATA, exist_ok=True)
    path = os.path.join(DATA, pack_name(ch, rate, quality, managed))
    write_vpk(path, arrays)
    c = arrays["info/counts"]
    print(f"{path}: template={bytes(arrays['info/template']).decode()} base={arrays['info/base_setting'][0]:.7f} "
          f"blocksizes={arrays['info/blocksizes']} modes/maps/floors/residues/books/psys={list(c)} "
          f"lowpassr={arrays['info/block_lowpassr']} size={os.path.getsize(path)}")
