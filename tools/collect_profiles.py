#!/usr/bin/env python3
"""Turn gpurun_out/final/ (tools/gpu_profile.sh) into the committed artifacts:
   profiles/<round>/bench_default.json, kernel_stats.csv, pmc_hbm_traffic.csv and profiles/pmc_traffic.json
   (per-stage HBM bytes per launch that bench.py reports as roofline.traffic)."""
import csv, json, os, re, sqlite3, subprocess, sys, collections, statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "final")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
# second argument: where to write (default: the repository's profiles/); the GPU box writes under gpurun_out/ and
# drops the databases, which are too large to travel back
out_root = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
dst = os.path.join(out_root, rnd)
os.makedirs(dst, exist_ok=True)

STAGE_OF = [("k_window_mdct", "window_mdct"), ("k_window_fft_log", "window_fft_log"), ("k_to_tiled", "transpose"),
            ("k_prologue", "prologue"), ("k_nm_", "noisemask"), ("k_noisemask", "noisemask"), ("k_tm_", "tonemask"), ("k_tonemask", "tonemask"), ("k_zero_u128", "pack"), ("k_mix", "offset_and_mix"),
            ("k_floor_prep", "floor_fit"), ("k_floor_fit", "floor_fit"), ("k_floor_interp", "floor_fit"),
            ("k_floor_encode", "floor_encode"), ("k_floor_render", "floor_encode"), ("k_block_state", "offset_and_mix"),
            ("k_nonzero_propagate", "pack"), ("k_bitrate_choose", "pack"), ("k_blob_gather", "packet_out"),
            ("k_couple_", "couple_quantize"), ("k_pack_head", "pack"), ("k_pack_fused", "pack"), ("k_rows_out", "packet_out"),
            ("k_res_", "pack"), ("k_from_tiled", "packet_out"), ("k_fe_", "front_end"), ("k_spread_flags", "prologue"),
            ("k_copy_counted", "packet_out")]

def short(name):
    name = re.sub(r"\(.*\)", "", name)
    return re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)

def stage_of(k):
    for pre, st in STAGE_OF:
        if k.startswith(pre): return st
    return None

def open_db(sub):
    d = os.path.join(src, sub)
    f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith(".db")][0]
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    return db, (lambda p: [t for t in tabs if t.startswith(p)][0])

# 1. bench line
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, "bench_default.json"), "w"), indent=1)

# 2. kernel stats of the same command
db, tab = open_db("stats")
kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
                  f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "stage", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "percent"])
    for name, c, a, mn, mx, t in rows:
        k = short(name)
        w.writerow([k, stage_of(k) or "", c, round(a), mn, mx, t, round(100 * t / tot, 2)])

def kernel_table(sub, path):
    db, tab = open_db(sub)
    kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "stage", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "percent"])
        for name, c, a, mn, mx, t in rows:
            k = short(name)
            if not k.startswith("k_") and "k_zero" not in k: continue
            w.writerow([k, stage_of(k.replace("_ZL11", "")) or ("pack" if "k_zero" in k else ""), c, round(a), mn, mx, t, round(100 * t / tot, 2)])

# 2b. every stage of the per-block path alone (bench.py --only solo): full-size launches only
if os.path.isdir(os.path.join(src, "solo")):
    kernel_table("solo", os.path.join(dst, "solo_kernel_stats.csv"))

def work_table(sub, path, per_label):
    db, tab = open_db(sub)
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    q = (f"select s.kernel_name, i.name, sum(x.v), count(*) from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pe} e group by e.event_id, e.pmc_id) x "
         f"join {kd} d on d.event_id = x.eid join {ks} s on d.kernel_id = s.id join {pi} i on i.id = x.pid group by s.kernel_name, i.name")
    out = collections.defaultdict(dict); calls = {}
    for name, ctr, v, c in db.execute(q):
        k = short(name)
        if not (k.startswith("k_") or "k_zero" in k): continue
        out[k][ctr] = v; calls[k] = c
    writes = max([c for k, c in calls.items() if k.startswith("k_fe_append")] or [0])
    div = writes if writes else max(calls.get(k, 0) for k in calls if k.startswith("k_noisemaskILi2"))
    tot = sum(d.get("SQ_INSTS_VALU", 0) for d in out.values())
    with open(path, "w") as f:
        f.write(f"# wave-instructions issued, summed over all launches of the run and divided by {div} {per_label}; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES\n")
        f.write(f"# issue bound: {tot / div / 1e6:.0f} M VALU instructions per {per_label[:-1]} / (1024 SIMDs x 2.4 GHz / 4 cycles = 614 G/s) = {tot / div / 614e9 * 1e3:.2f} ms\n")
        f.write(f"{'kernel':44s} {'calls':>6s} {'VALU_M':>9s} {'%':>6s} {'SALU_M':>9s} {'LDS_M':>8s} {'wave_Mcyc':>10s}\n")
        for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
            f.write(f"{k[:44]:44s} {calls[k]:6d} {d.get('SQ_INSTS_VALU',0)/div/1e6:9.2f} {100*d.get('SQ_INSTS_VALU',0)/tot:6.1f} {d.get('SQ_INSTS_SALU',0)/div/1e6:9.2f} {d.get('SQ_INSTS_LDS',0)/div/1e6:8.2f} {d.get('SQ_WAVE_CYCLES',0)/div/1e6:10.1f}\n")
        f.write(f"total VALU M per {per_label[:-1]}: {tot / div / 1e6:.1f}\n")
    return tot / div

valu_pcm = work_table("workp", os.path.join(dst, "valu_work_from_pcm.txt"), "writes") if os.path.isdir(os.path.join(src, "workp")) else None
valu_solo = work_table("works", os.path.join(dst, "valu_work_solo.txt"), "steps") if os.path.isdir(os.path.join(src, "works")) else None

# 3. HBM traffic per kernel and per stage: (2*FETCH_SIZE + WRITE_SIZE) KiB, MI355X_MICROARCH.md HBM section
def pmc(sub, ctr):
    db, tab = open_db(sub)
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    q = (f"select s.kernel_name, x.v from (select e.event_id eid, sum(e.value) v from {pe} e join {pi} i on i.id=e.pmc_id "
         f"where i.name='{ctr}' group by e.event_id) x join {kd} d on d.event_id=x.eid join {ks} s on d.kernel_id=s.id")
    per = collections.defaultdict(list)
    for name, v in db.execute(q): per[short(name)].append(v)
    return {k: (statistics.median(v), len(v)) for k, v in per.items()}

fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
stage_bytes = collections.defaultdict(float)
with open(os.path.join(dst, "pmc_hbm_traffic.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "stage", "launches_sampled", "FETCH_SIZE_KiB_median", "WRITE_SIZE_KiB_median", "hbm_bytes_per_launch"])
    for k in sorted(set(fetch) | set(write)):
        st = stage_of(k)
        if not st: continue
        fk, n = fetch.get(k, (0, 0)); wk, _ = write.get(k, (0, 0))
        by = (2 * fk + wk) * 1024
        # kernels that run twice per stage launch (k_to_tiled: mdct + logfft; templates listed separately)
        mult = 2 if k.startswith("k_to_tiled") else 1
        stage_bytes[st] += by * mult
        w.writerow([k, st, n, fk, wk, round(by)])
# 3b. the same over the from-PCM leg: totals per kernel divided by the number of writes (k_fe_append launches)
pcm_bytes = None
if os.path.isdir(os.path.join(src, "fetchp")) and os.path.isdir(os.path.join(src, "writep")):
    def pmc_sum(sub, ctr):
        db, tab = open_db(sub)
        kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
        q = (f"select s.kernel_name, sum(x.v), count(*) from (select e.event_id eid, sum(e.value) v from {pe} e join {pi} i on i.id=e.pmc_id "
             f"where i.name='{ctr}' group by e.event_id) x join {kd} d on d.event_id=x.eid join {ks} s on d.kernel_id=s.id group by s.kernel_name")
        return {short(n): (v, c) for n, v, c in db.execute(q)}
    fp, wp = pmc_sum("fetchp", "FETCH_SIZE"), pmc_sum("writep", "WRITE_SIZE")
    writes = max([c for k, (v, c) in fp.items() if k.startswith("k_fe_append")] or [1])
    pcm_bytes = collections.defaultdict(float)
    with open(os.path.join(dst, "pmc_hbm_traffic_from_pcm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "stage", "launches", "hbm_bytes_per_write"])
        rows = []
        for k in sorted(set(fp) | set(wp)):
            if not (k.startswith("k_") or "k_zero" in k): continue
            by = (2 * fp.get(k, (0, 0))[0] + wp.get(k, (0, 0))[0]) * 1024 / writes
            st = stage_of(k) or ("pack" if "k_zero" in k else "other")
            pcm_bytes[st] += by
            rows.append([k, st, fp.get(k, (0, 0))[1], round(by)])
        for r in sorted(rows, key=lambda r: -r[3]): w.writerow(r)
cfg = bench["config"]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --only block --steps 6 --warmup 2` "
                   "(per-block leg: every launch is one step's long blocks); "
                   "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE tallies 128-B requests at 64 B, "
                   "MI355X_MICROARCH.md HBM section); median over launches, summed over the kernels of a stage.",
           "commit": commit, "mode": "per_block_path",
           "channel_blocks_per_step": cfg.get("channel_blocks_per_step", 32768), "sub_batches": bench.get("sub_batches", 1),
           "hbm_bytes_per_launch": {k: round(v) for k, v in stage_bytes.items()},
           "from_pcm": None if pcm_bytes is None else {
               "note": "same counters over `bench.py --only pcm --steps 24 --warmup 4` (VBM_BENCH_PRIME=16): every kernel of the run "
                       "incl. the front end and the small batches, summed and divided by the number of writes",
               "hbm_bytes_per_write": {k: round(v) for k, v in pcm_bytes.items()},
               "total_bytes_per_write": round(sum(pcm_bytes.values()))},
           "valu_instructions_per_write_from_pcm": None if valu_pcm is None else round(valu_pcm),
           "valu_instructions_per_step_solo": None if valu_solo is None else round(valu_solo)},
          open(os.path.join(out_root, "pmc_traffic.json"), "w"), indent=1)
print("wrote", dst, "and profiles/pmc_traffic.json")
for k, v in sorted(stage_bytes.items(), key=lambda x: -x[1]): print(f"  {k:18s} {v/1e6:10.1f} MB/launch")
