#!/bin/bash
# per-kernel totals over the from-PCM leg: calls, VALU instructions, wave cycles, time (one PMC pass + one timing pass)
OUT=$GRAFT_REPO_ROOT/gpurun_out/work
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export VBM_BENCH_NO_STAGE_PASS=1 VBM_BENCH_PRIME=16
rm -rf $OUT/q; mkdir -p $OUT/q
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/q -o q -- python3 $GRAFT_REPO_ROOT/bench.py --only pcm --steps 24 --warmup 4 --no-cpu-baseline > $OUT/q.log 2>&1
python3 - $(ls $OUT/q/*/*.db $OUT/q/*.db 2>/dev/null | head -1) <<'PY'
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
tab = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
q = (f"select s.kernel_name, i.name, sum(x.v), count(*) from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pe} e group by e.event_id, e.pmc_id) x "
     f"join {kd} d on d.event_id = x.eid join {ks} s on d.kernel_id = s.id join {pi} i on i.id = x.pid group by s.kernel_name, i.name")
out = collections.defaultdict(dict); calls = {}
for name, ctr, v, c in db.execute(q):
    name = re.sub(r"\(.*\)", "", name); name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:40]
    out[name][ctr] = v; calls[name] = c
tot = sum(d.get("SQ_INSTS_VALU", 0) for d in out.values())
print(f"{'kernel':40s} {'calls':>6s} {'VALU_M':>9s} {'%':>6s} {'SALU_M':>9s} {'LDS_M':>8s} {'waveMcyc':>9s}")
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:45]:
    print(f"{k:40s} {calls[k]:6d} {d.get('SQ_INSTS_VALU',0)/1e6:9.1f} {100*d.get('SQ_INSTS_VALU',0)/tot:6.1f} {d.get('SQ_INSTS_SALU',0)/1e6:9.1f} {d.get('SQ_INSTS_LDS',0)/1e6:8.1f} {d.get('SQ_WAVE_CYCLES',0)/1e6:9.1f}")
print("total VALU M:", tot / 1e6)
PY
rm -rf $OUT/q
