#!/usr/bin/env python3
"""Per-kernel mean of every PMC counter in rocprofv3 results .db files: pmcstats.py a.db [b.db ...] [--filter substr]"""
import sqlite3, sys, re, collections
flt = None
dbs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--filter": flt = args.pop(0)
    else: dbs.append(a)
out = collections.defaultdict(dict)
for path in dbs:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
    kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    cols = [r[1] for r in db.execute(f"pragma table_info({pe})")]
    q = (f"select s.kernel_name, i.name, avg(v), count(*) from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pe} e group by e.event_id, e.pmc_id) x "
         f"join {kd} d on d.event_id = x.eid join {ks} s on d.kernel_id = s.id join {pi} i on i.id = x.pid group by s.kernel_name, i.name")
    for name, ctr, v, c in db.execute(q):
        name = re.sub(r"\(.*\)", "", name)
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:28]
        out[name][ctr] = v
ctrs = sorted({c for d in out.values() for c in d})
print("kernel".ljust(28), " ".join(c[-14:].rjust(14) for c in ctrs))
for k, d in sorted(out.items()):
    if flt and flt not in k: continue
    if not k.startswith("k_"): continue
    print(k.ljust(28), " ".join((f"{d[c]:14.0f}" if c in d else " " * 14) for c in ctrs))
