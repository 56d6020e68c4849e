#!/usr/bin/env python3
"""Merged timeline of HIP API calls (host) and kernel dispatches (device) over the last N ms of a rocprofv3 .db"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 14
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def tab(p): return [t for t in tabs if t.startswith(p)][0]
kd, ks, rg, st = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_region"), tab("rocpd_string")
ev = []
for s, e, q, name in db.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id"):
    name = re.sub(r"\(.*\)", "", name); name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:28]
    ev.append((s, e, f"  dev q{q}", name))
cols = [r[1] for r in db.execute(f"pragma table_info({rg})")]
for s, e, tid, name in db.execute(f"select r.start, r.end, r.tid, s.string from {rg} r join {st} s on r.name_id=s.id"):
    ev.append((s, e, f"host t{tid % 1000}", name[:28]))
ev.sort()
t1 = max(e for _, e, _, _ in ev)
t0 = t1 - last_ms * 1e6
for s, e, who, name in ev:
    if s < t0: continue
    print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:9.1f} {who} {name}")
