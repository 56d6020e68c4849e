#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches in a rocprofv3 .db: start(us) dur(us) queue kernel"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else cols[0])
rows = db.execute(f"select d.start, d.end, d.{qcol}, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
rows = rows[-N:]
t0 = rows[0][0]
for st, en, q, name in rows:
    name = re.sub(r"\(.*\)", "", name); name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:30]
    print(f"{(st-t0)/1e3:10.1f} {(en-st)/1e3:9.1f} q{q} {name}")
