#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 results .db (kernel-trace): name, calls, avg/min/max us, total ms."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
for name, c, a, mn, mx, t in rows:
    name = re.sub(r"\(.*\)", "", name)[:60]
    print(f"{name:60s} {c:6d} {a/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {t/1e6:10.2f} {100*t/tot:6.1f}")
