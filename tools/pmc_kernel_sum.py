"""Sum of PMC counters per kernel (per launch) from a rocprofv3 --pmc ... --kernel-trace database directory:
python3 tools/pmc_kernel_sum.py <dir> <kernel-substring> ..."""
import os, sqlite3, sys, collections
d = sys.argv[1]
f = [os.path.join(dp, x) for dp, _, fs in os.walk(d) for x in fs if x.endswith(".db")][0]
db = sqlite3.connect(f)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
tab = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, pi = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
q = (f"select s.kernel_name, i.name, sum(x.v), count(*) from (select e.event_id eid, e.pmc_id pid, sum(e.value) v from {pe} e group by e.event_id, e.pmc_id) x "
     f"join {kd} d on d.event_id = x.eid join {ks} s on d.kernel_id = s.id join {pi} i on i.id = x.pid group by s.kernel_name, i.name")
out = collections.defaultdict(dict)
for name, ctr, v, c in db.execute(q):
    for want in sys.argv[2:]:
        if want in name:
            out[want][ctr] = out[want].get(ctr, 0) + v / c
for k, dct in out.items():
    print(k, {c: round(v / 1e6, 2) for c, v in sorted(dct.items())}, "M per launch")
