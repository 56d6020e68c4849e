"""From a ktimeline dump: for every start of a big-batch front half, what ended just before it on the other queues."""
import sys, collections
rows=[l.split() for l in open(sys.argv[1])]
rows=[(float(r[0]),float(r[1]),r[2],r[3]) for r in rows if len(r)>=4]
byq=collections.defaultdict(list)
for r in rows: byq[r[2]].append(r)
# classify queues
role={}
for q,rs in byq.items():
    names=" ".join(x[3] for x in rs)
    if "fe_append" in names: role[q]="fe"
    elif "window_mdctILi2048" in names and "couple_fast" not in names and "tonemask" in names and max(x[1] for x in rs if "tonemask" in x[3])>500: role[q]="bigF"
    elif "couple_fast" in names and max(x[1] for x in rs if "couple_fast" in x[3])>300 and "tonemask" not in names: role[q]="bigB"
    else: role[q]="small"
print({q:role[q] for q in sorted(role)})
bigF=[q for q in role if role[q]=="bigF"]
if not bigF: sys.exit("no big front queue found")
F=byq[bigF[0]]
starts=[r for r in F if "spread_flags" in r[3]]
for st in starts:
    t=st[0]
    line=[f"F start {t:9.0f}"]
    for q in sorted(byq):
        prev=[r for r in byq[q] if r[0]+r[1] <= t+1]
        if prev:
            p=prev[-1]
            line.append(f"{q}({role[q]}): {p[3][:16]} end {p[0]+p[1]:9.0f} (-{t-p[0]-p[1]:5.0f})")
    print(" | ".join(line))
