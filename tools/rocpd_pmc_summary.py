"""Per-kernel averages of PMC counters from a rocprofv3 rocpd sqlite result (--pmc ... --kernel-trace)."""
import sqlite3, sys, re, collections

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
print("# counters_collection columns:", cols)
namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
ccol = "counter_name" if "counter_name" in cols else None
vcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
dcol = "dispatch_id" if "dispatch_id" in cols else None
if not (namecol and ccol and vcol):
    for r in db.execute("select * from counters_collection limit 5"):
        print(r)
    sys.exit(0)
rows = db.execute(f"select {namecol}, {ccol}, {dcol}, sum({vcol}) from counters_collection group by {namecol}, {ccol}, {dcol}").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for n, c, d, v in rows:
    agg[n][c].append(v)
dur = {}
try:
    for n, a, cnt in db.execute("select name, avg(duration), count(*) from kernels group by name"):
        dur[n] = (a, cnt)
except Exception:
    pass
ctrs = sorted({c for n in agg for c in agg[n]})
print("| kernel | calls | avg us | " + " | ".join(ctrs) + " |")
print("|---|---|---|" + "---|" * len(ctrs))
for n in sorted(agg, key=lambda k: -(dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1])):
    short = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:70]
    a, cnt = dur.get(n, (0, 0))
    print(f"| {short} | {cnt} | {a/1e3:.1f} | " + " | ".join(f"{sum(agg[n][c])/len(agg[n][c]):.4g}" if c in agg[n] else "-" for c in ctrs) + " |")
