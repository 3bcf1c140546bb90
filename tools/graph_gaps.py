"""Gaps between kernels under hipGraph replay: from a rocprofv3 --kernel-trace database of `bench.py` (graph mode), take the last
replayed step (the kernels between the last two launches of adam_kernel) and compare the sum of kernel durations with the span from the
first kernel's start to the last kernel's end.   python tools/graph_gaps.py <results.db>"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
if len(idx) < 3:
    print("not enough steps in the trace", len(idx)); sys.exit(0)
for a, b in zip(idx[-4:-1], idx[-3:]):
    seg = rows[a + 1:b + 1]
    busy = sum(e - s for _, s, e in seg)
    span = seg[-1][2] - seg[0][1]
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, seg[i][0][:60], seg[i + 1][0][:60]) for i in range(len(seg) - 1))
    neg = sum(1 for g in gaps if g[0] < 0)
    print(f"step: {len(seg)} kernels, span {span/1e6:.3f} ms, sum of durations {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms "
          f"({100*(span-busy)/span:.1f} %), median gap {gaps[len(gaps)//2][0]:.2f} us, overlapping pairs {neg}")
print("largest gaps (us, after kernel -> before kernel):")
for g in gaps[-8:]:
    print(f"  {g[0]:8.2f}  {g[1]}  ->  {g[2]}")
# per-kernel totals of the last step
import collections, re
seg = rows[idx[-2] + 1:idx[-1] + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    k = re.sub(r"\(anonymous namespace\)::|void ", "", n)[:70]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
small = sum(1 for _, s, e in seg if e - s < 10000), sum((e - s) / 1e3 for _, s, e in seg if e - s < 10000)
print(f"\nlast step: kernels shorter than 10 us: {small[0]} launches, {small[1]/1e3:.3f} ms")
print("| ms | calls | avg us | kernel |")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"| {t/1e3:6.3f} | {c:4d} | {t/c:7.1f} | {k} |")
