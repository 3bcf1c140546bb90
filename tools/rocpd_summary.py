"""Summarise a rocprofv3 rocpd sqlite result (kernel-trace) into a per-kernel stats table (markdown/CSV-ish text)."""
import sqlite3, sys, re

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:150]

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1].split('/')[-1]}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
print("| % | total ms | calls | avg us | min us | max us | kernel |")
print("|---|---|---|---|---|---|---|")
for n, c, s, a, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"| {100*s/tot:5.1f} | {s/1e6:8.3f} | {c:5d} | {a/1e3:9.1f} | {mn/1e3:8.1f} | {mx/1e3:8.1f} | {short(n)} |")
# per-(kernel, grid) breakdown of the GEMM kernels: one launch shape per row, so the dominant kernel at its
# train-step arguments (decoder FFN conv: grid 1024 x 256 threads) can be compared with bench.py's HIP-event timing
print("\n## GEMM kernels by launch shape (grid_x = workgroups x 256 threads)\n")
print("| total ms | calls | avg us | min us | max us | grid_x | grid_z | kernel |")
print("|---|---|---|---|---|---|---|---|")
g = db.execute("select name, grid_x, grid_z, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
               "where name like '%gemm_%' group by name, grid_x, grid_z order by sum(duration) desc limit 24").fetchall()
for n, gx, gz, c, s, a, mn, mx in g:
    print(f"| {s/1e6:8.3f} | {c:5d} | {a/1e3:9.1f} | {mn/1e3:8.1f} | {mx/1e3:8.1f} | {gx//256} | {gz} | {short(n)} |")
