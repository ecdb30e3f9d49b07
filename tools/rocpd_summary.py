"""Per-kernel summary of a rocprofv3 (rocpd sqlite) kernel trace: python tools/rocpd_summary.py DB [steps]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':80s} {'calls':>6s} {'ms/step':>9s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'%':>6s}")
for n, c, s, a, lo, hi in rows[:40]:
    nm = re.sub(r'\(anonymous namespace\)::', '', n)
    nm = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', nm)[:80]
    print(f"{nm:80s} {c:6d} {s/1e6/steps:9.3f} {a/1e3:9.1f} {lo/1e3:9.1f} {hi/1e3:9.1f} {100*s/tot:6.1f}")
print(f"total kernel time per step: {tot/1e6/steps:.3f} ms")
