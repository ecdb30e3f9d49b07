"""From a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv): wall span of the last N% of the trace, sum of kernel
durations in it, fraction of the span during which >= 2 kernels were resident, and average duration per kernel name.
  python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction=0.5]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
ev = ev[int(len(ev) * skip):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
pts = sorted([(s, 1) for s, e, _ in ev] + [(e, -1) for s, e, _ in ev])
depth, last, cov = 0, t0, collections.Counter()
for t, d in pts:
    cov[min(depth, 2)] += t - last
    last, depth = t, depth + d
span = t1 - t0
print(f"kernels {len(ev)}  span {span/1e6:.3f} ms  sum of durations {busy/1e6:.3f} ms  idle {cov[0]/span:.1%}  one kernel {cov[1]/span:.1%}  two or more {cov[2]/span:.1%}")
by = collections.defaultdict(list)
for s, e, n in ev:
    by[n].append(e - s)
for n, ds in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{sum(ds)/1e6:8.3f} ms  {len(ds):5d} x {sum(ds)/len(ds)/1e3:8.1f} us  {n[:100]}")
