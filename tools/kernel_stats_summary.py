"""Summarise a rocprofv3 *_kernel_stats.csv: total kernel time per step and the top kernels.
  python tools/kernel_stats_summary.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("kernel time total %.1f ms over %g steps -> %.2f ms/step; launches/step %.0f" % (tot, steps, tot / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:24]:
    print("%7.3f ms/step %6.1f/step %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
