"""Print a rocprofv3 kernel_stats.csv as a per-step table: python tools/kstats.py <csv> <steps_incl_warmup>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.3f ms (%d kernels listed)" % (tot / 1e6 / steps, len(rows)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-95s %6s calls %9.3f ms/step  avg %9.1f us  %6.2f%%" % (
        r["Name"][:95], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3,
        float(r["Percentage"])))
