"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-kernel time per step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.3f  launches/step %.0f" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
    print("%-50s calls/step %6.1f  avg %8.1f us  ms/step %.3f" % (n, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / steps / 1e6))
