"""Per-dispatch PMC summary for kernels matching a substring."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if pat not in r["Kernel_Name"]:
        continue
    a = agg.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    a[r["Counter_Name"]] = float(r["Counter_Value"])
seen = 0
for d_, v in agg.items():
    seen += 1
    if seen % int(sys.argv[3] if len(sys.argv) > 3 else 1):
        continue
    print(d_, "dur_us %.1f" % v["dur"], {k: "%.4g" % x for k, x in v.items() if k not in ("name", "dur")})
