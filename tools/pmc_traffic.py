"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs of the same command).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so it is doubled; WRITE_SIZE is exact. The doubling is calibrated in the same run on
softmax_nll_fwd_kernel, whose only traffic is one read of the [B*T, V] fp32 logits."""
import collections
import csv
import glob
import json
import re
import sys


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|cgn_gemm::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        key = "%s g%d" % (n[:110], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"unit": "bytes per launch", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count), WRITE_SIZE KiB x 1024",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --eager --no-cpu-baseline --steps 5 --warmup 2",
       "kernels": {}}
for k in fetch:
    if k not in write or len(fetch[k]) < 3:
        continue
    fr = sum(fetch[k]) / len(fetch[k])
    wr = sum(write[k]) / len(write[k])
    out["kernels"][k] = {"launches": len(fetch[k]), "fetch_raw": fr, "fetch": 2 * fr, "write": wr, "traffic": 2 * fr + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic"] * kv[1]["launches"])[:25]:
    print("%-70s n=%3d fetch %8.1f MB write %8.1f MB" % (k, v["launches"], v["fetch"] / 1e6, v["write"] / 1e6))
