"""MFMA-pipe utilisation per kernel symbol from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_F32 / _BF16 where the device offers them, GRBM_GUI_ACTIVE) over `python3 bench.py --eager`.
usage: pmc_mfma.py <counter_dir> <out.txt>
SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which an MFMA is executing, summed over the SIMDs that report (MI355X_MICROARCH.md:
it equals 32 x N for N v_mfma_f32_32x32x16_bf16); the utilisation quoted here is the RATIO to SQ_BUSY_CYCLES of the same
launch (both are sums over the same shader engines), i.e. the share of the kernel's busy time in which matrix instructions
were in flight. It is evidence beside the FLOP / time fractions of the gemm table, not a replacement."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|cgn_gemm::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    key = n[:86]
    d = rows.setdefault(key, collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        d["launches"] += 1
names = sorted({c for d in rows.values() for c in d if c != "launches"})
with open(sys.argv[2], "w") as out:
    out.write("# rocprofv3 --pmc %s -- python3 bench.py --eager --no-cpu-baseline --steps 5 --warmup 2\n" % " ".join(names))
    out.write("# per kernel symbol, summed over its launches; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES\n")
    out.write("%-88s %6s %14s %14s %9s\n" % ("kernel", "n", "MFMA_BUSY", "SQ_BUSY", "mfma_busy"))
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        busy = d.get("SQ_BUSY_CYCLES", 0.0)
        mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if mf <= 0:
            continue
        out.write("%-88s %6d %14.0f %14.0f %9.3f\n" % (k, d["launches"], mf, busy, mf / busy if busy else float("nan")))
print(open(sys.argv[2]).read()[:3000])
