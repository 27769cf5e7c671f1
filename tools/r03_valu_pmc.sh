#!/bin/bash
# Runs on the GPU box: which kernels of the step are bound by vector instructions? One rocprofv3 --pmc pass
# (SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES) over the eager step.
# usage: r03_valu_pmc.sh <name> <bench.py arguments...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
O=$R/gpurun_out/r03_valu_$NAME
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/pmc" -o v -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 3 --warmup 1 "$@" > "$O/pmc.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$O/pmc.log"; exit 1; }
python3 - "$O" <<'PY'
import collections, csv, glob, re, sys
O = sys.argv[1]
f = glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::|cgn_gemm::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
    d = rows.setdefault(n[:96], collections.defaultdict(float))
    d[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        d["n"] += 1
out = ["# per kernel symbol, summed over its launches (4 eager steps). busy = SQ_BUSY_CYCLES / 32 shader engines = shader cycles the",
       "# kernel was on the chip; valu = 4 x SQ_INSTS_VALU / (1024 SIMDs x busy): share of the SIMDs' cycles that issuing its",
       "# vector instructions takes at 4 cycles per wave64 instruction (transcendentals take more: a lower bound; MFMAs are",
       "# counted among the vector instructions); mfma = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x busy) as tools/pmc_mfma.py",
       "%-98s %5s %10s %7s %7s %9s" % ("kernel", "n", "busy Mcyc", "valu", "mfma", "VALU inst")]
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0))[:40]:
    busy = d.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    if busy <= 0:
        continue
    out.append("%-98s %5d %10.2f %7.3f %7.3f %9.2e" % (k, d["n"], busy / 1e6, 4.0 * d.get("SQ_INSTS_VALU", 0.0) / (1024 * busy),
                                                      d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * busy), d.get("SQ_INSTS_VALU", 0.0)))
open(O + "/valu.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find "$O" -name "*counter_collection.csv" -delete
