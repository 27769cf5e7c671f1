#!/bin/bash
# Runs on the GPU box: SQ counters of two GEMM shapes (one rocprofv3 --pmc pass).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/gemm_pmc
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS} --output-format csv -d "$O" -o p -- python3 "$R/tools/gemm_bench.py" ${CASES:-'(4096,4096,4096,"nt",128)' '(4096,4096,4096,"nt",64)' '(1280,2484,828,"nt",64)' '(1344,20000,512,"nt",128)'} > "$O/run.log" 2>&1 || { echo failed; tail -5 "$O/run.log"; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "gemm_kernel" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
    d = agg.setdefault(key, collections.defaultdict(list))
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in agg.items():
    print(key)
    for k, v in d.items():
        print("   %-28s %14.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
