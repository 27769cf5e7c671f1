#!/bin/bash
# counter passes over single bf16 x bf16 products (diagnosis of what paces the kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bb_pmc
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$O/avail.txt" 2>&1
pass() {
  name=$1; shift
  ctrs=$1; shift
  mkdir -p "$O/$name"
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$O/$name" -o p -- python3 "$R/tools/gemm_x6_pmc.py" "$@" > "$O/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$O/$name.log"; exit 1; }
  python3 - "$O/$name" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "bb_kernel" in r["Kernel_Name"]:
        rows[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        rows[r["Kernel_Name"][:60]]["_n"] += 1
for k, d in rows.items():
    n = d.pop("_n") / max(1, len(d))
    print(sys.argv[1].split("/")[-1], k, "launches %d" % n, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(d.items())))
PY
  find "$O/$name" -name "*counter_collection.csv" -delete
}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
pass sq1_8192 "$SQ1" nt 8192 8192 8192 bf16 || exit 1
pass sq2_8192 "$SQ2" nt 8192 8192 8192 bf16 || exit 1
pass tcc_8192 "$TCC" nt 8192 8192 8192 bf16 || exit 1
pass sq1_fc "$SQ1" nt 10496 50000 1024 bf16 || exit 1
pass tcc_fc "$TCC" nt 10496 50000 1024 bf16 || exit 1
