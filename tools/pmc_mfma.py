"""MFMA-pipe utilisation per kernel symbol from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_F32 / _BF16 where the device offers them, GRBM_GUI_ACTIVE) over `python3 bench.py --eager`.
usage: pmc_mfma.py <counter_dir> <out.txt> [steps_in_the_run ms_per_replayed_step]
SQ_VALU_MFMA_BUSY_CYCLES = cycles in which a SIMD's matrix pipe is executing, summed over the chip's 1024 SIMDs
(MI355X_MICROARCH.md: 32 per v_mfma_f32_32x32x16_bf16; measured here: 64 per v_mfma_f32_32x32x2_f32 - dW_fc's 6.72 M fp32
MFMAs give 432 M busy cycles per launch); SQ_BUSY_CYCLES = cycles the sequencers are busy, summed over the 32 shader
engines (4 per XCD): SQ_BUSY / 32 is the kernel's busy time in shader cycles (dW_fc: 616 k cycles = 257 us at 2.4 GHz).
  mfma_busy = MFMA_BUSY / (1024 SIMDs x SQ_BUSY / 32) = MFMA_BUSY / (32 x SQ_BUSY)
is the share of all matrix-pipe cycles of the launch that were busy - the counter-level MFMA utilisation. It counts the
padded tiles' MFMAs too (a 128-tile grid over M = 1344 executes 1408 rows), so it sits a few points above the
algorithmic FLOP / time fractions of the gemm table."""
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
    out.write("# per kernel symbol, summed over its launches; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)\n")
    out.write("%-88s %6s %14s %14s %9s\n" % ("kernel", "n", "MFMA_BUSY", "SQ_BUSY", "mfma_busy"))
    for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        busy = d.get("SQ_BUSY_CYCLES", 0.0)
        mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if mf <= 0:
            continue
        out.write("%-88s %6d %14.0f %14.0f %9.3f\n" % (k, d["launches"], mf, busy, mf / busy / 32.0 if busy else float("nan")))
    if len(sys.argv) > 4:
        steps, ms = int(sys.argv[3]), float(sys.argv[4])
        tot = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in rows.values()) / steps
        out.write("# whole step: %.0f M matrix-pipe busy cycles per step (all kernels, %d steps); over the replayed step's %.3f ms "
                  "x 1024 SIMDs x 2.4 GHz = %.3f of all matrix-pipe cycles\n" % (tot / 1e6, steps, ms, tot / (1024 * 2.4e9 * ms * 1e-3)))
print(open(sys.argv[2]).read()[:3000])
