"""Copies the results of the last tools/r02_measure.sh run (gpurun_out/r02, gpurun_out/prof_r02, prof_r02e) into
profiles/ (tracked): bench line, one-line-per-variant summary, per-shape GEMM tables, rocprofv3 kernel stats, timeline."""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r02"), os.path.join(R, "profiles")
lines = []
for name in ("bench_default", "bench_bf16", "bench_cfg4", "bench_cfg4_bf16", "bench_eager", "bench_forcedp",
             "bench_forcedp_eager"):
    d = json.load(open(os.path.join(O, name + ".json")))
    r, s = d["roofline"], d["step_mfma"]
    lines.append("%-20s %8.3f ms/step %8.0f sentences/s | launch: %s | roofline %s: %.1f TFLOP/s = %.3f of peak, %.1f us/launch"
                 " | step MFMA utilisation %.3f (%.1f GFLOP/step)" % (
                     name, d["ms_per_step"], d["value"], d["config"].get("launch", "?"), r["kernel"].split(" (")[0], r["achieved"],
                     r["frac"], r.get("avg_launch_us", float("nan")), s["utilisation"], s["gflop_per_step"]))
open(os.path.join(P, "r02_bench_summary.txt"), "w").write("\n".join(lines) + "\n")
shutil.copy(os.path.join(O, "bench_default.json"), os.path.join(P, "r02_bench_line.json"))
shutil.copy(os.path.join(O, "gemm_table.txt"), os.path.join(P, "r02_gemm_table.txt"))
shutil.copy(os.path.join(O, "gemm_table_bf16.txt"), os.path.join(P, "r02_gemm_table_bf16.txt"))
for tag, sfx in (("prof_r02", ""), ("prof_r02e", "_eager")):
    shutil.copy(os.path.join(R, "gpurun_out", tag, "kernel_stats.csv"), os.path.join(P, "r02_kernel_stats%s.csv" % sfx))
shutil.copy(os.path.join(R, "gpurun_out", "prof_r02", "timeline.txt"), os.path.join(P, "r02_step_timeline.txt"))
print("\n".join(lines))
