#!/bin/bash
# Copies the summaries of the last tools/collect_profiles.sh run (gpurun_out/prof) and the un-profiled bench lines
# (gpurun_out/bench_f32.log, gpurun_out/bench_bf16.log) into profiles/ (tracked).
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof
cp $P/f32/f32_kernel_stats.csv profiles/r01_kernel_stats.csv
cp $P/bf16/bf16_kernel_stats.csv profiles/r01_bf16_kernel_stats.csv
cp $P/pmc_traffic.json profiles/r01_pmc_traffic.json
cp $P/pmc_traffic.txt profiles/r01_pmc_traffic.txt
{
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline  (MI355X, round 1 final, tools/collect_profiles.sh; 5 warm-up + 30 timed graph replays + 10 instrumented eager steps)"
echo "# per-kernel table: profiles/r01_kernel_stats.csv ; HBM traffic (PMC passes): profiles/r01_pmc_traffic.txt"
echo "# bench line of the un-profiled run of the same tree (python bench.py):"
tail -1 gpurun_out/bench_f32.log
echo "# bench line printed under the profiler (tracing overhead included):"
grep -h '^{"metric' $P/f32.log
echo
echo "# timeline of one replayed step under the profiler (tools/prof_timeline.py)"
python3 tools/prof_timeline.py $P/f32/f32_results.db 20
} > profiles/r01_step_kernel_summary.txt
{
echo "# OPTIONAL bf16 mode: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --compute bf16  (MI355X, tools/collect_profiles.sh)"
echo "# per-kernel table: profiles/r01_bf16_kernel_stats.csv"
echo "# bench line of the un-profiled run (python bench.py --no-cpu-baseline --compute bf16):"
tail -1 gpurun_out/bench_bf16.log
echo "# bench line printed under the profiler:"
grep -h '^{"metric' $P/bf16.log
echo
echo "# timeline of one replayed step under the profiler (tools/prof_timeline.py)"
python3 tools/prof_timeline.py $P/bf16/bf16_results.db 20
} > profiles/r01_bf16_step_kernel_summary.txt
