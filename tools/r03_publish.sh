#!/bin/bash
# copies the round-3 summaries from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
P=gpurun_out/r03_prof
for f in kernel_stats.csv step_timeline.txt pmc_traffic.json pmc_traffic.txt mfma_busy.txt gemm_table.txt gemm_table_bf16.txt gemm_table_cfg4_bf16.txt lstm_bench.txt bb_probe.txt; do
  [ -f $P/$f ] && cp $P/$f profiles/r03_$f
done
for f in bench_line bench_line_bf16 bench_line_cfg4 bench_line_cfg4_bf16; do
  [ -f $P/$f.json ] && grep '^{' $P/$f.json | tail -1 > profiles/r03_$f.json
done
[ -f gpurun_out/r03_dp/summary.txt ] && cp gpurun_out/r03_dp/summary.txt profiles/r03_dp_one_rank.txt
[ -f gpurun_out/r03_dp/stamps_force_dp.txt ] && cp gpurun_out/r03_dp/stamps_force_dp.txt profiles/r03_step_stamps_cfg2_force_dp.txt
[ -f gpurun_out/r03_softemb/summary.txt ] && cat gpurun_out/r03_softemb/summary.txt gpurun_out/r03_softemb/gemm.txt > profiles/r03_softemb_tiles.txt
[ -f gpurun_out/r03_hwq/summary.txt ] && cp gpurun_out/r03_hwq/summary.txt profiles/r03_hw_queues.txt
[ -f gpurun_out/r03_cfg4/summary.txt ] && cp gpurun_out/r03_cfg4/summary.txt profiles/r03_ab_switches_cfg4.txt
ls profiles | grep r03_
