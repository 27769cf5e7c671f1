#!/bin/bash
# Runs on the GPU box: round-3 evidence. Kernel stats + timeline of the replayed bench, HBM traffic (FETCH / WRITE passes),
# MFMA-busy counters, the gemm table, the bench line. Everything lands under gpurun_out/r03_prof; tools/r03_publish.sh
# copies the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_prof
rm -rf "$O" && mkdir -p "$O/trace" "$O/fetch" "$O/write" "$O/mfma"
cd "$R" && timeout -k 10 300 python3 bench.py --gemm-table "$O/gemm_table.txt" > "$O/bench_line.json" 2> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
timeout -k 10 200 python3 bench.py --no-cpu-baseline --compute bf16 --gemm-table "$O/gemm_table_bf16.txt" > "$O/bench_line_bf16.json" 2>> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload cfg4 --steps 10 > "$O/bench_line_cfg4.json" 2>> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload cfg4 --compute bf16 --steps 10 --gemm-table "$O/gemm_table_cfg4_bf16.txt" > "$O/bench_line_cfg4_bf16.json" 2>> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
# the recurrences alone: cfg2 shapes (with and without the blocked hand-off scratch), configs[3] shapes in the three operand modes
{ echo "# tools/lstm_bench.py 64 512 20"; timeout -k 10 120 python3 tools/lstm_bench.py 64 512 20; echo "# CGN_LSTM_HANDOFF=0"; CGN_LSTM_HANDOFF=0 timeout -k 10 120 python3 tools/lstm_bench.py 64 512 20;
  echo "# tools/lstm_bench.py 256 1024 40 (fp32 mode)"; timeout -k 10 120 python3 tools/lstm_bench.py 256 1024 40;
  echo "# CGN_COMPUTE=bf16 CGN_LSTM_BF16_RT=1 (two-term operands)"; CGN_COMPUTE=bf16 CGN_LSTM_BF16_RT=1 timeout -k 10 120 python3 tools/lstm_bench.py 256 1024 40;
  echo "# CGN_COMPUTE=bf16 (plain bf16 operands, bf16 hand-off scratch)"; CGN_COMPUTE=bf16 timeout -k 10 120 python3 tools/lstm_bench.py 256 1024 40; } > "$O/lstm_bench.txt" 2>> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
timeout -k 10 400 python3 tools/gemm_bb_probe.py "$O/bb_probe.txt" > /dev/null 2>> "$O/bench.err" || { tail "$O/bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$O/trace" -o t -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$O/trace.log"; exit 1; }
DB=$(ls "$O"/trace/*.db "$O"/trace/*/*.db 2>/dev/null | head -1)
python3 "$R/tools/prof_timeline.py" "$DB" 20 > "$O/step_timeline.txt" 2>&1
cp "$(ls "$O"/trace/*kernel_stats.csv "$O"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)" "$O/kernel_stats.csv"
find "$O/trace" -name "*kernel_trace.csv" -delete; find "$O/trace" -name "*.db" -delete
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$O/fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$O/write.log"; exit 1; }
python3 "$R/tools/pmc_traffic.py" "$O/fetch" "$O/write" "$O/pmc_traffic.json" > "$O/pmc_traffic.txt" 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/mfma" -o m -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/mfma.log" 2>&1 || { echo "mfma pass failed"; tail -5 "$O/mfma.log"; exit 1; }
MS=$(python3 -c "import json; print(json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1])['ms_per_step'])")
python3 "$R/tools/pmc_mfma.py" "$O/mfma" "$O/mfma_busy.txt" 7 "$MS" > /dev/null 2>&1
find "$O" -name "*counter_collection.csv" -delete
head -12 "$O/pmc_traffic.txt"; head -14 "$O/mfma_busy.txt"; cat "$O"/bench_line*.json | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('%.3f ms/step %s %s' % (d['ms_per_step'], d['dtype'], d['config']['workload'][:90]))"
