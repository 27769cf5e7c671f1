#!/bin/bash
# round 3: step-level parity (oracle, golden, full size, replay) then bench + un-profiled stamp timeline
set -o pipefail
out=gpurun_out/${1:-r03_step}
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_step_gpu.py tests/test_step_golden_gpu.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py \
    tests/test_graph_fullsize_gpu.py tests/test_loop_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1
rc=$?
tail -15 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("ms/step %.3f  sentences/s %.0f  step_mfma %.3f  roofline %.3f" % (d["ms_per_step"], d["value"], d["step_mfma"]["utilisation"], d["roofline"]["frac"]))
PY
timeout -k 10 200 python tools/step_stamps.py --out $out/stamps_cfg2.txt 2>&1 | grep -v amdgpu.ids
