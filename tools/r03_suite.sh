#!/bin/bash
# round 3: the whole GPU tier + the default bench line (what the driver runs at round end)
set -o pipefail
out=gpurun_out/${1:-r03_suite}
mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1
rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("ms/step %.3f  sentences/s %.0f  step_mfma %.3f  roofline %.3f  cpu %s" % (
    d["ms_per_step"], d["value"], d["step_mfma"]["utilisation"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value")))
PY
