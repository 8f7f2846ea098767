#!/bin/bash
# whole GPU suite, then the headline and the record workload without the extras
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/suite_tests.log 2>&1 || { tail -30 gpurun_out/suite_tests.log; exit 1; }
tail -2 gpurun_out/suite_tests.log
for w in mixed8g records4g; do
timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --no-others --no-verify --steps 4 > gpurun_out/suite_$w.json 2> gpurun_out/suite_$w.err || { tail -20 gpurun_out/suite_$w.err; exit 1; }
python - $w <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/suite_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(l["value"], 3), "GB/s", round(l["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in l["kernels_ms_per_step"].items()})
PY
done
