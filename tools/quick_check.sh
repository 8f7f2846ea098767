#!/bin/bash
# GPU box: parity tests first, then short bench lines.  usage: tools/quick_check.sh <tag> "<workloads>" [pytest -k expression]
set -u
tag="$1"; wls="$2"; kexpr="${3:-}"
if [ -n "$kexpr" ]; then
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q --timeout 300 --timeout-method thread -k "$kexpr" > gpurun_out/qc_${tag}_tests.log 2>&1 || { tail -30 gpurun_out/qc_${tag}_tests.log; exit 1; }
else
  timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/qc_${tag}_tests.log 2>&1 || { tail -30 gpurun_out/qc_${tag}_tests.log; exit 1; }
fi
tail -1 gpurun_out/qc_${tag}_tests.log
for w in $wls; do
  timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/qc_${tag}_$w.json 2> gpurun_out/qc_${tag}_$w.err || { echo "$w failed"; tail -5 gpurun_out/qc_${tag}_$w.err; exit 1; }
  python3 - "$tag" "$w" <<'PY'
import json, sys
t, w = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/qc_{t}_{w}.json").read().strip().splitlines()[-1])
print(w, round(d["value"], 2), "GB/s", round(d["ms_per_step"], 2), "ms", {k: round(x, 2) for k, x in d["kernels_ms_per_step"].items() if x >= 0.3})
PY
done
