#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_inflate.py -x -q -m gpu --timeout 120 --timeout-method thread -k "record or batch or inflate or blocks" > gpurun_out/r3l_tests.log 2>&1 || { tail -30 gpurun_out/r3l_tests.log; exit 1; }
tail -2 gpurun_out/r3l_tests.log
timeout -k 10 300 python -u tools/time_inflate.py text 256 2 || exit 1
timeout -k 10 300 python -u tools/time_inflate.py mixed 8192 2 || exit 1
timeout -k 10 300 python -u bench.py --workload records4g --no-cpu-baseline --no-others --no-verify --steps 4 > gpurun_out/r3k_rec.json 2> gpurun_out/r3k_rec.err || { tail -20 gpurun_out/r3k_rec.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r3k_rec.json").read().strip().splitlines()[-1])
print("records4g", round(l["value"], 3), "GB/s", round(l["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in l["kernels_ms_per_step"].items()})
PY
