#!/bin/bash
# new shard-piece tests, the whole GPU suite, the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "shard or several_devices" > gpurun_out/full_shard.log 2>&1 || { tail -30 gpurun_out/full_shard.log; exit 1; }
tail -2 gpurun_out/full_shard.log
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/full_tests.log 2>&1 || { tail -30 gpurun_out/full_tests.log; exit 1; }
tail -2 gpurun_out/full_tests.log
timeout -k 10 560 python -u bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err || { tail -20 gpurun_out/full_bench.err; exit 1; }
python - <<'PY'
import json
l = json.loads(open("gpurun_out/full_bench.json").read().strip().splitlines()[-1])
print(round(l["value"], 3), "GB/s", round(l["ms_per_step"], 1), "ms", {k: round(v, 1) for k, v in l["kernels_ms_per_step"].items()})
print("inflate on device ms", l["config"]["inflate_on_device_ms"], "zlib", l["config"]["round_trip_system_zlib"])
for k, v in l.get("other_configs", {}).items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "what"})
print("cpu", l.get("cpu_baseline", {}).get("value"))
PY
