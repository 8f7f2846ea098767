#!/bin/bash
# inflate tests, then the device round trip of 1 GiB (bench.py reports its time)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_inflate.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3g_inflate.log 2>&1 || { tail -30 gpurun_out/r3g_inflate.log; exit 1; }
tail -2 gpurun_out/r3g_inflate.log
for w in mixed1g text256m; do
timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --no-others --steps 2 > gpurun_out/r3g_$w.json 2> gpurun_out/r3g_$w.err || { tail -20 gpurun_out/r3g_$w.err; exit 1; }
python - $w <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/r3g_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(l["value"], 3), "GB/s; inflate on device ms", l["config"]["inflate_on_device_ms"], "zlib", l["config"]["round_trip_system_zlib"])
PY
done
