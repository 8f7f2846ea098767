# bench.py with events around every launch, around the dominant kernel only, and with none in the timed steps
set -o pipefail
for w in mixed8g records4g text64m; do
for t in all dominant 0; do
ZG_BENCH_TIMERS=$t timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --no-others --no-verify --steps 4 > gpurun_out/tm_${w}_$t.json 2> gpurun_out/tm_${w}_$t.err || { tail -20 gpurun_out/tm_${w}_$t.err; exit 1; }
python - $w $t <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/tm_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "timers", sys.argv[2], round(l["value"], 3), "GB/s", round(l["ms_per_step"], 2), "ms")
PY
done
done
