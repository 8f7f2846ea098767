#!/bin/bash
# the multi-rank bench path rehearsed over gloo: 2 and 3 ranks share the one GPU of the box
set -o pipefail
mkdir -p gpurun_out
export ZG_BENCH_BACKEND=gloo
export ZG_BENCH_HANG_DUMP=100
for n in 2 3; do
for ex in overlapped serial; do
ZG_BENCH_EXCHANGE=$ex timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 2 --warmup 1 --workload mixed1g > gpurun_out/gloo_$n$ex.json 2> gpurun_out/gloo_$n$ex.err || { tail -20 gpurun_out/gloo_$n$ex.err; exit 1; }
python - $n$ex <<'PY'
import json, sys
l = json.loads(open(f"gpurun_out/gloo_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(l["value"], 3), "GB/s", round(l["ms_per_step"], 1), "ms; zlib", l["config"]["round_trip_system_zlib"], l["config"]["compressed_bytes"])
PY
done
done
