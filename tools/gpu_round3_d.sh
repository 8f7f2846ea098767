#!/bin/bash
set -u
for cap in 0 400 800 1100; do
for w in mixed1g text64m; do
ZLIB_AMD_INDEX_CAP_KB=$cap timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/r3d_$w.json 2> gpurun_out/r3d_$w.err || { tail -5 gpurun_out/r3d_$w.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3d_$w.json').read().strip().splitlines()[-1])
print('cap=$cap $w',round(d['value'],3),'GB/s',round(d['ms_per_step'],1),'ms',{k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if v>0.3})"
done; done
