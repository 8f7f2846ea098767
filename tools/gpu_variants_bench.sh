#!/bin/bash
# usage: tools/gpu_variants_bench.sh "<tags>" <workload>: per-kernel times of bench.py with prebuilt zlib_amd/libzlib_amd_<tag>.so variants ("base" = the regular build)
set -u
cp zlib_amd/libzlib_amd.so /tmp/keep.so
for v in $1; do
  [ "$v" = base ] || cp zlib_amd/libzlib_amd_$v.so zlib_amd/libzlib_amd.so
  timeout -k 10 200 python -u bench.py --workload $2 --no-cpu-baseline --no-others --no-verify --steps 4 > /tmp/vb.json 2> /tmp/vb.err || { tail /tmp/vb.err; cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }
  python - $v <<'PY'
import json, sys
l = json.loads(open("/tmp/vb.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(l["value"], 3), "GB/s", round(l["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in l["kernels_ms_per_step"].items()}, flush=True)
PY
  cp /tmp/keep.so zlib_amd/libzlib_amd.so
done
