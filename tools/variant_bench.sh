#!/bin/bash
# Runs bench.py on a few workloads once per library variant found under zlib_amd/variants/ (built beforehand with
# different -D switches); the variant is copied over the library in the scratch copy of the tree on the GPU box.
# usage: tools/variant_bench.sh "<workloads>" <variant.so>...
set -u
wls="$1"; shift
cp zlib_amd/libzlib_amd.so /tmp/lib_keep.so
for v in "$@"; do
  cp "zlib_amd/variants/$v" zlib_amd/libzlib_amd.so
  for w in $wls; do
    timeout -k 10 200 python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/var_${v%.so}_$w.json 2> gpurun_out/var_${v%.so}_$w.err || { echo "$v $w failed"; cp /tmp/lib_keep.so zlib_amd/libzlib_amd.so; exit 1; }
    python - "$v" "$w" <<'PY'
import json, sys
v, w = sys.argv[1:3]
d = json.loads(open(f"gpurun_out/var_{v[:-3]}_{w}.json").read().strip().splitlines()[-1])
print(v, w, round(d["value"], 2), round(d["ms_per_step"], 2), {k: round(x, 2) for k, x in d["kernels_ms_per_step"].items() if x >= 0.5})
PY
  done
done
cp /tmp/lib_keep.so zlib_amd/libzlib_amd.so
