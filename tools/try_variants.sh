#!/bin/bash
# usage: tools/try_variants.sh "1_1 1_2 ..." : benches each prebuilt zlib_amd/libzlib_amd_<v>.so variant
for v in $1; do
  cp zlib_amd/libzlib_amd_$v.so zlib_amd/libzlib_amd.so
  echo -n "$v: "
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['kernels_ms_per_step']['k_match'],2))"
done
