#!/bin/bash
# usage: tools/try_variants.sh "tag1 tag2 ..." "workload1 workload2 ..." : benches prebuilt zlib_amd/libzlib_amd_<tag>.so variants
cp zlib_amd/libzlib_amd.so /tmp/libzlib_amd_orig.so
for v in $1; do
  cp zlib_amd/libzlib_amd_$v.so zlib_amd/libzlib_amd.so
  for w in $2; do
    echo -n "$v $w: "
    timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(round(d['value'],3), {a: round(b,2) for a,b in k.items() if b > 0.2})"
  done
done
cp /tmp/libzlib_amd_orig.so zlib_amd/libzlib_amd.so
