#!/bin/bash
# usage: tools/walk_probe.sh [gap shifts...] : counters of the debug build, then throughput per kind of corpus
cp zlib_amd/libzlib_amd.so /tmp/orig.so && cp zlib_amd/libzlib_amd_dbg.so zlib_amd/libzlib_amd.so
for k in text binary random; do timeout -k 10 120 python tools/debug_walk.py $k || exit 1; done
cp /tmp/orig.so zlib_amd/libzlib_amd.so
for g in ${@:-4}; do
  for w in text256m binary256m random256m; do
    echo -n "gap shift $g: "
    ZLIB_AMD_WALK=1 ZLIB_AMD_WALK_GAP_SHIFT=$g timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['config']['name'], round(d['value'],3), {a: round(b,2) for a,b in k.items() if b > 0.2})" || exit 1
  done
done
