#!/bin/bash
# round-3 first GPU call: the GPU suite, a short headline bench, the index kernel's wave timeline
set -u
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3a_tests.log 2>&1 || { tail -40 gpurun_out/r3a_tests.log; exit 1; }
tail -2 gpurun_out/r3a_tests.log
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err || { tail -5 gpurun_out/r3a_bench.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3a_bench.json').read().strip().splitlines()[-1])
print(round(d['value'],3),'GB/s',round(d['ms_per_step'],1),'ms',{k:round(v,1) for k,v in d['kernels_ms_per_step'].items()})"
cp zlib_amd/libzlib_amd.so /tmp/keep.so && cp zlib_amd/libzlib_amd_idxdbg.so zlib_amd/libzlib_amd.so
for k in mixed; do timeout -k 10 120 python -u tools/debug_index.py $k 256 >> gpurun_out/r3a_index.txt 2>&1 || { tail -5 gpurun_out/r3a_index.txt; cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }; done
cp /tmp/keep.so zlib_amd/libzlib_amd.so
cat gpurun_out/r3a_index.txt
