#!/bin/bash
# parity of the rewritten index kernel, then its timing
set -u
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_huffman_limits.py -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r3c_tests.log 2>&1 || { tail -40 gpurun_out/r3c_tests.log; exit 1; }
tail -2 gpurun_out/r3c_tests.log
for w in mixed1g text256m binary256m random256m records256m; do
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/r3c_$w.json 2> gpurun_out/r3c_$w.err || { tail -5 gpurun_out/r3c_$w.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3c_$w.json').read().strip().splitlines()[-1])
print('$w',round(d['value'],3),'GB/s',round(d['ms_per_step'],1),'ms',{k:round(v,2) for k,v in d['kernels_ms_per_step'].items() if v>0.3})"
done
cp zlib_amd/libzlib_amd.so /tmp/keep.so && cp zlib_amd/libzlib_amd_idxdbg.so zlib_amd/libzlib_amd.so
rm -f gpurun_out/r3c_index.txt
for k in mixed binary random; do timeout -k 10 120 python -u tools/debug_index.py $k 256 >> gpurun_out/r3c_index.txt 2>&1 || { tail -5 gpurun_out/r3c_index.txt; cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }; done
cp /tmp/keep.so zlib_amd/libzlib_amd.so
grep -v amdgpu.ids gpurun_out/r3c_index.txt
