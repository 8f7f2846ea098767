#!/bin/bash
set -u
cp zlib_amd/libzlib_amd.so /tmp/keep.so && cp zlib_amd/libzlib_amd_idxdbg.so zlib_amd/libzlib_amd.so
rm -f gpurun_out/r3b_index.txt
for k in mixed text binary random; do timeout -k 10 120 python -u tools/debug_index.py $k 256 >> gpurun_out/r3b_index.txt 2>&1 || { tail -5 gpurun_out/r3b_index.txt; cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }; done
cp /tmp/keep.so zlib_amd/libzlib_amd.so
grep -v amdgpu.ids gpurun_out/r3b_index.txt
