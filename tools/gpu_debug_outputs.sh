#!/bin/bash
# phase timers and counters of the index and match kernels (debug builds of the library), written as files
set -u
OUT=${1:-gpurun_out/r3dbg}
mkdir -p $OUT
cp zlib_amd/libzlib_amd.so /tmp/keep.so
cp zlib_amd/libzlib_amd_idxdbg.so zlib_amd/libzlib_amd.so
for k in mixed text binary random; do
  timeout -k 10 120 python -u tools/debug_index.py $k 256 2>&1 | grep -v amdgpu.ids >> $OUT/index_timeline.txt || { cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }
done
cp zlib_amd/libzlib_amd_matchdbg.so zlib_amd/libzlib_amd.so
for k in text binary random recbatch; do
  timeout -k 10 120 python -u tools/debug_match.py $k 2>&1 | grep -v amdgpu.ids >> $OUT/match_phases.txt || { cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }
done
cp /tmp/keep.so zlib_amd/libzlib_amd.so
cat $OUT/index_timeline.txt $OUT/match_phases.txt
