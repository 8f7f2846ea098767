#!/bin/bash
# usage: tools/gpu_variants_index.sh "<tags>" [kinds]: index-only timing of prebuilt zlib_amd/libzlib_amd_<tag>.so variants
set -u
cp zlib_amd/libzlib_amd.so /tmp/keep.so
for v in $1; do
  cp zlib_amd/libzlib_amd_$v.so zlib_amd/libzlib_amd.so
  TAG=$v timeout -k 10 120 python -u tools/debug_index_only.py ${2:-} 2>&1 | grep -v amdgpu.ids || { cp /tmp/keep.so zlib_amd/libzlib_amd.so; exit 1; }
done
cp /tmp/keep.so zlib_amd/libzlib_amd.so
