#!/bin/bash
# usage: tools/try_test.sh TAG "pytest -k expr": runs selected GPU tests against zlib_amd/libzlib_amd_TAG.so
cp zlib_amd/libzlib_amd.so /tmp/libzlib_amd_orig.so
cp zlib_amd/libzlib_amd_$1.so zlib_amd/libzlib_amd.so
timeout -k 5 100 python -u -m pytest tests -m gpu -x -q -k "$2" --timeout 80 --timeout-method thread 2>&1 | tail -5
rc=$?
cp /tmp/libzlib_amd_orig.so zlib_amd/libzlib_amd.so
exit $rc
