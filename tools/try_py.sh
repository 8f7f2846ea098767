#!/bin/bash
# usage: tools/try_py.sh TAG script args... : runs a python script against zlib_amd/libzlib_amd_TAG.so
TAG=$1; shift
cp zlib_amd/libzlib_amd.so /tmp/libzlib_amd_orig.so
cp zlib_amd/libzlib_amd_$TAG.so zlib_amd/libzlib_amd.so
timeout -k 5 120 python -u "$@" 2>&1 | tail -30
cp /tmp/libzlib_amd_orig.so zlib_amd/libzlib_amd.so
