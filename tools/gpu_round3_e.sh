#!/bin/bash
set -u
timeout -k 10 600 python -u -m pytest tests/test_gpu_inflate.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e_tests.log 2>&1 || { tail -40 gpurun_out/r3e_tests.log; exit 1; }
tail -3 gpurun_out/r3e_tests.log
