#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_inflate.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
timeout -k 10 300 python -u tools/time_inflate.py text 256 2 || exit 1
timeout -k 10 300 python -u tools/time_inflate.py mixed 4096 2 || exit 1
