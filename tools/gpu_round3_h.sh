#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_inflate.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3h_inflate.log 2>&1 || { tail -30 gpurun_out/r3h_inflate.log; exit 1; }
tail -2 gpurun_out/r3h_inflate.log
timeout -k 10 200 python -u tools/time_inflate.py text 256 3 || exit 1
timeout -k 10 200 python -u tools/time_inflate.py mixed 1024 2 || exit 1
rm -rf gpurun_out/r3h/*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3h -o inf -- python3 tools/time_inflate.py mixed 1024 2 > gpurun_out/r3h/run.log 2>&1 || { tail gpurun_out/r3h/run.log; exit 1; }
python3 - <<'PY'
import sqlite3
c = sqlite3.connect('gpurun_out/r3h/inf_results.db')
for r in c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6 from kernels group by name order by 3 desc limit 8"):
    print(r[0][:50], r[1], round(r[2], 2), round(r[3], 3))
PY
