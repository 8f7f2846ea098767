#!/bin/bash
# usage: tools/run_stats.sh TAG [bench args...] -- rocprofv3 kernel trace + stats of bench.py into gpurun_out/stats_TAG
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG} -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify "$@" > gpurun_out/stats_${TAG}.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/stats_${TAG}/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -1 gpurun_out/stats_${TAG}.log | cut -c1-200
