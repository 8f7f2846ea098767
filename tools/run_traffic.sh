#!/bin/bash
# usage: tools/run_traffic.sh TAG WORKLOAD -- FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), 1 step
set -e
TAG=$1; WL=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traf_${TAG}_f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --workload $WL > gpurun_out/traf_${TAG}_f.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traf_${TAG}_w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --workload $WL > gpurun_out/traf_${TAG}_w.log 2>&1
