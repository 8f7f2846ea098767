#!/bin/bash
# usage: tools/run_pmc.sh TAG  -- runs the two SQ counter passes on bench.py (1 step) into gpurun_out/pmc_TAG{a,b}
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${TAG}a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --workload text64m > gpurun_out/pmc_${TAG}a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_${TAG}b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --workload text64m > gpurun_out/pmc_${TAG}b.log 2>&1
