#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh OUTDIR -- bench lines, kernel-trace stats, SQ counters and HBM
# traffic counters of bench.py for the workloads the design document quotes; one rocprofv3 run per counter set.
set -e
OUT=$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-verify --no-others"
echo "== the default bench line"
timeout -k 10 560 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for WL in mixed8g text64m raw4g records4g text256m binary256m random256m; do
  echo "== bench line $WL"
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-others --steps 5 --warmup 1 --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
for WL in mixed8g text64m raw4g records256m records4g; do
  echo "== stats $WL"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o r -- $B --steps 3 --warmup 1 --workload $WL > $OUT/stats_$WL.log 2>&1
done
for WL in mixed8g text64m raw4g records4g; do
  echo "== traffic $WL"
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$WL -- $B --steps 1 --warmup 0 --workload $WL > $OUT/fetch_$WL.log 2>&1
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$WL -- $B --steps 1 --warmup 0 --workload $WL > $OUT/write_$WL.log 2>&1
done
for WL in mixed8g text64m records4g; do
  echo "== sq $WL"
  timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sqa_$WL -- $B --steps 1 --warmup 0 --workload $WL > $OUT/sqa_$WL.log 2>&1
  timeout -s KILL 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sqb_$WL -- $B --steps 1 --warmup 0 --workload $WL > $OUT/sqb_$WL.log 2>&1
done
echo done
