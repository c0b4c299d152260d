#!/bin/bash
# SQ counters of the two compress-stage kernels (separate --pmc passes, never combined with a trace), on
# PMC_CHUNKS (default 32768) chunks of the bench workload, steady state (--warmup 1, one batch at a time).  $1 = tag.
# Every pass's exit code is recorded; a failed or empty pass stops the script.
set -u
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "sqz::" --output-format csv -d $O/p$i -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs --no-pipeline > $O/bench_p$i.json 2> $O/p$i.err
  rc=$?
  echo "pass $i rc=$rc" >> $O/progress.log
  if [ $rc -ne 0 ] || [ ! -s $O/bench_p$i.json ]; then echo "pass $i failed (rc=$rc)"; tail -5 $O/p$i.err; exit 1; fi
done
find $O -name "*counter_collection.csv" | head
