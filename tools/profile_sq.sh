#!/bin/bash
# SQ counters of the two compress-stage kernels (separate --pmc passes, never combined with a trace), on
# PMC_CHUNKS (default 32768) chunks of the bench workload, one launch each.  $1 = tag.
set -e
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "sqz::" --output-format csv -d $R/gpurun_out/sq_$TAG/p$i -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/sq_$TAG/bench_p$i.json 2> $R/gpurun_out/sq_$TAG/p$i.err || true
  echo "pass $i done" >> $R/gpurun_out/sq_$TAG/progress.log
done
find $R/gpurun_out/sq_$TAG -name "*counter_collection.csv" | head
