#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun).  $1 = tag.  Kernel trace + stats of
# the default bench line, then FETCH_SIZE and WRITE_SIZE in two separate counter passes (never
# combined with a trace) on PMC_CHUNKS (default 32768) chunks of the same workload, one launch each;
# only this library's kernels are instrumented (the synthetic-data generator launches ~10^5 tiny ones).
set -e
TAG=${1:-r1}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG/bench_under_trace.json 2> $R/gpurun_out/prof_$TAG/trace.err || true
echo "trace done" >> $R/gpurun_out/prof_$TAG/progress.log
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_fetch -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_$TAG/bench_pmc_fetch.json 2> $R/gpurun_out/prof_$TAG/pmc_fetch.err || true
echo "fetch done" >> $R/gpurun_out/prof_$TAG/progress.log
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $R/gpurun_out/prof_$TAG/pmc_write -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof_$TAG/bench_pmc_write.json 2> $R/gpurun_out/prof_$TAG/pmc_write.err || true
echo "write done" >> $R/gpurun_out/prof_$TAG/progress.log
find $R/gpurun_out/prof_$TAG -name "*.csv" | head -30
