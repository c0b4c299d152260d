#!/bin/bash
# rocprofv3 evidence for bench.py, round 2 (run on the GPU box via gpurun).  $1 = tag.
#   1. kernel trace + stats of the DEFAULT bench line (two batches in flight)            -> trace/
#   2. kernel trace + stats of the same workload with one batch at a time (--no-pipeline) -> trace_solo/
#   3. FETCH_SIZE and 4. WRITE_SIZE in two separate counter passes (never combined with a trace) on PMC_CHUNKS
#      chunks of the same workload, --warmup 1 --steps 1 --no-pipeline: tools/pmc_summary.py takes the LAST launch of
#      every kernel, i.e. the steady state (table pools already cleared and tagged by the warm-up launch).
# Only this library's kernels are instrumented.  Every pass's exit code goes to progress.log; a failed pass stops
# the script (round 1 hid a crashed pass behind `|| true`, see DESIGN.md "Incidents").
set -u
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, then the command
    local name=$1; shift
    "$@" > $O/$name.json 2> $O/$name.err
    local rc=$?
    echo "$name rc=$rc" >> $O/progress.log
    if [ $rc -ne 0 ] || [ ! -s $O/$name.json ]; then echo "pass $name failed (rc=$rc)"; tail -5 $O/$name.err; exit 1; fi
}
run bench_under_trace rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-configs
run bench_under_trace_solo rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_solo -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-pipeline
run bench_pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-pipeline
run bench_pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_write -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-pipeline
find $O -name "*.csv" | head -30
