#!/bin/bash
# Round 4 rocprofv3 evidence for bench.py (GPU box, via gpurun).  $1 = tag.  Passes (each its own run; counters never with a trace):
#   trace       kernel trace + stats of the DEFAULT bench command (two batches in flight)
#   trace_solo  the same workload one batch at a time (--no-pipeline)
#   pmc_fetch / pmc_write   FETCH_SIZE / WRITE_SIZE on PMC_CHUNKS chunks, --no-pipeline --warmup 2 --steps 2:
#               sq_pack_submit alternates the context's two lanes, so two warm-up steps touch BOTH lanes' table pools
#               and the two counted steps are one steady-state launch per lane (round 3 counted lane 1's first launch,
#               i.e. its 2 MiB-per-wave table clears); tools/pmc_summary.py averages each kernel's last two launches.
# Only this library's kernels are instrumented.  Every pass's exit code goes to progress.log; a failed pass stops the script.
set -u
TAG=${1:-round4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; local rc=$?; echo "$name rc=$rc" >> $O/progress.log
        if [ $rc -ne 0 ] || [ ! -s $O/$name.json ]; then echo "pass $name failed (rc=$rc)"; tail -5 $O/$name.err; exit 1; fi; }
rm -rf $O/trace $O/trace_solo $O/pmc_fetch $O/pmc_write
PMC="--chunks ${PMC_CHUNKS:-32768} --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs --no-pipeline"
if [ -z "${SKIP_TRACE:-}" ]; then
run bench_under_trace rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-configs
run bench_under_trace_solo rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_solo -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-pipeline
fi
run bench_pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PMC
run bench_pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PMC
cat $O/progress.log
