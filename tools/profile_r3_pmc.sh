#!/bin/bash
# Round 3: the two PMC passes of tools/profile_r2.sh again (FETCH_SIZE, WRITE_SIZE, separate runs, never with a trace) plus
# the un-overlapped kernel trace, after bench.py's --no-pipeline steps went through sq_pack_submit / sq_pack_wait (the
# fused form of the batch).  $1 = tag (reuses gpurun_out/prof_$1 of profile_r2.sh).
set -u
TAG=${1:-round3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err; local rc=$?; echo "$name rc=$rc" >> $O/progress.log
        if [ $rc -ne 0 ] || [ ! -s $O/$name.json ]; then echo "pass $name failed (rc=$rc)"; tail -5 $O/$name.err; exit 1; fi; }
rm -rf $O/trace_solo $O/pmc_fetch $O/pmc_write
run bench_under_trace_solo rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_solo -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --no-pipeline
run bench_pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-pipeline
run bench_pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "sq[xzd]::" --output-format csv -d $O/pmc_write -- python3 $R/bench.py --chunks ${PMC_CHUNKS:-32768} --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-pipeline
