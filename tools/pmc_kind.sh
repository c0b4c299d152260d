#!/bin/bash
# SQ / instruction-cache counters of the compress stage on chunks of a corpus kind and size ($2 chunks, $3 kind, $4 bytes; default 2,048 x 256 KiB text) (experiment tooling; GPU box).
# One rocprofv3 --pmc pass per counter group, release library, tools/prof_mixed.py as the workload.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_kind_${1:-a}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"; do
  echo "pass $i: $SET" >> $O/progress.log; SQ_NO_FORK=1 timeout -k 10 200 rocprofv3 --pmc $SET --kernel-include-regex "sqz::k_compress" --output-format csv -d $O/p$i -- python3 $R/tools/prof_mixed.py ${2:-2048} 64 ${3:-text} ${4:-262144} > $O/p$i.log 2> $O/p$i.err
  echo "pass $i rc=$?" >> $O/progress.log
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    last=max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"])==last: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
cat $O/progress.log
