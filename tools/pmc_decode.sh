#!/bin/bash
# SQ counters of the decode kernel (sqdz::k_decompress_batch) on frames of a corpus kind (experiment tooling; GPU box).
# usage: tools/pmc_decode.sh tag [chunks] [kind] [bytes]   -- one rocprofv3 --pmc pass per counter group, release library,
# tools/bench_decode.py as the workload (it packs the chunks on the GPU, then decodes every frame twice; the LAST decode launch is summed).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_decode_${1:-a}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAVES"; do
  echo "pass $i: $SET" >> $O/progress.log
  SQ_NO_FORK=1 timeout -k 10 150 rocprofv3 --pmc $SET --kernel-include-regex "sqdz::k_decompress" --output-format csv -d $O/p$i -- python3 $R/tools/bench_decode.py ${2:-2048} 32 ${3:-mixed} ${4:-2097152} > $O/p$i.log 2> $O/p$i.err
  echo "pass $i rc=$?" >> $O/progress.log
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    if not rows: continue
    last=max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"])==last: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
grep -h "decoded in" $O/p*.log
cat $O/progress.log
