#!/bin/bash
# Memory-side counters (vector L1 address translation, L2 hits / misses, requests to the fabric) of the compress stage on chunks of
# a corpus kind and size (experiment tooling; GPU box).  usage: tools/pmc_mem_kind.sh tag [chunks] [kind] [bytes]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mem_${1:-a}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  echo "pass $i: $SET" >> $O/progress.log
  SQ_NO_FORK=1 timeout -k 10 200 rocprofv3 --pmc $SET --kernel-include-regex "sqz::k_compress" --output-format csv -d $O/p$i -- python3 $R/tools/prof_mixed.py ${2:-2048} 64 ${3:-text} ${4:-262144} > $O/p$i.log 2> $O/p$i.err
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
cat $O/progress.log
