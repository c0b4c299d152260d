#!/bin/bash
# HBM traffic of the compress kernel on 2 MiB mixed / text chunks (rocprofv3 FETCH_SIZE and WRITE_SIZE in separate
# passes, release library, tools/prof_mixed.py; last launch = steady state).  GPU box.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mixed_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kind in mixed text; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-include-regex "sqz::k_compress" --output-format csv -d $O/${kind}_$C -- python3 $R/tools/prof_mixed.py 2048 64 $kind > $O/${kind}_$C.log 2> $O/${kind}_$C.err
    echo "$kind $C rc=$?" >> $O/progress.log
  done
done
python3 - <<PY
import csv,glob,re
for kind in ("mixed","text"):
    vals={}
    for C in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv"%(kind,C), recursive=True):
            rows=[r for r in csv.DictReader(open(f)) if r["Counter_Name"]==C]
            last=max(int(r["Dispatch_Id"]) for r in rows)
            vals[C]=sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last)
    ms=None
    for line in open("$O/%s_FETCH_SIZE.log"%kind):
        m=re.search(r"stage 2 ([0-9.]+) ms", line)
        if m: ms=float(m.group(1))
    print(kind, "FETCH_SIZE KB", vals.get("FETCH_SIZE"), "WRITE_SIZE KB", vals.get("WRITE_SIZE"), "kernel ms (under counters)", ms)
PY
cat $O/progress.log
