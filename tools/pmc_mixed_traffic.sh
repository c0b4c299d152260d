#!/bin/bash
# HBM traffic of the compress kernel on 2 MiB mixed / text chunks (rocprofv3 FETCH_SIZE and WRITE_SIZE in separate
# passes, release library, tools/prof_mixed.py: three launches on the same lane, the LAST one = steady state).  GPU box.
# Writes gpurun_out/pmc_mixed_traffic/summary.json (copy to profiles/pmc_traffic_mixed.json: bench.py quotes it in the
# configs[3] leg as traffic_ratio = (fetched + written) / algorithmic bytes).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mixed_traffic
mkdir -p $O
N=${PMC_MIXED_CHUNKS:-2048}
cd /tmp && export TMPDIR=/tmp
for kind in ${PMC_MIXED_KINDS:-mixed text}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${kind}_$C
    echo "$kind $C ..." >> $O/progress.log
    SQ_NO_FORK=1 PROF_WARMUPS=2 timeout -k 10 300 rocprofv3 --pmc $C --kernel-include-regex "sqz::k_compress" --output-format csv -d $O/${kind}_$C -- python3 $R/tools/prof_mixed.py $N 64 $kind > $O/${kind}_$C.log 2> $O/${kind}_$C.err
    echo "$kind $C rc=$?" >> $O/progress.log
  done
done
python3 - <<PY
import csv,glob,re,json
out={}
for kind in "${PMC_MIXED_KINDS:-mixed text}".split():
    vals={}
    for C in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv"%(kind,C), recursive=True):
            rows=[r for r in csv.DictReader(open(f)) if r["Counter_Name"]==C]
            last=max(int(r["Dispatch_Id"]) for r in rows)
            vals[C]=sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last)
    ms=ratio=None
    for line in open("$O/%s_FETCH_SIZE.log"%kind):
        m=re.search(r"stage 2 ([0-9.]+) ms", line)
        if m: ms=float(m.group(1))
        m=re.search(r"ratio ([0-9.]+)", line)
        if m: ratio=float(m.group(1))
    n=$N; L=2<<20
    alg=n*L*(1+(ratio or 0))
    tr=(vals.get("FETCH_SIZE",0)+vals.get("WRITE_SIZE",0))*1024
    out[kind]={"chunks":n,"chunk_bytes":L,"FETCH_SIZE_KB":vals.get("FETCH_SIZE"),"WRITE_SIZE_KB":vals.get("WRITE_SIZE"),
               "kernel_ms_under_counters":ms,"compression_ratio":ratio,"algorithmic_bytes":alg,"traffic_bytes_uncorrected":tr,
               "traffic_ratio":tr/alg if alg else None,
               "note":"k_compress_batch, last (steady-state) launch; FETCH_SIZE as reported (random sector reads are counted in full; only wide coalesced streams are half-counted on gfx950, a minor part here)"}
    print(kind, json.dumps(out[kind]))
json.dump(out, open("$O/summary.json","w"), indent=1)
PY
cat $O/progress.log
