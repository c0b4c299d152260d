#!/bin/bash
# stage-1 timing experiments on the headline workload (SQ_FAST_DBG switches of sq_zstd_fast.h); experiment tooling.
# usage: tools/ab_dbg.sh out_tag dbg1 dbg2 ...
set -e
tag=$1; shift
: > gpurun_out/abdbg_$tag.txt
for d in "$@"; do
  SQ_FAST_DBG=$d timeout -k 10 300 python bench.py --chunks 32768 --steps 2 --warmup 1 --no-pipeline --no-other-configs --no-cpu-baseline --no-frame-check > gpurun_out/ab_line.json 2> gpurun_out/ab_err.txt
  python - "$d" >> gpurun_out/abdbg_$tag.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read())
r = d["roofline"]
print(f"SQ_FAST_DBG={sys.argv[1]}: step {d['ms_per_step']:.2f} ms; stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f} stage2 {r['compress_stage']['k_compress_batch_ms']:.2f} decided {r['compress_stage']['chunks_decided_by_stage1']}")
PY
done
cat gpurun_out/abdbg_$tag.txt
