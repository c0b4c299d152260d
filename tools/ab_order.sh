#!/bin/bash
# A/B on ONE box: stage 1 of batch k+1 ordered behind the launch of batch k's stage 2 (default) or not (SQ_S1_AFTER_S2=0),
# three runs each, alternating (experiment tooling; usage through gpurun: bash tools/ab_order.sh).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_order
mkdir -p $O
for i in 1 2 3; do
  for v in 1 0; do
    SQ_S1_AFTER_S2=$v python3 $R/bench.py --no-cpu-baseline --no-other-configs > $O/o${v}_$i.json 2> $O/o${v}_$i.err || { echo "order=$v run $i FAILED"; exit 1; }
    python3 - $v $i $O/o${v}_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[3])); r=d["roofline"]
print(f"order={sys.argv[1]} run {sys.argv[2]}: step {d['ms_per_step']:.2f} ms, whole-step {r['whole_step']['frac']:.3f}; timed stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f} (frac {r['frac']:.3f}) stage2 {r['compress_stage']['k_compress_batch_ms']:.2f}", flush=True)
PY
  done
done
