#!/bin/bash
# A/B of builds of the library on the headline workload (experiment tooling): one bench line per library given.
# usage: tools/ab_libs.sh out_tag lib1.so lib2.so ...   (paths relative to the repo root)
set -e
tag=$1; shift
mkdir -p gpurun_out
: > gpurun_out/ab_$tag.txt
for lib in "$@"; do
  SQUISH_GPU_LIB=$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/ab_line.json 2> gpurun_out/ab_err.txt
  python - "$lib" >> gpurun_out/ab_$tag.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read())
r = d["roofline"]; s = r["solo_step"]
print(f"{sys.argv[1]}: step {d['ms_per_step']:.2f} ms; solo stage1 {s['k_incompressible_batch_ms']:.2f} stage2 {s['k_compress_batch_ms']:.2f} hash {s['hash_ms']:.2f}; "
      f"timed stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f}; frac {r['frac']:.3f}")
PY
done
cat gpurun_out/ab_$tag.txt
