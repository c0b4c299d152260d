#!/bin/bash
# A/B of builds of the library on the headline AND the configs[2] / configs[3] legs (experiment tooling).
# usage: tools/ab_full.sh out_tag lib1.so lib2.so ...
set -e
tag=$1; shift
: > gpurun_out/abfull_$tag.txt
for lib in "$@"; do
  SQUISH_GPU_LIB=$lib timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --config3-chunks 2048 --config3-steps 1 > gpurun_out/ab_line.json 2> gpurun_out/ab_err.txt
  python - "$lib" >> gpurun_out/abfull_$tag.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read())
r = d["roofline"]; s = r["solo_step"]; o = d["other_configs"]
print(f"{sys.argv[1]}: step {d['ms_per_step']:.2f} ms; solo stage1 {s['k_incompressible_batch_ms']:.2f} stage2 {s['k_compress_batch_ms']:.2f} hash {s['hash_ms']:.2f}; "
      f"configs[2] {o['configs[2]']['ms_per_step']:.2f} ms; configs[3] sample {o['configs[3] sample']['ms_per_step']:.0f} ms = {o['configs[3] sample']['MiB_per_s']:.0f} MiB/s")
PY
done
cat gpurun_out/abfull_$tag.txt
