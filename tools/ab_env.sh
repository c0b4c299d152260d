#!/bin/bash
# A/B of one environment switch on the headline bench line (experiment tooling).
# usage: tools/ab_env.sh out_tag VAR value1 value2 ...   ("-" = unset)
set -e
tag=$1; var=$2; shift 2
: > gpurun_out/abenv_$tag.txt
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/ab_line.json 2> gpurun_out/ab_err.txt
  python - "$var=$v" >> gpurun_out/abenv_$tag.txt <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_line.json").read())
r = d["roofline"]; s = r["solo_step"]
print(f"{sys.argv[1]}: step {d['ms_per_step']:.2f} ms; solo stage1 {s['k_incompressible_batch_ms']:.2f} stage2 {s['k_compress_batch_ms']:.2f} hash {s['hash_ms']:.2f}; "
      f"timed stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f} stage2 {r['compress_stage']['k_compress_batch_ms']:.2f} hash {r['hash_kernel']['ms']:.2f}")
PY
done
cat gpurun_out/abenv_$tag.txt
