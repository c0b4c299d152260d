#!/bin/bash
# A/B of the fused-hash form on the GPU box: split form, fused form (depth 4), fused form (depth 3 build if present).
# usage (through gpurun): bash tools/ab_fused.sh [extra bench args]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_fused
mkdir -p $O
one() { # tag, env...
    local tag=$1; shift
    env "$@" python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 20 > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -5 $O/$tag.err; return 1; }
    python3 - $tag $O/$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]; s=r["solo_step"]
print(f"{sys.argv[1]}: step {d['ms_per_step']:.2f} ms whole-step frac {r['whole_step']['frac']:.3f}; solo stage1 {s['k_incompressible_batch_ms']:.2f} stage2 {s['k_compress_batch_ms']:.2f} hash {s['hash_ms']:.2f}; "
      f"timed stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f} stage2 {r['compress_stage']['k_compress_batch_ms']:.2f} hash {r['hash_kernel']['ms']:.2f}")
PY
}
one split SQ_FUSE_HASH=0 && one fused SQ_FUSE_HASH=1
one fused_noorder SQ_FUSE_HASH=1 SQ_S1_AFTER_S2=0
one fused_s2w6 SQ_FUSE_HASH=1 SQ_STAGE2_WAVES_PER_CU=6
one fused_s2w4 SQ_FUSE_HASH=1 SQ_STAGE2_WAVES_PER_CU=4
