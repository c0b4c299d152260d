# Round 4h: the lanes' second streams at the highest stream priority (SQ_AUX_PRIO=1) under the early-stage-2 order, against the default order.
set -e
mkdir -p gpurun_out/r4p
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-other-configs --no-cpu-baseline > gpurun_out/r4p/$tag.json 2> gpurun_out/r4p/$tag.err
python - "$tag $*" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4p/%s.json" % sys.argv[1].split()[0]).read().strip().splitlines()[-1])
r = d["roofline"]; s = r["solo_step"]
print(f"{sys.argv[1]}: step {d['ms_per_step']:.2f} ms frac {r['whole_step']['frac']:.3f}; timed stage1 {r['compress_stage']['k_incompressible_batch_ms']:.2f} stage2 {r['compress_stage']['k_compress_batch_ms']:.2f} hash {r['hash_kernel']['ms']:.2f}", flush=True)
PY
}
run default SQ_DUMMY=1
run s2_early SQ_S2_EARLY=1
run s2_early_prio SQ_S2_EARLY=1 SQ_AUX_PRIO=1
run s2_early_prio_noafter SQ_S2_EARLY=1 SQ_AUX_PRIO=1 SQ_S1_AFTER_S2=0
run default2 SQ_DUMMY=1
