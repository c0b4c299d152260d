set -e
mkdir -p gpurun_out/r4n
for w in 8 4 2 3 6; do
  SQ_STAGE2_WAVES_PER_CU=$w timeout -k 10 240 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r4n/s2w_$w.json 2> gpurun_out/r4n/s2w_$w.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r4n/s2w_$w.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("s2 waves/CU $w: step %.2f ms  s1 timed %.2f  s2 timed %.2f  solo s1 %.2f s2 %.2f" % (d["ms_per_step"], r["compress_stage"]["k_incompressible_batch_ms"], r["compress_stage"]["k_compress_batch_ms"], r["solo_step"]["k_incompressible_batch_ms"], r["solo_step"]["k_compress_batch_ms"]), flush=True)
P
done
