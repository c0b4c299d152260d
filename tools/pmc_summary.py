"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes written by tools/profile_r1.sh into
profiles/<tag>_pmc.json (per kernel, as reported) and refresh profiles/pmc_traffic.json, the
per-launch HBM traffic of k_compress_batch that bench.py quotes in roofline.traffic.
usage: python tools/pmc_summary.py <tag> <chunks> <chunk_bytes> <launches in the profiled run>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n, L, launches = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
               f"`python3 bench.py --chunks {n} --chunk-bytes {L} --steps {launches} --warmup 0 --no-cpu-baseline`; "
               "counter values are KB as reported, summed over the run's launches.  gfx950: FETCH_SIZE counts a "
               "wide coalesced streaming read at half its bytes (MI355X_MICROARCH.md, HBM section); the raw "
               "values are kept here and the correction is applied where a kernel is a pure streaming read.",
       "kernels": collections.defaultdict(dict)}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", f"pmc_{kind}", "**", "*counter_collection.csv"),
                      recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("sq") and r["Counter_Name"] == cname:
                out["kernels"][k][cname] = out["kernels"][k].get(cname, 0.0) + float(r["Counter_Value"])
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
ck = out["kernels"].get("sqz::k_compress_batch")
if ck and "FETCH_SIZE" in ck and "WRITE_SIZE" in ck:
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    t = json.load(open(p)) if os.path.exists(p) else {}
    t[f"{n}x{L}"] = {"fetch_bytes": ck["FETCH_SIZE"] * 1024 / launches, "write_bytes": ck["WRITE_SIZE"] * 1024 / launches,
                     "source": f"profiles/{tag}_pmc.json", "launches": launches,
                     "note": "per k_compress_batch launch; FETCH_SIZE uncorrected (the kernel mixes streaming reads, "
                             "counted at half, with 4-8 byte gathers)"}
    json.dump(t, open(p, "w"), indent=1)
    print(json.dumps(t[f"{n}x{L}"]))
