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
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = {}
for k, ck in out["kernels"].items():
    if "FETCH_SIZE" in ck and "WRITE_SIZE" in ck and k.startswith("sqz::"):
        t[f"{k}@{L}"] = {"fetch_bytes_per_chunk": ck["FETCH_SIZE"] * 1024 / launches / n,
                         "write_bytes_per_chunk": ck["WRITE_SIZE"] * 1024 / launches / n,
                         "measured_on_chunks": n, "source": f"profiles/{tag}_pmc.json",
                         "note": "FETCH_SIZE uncorrected (streaming reads are counted at half on gfx950; these "
                                 "kernels mix them with 4-8 byte gathers and 8-byte atomics)"}
        print(k, json.dumps(t[f"{k}@{L}"]))
json.dump(t, open(p, "w"), indent=1)
