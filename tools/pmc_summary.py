"""Fold the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes written by tools/profile_r2.sh into profiles/<tag>_pmc.json
(per kernel, as reported) and refresh profiles/pmc_traffic.json, the per-launch HBM traffic bench.py quotes in
roofline.traffic.  For every kernel the LAST TWO launches of the profiled run are averaged: bench.py's steps alternate
the context's two lanes, the run warms both (--warmup 2) and counts one steady-state launch per lane (--steps 2); every
launch's value is kept in the json ("launches") so that a cold one would show.  Also copies the kernel-stats CSVs of
the two trace passes into profiles/.
usage: python tools/pmc_summary.py <tag> <chunks> <chunk_bytes> [name under profiles/, default = tag]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
name_out = sys.argv[4] if len(sys.argv) > 4 else tag
O = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
for name in ("bench_pmc_fetch", "bench_pmc_write"):
    p = os.path.join(O, name + ".json")
    assert os.path.exists(p) and os.path.getsize(p) > 0, f"{p} is missing or empty: that counter pass failed"
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
               f"`python3 bench.py --chunks {n} --chunk-bytes {L} --steps 2 --warmup 2 --no-pipeline --no-other-configs "
               "--no-cpu-baseline`; per kernel the mean of the LAST TWO launches (one per lane of the context, both lanes "
               "warmed by the two warm-up steps: steady state); `launches` lists every launch in dispatch order; counter values are KB as reported.  "
               "gfx950: FETCH_SIZE counts a wide coalesced streaming read at half its bytes (MI355X_MICROARCH.md, HBM "
               "section); raw values here, the correction is applied in profiles/pmc_traffic.json / bench.py.",
       "kernels": {}}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(O, f"pmc_{kind}", "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter CSV for {kind}"
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("sq") and r["Counter_Name"] == cname:
                d = int(r["Dispatch_Id"])
                per.setdefault(k, {})
                per[k][d] = per[k].get(d, 0.0) + float(r["Counter_Value"])   # (a dispatch may be reported per XCD / agent row)
    for k, dv in per.items():
        vals = [dv[d] for d in sorted(dv)]
        tail = vals[-2:] if len(vals) >= 4 else vals[-1:]
        out["kernels"].setdefault(k, {})[cname] = sum(tail) / len(tail)
        out["kernels"][k].setdefault("launches", {})[cname] = vals
json.dump(out, open(os.path.join(ROOT, "profiles", f"{name_out}_pmc.json"), "w"), indent=1)
# streaming-read bytes per chunk (one 16 B/lane pass over the chunk), the part FETCH_SIZE tallies at half:
# (in the fused form of a batch -- bench.py's default on this workload -- the hash kernel reads stage 1's block sums, not the chunks)
fused = False
try:
    fused = json.load(open(os.path.join(O, "bench_pmc_fetch.json")))["roofline"]["form"].startswith("fused")
except (OSError, ValueError, KeyError):
    pass
STREAM = {"sqz::k_incompressible_batch": L, "sqx::k_xxh3_128_rows": 0 if fused else L}
t = {}
for k, ck in out["kernels"].items():
    if "FETCH_SIZE" in ck and "WRITE_SIZE" in ck:
        t[f"{k}@{L}"] = {"fetch_bytes_per_chunk": ck["FETCH_SIZE"] * 1024 / n,
                         "write_bytes_per_chunk": ck["WRITE_SIZE"] * 1024 / n,
                         "stream_read_bytes_per_chunk": STREAM.get(k, 0),
                         "measured_on_chunks": n, "source": f"profiles/{name_out}_pmc.json",
                         "note": "per chunk of the BATCH (stage 2 handles the chunks stage 1 leaves undecided: 0.7 % of random ones since round 4); corrected traffic = fetch + "
                                 "stream_read/2 + write (FETCH_SIZE tallies a coalesced streaming read at half on gfx950)"}
        print(k, json.dumps(t[f"{k}@{L}"]))
json.dump(t, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for sub, suffix in (("trace", "bench_kernel_stats.csv"), ("trace_solo", "bench_solo_kernel_stats.csv")):
    fs = glob.glob(os.path.join(O, sub, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(ROOT, "profiles", f"{name_out}_{suffix}"))
for name in ("bench_under_trace", "bench_under_trace_solo"):
    p = os.path.join(O, name + ".json")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(ROOT, "profiles", f"{name_out}_{name}_line.json"))
