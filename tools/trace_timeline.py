"""Timeline of the headline step from a rocprofv3 kernel trace (csv): for the last steps of the timed region, when each kernel of
this library started and ended relative to the first stage-1 launch listed, and the time in which NO stage-1 kernel was running.
usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [steps to show]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
show = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"]
        if "sq" not in name:
            continue
        short = name.split("(")[0].split("::")[-1]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
s1 = [r for r in rows if r[2].startswith("k_incompressible_batch")]
tail = s1[-(show + 1):]
t0 = tail[0][0]
print(f"{len(rows)} launches of this library's kernels, {len(s1)} of stage 1; the last {show} steps, ms from the start of stage 1 of the first:")
for st, en, nm in rows:
    if st >= t0:
        print(f"  {nm:28s} {(st - t0) / 1e6:9.3f} -> {(en - t0) / 1e6:9.3f}   ({(en - st) / 1e6:7.3f} ms)")
gaps = []
for a, b in zip(tail, tail[1:]):
    gaps.append((b[0] - a[1]) / 1e6)
per = [(b[0] - a[0]) / 1e6 for a, b in zip(tail, tail[1:])]
print("start-to-start of consecutive stage-1 launches, ms:", [round(x, 3) for x in per])
print("end of one stage-1 launch to the start of the next, ms (negative: they overlap):", [round(x, 3) for x in gaps])
