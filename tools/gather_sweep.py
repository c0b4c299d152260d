"""Random-request roof of the memory system: sq_calibrate_gather2 swept over request size (16 / 32 / 64
bytes) x requests in flight per lane (1 .. 16) x resident waves per CU x buffer size (inside L2's 32 MiB, inside the
256 MiB Infinity Cache, far beyond it).  Prints a table (G requests / s and the line rate in GB/s at 64 B per request
line).  Run on the GPU box: python tools/gather_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import squishrs_amd

st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1024)
buf = torch.empty(16 << 30, dtype=torch.uint8, device="cuda")
buf.random_(0, 256)
print("buffer        req  depth waves/CU   G req/s   sector GB/s (64 B per request)")
for name, nbytes in (("16 MiB (L2)", 16 << 20), ("128 MiB (Infinity Cache)", 128 << 20), ("4 GiB", 4 << 30), ("16 GiB", 16 << 30)):
    for req in (16, 32, 64):
        for depth in (1, 4, 8, 16):
            for w in (8, 32):
                if depth in (1, 16) and w == 8:
                    continue
                r = max(st.gather_rate2(buf, nbytes, req, depth, w, per_lane=256 if depth > 1 else 64) for _ in range(2))
                print(f"{name:24s} {req:4d} {depth:5d} {w:7d} {r / 1e9:9.1f} {r * 64 / 1e9:10.0f}", flush=True)
st.close()
