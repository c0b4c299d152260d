"""Experiment (not the contract bench): BASELINE configs[1] with TWO contexts in flight.  Each step is the
same full pass as bench.py's (hash -> dedup -> compress of 100k x 256 KiB chunks resident in HBM); two host
threads drive one sq_ctx each (own stream, own buffers, own output slots), so the low-occupancy tail of one
context's stage 2 overlaps the other's hash / stage 1.  Kernel durations measured under this overlap are not
comparable with the roofline numbers of bench.py, which is why bench.py keeps one context.
usage: python tools/bench_inflight.py [chunks] [steps per context] [contexts]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus_gpu
import squishrs_amd
from squishrs_amd.parallel import ShardedPacker

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = 262144
data = corpus_gpu.random_chunks(1, 0, n, L, device="cuda", group=16384)
offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
lens = np.full(n, L, dtype=np.uint32)
ctxs = []
for k in range(nctx):
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
    ctxs.append({"store": st, "packer": ShardedPacker(st),
                 "hashes": torch.zeros(n * 16, dtype=torch.uint8, device="cuda"),
                 "is_new": torch.zeros(n, dtype=torch.uint8, device="cuda"),
                 "comp": torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda"), "res": None})
torch.cuda.synchronize()


def run(c, k):
    for _ in range(k):
        c["store"].reset()
        c["res"] = c["packer"].pack(data, offs, lens, c["hashes"], c["is_new"], c["comp"], layout="slots")


for c in ctxs:
    run(c, 1)  # warm-up, one context at a time
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(c, steps)) for c in ctxs]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tot = steps * nctx
print(f"{nctx} contexts in flight: {tot} steps in {dt * 1e3:.1f} ms = {dt / tot * 1e3:.2f} ms per step, "
      f"{tot * n * L / 2**20 / dt:.0f} MiB/s", flush=True)
ref = [int(x) for x in ctxs[0]["res"][1][:8]]
for c in ctxs:
    off, ln = c["res"]
    assert int(ln.astype(np.uint64).sum()) == n * 262159, "a frame is not the raw-block frame expected for random data"
    assert [int(x) for x in ln[:8]] == ref
    c["store"].close()
print("all frame lengths as expected (262,159 bytes each)")
