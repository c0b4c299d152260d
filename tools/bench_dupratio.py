"""Which form of a batch is faster at which duplicate ratio (experiment tooling; picks the threshold of sq_api.hip's
pack_enqueue): n x 256 KiB random chunks whose last d * n are exact copies of earlier ones, two batches in flight, the
fused form (stage 1 before the dedup decision, hashing what it streams) against the split form.
usage: python tools/bench_dupratio.py [chunks] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus_gpu
import squishrs_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L = 262144
base = corpus_gpu.random_chunks(3, 0, n, L, device="cuda", group=16384)
offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
lens = np.full(n, L, dtype=np.uint32)
for d in (0.0, 0.1, 0.15, 0.2, 0.25, 0.3):
    data = base.clone()
    nd = int(n * d)
    if nd:
        src = torch.from_numpy(np.random.default_rng(5).integers(0, n - nd, size=nd)).cuda()
        v = data.view(n, L)
        for s in range(0, nd, 8192):
            v[n - nd + s:n - nd + s + 8192] = v.index_select(0, src[s:s + 8192])
    res = {}
    for fuse in (True, False):
        st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024, fuse_hash=fuse)
        bufs = [(torch.zeros(n * 16, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"),
                 torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")) for _ in range(2)]
        for k in range(2):
            st.pack_submit(data, offs, lens, *bufs[k & 1], reset=True)
        for k in range(2):
            st.pack_wait(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if k >= 2:
                st.pack_wait(n)
            st.pack_submit(data, offs, lens, *bufs[k & 1], reset=True)
        st.pack_wait(n)
        st.pack_wait(n)
        torch.cuda.synchronize()
        res[fuse] = (time.perf_counter() - t0) / steps * 1e3
        assert st.len() == n - nd or nd == 0 or st.len() <= n - nd + 1, (st.len(), n - nd)
        st.close()
        del bufs
    print(f"duplicates {d:.2f}: fused {res[True]:.2f} ms per step, split {res[False]:.2f} ms per step ({n} chunks)", flush=True)
    del data
