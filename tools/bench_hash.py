"""Quick hash-stage microbenchmark (GPU box): XXH3-128 kernel GB/s vs the read-bandwidth probe."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import squishrs_amd  # noqa: E402

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 262144
st = squishrs_amd.ChunkStore(device=0)
data = torch.randint(0, 256, (n * L,), dtype=torch.uint8, device="cuda")
offs = (np.arange(n, dtype=np.uint64) * L)
lens = np.full(n, L, dtype=np.uint32)
h = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
for _ in range(3):
    st.hash_device(data, offs, lens, h)
ts = []
for _ in range(10):
    st.hash_device(data, offs, lens, h)
    ts.append(st.timings().hash_ms)
ms = float(np.median(ts))
bw = [st.read_bw_ms(data) for _ in range(5)]
print(f"hash: n={n} L={L} {ms:.3f} ms -> {n * L / ms / 1e6:.1f} GB/s ; read probe {n * L / min(bw) / 1e6:.1f} GB/s")
