"""BASELINE configs[2] on one GPU (experiment, not the contract bench): 100k x 256 KiB chunks of which the second
half are exact duplicates of the first half in a seeded permutation -- the hash kernel sees every byte, the
compress stage only the 50,000 new chunks.  usage: python tools/bench_dups.py [chunks] [steps]"""
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
from squishrs_amd.parallel import ShardedPacker

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 262144
half = n // 2
data = torch.empty(n * L, dtype=torch.uint8, device="cuda")
data[: half * L] = corpus_gpu.random_chunks(2, 0, half, L, device="cuda", group=16384)
perm = torch.from_numpy(np.random.default_rng(7).permutation(half)).cuda()
first = data[: half * L].view(half, L)
for s in range(0, half, 8192):  # duplicates, in slices to bound the temporary
    idx = perm[s:s + 8192]
    data[(half + s) * L:(half + s + len(idx)) * L] = first.index_select(0, idx).reshape(-1)
offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
lens = np.full(n, L, dtype=np.uint32)
st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
packer = ShardedPacker(st)
hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
for _ in range(1):
    st.reset()
    packer.pack(data, offs, lens, hashes, is_new, comp, layout="slots")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    st.reset()
    off, ln = packer.pack(data, offs, lens, hashes, is_new, comp, layout="slots")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
t = st.timings()
nn = is_new.cpu().numpy()
assert st.len() == half and int(nn[:half].sum()) == half and int(nn[half:].sum()) == 0, "dedup: first occurrences only"
assert int((ln > 0).sum()) == half
print(f"configs[2]: {n} chunks, {half} unique: {dt * 1e3:.2f} ms per step = {n * L / 2**20 / dt:.0f} MiB/s "
      f"(hash {t.hash_ms:.2f} ms, dedup {t.dedup_ms:.3f} ms, stage 1 {t.fast_ms:.2f} ms, stage 2 {t.compress_ms:.2f} ms)", flush=True)
st.close()
