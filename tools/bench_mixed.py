"""BASELINE configs[3] in small (experiment, not the contract bench): N Silesia-style mixed-entropy chunks of
2 MiB (tools/corpus.py `mixed`, a different seed per chunk: text-like, records, runs, skewed and random regions)
through hash -> dedup -> compress on one GPU; a sample of frames is compared with the oracle, which is also
timed on one host thread.  usage: python tools/bench_mixed.py [chunks] [steps] [chunk_bytes] [kind]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import oracle_lib
import squishrs_amd
from squishrs_amd.parallel import ShardedPacker

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = int(sys.argv[3]) if len(sys.argv) > 3 else 2 << 20
kind = sys.argv[4] if len(sys.argv) > 4 else "mixed"
t0 = time.time()
chunks = [corpus.make(kind, 1000 + i, L) for i in range(n)]
print(f"generated {n} chunks in {time.time() - t0:.1f} s", flush=True)
data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
lens = np.full(n, L, dtype=np.uint32)
st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
packer = ShardedPacker(st)
hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
st.reset()
off, ln = packer.pack(data, offs, lens, hashes, is_new, comp, layout="slots")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    st.reset()
    off, ln = packer.pack(data, offs, lens, hashes, is_new, comp, layout="slots")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
t = st.timings()
o = oracle_lib.load()
tc = time.perf_counter()
k = 0
for i in (0, n // 2, n - 1):
    fr = comp[int(off[i]):int(off[i]) + int(ln[i])].cpu().numpy().tobytes()
    assert fr == o.compress_chunk(chunks[i]), f"frame {i} differs from the oracle"
    k += 1
cpu_dt = (time.perf_counter() - tc) / k
ratio = float(ln.astype(np.uint64).sum()) / (n * L)
print(f"configs[3] sample: {n} x {L} B {kind} chunks, ratio {ratio:.3f}: {dt * 1e3:.1f} ms per step = {n * L / 2**20 / dt:.0f} MiB/s "
      f"(hash {t.hash_ms:.2f} ms, stage 2 {t.compress_ms:.1f} ms); oracle on one host thread: {L / 2**20 / cpu_dt:.1f} MiB/s; "
      f"3 sampled frames bit-exact", flush=True)
st.close()
