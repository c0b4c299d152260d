"""One-off GPU validation at scale of the hand-over paths: N incompressible 256 KiB chunks with planted repeats
(every variant of corpus.planted, many seeds; aligned starts) plus the simulator fuzz's multi-coincidence chunks,
through the two-stage compress path, EVERY frame compared with the oracle.
usage: python tools/gpu_planted_check.py [chunks]"""
import os
import random
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = 262144
rng = random.Random(5)
sched = corpus.lazy_schedule(L)
chunks = []
t0 = time.time()
for i in range(n):
    if i % 3:
        chunks.append(corpus.planted(1000 + i, L, i % 10))
    else:  # several short coincidences, many in the dense start of block 1
        b = bytearray(corpus.random_bytes(78, i, L))
        for k in range(rng.randrange(1, 7)):
            t = rng.choice(sched[len(sched) // 8:]) + rng.choice([0, 0, 0, 1, -1])
            if rng.random() < 0.5:
                t = rng.choice([x for x in sched if 131072 <= x < 131072 + 4096]) + rng.choice([0, 0, 1])
            ln = rng.choice([3, 4, 4, 4, 5, 6, 8, 20])
            if t < 100 or t + ln > L:
                continue
            src = rng.randrange(2, t - ln - 1)
            b[t:t + ln] = b[src:src + ln]
        chunks.append(bytes(b))
print(f"generated {n} chunks in {time.time() - t0:.1f} s", flush=True)
o = oracle_lib.load()
st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
data = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
lens = np.full(n, L, dtype=np.uint32)
comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
for rep in range(2):  # the second pass runs on persisted table generations
    off, ln = st.compress_device_slots(data, offs, lens, None, comp)
    t = st.timings()
    host = comp.cpu().numpy()
    bad = 0
    for i in range(n):
        fr = host[int(off[i]):int(off[i]) + int(ln[i])].tobytes()
        if fr != o.compress_chunk(chunks[i]):
            bad += 1
            if bad < 5:
                print("MISMATCH chunk", i, "variant", i % 10 if i % 3 else "multi", flush=True)
    print(f"pass {rep}: {n} frames checked, {bad} differ; stage 1 decided {t.fast_decided}, "
          f"stage 1 {t.fast_ms:.2f} ms, stage 2 {t.compress_ms:.2f} ms", flush=True)
    assert bad == 0
st.close()
