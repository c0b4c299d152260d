"""Differential fuzz on the GPU box: many chunks of random kinds / sizes / compositions through the compress stage,
EVERY frame compared with the oracle (oracle calls on host threads; ctypes releases the GIL), then every frame decoded on the
GPU and compared with its chunk.
usage: python tools/gpu_fuzz.py [chunks] [seed]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
kinds = ["text", "records", "runs", "skewed", "mixed", "random", "zeros", "planted0", "planted3", "planted6", "planted9",
         "plantedL1", "plantedL4", "plantedL7", "random"]   # plantedL*: chunks above 256 KiB (stage 1's certificate per block pair)
edges = [1, 2, 7, 8, 9, 255, 256, 257, 16383, 16384, 16385, 40960, 65535, 65536, 65537, 131071, 131072, 131073, 131080,
         262143, 262144, 262145, 262160, 393216, 524288, 524289, 1048575, 1048576, 2097151, 2097152]


def make(i):
    r = np.random.default_rng(seed * 1000003 + i)
    style = int(r.integers(0, 10))
    if style < 2:
        size = int(r.choice(edges))
    elif style < 5:
        size = int(r.integers(262145, 2097153))
    elif style < 8:
        size = int(r.integers(16385, 262145))
    else:
        size = int(r.integers(1, 16385))
    if style % 3 == 0 and size > 5000:     # a composition: the parser changes mode every few KiB
        parts, got, j = [], 0, 0
        while got < size:
            ln = int(r.integers(500, 20000))
            parts.append(corpus.make(kinds[int(r.integers(0, 7))], seed * 77 + i * 31 + j, ln))
            got += ln
            j += 1
        return b"".join(parts)[:size]
    k = kinds[int(r.integers(0, len(kinds)))]
    if k.startswith("plantedL") and size <= 300000:
        k = "random"
    elif k.startswith("planted") and not k.startswith("plantedL") and (size < 20000 or size > 262144):
        k = "random"
    try:
        return corpus.make(k, seed * 13 + i, size)
    except IndexError:      # plantedL*: the last block is too short to hold the plant
        return corpus.make("random", seed * 13 + i, size)


t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    chunks = list(ex.map(make, range(n)))
print(f"{n} chunks, {sum(len(c) for c in chunks) / 2**20:.0f} MiB generated in {time.time() - t0:.1f} s", flush=True)
import torch
import squishrs_amd
from test_compress_gpu import _batch_compress
st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 16)
t0 = time.time()
frames = []
for b0 in range(0, n, 512):
    frames += _batch_compress(st, chunks[b0:b0 + 512])
gpu_s = time.time() - t0
o = oracle_lib.load()
t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    want = list(ex.map(o.compress_chunk, chunks))
bad = [(i, len(chunks[i])) for i in range(n) if frames[i] != want[i]]
print(f"gpu {gpu_s:.1f} s, oracle (16 threads) {time.time() - t0:.1f} s; mismatches: {len(bad)} {bad[:10]}", flush=True)
# the same through the pipelined host path (sq_pack_submit / sq_pack_wait), dedup included
res = []
st.reset()
for b0 in range(0, n, 300):
    res += st.insert_batch(chunks[b0:b0 + 300])
seen, bad2 = set(), []
for i, (c, r) in enumerate(zip(chunks, res)):
    if c in seen:
        if r.compressed_data is not None:
            bad2.append(i)
    else:
        seen.add(c)
        if r.compressed_data != want[i] or r.hash != o.hash_chunk(c):
            bad2.append(i)
print(f"insert_batch path: mismatches {len(bad2)} {bad2[:10]}; unique {st.len()} == {len(seen)}", flush=True)
# the unpack side: every frame back through sq_decompress_batch with the capacity the reader gives it (the chunk's recorded size,
# reader.rs:302-303); the output must be the chunk
t0 = time.time()
bad3 = []
for b0 in range(0, n, 512):
    got = st.decompress_batch(frames[b0:b0 + 512], [len(c) for c in chunks[b0:b0 + 512]])
    bad3 += [b0 + j for j, g in enumerate(got) if g != chunks[b0 + j]]
print(f"sq_decompress_batch of the {n} frames: {time.time() - t0:.1f} s; outputs that differ from their chunk: {len(bad3)} {bad3[:10]}", flush=True)
st.close()
assert not bad and not bad2 and not bad3 and st is not None
