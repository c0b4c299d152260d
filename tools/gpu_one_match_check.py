"""GPU check of stage 1's second pass ("one short match", sq_zstd_fast.h fastpath_blocks): chunks of the btlazy2 size class with ONE
planted coincidence on a searched position and the variants the pass must refuse (tools/fuzz_one_match_sim.py make_case), through
the compress stage (split form) and through sq_pack_submit (fused form), EVERY frame against the oracle; prints how many chunks
stage 1 decided.  usage: python tools/gpu_one_match_check.py [chunks] [seed]"""
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor
import multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def make(i):
    import fuzz_one_match_sim as f   # (loads the oracle and the simulator library in the worker: only make_case is used)
    rng = random.Random(seed * 1000003 + i)
    for _ in range(8):
        d, kind = f.make_case(rng)
        if d is not None:
            return d, kind
    raise RuntimeError("no case")


if __name__ == "__main__":
    t0 = time.time()
    with mp.get_context("fork").Pool(min(16, len(os.sched_getaffinity(0)))) as pool:
        cases = pool.map(make, range(n), chunksize=16)
    chunks = [c for c, _ in cases]
    kinds = [k for _, k in cases]
    print(f"{n} chunks, {sum(len(c) for c in chunks) / 2**20:.0f} MiB generated in {time.time() - t0:.1f} s", flush=True)
    import oracle_lib
    import torch
    import squishrs_amd
    from test_compress_gpu import _batch_compress
    o = oracle_lib.load()
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(o.compress_chunk, chunks))
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 16)
    frames, decided = [], 0
    for b0 in range(0, n, 512):
        frames += _batch_compress(st, chunks[b0:b0 + 512])
        decided += st.timings().fast_decided if hasattr(st.timings(), "fast_decided") else 0
    bad = [(i, kinds[i], len(chunks[i])) for i in range(n) if frames[i] != want[i]]
    print(f"compress stage (split form): mismatches {len(bad)} {bad[:10]}; stage 1 decided {decided} of {n}", flush=True)
    res = []
    st.reset()
    for b0 in range(0, n, 300):
        res += st.insert_batch(chunks[b0:b0 + 300])
    seen, bad2 = set(), []
    for i, (c, r) in enumerate(zip(chunks, res)):
        if c in seen:
            if r.compressed_data is not None:
                bad2.append((i, kinds[i]))
        else:
            seen.add(c)
            if r.compressed_data != want[i] or r.hash != o.hash_chunk(c):
                bad2.append((i, kinds[i]))
    print(f"sq_pack_submit (fused form): mismatches {len(bad2)} {bad2[:10]}; unique {st.len()} == {len(seen)}", flush=True)
    per = {}
    for k in kinds:
        per[k] = per.get(k, 0) + 1
    print("kinds:", dict(sorted(per.items())), flush=True)
    st.close()
    assert not bad and not bad2
