"""Decode throughput of sq_decompress_batch (experiment tooling, GPU box): N chunks of a corpus kind are packed on the GPU,
then all frames decoded on the GPU; prints the decode kernel's time.  usage: python tools/bench_decode.py [chunks] [distinct] [kind] [chunk_bytes]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "mixed"
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2 << 20


def one(seed):
    return corpus.make(kind, seed, L)


if __name__ == "__main__":
    if os.environ.get("SQ_NO_FORK"):   # under rocprofv3 --pmc the GPU is initialised before main(): no fork then
        chunks = [one(s) for s in range(41000, 41000 + k)]
    else:
        with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
            chunks = pool.map(one, range(41000, 41000 + k))
    import torch
    import squishrs_amd
    base = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    data = base.view(k, L).repeat((n + k - 1) // k, 1)[:n].contiguous().view(-1)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
    comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
    off, ln = st.compress_device_slots(data, offs, lens, None, comp)
    out = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ol, status = st.decompress_device(comp, off, ln, out, offs, lens)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert (status == 0).all() and (ol == lens).all() and torch.equal(out, data)
    ratio = float(ln.astype(np.uint64).sum()) / (n * L)
    print(f"{n} x {L} B {kind} (ratio {ratio:.3f}): decoded in {dt * 1e3:.1f} ms = {n * L / 2**20 / dt:.0f} MiB/s of output "
          f"({n * L / 2**20 / dt / min(n, 3072):.1f} MiB/s per wave); kernel {st.timings().compact_ms:.1f} ms; outputs equal the chunks", flush=True)
    st.close()
