"""Phase profile of the compress stage on 2 MiB mixed chunks with every wave busy (experiment tooling).
usage: SQUISH_GPU_LIB=squishrs_amd/libsquish_gpu_prof.so python tools/prof_mixed.py [chunks] [distinct] [kind] [chunk_bytes]
`distinct` contents are generated on the host by forked workers (before the GPU is touched) and replicated on the
device; the compress stage runs without dedup, so every copy is compressed."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
kind = sys.argv[3] if len(sys.argv) > 3 else "mixed"
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2 << 20


def one(seed):
    return corpus.make(kind, seed, L)


if __name__ == "__main__":
    t0 = time.time()
    if os.environ.get("SQ_NO_FORK"):   # under rocprofv3 --pmc the GPU is initialised before main(): no fork then
        chunks = [one(s) for s in range(41000, 41000 + k)]
    else:
        with mp.get_context("fork").Pool(min(32, len(os.sched_getaffinity(0)))) as pool:
            chunks = pool.map(one, range(41000, 41000 + k))
    print(f"generated {k} x {L} B {kind} chunks in {time.time() - t0:.1f} s", flush=True)
    import torch
    import oracle_lib
    import squishrs_amd
    base = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    data = base.view(k, L).repeat((n + k - 1) // k, 1)[:n].contiguous().view(-1)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
    comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
    for _ in range(int(os.environ.get("PROF_WARMUPS", "1"))):   # (2 for counter runs: the second launch may change the kernel's build
        st.compress_device_slots(data, offs, lens, None, comp)    #  and clear the tables of the waves it adds; the third is steady state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    off, ln = st.compress_device_slots(data, offs, lens, None, comp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = st.timings()
    o = oracle_lib.load()
    for i in (0, k // 2, k - 1, n - 1):
        fr = comp[int(off[i]):int(off[i]) + int(ln[i])].cpu().numpy().tobytes()
        assert fr == o.compress_chunk(chunks[i % k]), f"frame {i} differs from the oracle"
    ratio = float(ln.astype(np.uint64).sum()) / (n * L)
    print(f"{n} x {L} B {kind}: ratio {ratio:.3f}, {dt * 1e3:.1f} ms = {n * L / 2**20 / dt:.0f} MiB/s "
          f"({n * L / 2**20 / dt / min(n, 2048):.2f} MiB/s per wave; stage 1 {t.fast_ms:.1f} ms, stage 2 {t.compress_ms:.1f} ms); 4 frames bit-exact", flush=True)
    st.close()
