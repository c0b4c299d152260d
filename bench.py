#!/usr/bin/env python3
"""bench.py -- pack-path throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (XXH3-128 hash -> [all-gather of hashes] -> dedup ->
zstd-12 compress) over one batch that is already resident in HBM.  Output layout "slots" (default):
every frame stays in the slot it was produced in (offset + length per chunk), as the reference hands
its writer one buffer per chunk; "packed" adds the pass that moves the frames back to back.
Workload at N=1: BASELINE.json configs[1], "100k x 256 KiB unique random chunks" (per GPU; for
N>1 every rank holds its own 100k globally-unique chunks => weak scaling, and the hash all-gather
over RCCL is inside the timed region).

Prints ONE JSON line on rank 0.  The CPU oracle is used here only as the checker (sample parity,
outside the timed region) and for the `cpu_baseline` leg."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _pmc_traffic(kernel, n, L):
    """HBM traffic of one launch of `kernel` (bytes), from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE collected in separate runs of this same script on fewer chunks of the
    same workload, see tools/profile_r1.sh, tools/pmc_summary.py and profiles/), scaled linearly to n
    chunks: counters cannot be read from inside the process.  None when no committed measurement
    matches this kernel and chunk size."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(f"{kernel}@{L}")
        return None if e is None else float(e["fetch_bytes_per_chunk"] + e["write_bytes_per_chunk"]) * n
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=100000, help="chunks per GPU")
    ap.add_argument("--chunk-bytes", type=int, default=262144)
    ap.add_argument("--waves", type=int, default=0, help="persistent compressor waves (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all logical CPUs)")
    ap.add_argument("--layout", choices=("slots", "packed"), default="slots")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"launched with WORLD_SIZE={world} but --gpus {a.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")

    import corpus_gpu
    import squishrs_amd
    from squishrs_amd.parallel import ShardedPacker

    n, L = a.chunks, a.chunk_bytes
    store = squishrs_amd.ChunkStore(device=local_rank, max_chunk=L, dedup_capacity=world * n + 1024,
                                    compress_waves=a.waves)
    store.set_stream(torch.cuda.current_stream().cuda_stream)
    packer = ShardedPacker(store)

    # ---- synthetic workload, generated in HBM (config 1: unique random chunks)
    t0 = time.time()
    data = corpus_gpu.random_chunks(1, rank * n, n, L, device="cuda", group=16384)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
    bound = squishrs_amd.compress_bound(L)
    slots = a.layout == "slots"
    comp = torch.empty(store.slots_bound(lens) if slots else n * (bound + 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    def step():
        store.reset()
        return packer.pack(data, offs, lens, hashes, is_new, comp, layout=a.layout)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        comp_off = step()
    barrier()
    kern = {"hash": [], "dedup": [], "compress": [], "compact": [], "fast": []}
    t_start = time.perf_counter()
    for _ in range(a.steps):
        comp_off = step()
        t = store.timings()   # HIP-event durations on the ctx stream (no extra sync: step() syncs)
        kern["hash"].append(t.hash_ms)
        kern["dedup"].append(t.dedup_ms)
        kern["compress"].append(t.compress_ms)
        kern["compact"].append(t.compact_ms)
        kern["fast"].append(t.fast_ms)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / a.steps * 1e3

    # ---- outside the timed region: parity sample against the CPU oracle, roofline, CPU baseline
    unique = store.len()
    if slots:
        comp_off, comp_len = comp_off
        frame_end = comp_off + comp_len.astype(np.uint64)
        clen_total = int(comp_len.astype(np.uint64).sum())
    else:
        frame_end = comp_off[1:]
        clen_total = int(comp_off[-1])
    out = None
    if rank == 0:
        import corpus
        import oracle_lib
        o = oracle_lib.load()
        hh = hashes.cpu().numpy().reshape(-1, 16)
        cc = comp
        nn = is_new.cpu().numpy()
        ok = True
        for i in sorted(set([0, 1, n // 2, n - 1])):
            c = corpus.random_bytes(1, rank * n + i, L)
            ok &= hh[i].tobytes() == o.hash_chunk(c)
            fr = cc[int(comp_off[i]):int(frame_end[i])].cpu().numpy().tobytes()
            ok &= bool(nn[i]) and fr == o.compress_chunk(c)
        assert ok, "bench sample is not bit-exact vs the oracle"

        value = world * n * L / 2**20 / (ms_per_step / 1e3)
        cms = float(np.mean(kern["compress"]))
        hms = float(np.mean(kern["hash"]))
        # The compress stage is two kernels: stage 1 (k_incompressible_batch) looks at every new
        # chunk once and writes the frames it can decide; stage 2 (k_compress_batch) does the rest.
        # Algorithmic bytes (SURVEY.md 8d "compress-only"): every unique input byte read once +
        # every frame byte written once, attributed to the kernel that produced the frame.
        fms = float(np.mean(kern["fast"]))
        decided = int(store.timings().fast_decided)
        frame_avg = clen_total / max(1, int(unique if world == 1 else n))
        fast_bytes = n * L + decided * frame_avg
        slow_bytes = (n - decided) * (L + frame_avg)
        hash_bytes = n * L + 16 * n
        bw_probe_ms = min(store.read_bw_ms(data) for _ in range(3))
        stages = {"sqz::k_incompressible_batch": (fms, fast_bytes), "sqz::k_compress_batch": (cms, slow_bytes)}
        dom = max(stages, key=lambda k: stages[k][0])
        dms, dbytes = stages[dom]
        roof = {"bound": "hbm", "kernel": dom,
                "achieved": dbytes / (dms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": dbytes / (dms / 1e3) / 1e9 / 8000.0, "traffic": _pmc_traffic(dom, n, L),
                "kernel_ms": dms,
                "compress_stage": {"k_incompressible_batch_ms": fms, "k_compress_batch_ms": cms,
                                   "chunks_decided_by_stage1": decided,
                                   "stage_GBps": (n * L + clen_total) / ((fms + cms) / 1e3) / 1e9},
                "hash_kernel": {"kernel": "sqx::k_xxh3_128_rows", "ms": hms,
                                "achieved": hash_bytes / (hms / 1e3) / 1e9,
                                "frac": hash_bytes / (hms / 1e3) / 1e9 / 8000.0},
                "dedup_ms": float(np.mean(kern["dedup"])), "compact_ms": float(np.mean(kern["compact"])),
                "read_probe_GBps": n * L / (bw_probe_ms / 1e3) / 1e9}
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            done = 0
            busy = 0.0
            ost = oracle_lib.OracleStore(o)
            wall0 = time.perf_counter()
            while busy < a.cpu_seconds and done < n and time.perf_counter() - wall0 < 6 * a.cpu_seconds:
                c = corpus.random_bytes(1, done, L)          # generation is not timed
                tb = time.perf_counter()
                ost.insert(c)                                # hash + dedup + compress (chunk.rs:80-100)
                busy += time.perf_counter() - tb
                done += 1
            dt = busy
            single = done * L / 2**20 / max(dt, 1e-9)
            # ... and on every host core at once, as the reference's Rayon pool would run it (one chunk per
            # task: hash + compress; the ctypes calls release the GIL).  Each thread works through its own
            # 32 chunks of the workload (copied from HBM before the clock starts) until the time is up.
            import ctypes
            try:    # CPUs this process may really use: affinity mask, and the cgroup's CPU quota if there is one
                T = len(os.sched_getaffinity(0))
            except (AttributeError, OSError):
                T = os.cpu_count() or 1
            try:
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    T = max(1, min(T, int(int(q) / int(per))))
            except (OSError, ValueError):
                pass
            if a.cpu_threads > 0:
                T = a.cpu_threads
            per_thread = 32
            pool_n = min(n, T * per_thread)
            host = data[:pool_n * L].cpu().numpy()
            o.lib.sqo_bench_insert_threads.restype = ctypes.c_uint64
            o.lib.sqo_bench_insert_threads.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_double]
            par_s = max(2.0, a.cpu_seconds * 0.75)
            tp0 = time.perf_counter()
            total = int(o.lib.sqo_bench_insert_threads(host.ctypes.data, pool_n, L, T, par_s))
            par_dt = time.perf_counter() - tp0
            del host
            par = total * L / 2**20 / par_dt
            counts = [total]
            cpu = {"value": par, "unit": "MiB/s", "cores": T, "kind": "port",
                   "sample": f"{sum(counts)} chunk compressions in {par_dt:.1f} s on {T} pthreads (oracle/bench_threads.c: each thread cycles "
                             f"through its own 32 chunks of the same workload, XXH3-128 + zstd-12 per chunk, one chunk per task "
                             f"like the reference's Rayon pool; T = CPUs this process may use); single thread through the oracle's "
                             f"ChunkStore::insert on the first {done} chunks: {single:.0f} MiB/s; host has "
                             f"{os.cpu_count()} logical CPUs"}
        out = {"metric": "pack throughput MiB/s", "value": value, "unit": "MiB/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"{n} x {L // 1024} KiB unique random chunks per GPU "
                                      f"(BASELINE configs[1]), xoshiro256** seeded, resident in HBM",
                          "chunks_per_gpu": n, "chunk_bytes": L, "unique_chunks": int(unique),
                          "compressed_bytes_rank0": clen_total, "parallelism": f"shard{world}",
                          "output_layout": a.layout,
                          "parity_sample": "bit-exact vs oracle", "gen_seconds": gen_s},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    store.close()


if __name__ == "__main__":
    main()
