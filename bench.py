#!/usr/bin/env python3
"""bench.py -- pack-path throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (XXH3-128 hash -> [all-gather of hashes] -> dedup -> zstd-12 compress of the
new chunks) over one batch that is already resident in HBM.  Output layout "slots" (default): every frame stays in
the slot it was produced in (offset + length per chunk), as the reference hands its writer one buffer per chunk;
"packed" adds the pass that moves the frames back to back.

Headline workload (the ONE JSON line's `value`):
  N = 1 : BASELINE.json configs[1], "100k x 256 KiB unique random chunks".  The timed steps keep two batches in
          flight on the context's two lanes (sq_pack_submit / sq_pack_wait; --no-pipeline: one at a time); one more
          step WITHOUT overlap follows the timed region and supplies `roofline.solo_step` and the outputs the parity
          checks look at (ALL frames on the device, a sample -- including chunks from stage 2's work list --
          against the oracle).
  N > 1 : BASELINE configs[4]'s shape at --chunks per GPU: 30 % of all chunks are exact copies of a chunk that lives
          on ANOTHER rank (seeded plan, generated on device); the hash all-gather over RCCL is inside the timed
          region (two batches in flight here too: sq_shard_submit_hash / the all-gather / sq_shard_submit_rest, so
          the exchange of step k+1 runs under step k's compression); every rank checks its is_new flags against a
          host model and unique == 0.7 N n.
Defaults: 30 timed steps after 2 warm-up steps (a step is 21 ms; with two batches in flight the first step's hash and
the last step's thin tail are not overlapped, which 3 steps would weigh at a third each).
After the headline (N = 1 only, outside its timed region) two more legs are measured and reported under
`other_configs`: configs[2] (100k x 256 KiB, 50 % exact duplicates) and a configs[3] sample (4,096 Silesia-style
mixed-entropy chunks of 2 MiB, two per compressor wave; generated on the host by forked workers before the GPU is
touched), each with its own parity checks, CPU baseline and -- configs[3] -- the measured random-request roof.

Prints ONE JSON line on rank 0.  The CPU oracle is used here only as the checker (outside the timed regions) and
for the `cpu_baseline` leg."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

RAW_FRAME_256K = 262159      # 9-byte frame header + 2 x (3-byte block header + 131,072 raw bytes)


def _pmc_traffic(kernel, n, L):
    """HBM traffic of one launch of `kernel` (bytes) from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this script on fewer chunks of the same workload: tools/profile_r2.sh,
    tools/pmc_summary.py, profiles/), scaled linearly to n chunks -- counters cannot be read from inside the
    process.  CORRECTED as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies a wide coalesced
    streaming read at half its bytes, so the kernel's streaming-read bytes (stated per kernel in the json) are
    added once more at half.  None when no committed measurement matches."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(f"{kernel}@{L}")
        if e is None:
            return None
        return float(e["fetch_bytes_per_chunk"] + 0.5 * e.get("stream_read_bytes_per_chunk", 0.0) + e["write_bytes_per_chunk"]) * n
    except (OSError, ValueError, KeyError):
        return None


def _whole_step_traffic(n, L):
    """Sum of _pmc_traffic over every kernel of a step that the committed PMC passes saw at this chunk size."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    tot, seen = 0.0, 0
    for key in t:
        k, _, ls = key.partition("@")
        if ls == str(L) and k != "sqd::k_read_bw":
            v = _pmc_traffic(k, n, L)
            if v is not None:
                tot += v
                seen += 1
    return tot if seen else None


def _mixed_traffic():
    """configs[3]: HBM bytes the compress kernel moves per algorithmic byte on 2 MiB mixed chunks, from the committed
    rocprofv3 PMC passes (tools/pmc_mixed_traffic.sh -> profiles/pmc_traffic_mixed.json); None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_mixed.json")) as f:
            m = json.load(f)["mixed"]
        return {"traffic_ratio": m["traffic_ratio"], "fetched_bytes_per_chunk": m["FETCH_SIZE_KB"] * 1024 / m["chunks"],
                "written_bytes_per_chunk": m["WRITE_SIZE_KB"] * 1024 / m["chunks"], "measured_on_chunks": m["chunks"],
                "source": "profiles/pmc_traffic_mixed.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, steady-state launch of "
                          "sqz::k_compress_batch on 2 MiB mixed chunks)"}
    except (OSError, ValueError, KeyError, TypeError):
        return None


def _usable_cpus():
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
    return T


def _mixed_one(seed):
    import corpus
    return corpus.mixed(seed, 2 << 20)


def _gen_mixed_host(n, workers):
    """n distinct Silesia-style mixed-entropy chunks of 2 MiB (tools/corpus.py `mixed`, one seed per chunk) in one
    host buffer.  Generated by forked workers BEFORE this process touches the GPU."""
    import multiprocessing as mp
    buf = np.empty(n * (2 << 20), dtype=np.uint8)
    with mp.get_context("fork").Pool(workers) as pool:
        for i, b in enumerate(pool.imap(_mixed_one, range(31000, 31000 + n), chunksize=4)):
            buf[i * (2 << 20):(i + 1) * (2 << 20)] = np.frombuffer(b, dtype=np.uint8)
    return buf


def _cpu_pipeline(o, host, n_chunks, L, threads, seconds):
    """oracle/bench_threads.c sqo_bench_pipeline: the reference's pack pipeline on host threads."""
    o.lib.sqo_bench_pipeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                         ctypes.c_double, ctypes.POINTER(ctypes.c_uint64)]
    st = (ctypes.c_uint64 * 4)()
    t0 = time.perf_counter()
    rc = o.lib.sqo_bench_pipeline(host.ctypes.data, n_chunks, L, threads, seconds, st)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {"MiB_per_s": st[0] * L / 2**20 / dt, "chunks": int(st[0]), "seconds": dt, "threads": threads,
            "archive_bytes": int(st[1]), "max_channel_backlog": int(st[3])}


def _real_libzstd(host, n_chunks, L, threads, seconds):
    """Timing only, where the image carries it: the REAL libzstd 1.5.7 (the library zstd-sys wraps; found in the
    pillow wheel by tools/zref.py) + python-xxhash on `threads` Python threads (both release the GIL): XXH3-128 +
    ZSTD_compress(level 12) per chunk, each thread cycling through its own chunks of the buffer.  None if absent."""
    try:
        import threading
        import xxhash
        import zref
        z = zref.lib()
    except Exception:  # noqa: BLE001 -- LookupError / ImportError: not in this image
        return None
    cap = z.ZSTD_compressBound(L)
    counts = [0] * threads
    t_end = time.perf_counter() + seconds
    base = host.ctypes.data

    def work(t):
        dst = ctypes.create_string_buffer(cap)
        i = t % n_chunks
        while time.perf_counter() < t_end:
            xxhash.xxh3_128_digest(host[i * L:(i + 1) * L])
            z.ZSTD_compress(dst, cap, ctypes.c_void_p(base + i * L), L, 12)
            counts[t] += 1
            i += threads
            if i >= n_chunks:
                i = t % n_chunks
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return {"MiB_per_s": sum(counts) * L / 2**20 / dt, "chunks": sum(counts), "seconds": dt, "threads": threads,
            "note": "real libzstd 1.5.7 (pillow wheel) + python-xxhash, hash + compress only, timing only"}


def _text_one(args):
    import corpus
    seed, nbytes = args
    return corpus.text_like(seed, nbytes)


def _gen_text_host(n_bytes, workers):
    """n_bytes of text-like bytes (tools/corpus.py text_like, one seed per MiB) in one host buffer, by forked workers."""
    import multiprocessing as mp
    piece = 1 << 20
    k = (n_bytes + piece - 1) // piece
    buf = np.empty(k * piece, dtype=np.uint8)
    with mp.get_context("fork").Pool(workers) as pool:
        for i, b in enumerate(pool.imap(_text_one, [(52000 + i, piece) for i in range(k)], chunksize=4)):
            buf[i * piece:(i + 1) * piece] = np.frombuffer(b, dtype=np.uint8)
    return buf[:n_bytes]


def _bind_oracle_bench(o):
    L = o.lib
    L.sqo_bench_pipeline2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_double,
                                      ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p]
    L.sqo_bench_decode_verify.restype = ctypes.c_int64
    L.sqo_bench_decode_verify.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_uint, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64)]


def _cpu_pipeline2(o, host, offs, lens, threads, seconds):
    """sqo_bench_pipeline2: the reference's pack pipeline on host threads over ragged chunks; also returns, per chunk
    its first pass reached, (frame length, XXH3-128 of the frame)."""
    n = len(lens)
    st = (ctypes.c_uint64 * 4)()
    out_len = np.zeros(n, dtype=np.uint32)
    out_dig = np.zeros(n * 16, dtype=np.uint8)
    t0 = time.perf_counter()
    rc = o.lib.sqo_bench_pipeline2(host.ctypes.data, offs.ctypes.data, lens.ctypes.data, n, threads, seconds, st, out_len.ctypes.data, out_dig.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    covered = int(min(st[0], n))
    return {"MiB_per_s": float(lens.astype(np.float64).mean()) * st[0] / 2**20 / dt, "chunks": int(st[0]), "seconds": dt, "threads": threads}, out_len, out_dig.reshape(n, 16), covered


def _decode_verify(torch, o, comp, co, cl, idx, want_len, want_hash, threads, seconds):
    """Frames idx (device slots comp[co[i] : co[i] + cl[i]]) are brought to the host in groups and decoded by the
    oracle's RFC 8878 decoder on `threads` threads: decoded length == want_len and XXH3-128(decoded) == want_hash.
    Stops when `seconds` are used up.  Returns (frames verified, frames that failed)."""
    t_end = time.perf_counter() + seconds
    done_total, bad_total = 0, 0
    G = 1024
    for g0 in range(0, len(idx), G):
        left = t_end - time.perf_counter()
        if left <= 0.2:
            break
        ids = idx[g0:g0 + G]
        parts = [comp[int(co[i]):int(co[i]) + int(cl[i])] for i in ids]
        host = torch.cat(parts).cpu().numpy()
        fl = cl[ids].astype(np.uint32)
        fo = np.zeros(len(ids), dtype=np.uint64)
        fo[1:] = np.cumsum(fl.astype(np.uint64))[:-1]
        wl = np.ascontiguousarray(want_len[ids], dtype=np.uint32)
        wh = np.ascontiguousarray(want_hash[ids]).reshape(-1)
        dn = ctypes.c_uint64(0)
        bad = o.lib.sqo_bench_decode_verify(host.ctypes.data, fo.ctypes.data, fl.ctypes.data, wl.ctypes.data, wh.ctypes.data, len(ids), threads,
                                            max(0.2, t_end - time.perf_counter()), ctypes.byref(dn))
        assert bad >= 0
        done_total += int(dn.value)
        bad_total += int(bad)
    return done_total, bad_total


def _run_cli(cli, *args, env=None):
    import subprocess
    t0 = time.perf_counter()
    r = subprocess.run([cli, *args], capture_output=True, text=True, env=dict(os.environ, SQUISH_TIMING="1", **(env or {})))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-400:]
    stages = None
    for line in r.stderr.splitlines():
        if "[squishrs timing]" in line:
            import re
            m = re.search(r"ctx_create ([0-9.]+) s; overlapped stages: file reads ([0-9.]+) s, sq_pack_submit/wait ([0-9.]+) s, record writes ([0-9.]+) s", line)
            if m:
                stages = {"ctx_create_s": float(m.group(1)), "file_reads_busy_s": float(m.group(2)), "gpu_stage_busy_s": float(m.group(3)),
                          "record_writes_busy_s": float(m.group(4))}
            w = re.search(r"wall inside pack\(\): walk \+ stat ([0-9.]+) s, ctx_create ([0-9.]+) s, the three stages ([0-9.]+) s", line)
            if w and stages is not None:   # (the rest of pack_seconds is the process's start and its teardown)
                stages["wall_inside_pack_s"] = {"walk_and_stat": float(w.group(1)), "ctx_create": float(w.group(2)), "three_stages": float(w.group(3))}
    return dt, stages, r.stdout


def _end_to_end(torch, o, data, L, n_big):
    """the product number -- the C++ driver (squishrs_amd/host/squishrs) packing a tree of files
    from /dev/shm: file reads, H2D, kernels, D2H, the record writer, all in.  Two trees: BASELINE configs[0]'s shape
    (1,000 x 64 KiB random files; the archive is compared with the oracle's restatement byte for byte, timestamp
    aside) and n_big files of L bytes taken from the headline workload.  The path timed is
    /root/reference/src/lib.rs:25-56 -> src/archive/writer.rs:151-195."""
    import shutil
    import tempfile
    import corpus
    cli = os.path.join(ROOT, "squishrs_amd", "host", "squishrs")
    if not os.path.exists(cli):
        return {"skipped": "squishrs_amd/host/squishrs not built"}
    base = tempfile.mkdtemp(prefix="sqe2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"where": base}
    try:
        # ---- configs[0]
        tree0 = os.path.join(base, "c0")
        os.makedirs(tree0)
        files = {}
        for i in range(1000):
            files["f%06d.bin" % i] = corpus.random_bytes(0, i, 65536)
            with open(os.path.join(tree0, "f%06d.bin" % i), "wb") as f:
                f.write(files["f%06d.bin" % i])
        a0 = os.path.join(base, "c0.squish")
        _run_cli(cli, "pack", tree0, "-o", a0)                       # warm-up: library load, page cache
        os.remove(a0)
        dt0, st0, _ = _run_cli(cli, "pack", tree0, "-o", a0)
        arch = open(a0, "rb").read()
        # manifest order = the driver's walk order; the oracle packs the same files in that order
        count = int.from_bytes(arch[19:27], "little")
        pos = 27
        for _ in range(count):
            pos += 32 + int.from_bytes(arch[pos + 24:pos + 32], "little")
        nfiles = int.from_bytes(arch[pos:pos + 4], "little")
        pos += 4
        order = []
        for _ in range(nfiles):
            pl = int.from_bytes(arch[pos:pos + 4], "little")
            order.append(arch[pos + 4:pos + 4 + pl].decode())
            pos += 4 + pl + 8
            pos += 4 + 16 * int.from_bytes(arch[pos:pos + 4], "little")
        exp = o.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)
        assert count == 1000 and nfiles == 1000 and arch[19:] == exp[19:], "configs[0] archive differs from the oracle's"
        dtl, _, txt = _run_cli(cli, "list", a0, "--simple")
        assert "number_of_files: 1000" in txt
        out["configs[0]"] = {"workload": "squishrs pack of 1,000 files x 65,536 random bytes from /dev/shm", "pack_seconds": dt0,
                             "MiB_per_s": 1000 * 65536 / 2**20 / dt0, "stages": st0, "archive_bytes": len(arch),
                             "parity": "archive == oracle .squish restatement (all but the 8 timestamp bytes); list --simple: 1000 files"}
        # ---- the big tree
        treeb = os.path.join(base, "big")
        host = data[:n_big * L].cpu().numpy()
        t0 = time.perf_counter()
        for i in range(n_big):
            d = os.path.join(treeb, "d%03d" % (i // 512))
            if i % 512 == 0:
                os.makedirs(d)
            host[i * L:(i + 1) * L].tofile(os.path.join(d, "f%06d.bin" % i))
        gen_s = time.perf_counter() - t0
        del host
        ab = os.path.join(base, "big.squish")
        _run_cli(cli, "pack", treeb, "-o", ab)
        os.remove(ab)
        dtb, stb, _ = _run_cli(cli, "pack", treeb, "-o", ab)
        _, _, txt = _run_cli(cli, "list", ab, "--simple")
        assert f"number_of_files: {n_big}" in txt
        out["big"] = {"workload": f"squishrs pack of {n_big} files x {L // 1024} KiB random bytes ({n_big * L / 2**30:.1f} GiB) from /dev/shm, archive into /dev/shm",
                      "pack_seconds": dtb, "MiB_per_s": n_big * L / 2**20 / dtb, "stages": stb, "archive_bytes": os.path.getsize(ab),
                      "tree_write_seconds": gen_s,
                      "note": "second run (page cache warm); stage times are each stage's busy time, the stages overlap; includes process start, "
                              "HIP initialisation, the directory walk and teardown"}
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out


def _check_raw_frames(torch, comp, stride, data, n, L, rows):
    """Device-side: every frame in `rows` (slots layout, constant stride) is the two-raw-block frame of its input:
    9 header bytes, block headers, payload == input.  Returns the number of bad rows."""
    hdr = torch.tensor([0x28, 0xB5, 0x2F, 0xFD, 0xA0, 0x00, 0x00, 0x04, 0x00, 0x00, 0x00, 0x10], dtype=torch.uint8, device=comp.device)
    bh1 = torch.tensor([0x01, 0x00, 0x10], dtype=torch.uint8, device=comp.device)
    fr = comp[:n * stride].view(n, stride)
    dv = data[:n * L].view(n, L)
    bad = 0
    for k0 in range(0, len(rows), 2048):
        r = rows[k0:k0 + 2048]
        f = fr[r]
        d = dv[r]
        ok = (f[:, :12] == hdr).all(dim=1) & (f[:, 12 + 131072:12 + 131072 + 3] == bh1).all(dim=1)
        ok &= (f[:, 12:12 + 131072] == d[:, :131072]).all(dim=1)
        ok &= (f[:, 12 + 131072 + 3:RAW_FRAME_256K] == d[:, 131072:]).all(dim=1)
        bad += int((~ok).sum().item())
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunks", type=int, default=100000, help="chunks per GPU")
    ap.add_argument("--chunk-bytes", type=int, default=262144)
    ap.add_argument("--waves", type=int, default=0, help="persistent compressor waves (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all usable CPUs)")
    ap.add_argument("--layout", choices=("slots", "packed"), default="slots")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--no-pipeline", action="store_true", help="one batch at a time (kernel durations without overlap)")
    ap.add_argument("--no-frame-check", action="store_true",
                    help="timing experiments only (SQ_FAST_DBG switches make wrong frames on purpose): the line says so")
    ap.add_argument("--config3-chunks", type=int, default=4096, help="DISTINCT 2 MiB mixed chunks generated on the host for the configs[3] leg")
    ap.add_argument("--config3-replicas", type=int, default=4, help="the leg's batch = this many copies of them, each rotated by a different "
                                                                     "number of bytes (no two chunks equal): 4 x 4096 x 2 MiB = 32 GiB")
    ap.add_argument("--config3-steps", type=int, default=1)
    ap.add_argument("--verify-seconds", type=float, default=20.0, help="time budget of the decode-and-verify pass over the configs[3] frames")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--e2e-files", type=int, default=32768, help="files of the end-to-end leg's big tree (x --chunk-bytes)")
    ap.add_argument("--stripe", type=int, default=16, help="N > 1: chunks per stripe of the block-cyclic assignment of the global chunk "
                                                           "order to the ranks (0 = contiguous ranges: rank 0 then holds first occurrences only)")
    ap.add_argument("--no-small-classes", action="store_true")
    ap.add_argument("--no-random-2m", action="store_true")
    ap.add_argument("--oracle-sample", type=int, default=16384,
                    help="headline chunks whose frames the CPU-baseline pass (the oracle's pack pipeline) also produces; each is compared "
                         "with the GPU's frame by (length, XXH3-128 of the frame)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"launched with WORLD_SIZE={world} but --gpus {a.gpus}"
    other = (world == 1) and not a.no_other_configs
    T_cpu = a.cpu_threads if a.cpu_threads > 0 else _usable_cpus()

    # configs[3] corpus: host generation with forked workers, before anything touches the GPU
    mixed_host, mixed_gen_s = None, 0.0
    if other and a.config3_chunks > 0:
        t0 = time.time()
        mixed_host = _gen_mixed_host(a.config3_chunks, max(1, min(T_cpu, 32)))
        mixed_gen_s = time.time() - t0
    text_host = None
    if other and not a.no_small_classes:
        text_host = _gen_text_host(128 << 20, max(1, min(T_cpu, 32)))

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # SQUISH_DEVICE / SQUISH_BACKEND: rehearsal of the N > 1 path on ONE GPU (every rank on that device, gloo for
    # the exchange -- RCCL refuses two ranks per device); the driver's runs use one GPU per rank over RCCL.
    local_dev = int(os.environ.get("SQUISH_DEVICE", local_rank))
    backend = os.environ.get("SQUISH_BACKEND", "nccl")
    torch.cuda.set_device(local_dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)

    import corpus_gpu
    import squishrs_amd
    from squishrs_amd.parallel import ShardedPacker

    n, L = a.chunks, a.chunk_bytes
    store = squishrs_amd.ChunkStore(device=local_dev, max_chunk=max(L, 2 << 20) if other else L,
                                    dedup_capacity=world * n + 1024, compress_waves=a.waves)
    store.set_stream(torch.cuda.current_stream().cuda_stream)
    packer = ShardedPacker(store)

    # ---- synthetic workload, generated in HBM by one kernel launch (tools/corpus_gen.hip)
    t0 = time.time()
    N = world * n
    import corpus
    plan = corpus.cross_shard_plan(world, n)                     # world == 1: identity = configs[1], every chunk unique
    cfg_id = 1 if world == 1 else 4                              # world > 1: configs[4] shape, 30 % cross-shard copies
    # which chunks of the global order are this rank's: block-cyclic stripes (the first occurrences, i.e. the chunks that
    # have to be compressed, then spread evenly) or contiguous ranges (SURVEY 8(e) as first written)
    from squishrs_amd.parallel import stripe_assign
    cyclic = world > 1 and a.stripe > 0 and n % a.stripe == 0 and a.layout == "slots"
    if cyclic:
        owner, _counts, perm_np = stripe_assign(N, world, a.stripe)
        my_gidx = np.nonzero(owner == rank)[0]
    else:
        owner = np.repeat(np.arange(world), n)
        perm_np = None
        my_gidx = np.arange(rank * n, (rank + 1) * n)
    my_plan = plan[my_gidx] if world > 1 else plan
    data = corpus_gpu.random_chunks_indexed(cfg_id, my_plan, L)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
    bound = squishrs_amd.compress_bound(L)
    slots = a.layout == "slots"
    comp = torch.empty(store.slots_bound(lens) if slots else n * (bound + 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    # Two batches in flight (sq_pack_submit / sq_pack_wait; N > 1: sq_shard_submit_hash / _rest with the all-gather in
    # between): step k+1's hash, exchange and first compress stage run under the thin tail of step k's second
    # compress stage.  Every in-flight step has its own output buffers.
    pipelined = slots and not a.no_pipeline   # (N > 1: the all-gather of step k+1 runs under step k's compression)
    if pipelined:
        hashes_b = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        is_new_b = torch.zeros(n, dtype=torch.uint8, device="cuda")
        comp_b = torch.empty(comp.numel(), dtype=torch.uint8, device="cuda")
        out_bufs = [(hashes, is_new, comp), (hashes_b, is_new_b, comp_b)]

    perm_t = torch.from_numpy(perm_np).cuda() if cyclic else None
    idx_t = torch.from_numpy(my_gidx.astype(np.int32)).cuda() if cyclic else None

    def step():
        """one batch at a time, through the same entry points as the overlapped steps (so the same form of the batch:
        fused or split)"""
        if cyclic:
            packer.submit(data, offs, lens, hashes, is_new, comp, reset=True, uniform=True, perm=perm_t, my_idx=idx_t)
            return packer.wait()
        if world == 1 and slots:
            store.pack_submit(data, offs, lens, hashes, is_new, comp, reset=True)
            return store.pack_wait(n)
        store.reset()
        return packer.pack(data, offs, lens, hashes, is_new, comp, layout=a.layout, uniform=True)   # n chunks on every rank

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rank_stats = []   # N > 1: one entry per rank, filled by timed()

    def timed(K, W):
        kern = {"hash": [], "dedup": [], "compress": [], "compact": [], "fast": []}

        def note():
            t = store.timings()   # HIP-event durations on the lane's stream (no extra sync)
            kern["hash"].append(t.hash_ms)
            kern["dedup"].append(t.dedup_ms)
            kern["compress"].append(t.compress_ms)
            kern["compact"].append(t.compact_ms)
            kern["fast"].append(t.fast_ms)

        def run(count, record):
            r = None
            if pipelined and count > 1:
                inflight = 0
                for k in range(count):
                    if inflight == 2:
                        r = packer.wait() if world > 1 else store.pack_wait(n)
                        if record:
                            note()
                        inflight -= 1
                    h_, n_, c_ = out_bufs[k & 1]
                    if world > 1:   # the sharded form: hash | exchange of hashes | global dedup + compress
                        packer.submit(data, offs, lens, h_, n_, c_, reset=True, uniform=True, perm=perm_t, my_idx=idx_t)
                    else:
                        store.pack_submit(data, offs, lens, h_, n_, c_, reset=True)   # == store.reset() + pack, enqueued
                    inflight += 1
                while inflight:
                    r = packer.wait() if world > 1 else store.pack_wait(n)
                    if record:
                        note()
                    inflight -= 1
            else:
                for _ in range(count):
                    r = step()
                    if record:
                        note()
            return r

        run(max(W, 2) if (pipelined or (world == 1 and slots) or cyclic) else W, False)   # (both lanes of the context warm: submit alternates them)
        barrier()
        packer.reset_stats()
        t_start = time.perf_counter()
        r = run(K, True)
        barrier()
        el = time.perf_counter() - t_start
        if world > 1:
            cdev = "cpu" if backend == "gloo" else "cuda"
            # every rank's own view of the timed region, gathered for the line: its wall time per step, the host seconds per
            # step it spent in each phase (hash + wait, all-gather of the hashes, enqueue of dedup + compression, waiting for
            # frames) and its mean kernel times -- so that the first run on real links shows WHICH rank and WHICH phase
            # sets the step
            ps = packer.stats
            mine = torch.tensor([el / K * 1e3, ps["hash_wait"] / K * 1e3, ps["exchange"] / K * 1e3, ps["rest"] / K * 1e3, ps["wait"] / K * 1e3,
                                 float(np.mean(kern["hash"])), float(np.mean(kern["fast"])), float(np.mean(kern["compress"]))],
                                dtype=torch.float64, device=cdev)
            allr = torch.zeros(world * mine.numel(), dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(allr, mine)
            rank_stats.clear()
            for rr in allr.view(world, -1).cpu().tolist():
                rank_stats.append({"ms_per_step": rr[0], "host_ms_per_step": {"hash_and_wait": rr[1], "all_gather_of_hashes": rr[2],
                                                                            "enqueue_dedup_compress": rr[3], "wait_for_frames": rr[4]},
                                   "kernels_ms": {"hash": rr[5], "stage1": rr[6], "stage2": rr[7]}})
            tt = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return r, el / K * 1e3, {k: float(np.mean(v)) for k, v in kern.items()}

    comp_off, ms_per_step, kern = timed(a.steps, a.warmup)
    kern_solo = None
    if pipelined:
        # one more step WITHOUT overlap (outside the timed region): kernel durations that do not include time shared
        # with the neighbouring step's kernels; its outputs are what the parity checks below look at
        # (the last timed step's flags and frame lengths are kept and must equal this step's: the overlapped form and
        # the one-at-a-time form are the same computation)
        last_new = out_bufs[(a.steps - 1) & 1][1].clone() if a.steps > 1 else None
        last_len = comp_off[1].copy() if a.steps > 1 else None
        comp_off = step()
        if last_new is not None:
            assert torch.equal(last_new, is_new), "is_new of the overlapped step differs from the un-overlapped step's"
            assert (last_len == comp_off[1]).all(), "frame lengths of the overlapped step differ from the un-overlapped step's"
        ts = store.timings()
        kern_solo = {"hash": ts.hash_ms, "dedup": ts.dedup_ms, "compress": ts.compress_ms, "compact": ts.compact_ms, "fast": ts.fast_ms}

    # ---- outside the timed region: parity, roofline, other configs, CPU baseline
    unique = store.len()
    if slots:
        comp_off, comp_len = comp_off
        clen_total = int(comp_len.astype(np.uint64).sum())
    else:
        comp_len = np.diff(comp_off.astype(np.int64)).astype(np.uint32)
        comp_off = comp_off[:-1]
        clen_total = int(comp_len.astype(np.uint64).sum())
    nn = is_new.cpu().numpy()
    # dedup against the host model of "first occurrence in global index order" (chunk.rs:80-100, Appendix E)
    _, first_idx = np.unique(plan, return_index=True)
    model_new = np.zeros(len(plan), dtype=np.uint8)
    model_new[first_idx] = 1
    mine_new = model_new[my_gidx] if world > 1 else model_new
    new_per_rank = [int(model_new[owner == r].sum()) for r in range(world)] if world > 1 else [int(model_new.sum())]
    assert (nn == mine_new).all(), "is_new differs from the first-occurrence model"
    assert unique == len(first_idx), f"unique {unique} != distinct contents {len(first_idx)}"
    if world > 1:
        assert unique == N - int(0.3 * N)
    assert ((comp_len > 0) == (nn == 1)).all(), "a new chunk without a frame / a duplicate with one"

    import oracle_lib
    o = oracle_lib.load()
    parity = {}
    if a.no_frame_check:
        parity = {"skipped": "--no-frame-check: a timing experiment, not a result"}
    elif slots and L == 262144:
        # every frame, on the device: raw-block frames carry their input byte for byte
        stride = int(comp_off[1] - comp_off[0]) if n > 1 else comp.numel()
        raw_rows = torch.from_numpy(np.nonzero(comp_len == RAW_FRAME_256K)[0]).cuda()
        bad = _check_raw_frames(torch, comp, stride, data, n, L, raw_rows)
        assert bad == 0, f"{bad} raw-block frames differ from their input"
        # the others (a match found, block compressed): decoded AND re-compressed by the oracle, all of them
        odd = np.nonzero((comp_len != RAW_FRAME_256K) & (comp_len > 0))[0]
        s2 = store.stage2_list()
        pick = list(odd[:256]) + [int(x) for x in s2[:: max(1, len(s2) // 24)][:24]] + [0, n // 2, n - 1]
        pick = sorted({int(i) for i in pick if nn[int(i)]})
        for i in pick:
            c = corpus.random_bytes(cfg_id, int(my_plan[i]), L)
            fr = comp[int(comp_off[i]):int(comp_off[i]) + int(comp_len[i])].cpu().numpy().tobytes()
            assert fr == o.compress_chunk(c), f"frame {i} differs from the oracle"
        hh = hashes.cpu().numpy().reshape(-1, 16)
        for i in pick[:32]:
            assert hh[i].tobytes() == o.hash_chunk(corpus.random_bytes(cfg_id, int(my_plan[i]), L))
        parity = {"raw_block_frames_equal_to_the_raw_block_form_of_their_input_on_device": int(raw_rows.numel()),
                  "what_that_proves": "the frame is a valid frame of raw blocks of ITS chunk; that libzstd would also emit raw blocks for the chunk "
                                      "is what the comparisons with the oracle's frames show (next keys), on a sample",
                  "frames_with_compressed_blocks": int(len(odd)),
                  "frames_compared_with_oracle_bytes": len(pick), "of_them_from_stage2_list": int(min(len(s2), 24)),
                  "stage2_list_len": int(len(s2))}
    else:
        for i in sorted(set([0, 1, n // 2, n - 1])):
            if nn[i]:
                c = corpus.random_bytes(cfg_id, int(my_plan[i]), L)
                fr = comp[int(comp_off[i]):int(comp_off[i]) + int(comp_len[i])].cpu().numpy().tobytes()
                assert fr == o.compress_chunk(c)
        parity = {"frames_compared_with_oracle": 4}

    # (length, XXH3-128) of the headline's first frames by the product's own hash kernel (pinned by tests/test_hash_gpu.py),
    # taken now: the other legs reuse `data` and `comp`.  The CPU-baseline pass below pushes the same chunks through the
    # oracle's pack pipeline and every frame it produces is compared with these.
    pool_n = min(n, a.oracle_sample) if (world == 1 and slots and not a.no_cpu_baseline and not a.no_frame_check) else 0
    head_dig = head_len = None
    if pool_n:
        dg = torch.zeros(pool_n * 16, dtype=torch.uint8, device="cuda")
        store.hash_device(comp, comp_off[:pool_n].astype(np.uint64), comp_len[:pool_n].astype(np.uint32), dg)
        head_dig = dg.cpu().numpy().reshape(pool_n, 16)
        head_len = comp_len[:pool_n].copy()
        del dg

    out = None
    if rank == 0:
        value = world * n * L / 2**20 / (ms_per_step / 1e3)
        cms, hms, fms = kern["compress"], kern["hash"], kern["fast"]
        # The compress stage is two kernels: stage 1 (k_incompressible_batch) looks at every new chunk once and
        # writes the frames it can decide; stage 2 (k_compress_batch) does the rest.  Algorithmic bytes (SURVEY.md
        # 8d): every input byte read once + every frame byte written once, attributed to the kernel that produced
        # the frame.
        n_new = int(nn.sum())
        decided = int(store.timings().fast_decided)
        frame_avg = clen_total / max(1, n_new)
        fast_bytes = n_new * L + decided * frame_avg
        slow_bytes = (n_new - decided) * (L + frame_avg)
        hash_bytes = n * L + 16 * n
        path_bytes = n * L + clen_total + 16 * n           # the whole step (SURVEY.md 8d formula), this rank
        bw_probe_ms = min(store.read_bw_ms(data) for _ in range(3))
        # what a plain device-to-device copy reaches here (read + write): the practical roof of a kernel that writes
        # as much as it reads, as stage 1 does with its raw blocks
        cp_n = min(data.numel(), comp.numel(), 8 << 30)
        cp_ms = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            comp[:cp_n].copy_(data[:cp_n])
            e1.record()
            torch.cuda.synchronize()
            cp_ms = min(cp_ms, e0.elapsed_time(e1))
        stages = {"sqz::k_incompressible_batch": (fms, fast_bytes), "sqz::k_compress_batch": (cms, slow_bytes)}
        # which kernel dominates is decided on the durations of the un-overlapped step (with two batches in flight a
        # kernel's event-bracketed time includes what it shared with the other batch's kernels); the duration that
        # prices it is still the one measured over the timed region
        solo_ms = {"sqz::k_incompressible_batch": kern_solo["fast"], "sqz::k_compress_batch": kern_solo["compress"]} \
            if kern_solo is not None else {k: v[0] for k, v in stages.items()}
        dom = max(stages, key=lambda k: solo_ms[k])
        dms, dbytes = stages[dom]
        fused = bool(store.timings().fused)
        roof = {"bound": "hbm", "kernel": dom,
                "form": ("fused: stage 1 runs before the dedup decision on every chunk and computes XXH3's block sums beside its scan; "
                         "the hash kernel only runs the scramble chain over them (one read of every chunk)") if fused else
                        "split: hash kernel -> dedup -> stage 1 on the new chunks (two reads of every new chunk)",
                "achieved": dbytes / (dms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": dbytes / (dms / 1e3) / 1e9 / 8000.0, "traffic": _pmc_traffic(dom, n_new, L),
                "traffic_note": "rocprofv3 FETCH_SIZE + WRITE_SIZE per launch, gfx950 streaming-read correction applied",
                "kernel_ms": dms,
                "pipelined": bool(pipelined),
                "solo_step": None if kern_solo is None else {
                    "note": "the same step run once with nothing else in flight: kernel durations without the overlap of the timed region",
                    "k_incompressible_batch_ms": kern_solo["fast"], "k_compress_batch_ms": kern_solo["compress"],
                    "hash_ms": kern_solo["hash"],
                    "dominant_kernel_frac": fast_bytes / (kern_solo["fast"] / 1e3) / 1e9 / 8000.0 if kern_solo["fast"] > 0 else None},
                "whole_step": {"algorithmic_bytes": path_bytes, "ms": ms_per_step,
                               "achieved": path_bytes / (ms_per_step / 1e3) / 1e9,
                               "frac": path_bytes / (ms_per_step / 1e3) / 1e9 / 8000.0,
                               "traffic": _whole_step_traffic(n_new, L),
                               "traffic_note": "sum over the step's kernels (stage 1, hash finish, dedup, stage 2) of the committed steady-state PMC "
                                               "passes (profiles/pmc_traffic.json: both lanes warm, the last launch of each lane averaged), corrected "
                                               "for gfx950's half-counted streaming reads, scaled to this batch"},
                "compress_stage": {"k_incompressible_batch_ms": fms, "k_compress_batch_ms": cms,
                                   "chunks_decided_by_stage1": decided,
                                   "stage_GBps": (n_new * L + clen_total) / ((fms + cms) / 1e3) / 1e9},
                "hash_kernel": ({"kernel": "sqx::k_xxh3_128_rows", "ms": hms, "solo_ms": None if kern_solo is None else kern_solo["hash"],
                                 "note": "fused form: reads the 64 B of sums per KiB that stage 1 left (scramble chain + tail of the long path), not the chunks; "
                                         "timed under two batches in flight it includes waiting for wave slots"} if fused else
                                {"kernel": "sqx::k_xxh3_128_rows", "ms": hms,
                                 "achieved": hash_bytes / (hms / 1e3) / 1e9,
                                 "frac": hash_bytes / (hms / 1e3) / 1e9 / 8000.0}),
                "dedup_ms": kern["dedup"], "compact_ms": kern["compact"],
                "read_probe_GBps": n * L / (bw_probe_ms / 1e3) / 1e9,
                "copy_probe_GBps": 2 * cp_n / (cp_ms / 1e3) / 1e9}

        others = None
        cpu = None
        host_random = None
        if world == 1 and not a.no_cpu_baseline:
            host_random = data[:max(pool_n, min(n, 4 * T_cpu + 64)) * L].cpu().numpy()
        e2e = None
        if other and not a.no_end_to_end and L == 262144:
            e2e = _end_to_end(torch, o, data, L, min(a.e2e_files, n))
        if other:
            others = {}
            # ---- configs[2]: the second half of the chunks are exact duplicates of the first half (seeded permutation)
            half = n // 2
            plan2 = np.arange(n, dtype=np.int64)
            plan2[half:] = np.random.default_rng(7).permutation(half)[:n - half]
            corpus_gpu.random_chunks_indexed(2, plan2, L, out=data)
            _, ms2, k2 = timed(3, 1)
            nn2 = is_new.cpu().numpy()
            assert store.len() == half and int(nn2[:half].sum()) == half and int(nn2[half:].sum()) == 0, "configs[2] dedup"
            # frames of the new half: raw-block frames checked on the device, a sample against the oracle
            store.reset()
            co2, cl2 = packer.pack(data, offs, lens, hashes, is_new, comp, layout="slots")
            assert ((cl2 > 0) == (nn2 == 1)).all()
            rows2 = torch.from_numpy(np.nonzero(cl2 == RAW_FRAME_256K)[0]).cuda()
            assert _check_raw_frames(torch, comp, int(co2[1] - co2[0]), data, n, L, rows2) == 0
            for i in (0, half // 2, half - 1):
                fr = comp[int(co2[i]):int(co2[i]) + int(cl2[i])].cpu().numpy().tobytes()
                assert fr == o.compress_chunk(corpus.random_bytes(2, int(plan2[i]), L))
            cl2_total = int(cl2.astype(np.uint64).sum())
            bytes2 = n * L + cl2_total + 16 * n
            others["configs[2]"] = {
                "workload": f"{n} x {L // 1024} KiB chunks, the second half exact duplicates of the first (seeded permutation)",
                "ms_per_step": ms2, "MiB_per_s": n * L / 2**20 / (ms2 / 1e3), "steps": 3, "warmup": 1,
                "unique_chunks": half, "algorithmic_bytes": bytes2,
                "achieved_GBps": bytes2 / (ms2 / 1e3) / 1e9, "frac": bytes2 / (ms2 / 1e3) / 1e9 / 8000.0,
                "kernels_ms": {"hash": k2["hash"], "dedup": k2["dedup"], "stage1": k2["fast"], "stage2": k2["compress"]},
                "parity": f"is_new == first-occurrence model; {int(rows2.numel())} raw-block frames checked on device; 3 frames vs oracle"}

        if world == 1 and not a.no_cpu_baseline:
            # ---- CPU baseline: the oracle's restatement of the reference's whole pipeline -- pool of
            # workers over the files, sharded dedup map with the lock held during compression, ONE writer thread --
            # at -j = usable CPUs and at -j 25 (the reference's default, src/cmd/mod.rs:16), on a bounded sample of
            # the headline workload (the buffer is cycled; each pass counts as new files).
            pool_h = host_random.size // L
            if pool_n:
                # the timed pass also keeps (length, XXH3-128) of every frame of its first sweep over the sample: parity
                # against the GPU's frames of the same chunks, thousands of them instead of a hand-picked few
                _bind_oracle_bench(o)
                pj, ol, od, _cov = _cpu_pipeline2(o, host_random, offs[:pool_h], lens[:pool_h], T_cpu, max(2.0, a.cpu_seconds * 0.5))
                got = np.nonzero(ol[:pool_n])[0]
                assert (ol[got] == head_len[got]).all(), "headline: a frame's length differs from the oracle pipeline's"
                assert (od[got] == head_dig[got]).all(), "headline: a frame differs from the oracle pipeline's (XXH3-128 of the frame)"
                parity["frames_equal_to_oracle_frame_length_and_xxh3"] = int(len(got))
            else:
                pj = _cpu_pipeline(o, host_random, pool_h, L, T_cpu, max(2.0, a.cpu_seconds * 0.5))
            pool_n = pool_h
            p25 = _cpu_pipeline(o, host_random, pool_n, L, 25, max(2.0, a.cpu_seconds * 0.4))
            # and the bare hash + compress loop of round 1 (no map, no writer), for continuity
            o.lib.sqo_bench_insert_threads.restype = ctypes.c_uint64
            o.lib.sqo_bench_insert_threads.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint, ctypes.c_double]
            tp0 = time.perf_counter()
            tot = int(o.lib.sqo_bench_insert_threads(host_random.ctypes.data, pool_n, L, T_cpu, 3.0))
            bare = tot * L / 2**20 / (time.perf_counter() - tp0)
            cpu = {"value": pj["MiB_per_s"], "unit": "MiB/s", "cores": T_cpu, "kind": "port",
                   "sample": f"{pj['chunks']} chunks of the headline workload through the oracle's pack pipeline "
                             f"(oracle/bench_threads.c sqo_bench_pipeline: {T_cpu} worker threads over the files, 64-shard "
                             f"dedup map with the shard lock held during compression, one writer thread into a memory sink) "
                             f"in {pj['seconds']:.1f} s; host has {os.cpu_count()} logical CPUs, {T_cpu} usable by this process",
                   "j25": {"MiB_per_s": p25["MiB_per_s"], "chunks": p25["chunks"], "seconds": p25["seconds"],
                           "note": "-j 25 = the reference's default max_threads (src/cmd/mod.rs:16), same pipeline"},
                   "hash_compress_only_MiB_per_s": bare,
                   "real_libzstd_1_5_7": _real_libzstd(host_random, pool_n, L, T_cpu, 3.0)}

        if other and not a.no_random_2m:
            # ---- incompressible chunks of the reference's OWN chunk size (CHUNK_SIZE = 2 MiB, src/util/chunk.rs:11): stage 1's
            # certificate by pairs of blocks (lazy2 / row-hash finder, minMatch 5).  4,096 distinct random chunks, two batches
            # in flight; every frame must be the raw-block frame of its chunk (payload compared on the device).
            data = comp = None
            torch.cuda.empty_cache()
            nR, LR = 4096, 2 << 20
            dR = corpus_gpu.random_chunks(5, 0, nR, LR)
            offsR = np.arange(nR, dtype=np.uint64) * np.uint64(LR)
            lensR = np.full(nR, LR, dtype=np.uint32)
            bufsR = [(torch.zeros(nR * 16, dtype=torch.uint8, device="cuda"), torch.zeros(nR, dtype=torch.uint8, device="cuda"),
                      torch.empty(store.slots_bound(lensR), dtype=torch.uint8, device="cuda")) for _ in range(2)]
            def runR(count):
                inflight, r, ks = 0, None, {"fast": [], "compress": [], "hash": []}
                for k in range(count):
                    if inflight == 2:
                        r = store.pack_wait(nR)
                        t = store.timings()
                        ks["fast"].append(t.fast_ms); ks["compress"].append(t.compress_ms); ks["hash"].append(t.hash_ms)
                        inflight -= 1
                    store.pack_submit(dR, offsR, lensR, *bufsR[k & 1], reset=True)
                    inflight += 1
                while inflight:
                    r = store.pack_wait(nR)
                    t = store.timings()
                    ks["fast"].append(t.fast_ms); ks["compress"].append(t.compress_ms); ks["hash"].append(t.hash_ms)
                    inflight -= 1
                return r, ks
            runR(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stepsR = 6
            (coR, clR), ksR = runR(stepsR)
            torch.cuda.synchronize()
            msR = (time.perf_counter() - t0) / stepsR * 1e3
            decidedR = int(store.timings().fast_decided)
            frameR = 9 + 16 * 3 + LR
            assert (clR == frameR).all() and store.len() == nR
            hB, fB, cB = bufsR[(stepsR - 1) & 1]
            strideR = int(coR[1] - coR[0])
            frv = cB[:nR * strideR].view(nR, strideR)
            dvv = dR.view(nR, LR)
            for k0 in range(0, nR, 256):
                for b in range(16):
                    o_ = 9 + b * 131075 + 3
                    assert torch.equal(frv[k0:k0 + 256, o_:o_ + 131072], dvv[k0:k0 + 256, b * 131072:(b + 1) * 131072]), "random 2 MiB: a raw block differs from its input"
            for i in (0, nR - 1):
                c_ = dR[i * LR:(i + 1) * LR].cpu().numpy().tobytes()
                assert cB[int(coR[i]):int(coR[i]) + frameR].cpu().numpy().tobytes() == o.compress_chunk(c_)
                assert hB[16 * i:16 * i + 16].cpu().numpy().tobytes() == o.hash_chunk(c_)
            s1 = float(np.mean(ksR["fast"]))
            others["random_2MiB_chunks"] = {
                "workload": f"{nR} x 2 MiB distinct random chunks (the reference's CHUNK_SIZE), strategy lazy2 + row hash, two batches in flight",
                "ms_per_step": msR, "MiB_per_s": nR * LR / 2**20 / (msR / 1e3), "steps": stepsR, "warmup": 2,
                "chunks_decided_by_stage1": decidedR,
                "kernels_ms": {"hash": float(np.mean(ksR["hash"])), "stage1": s1, "stage2": float(np.mean(ksR["compress"]))},
                "frac": (nR * LR + nR * frameR + 16 * nR) / (msR / 1e3) / 1e9 / 8000.0,
                "stage1_frac": (nR * LR + decidedR * frameR) / (s1 / 1e3) / 1e9 / 8000.0,
                "parity": f"all {nR} frames are the raw-block frame of their chunk (16 block payloads each compared on the device); 2 frames and hashes vs the oracle"}
            del dR, bufsR, frv, dvv, cB, hB, fB
            torch.cuda.empty_cache()

        if other and text_host is not None:
            # ---- compressible chunks of the two small size classes (the strategy of every BASELINE chunk size once the
            # data is not random): text-like chunks of 256 KiB and 64 KiB (btlazy2) and 12 KiB (btopt), every
            # compressor wave busy.  Distinct chunks: the host text (128 MiB) cut into chunks, and for the 256 KiB
            # class rotated copies of it (rotation by a different number of bytes per copy).
            _bind_oracle_bench(o)
            data = comp = None
            torch.cuda.empty_cache()
            tdev = torch.from_numpy(text_host).cuda()
            small = {}
            # (chunk counts: three, four and eight rounds of the 2,816 compressor waves the kernel's second build keeps resident --
            # squish_s2_w3.co, chosen for the timed batch because the warm-up batch before it was compressible)
            for name, Ls, ns in (("text_256KiB_btlazy2", 262144, 8448), ("text_64KiB_btlazy2", 65536, 11264), ("text_12KiB_btopt", 12288, 22528)):
                per = (128 << 20) // Ls
                reps = (ns + per - 1) // per
                src = tdev[:per * Ls].view(per, Ls)
                ds = torch.empty(reps * per * Ls, dtype=torch.uint8, device="cuda")
                for r in range(reps):
                    ds.view(reps, per, Ls)[r] = src if r == 0 else torch.roll(src, shifts=1 + 4099 * r, dims=1)
                offs_s = np.arange(ns, dtype=np.uint64) * np.uint64(Ls)
                lens_s = np.full(ns, Ls, dtype=np.uint32)
                hs = torch.zeros(ns * 16, dtype=torch.uint8, device="cuda")
                fs = torch.zeros(ns, dtype=torch.uint8, device="cuda")
                cs = torch.empty(store.slots_bound(lens_s), dtype=torch.uint8, device="cuda")
                store.pack_submit(ds, offs_s, lens_s, hs, fs, cs, reset=True)     # warm-up
                store.pack_wait(ns)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                store.pack_submit(ds, offs_s, lens_s, hs, fs, cs, reset=True)
                co_s, cl_s = store.pack_wait(ns)
                torch.cuda.synchronize()
                ms_s = (time.perf_counter() - t0) * 1e3
                tm = store.timings()
                assert store.len() == ns and int(fs.sum().item()) == ns
                hh = hs.cpu().numpy().reshape(ns, 16)
                samp = sorted({int(x) for x in np.linspace(0, ns - 1, 12)})
                for i in samp:
                    c = ds[i * Ls:(i + 1) * Ls].cpu().numpy().tobytes()
                    assert hh[i].tobytes() == o.hash_chunk(c)
                    assert cs[int(co_s[i]):int(co_s[i]) + int(cl_s[i])].cpu().numpy().tobytes() == o.compress_chunk(c), f"{name} frame {i} differs from the oracle"
                done, bad = _decode_verify(torch, o, cs, co_s, cl_s, np.arange(ns), lens_s, hh, T_cpu, 6.0)
                assert bad == 0, f"{name}: {bad} frames do not decode to their chunk"
                cl_tot = int(cl_s.astype(np.uint64).sum())
                small[name] = {"chunks": ns, "chunk_bytes": Ls, "ms_per_step": ms_s, "MiB_per_s": ns * Ls / 2**20 / (ms_s / 1e3),
                               "ratio": cl_tot / (ns * Ls), "kernels_ms": {"hash": tm.hash_ms, "stage1": tm.fast_ms, "stage2": tm.compress_ms},
                               "stage2_build": "3 waves per SIMD, 11 per CU (squish_s2_w3.co)" if store.stage2_kernel()[1] else "2 waves per SIMD, 8 per CU",
                               "frac": (ns * Ls + cl_tot + 16 * ns) / (ms_s / 1e3) / 1e9 / 8000.0,
                               "parity": f"{len(samp)} frames bit-exact vs the oracle; {done} of {ns} frames decoded on the host: length and XXH3-128 of the output equal the chunk's"}
                del ds, cs
            others["compressible_small_classes"] = small
            del tdev
            torch.cuda.empty_cache()

        if other and mixed_host is not None:
            # ---- configs[3]: Silesia-style mixed-entropy corpus replicated to ~32 GiB: 2 MiB chunks + ragged tails
            _bind_oracle_bench(o)
            nb, reps, L3 = a.config3_chunks, max(1, a.config3_replicas), 2 << 20
            n3 = nb * reps
            data = comp = None
            torch.cuda.empty_cache()
            base3 = torch.from_numpy(mixed_host).cuda().view(nb, L3)
            d3 = torch.empty(n3 * L3, dtype=torch.uint8, device="cuda")
            shifts = [0] + [1 + 4099 * r for r in range(1, reps)]        # copy r = every chunk rotated by shifts[r] bytes
            for r in range(reps):
                d3.view(reps, nb, L3)[r] = base3 if r == 0 else torch.roll(base3, shifts=shifts[r], dims=1)
            del base3
            offs3 = np.arange(n3, dtype=np.uint64) * np.uint64(L3)
            lens3 = np.full(n3, L3, dtype=np.uint32)
            # tails: files of 1-50 MiB end in a chunk of any length -- 1.2 % of the chunks are cut short, a third each in
            # the size class of every strategy (btopt <= 16 KiB, btlazy2 <= 256 KiB, lazy2 above)
            rng3 = np.random.default_rng(33)
            tails = rng3.choice(n3, size=max(3, (n3 // 256) * 3), replace=False)
            for k, i in enumerate(tails):
                lens3[i] = int(rng3.integers(1, 16385)) if k % 3 == 0 else int(rng3.integers(16385, 262145)) if k % 3 == 1 else int(rng3.integers(262145, L3))
            h3 = torch.zeros(n3 * 16, dtype=torch.uint8, device="cuda")
            new3 = torch.zeros(n3, dtype=torch.uint8, device="cuda")
            c3 = torch.empty(store.slots_bound(lens3), dtype=torch.uint8, device="cuda")

            def step3():
                store.pack_submit(d3, offs3, lens3, h3, new3, c3, reset=True)
                return store.pack_wait(n3)

            step3()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ks = {"hash": [], "fast": [], "compress": []}
            for _ in range(a.config3_steps):
                co3, cl3 = step3()
                t = store.timings()
                ks["hash"].append(t.hash_ms)
                ks["fast"].append(t.fast_ms)
                ks["compress"].append(t.compress_ms)
            torch.cuda.synchronize()
            ms3 = (time.perf_counter() - t0) / a.config3_steps * 1e3
            assert store.len() == n3 and int(new3.sum().item()) == n3
            cl3_total = int(cl3.astype(np.uint64).sum())
            in3_total = int(lens3.astype(np.uint64).sum())
            # (length, XXH3-128) of every frame, by the product's own hash kernel (pinned by tests/test_hash_gpu.py)
            dig3 = torch.zeros(n3 * 16, dtype=torch.uint8, device="cuda")
            store.hash_device(c3, co3, cl3, dig3)
            dig3 = dig3.cpu().numpy().reshape(n3, 16)
            hh3 = h3.cpu().numpy().reshape(n3, 16)
            gather = max(store.gather_rate(d3) for _ in range(2))   # random 16-byte loads per second over this corpus
            bytes3 = in3_total + cl3_total + 16 * n3
            # CPU baseline = the oracle's pack pipeline over the first copy (the host corpus, same tails); every chunk it
            # gets through is also a parity check: its frame's length and XXH3-128 against the GPU frame's
            cmix, n_cmp = None, 0
            covered = np.zeros(n3, dtype=bool)
            if not a.no_cpu_baseline:
                pm, ol, od, _ = _cpu_pipeline2(o, mixed_host, offs3[:nb], lens3[:nb], T_cpu, max(3.0, a.cpu_seconds * 0.6))
                got = np.nonzero(ol)[0]
                assert (ol[got] == cl3[got]).all(), "configs[3]: a frame's length differs from the oracle's"
                assert (od[got] == dig3[got]).all(), "configs[3]: a frame differs from the oracle's (XXH3-128 of the frame)"
                covered[got] = True
                n_cmp = len(got)
                cmix = {"value": pm["MiB_per_s"], "unit": "MiB/s", "cores": T_cpu, "kind": "port",
                        "real_libzstd_1_5_7": _real_libzstd(mixed_host, min(nb, 8 * T_cpu), L3, T_cpu, 4.0),
                        "sample": f"{pm['chunks']} of these chunks through the oracle's pack pipeline in {pm['seconds']:.1f} s"}
            # a few chunks of the rotated copies and tails of every class against the oracle directly
            direct = [int(i) for i in list(tails[:6]) + [nb + 1, n3 - 1] if not covered[int(i)]]
            for i in direct:
                c = d3[i * L3:i * L3 + int(lens3[i])].cpu().numpy().tobytes()
                assert hh3[i].tobytes() == o.hash_chunk(c)
                assert c3[int(co3[i]):int(co3[i]) + int(cl3[i])].cpu().numpy().tobytes() == o.compress_chunk(c), f"configs[3] frame {i} differs from the oracle"
                covered[i] = True
            # every other frame: decoded on the host (the oracle's RFC 8878 decoder), length and XXH3-128 of the output
            # must be the chunk's (the chunk's hash as the GPU computed it), as far as the time budget goes
            rest = np.nonzero(~covered)[0]
            done, bad = _decode_verify(torch, o, c3, co3, cl3, rest, lens3, hh3, T_cpu, a.verify_seconds)
            assert bad == 0, f"configs[3]: {bad} frames do not decode to their chunk"
            # ---- the way back: every frame of the leg decoded by the GPU decoder (sq_decompress_batch, one
            # wave per record) into a third buffer; XXH3-128 of every output must be the chunk's hash
            unpack3 = None
            try:
                store.trim()   # (the compressor's workspace pools: 65 GB per lane for 2 MiB chunks; the decoder needs none of it)
                o3 = torch.empty(n3 * L3, dtype=torch.uint8, device="cuda")
                store.decompress_device(c3, co3, cl3, o3, offs3, lens3)            # warm-up (buffers, code)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ol3, st3 = store.decompress_device(c3, co3, cl3, o3, offs3, lens3)
                torch.cuda.synchronize()
                msd = (time.perf_counter() - t0) * 1e3
                assert (st3 == 0).all() and (ol3 == lens3).all(), "configs[3]: a frame did not decode on the GPU"
                dg = torch.zeros(n3 * 16, dtype=torch.uint8, device="cuda")
                store.hash_device(o3, offs3, lens3, dg)
                assert torch.equal(dg, h3), "configs[3]: GPU pack -> GPU unpack does not give the chunks back"
                unpack3 = {"ms": msd, "MiB_per_s": in3_total / 2**20 / (msd / 1e3), "kernel_ms": float(store.timings().compact_ms),
                           "compressed_MiB_per_s": cl3_total / 2**20 / (msd / 1e3),
                           "parity": f"all {n3} frames decoded by sq_decompress_batch; XXH3-128 of every output equals the chunk's hash (computed by the GPU before packing)"}
                del o3, dg
            except torch.OutOfMemoryError:
                unpack3 = {"skipped": "not enough device memory for a third copy of the corpus"}
            others["configs[3]"] = {
                "workload": f"{n3} chunks = {reps} copies (each rotated by a different number of bytes: no two chunks equal) of {nb} distinct 2 MiB "
                            f"Silesia-style mixed-entropy chunks (text / records / runs / skewed / random regions), {len(tails)} of them cut short as "
                            f"file tails of every strategy's size class: {in3_total / 2**30:.1f} GiB, all unique",
                "ms_per_step": ms3, "MiB_per_s": in3_total / 2**20 / (ms3 / 1e3), "steps": a.config3_steps, "warmup": 1,
                "ratio": cl3_total / in3_total, "algorithmic_bytes": bytes3,
                "achieved_GBps": bytes3 / (ms3 / 1e3) / 1e9, "frac": bytes3 / (ms3 / 1e3) / 1e9 / 8000.0,
                "kernels_ms": {"hash": float(np.mean(ks["hash"])), "stage1": float(np.mean(ks["fast"])), "stage2": float(np.mean(ks["compress"]))},
                "MiB_per_s_per_wave": in3_total / 2**20 / (float(np.mean(ks["compress"])) / 1e3) / min(n3, 2048),
                "parity": {"frames": n3, "equal_to_oracle_frame_length_and_xxh3": n_cmp, "equal_to_oracle_bytes": len(direct),
                           "decoded_and_rehashed_on_host": done, "unchecked_time_budget": int(len(rest) - done)},
                "traffic": _mixed_traffic(),
                "gpu_unpack": unpack3,
                "random_access_probe": {"requests_per_s": gather,
                                        "note": "sq_calibrate_gather: 16-byte loads at random places of this corpus, 8 in flight per lane (DESIGN.md 4.3)"},
                "cpu_baseline": cmix, "gen_seconds_host": mixed_gen_s}
            del d3, c3

        if world == 1:
            wl = f"{n} x {L // 1024} KiB unique random chunks (BASELINE configs[1]), xoshiro256** seeded, resident in HBM"
        else:
            wl = (f"{n} x {L // 1024} KiB random chunks per GPU, 30 % of all {N} chunks exact copies of another chunk "
                  f"(BASELINE configs[4] shape), resident in HBM; the global chunk order is dealt to the ranks in "
                  + (f"block-cyclic stripes of {a.stripe} chunks" if cyclic else "contiguous ranges (rank 0 holds first occurrences only)")
                  + f"; chunks each rank has to compress (first occurrences): {new_per_rank}")
        out = {"metric": "pack throughput MiB/s", "value": value, "unit": "MiB/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": wl, "chunks_per_gpu": n, "chunk_bytes": L, "unique_chunks": int(unique),
                          "compressed_bytes_rank0": clen_total, "parallelism": f"shard{world}",
                          "output_layout": a.layout, "parity": parity, "gen_seconds": gen_s},
               "roofline": roof, "cpu_baseline": cpu, "end_to_end": e2e, "other_configs": others}
        if world > 1:
            out["per_rank"] = {"note": "each rank's own clock over the timed region (the line's ms_per_step is their maximum); host times are "
                                       "what the rank's Python thread spent per step in each phase of ShardedPacker.submit / wait; chunks to "
                                       "compress per rank are in config.workload",
                               "ranks": list(rank_stats), "first_occurrences_per_rank": new_per_rank}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    store.close()


if __name__ == "__main__":
    main()
