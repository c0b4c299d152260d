"""CPU (gloo, world_size 2): the multi-GPU sharding + hash all-gather + global first-occurrence
logic of squishrs_amd/parallel.py, with an oracle-backed stand-in for the device stages (test
infrastructure; the product passes a HIP ChunkStore)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleStages:
    """Stand-in with the ChunkStore stage API; hashing/compression by the CPU oracle."""

    def __init__(self):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        self.o = oracle_lib.load()
        self.seen = set()

    def hash_device(self, data, offs, lens, hashes):
        d = data.numpy()
        for i, (o_, l_) in enumerate(zip(offs, lens)):
            h = self.o.hash_chunk(d[int(o_):int(o_) + int(l_)].tobytes())
            hashes[16 * i:16 * i + 16] = torch.frombuffer(bytearray(h), dtype=torch.uint8)

    def dedup_device(self, all_hashes, n_total, my_first, my_n, is_new):
        hs = all_hashes.numpy().reshape(-1, 16)
        for g in range(n_total):            # commit ALL hashes in global order (sq_dedup_global)
            k = hs[g].tobytes()
            new = k not in self.seen
            self.seen.add(k)
            if my_first <= g < my_first + my_n:
                is_new[g - my_first] = 1 if new else 0

    def compress_device(self, data, offs, lens, select, comp):
        d = data.numpy()
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        pos = 0
        for i, (o_, l_) in enumerate(zip(offs, lens)):
            off[i] = pos
            if select[i]:
                f = self.o.compress_chunk(d[int(o_):int(o_) + int(l_)].tobytes())
                comp[pos:pos + len(f)] = torch.frombuffer(bytearray(f), dtype=torch.uint8)
                pos += len(f)
        off[len(lens)] = pos
        return off

    def compress_device_slots(self, data, offs, lens, select, comp):
        """slots layout (sq_compress_batch_slots): frame i stays at its slot offset"""
        d = data.numpy()
        off = np.zeros(len(lens), dtype=np.uint64)
        ln = np.zeros(len(lens), dtype=np.uint32)
        pos = 0
        for i, (o_, l_) in enumerate(zip(offs, lens)):
            off[i] = pos
            if select[i]:
                f = self.o.compress_chunk(d[int(o_):int(o_) + int(l_)].tobytes())
                comp[pos:pos + len(f)] = torch.frombuffer(bytearray(f), dtype=torch.uint8)
                ln[i] = len(f)
            pos += (int(l_) + (int(l_) >> 8) + 64 + 32 + 15) // 16 * 16
        return off, ln


    # ---- the split form of the pipelined sharded packer (sq_shard_submit_hash / _wait_hash / _submit_rest / sq_pack_wait)
    def shard_submit_hash(self, data, offs, lens, hashes):
        assert getattr(self, "_pending", None) is None, "two hash steps without the rest of the first"
        self.hash_device(data, offs, lens, hashes)
        self._pending = (data, offs, lens)

    def shard_wait_hash(self):
        assert self._pending is not None

    def shard_submit_rest(self, all_hashes, n_total, my_first, is_new, comp, reset=False):
        data, offs, lens = self._pending
        self._pending = None
        if reset:
            self.seen = set()
        self.dedup_device(all_hashes, n_total, my_first, len(lens), is_new)
        if not hasattr(self, "_done"):
            self._done = []
        assert len(self._done) < 2, "more than two batches in flight"
        self._done.append(self.compress_device_slots(data, offs, lens, is_new, comp))

    def shard_submit_rest_idx(self, all_hashes, n_total, my_idx, is_new, comp, reset=False):
        """all_hashes in GLOBAL order; this rank's chunk i is entry my_idx[i] (sq_shard_submit_rest_idx)"""
        data, offs, lens = self._pending
        self._pending = None
        if reset:
            self.seen = set()
        flags = torch.zeros(max(n_total, 1), dtype=torch.uint8)
        self.dedup_device(all_hashes, n_total, 0, n_total, flags)
        for i, g in enumerate(my_idx.tolist()):
            is_new[i] = flags[g]
        if not hasattr(self, "_done"):
            self._done = []
        assert len(self._done) < 2, "more than two batches in flight"
        self._done.append(self.compress_device_slots(data, offs, lens, is_new, comp))

    def pack_wait(self, n):
        off, ln = self._done.pop(0)
        assert len(off) == n
        return off, ln


def _make_chunks():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpus
    base = [corpus.text_like(i, 3000 + 500 * i) for i in range(6)]
    # global order with cross-shard duplicates: 11 chunks, shard0 gets 6, shard1 gets 5
    order = [0, 1, 2, 0, 3, 1, 4, 2, 5, 5, 3]
    return [base[i] for i in order], order


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from squishrs_amd.parallel import ShardedPacker, shard_range
    chunks, order = _make_chunks()
    first, n = shard_range(len(chunks), rank, world)
    mine = chunks[first:first + n]
    lens = np.array([len(c) for c in mine], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    data = torch.frombuffer(bytearray(b"".join(mine)), dtype=torch.uint8)
    hashes = torch.zeros(n * 16, dtype=torch.uint8)
    is_new = torch.zeros(n, dtype=torch.uint8)
    comp = torch.zeros(int(lens.sum()) + 4096 + 128 * n, dtype=torch.uint8)
    st = OracleStages()
    p = ShardedPacker(st)
    coff = p.pack(data, offs, lens, hashes, is_new, comp)
    # the same batch again through the slots layout (bench.py's default) on a fresh dedup set
    st2 = OracleStages()
    is_new2 = torch.zeros(n, dtype=torch.uint8)
    soff, slen = ShardedPacker(st2).pack(data, offs, lens, hashes, is_new2, comp, layout="slots")
    assert is_new2.tolist() == is_new.tolist()
    assert [int(x) for x in slen] == [int(coff[j + 1] - coff[j]) for j in range(n)]
    assert all(int(soff[j + 1]) >= int(soff[j]) + int(slen[j]) for j in range(n - 1))
    q.put((rank, first, is_new.tolist(), len(st.seen), [int(x) for x in coff]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_contiguous():
    sys.path.insert(0, ROOT)
    from squishrs_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 4, 8):
            tot = 0
            for r in range(w):
                f, c = shard_range(n, r, w)
                assert f == tot
                tot += c
            assert tot == n


def test_stripe_assign_balances_first_occurrences():
    """Block-cyclic stripes: every chunk has exactly one owner, perm maps the global order onto the rank-major
    concatenation of an all-gather, and on BASELINE configs[4]'s plan (30 % copies) the chunks a rank has to compress
    -- first occurrences in global order -- are spread evenly, where contiguous ranges leave rank 0 with all of its
    chunks and the last rank with the fewest."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpus
    from squishrs_amd.parallel import stripe_assign
    for n_total, world, stripe in ((100, 4, 16), (37, 3, 1), (5, 4, 2), (64, 8, 8), (0, 2, 4)):
        owner, counts, perm = stripe_assign(n_total, world, stripe)
        assert len(owner) == n_total and counts.sum() == n_total and sorted(perm.tolist()) == list(range(n_total))
        # rank-major concatenation: rank r's chunks in global order
        rank_major = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)]) if n_total else np.zeros(0, dtype=np.int64)
        assert (rank_major[perm] == np.arange(n_total)).all()
        if n_total >= world * stripe:
            assert counts.max() - counts.min() <= stripe
    world, n = 8, 4096
    plan = corpus.cross_shard_plan(world, n)
    _, first_idx = np.unique(plan, return_index=True)
    new = np.zeros(world * n, dtype=bool)
    new[first_idx] = True
    owner, _, _ = stripe_assign(world * n, world, 16)
    cyc = [int(new[owner == r].sum()) for r in range(world)]
    con = [int(new[r * n:(r + 1) * n].sum()) for r in range(world)]
    assert con[0] > 0.95 * n and con[-1] < 0.5 * n                 # contiguous: rank 0 compresses (nearly) everything it has
    assert max(cyc) - min(cyc) < 0.05 * n and max(cyc) < 0.75 * n  # stripes: 70 % each, within a few percent


def test_two_rank_global_first_occurrence():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chunks, order = _make_chunks()
    seen, exp = set(), []
    for i in order:
        exp.append(0 if i in seen else 1)
        seen.add(i)
    got = res[0][2] + res[1][2]
    assert got == exp                                   # first occurrence in GLOBAL index order
    assert res[0][3] == res[1][3] == len(seen)          # every rank ends with the same set
    # duplicates produce empty frames; new chunks non-empty
    for r in res:
        off = r[4]
        for j, flag in enumerate(r[2]):
            assert (off[j + 1] - off[j] > 0) == bool(flag)


# ---- end-to-end sharded pack: N ranks -> ONE archive, byte-identical to the `-j 1` archive -----------------------
def _sharded_files():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpus
    big = corpus.mixed(11, 2 * 2097152 + 12345)              # 3 chunks: two full + a tail
    files = [("a/big.bin", big), ("a/empty", b""), ("b/t1.txt", corpus.text_like(1, 70000)),
             ("b/copy_of_big.bin", big),                       # cross-shard duplicates of every chunk of big.bin
             ("c/r.bin", corpus.random_bytes(3, 1, 40000)), ("c/t1_again.txt", corpus.text_like(1, 70000)),
             ("d/small", b"hello squish"), ("d/t2.txt", corpus.text_like(2, 150000)), ("e/z", bytes(300000))]
    return files


def _sharded_worker(rank, world, port, out_path, batch_chunks, stripe=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from squishrs_amd.sharded_pack import pack_sharded
    files = _sharded_files()
    pack_sharded([(p, None) for p, _ in files], out_path, OracleStages(), device="cpu", timestamp=1234567,
                 batch_chunks=batch_chunks, contents=[c for _, c in files], stripe=stripe)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,batch_chunks,stripe", [(2, 4096, 2), (4, 3, 1), (2, 5, 16), (4, 6, 2), (3, 4096, 1)])
def test_sharded_pack_archive_equals_single_writer_archive(tmp_path, world, batch_chunks, stripe):
    """gloo ranks with CPU stand-ins for the device stages: the ONE archive the ranks write together (every rank
    pwrite()s its own records at the offsets the exchanged record sizes give; block-cyclic stripes; two super-batches
    in flight) must be the oracle's `.squish` restatement byte for byte (same timestamp) -- records in global
    chunk-index order, duplicates that live on other ranks (and in other super-batches) dropped, count patched,
    manifest complete.  Super-batches smaller than the world (ranks without a chunk), stripes larger than a batch."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / f"w{world}.squish")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, out, batch_chunks, stripe)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    want = oracle_lib.load().pack_archive(_sharded_files(), timestamp=1234567)
    got = open(out, "rb").read()
    assert got == want


def test_sharded_pack_single_process_and_walk_order(tmp_path):
    """world size 1 (no process group): same bytes; and walk_files follows the driver's DFS (stack.pop, OS order)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from squishrs_amd.sharded_pack import pack_sharded, walk_files
    files = _sharded_files()
    root = tmp_path / "tree"
    for p, c in files:
        (root / p).parent.mkdir(parents=True, exist_ok=True)
        (root / p).write_bytes(c)
    walked = walk_files(str(root) + "/")
    assert sorted(r for r, _ in walked) == sorted(p for p, _ in files)
    out = str(tmp_path / "one.squish")
    size = pack_sharded(walked, out, OracleStages(), device="cpu", timestamp=99, batch_chunks=3)
    body = {p: c for p, c in files}
    want = oracle_lib.load().pack_archive([(r, body[r]) for r, _ in walked], timestamp=99)
    assert open(out, "rb").read() == want and size == len(want)


def _uniform_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from squishrs_amd.parallel import ShardedPacker
    p = ShardedPacker(OracleStages())
    n = 5
    mine = torch.arange(n * 16, dtype=torch.uint8) + rank * 80
    a = p.gather_hashes(mine, n, uniform=True)
    b = p.gather_hashes(mine, n)
    q.put((rank, a[0].tolist() == b[0].tolist(), a[1], a[2], b[1], b[2]))
    dist.barrier()
    dist.destroy_process_group()


def test_uniform_gather_skips_the_count_exchange():
    """bench.py passes uniform=True (every rank holds n chunks): same gathered list, totals and first index."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uniform_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, nt_a, mf_a, nt_b, mf_b in res:
        assert same and nt_a == nt_b == 15 and mf_a == mf_b == 5 * rank


def _pipelined_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from squishrs_amd.parallel import ShardedPacker, shard_range
    chunks, order = _make_chunks()
    first, n = shard_range(len(chunks), rank, world)
    mine = chunks[first:first + n]
    lens = np.array([len(c) for c in mine], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    data = torch.frombuffer(bytearray(b"".join(mine)), dtype=torch.uint8)
    bufs = [(torch.zeros(n * 16, dtype=torch.uint8), torch.zeros(n, dtype=torch.uint8),
             torch.zeros(int(lens.sum()) + 4096 + 256 * n, dtype=torch.uint8)) for _ in range(2)]
    st = OracleStages()
    p = ShardedPacker(st)
    # three batches, two in flight: the first on a fresh set, the second WITHOUT a reset (everything is a duplicate
    # now), the third with a reset again
    got = []
    p.submit(data, offs, lens, *bufs[0], reset=True)
    p.submit(data, offs, lens, *bufs[1], reset=False)
    got.append((bufs[0][1].tolist(), [int(x) for x in p.wait()[1]]))
    got.append((bufs[1][1].tolist(), [int(x) for x in p.wait()[1]]))
    p.submit(data, offs, lens, *bufs[0], reset=True)
    got.append((bufs[0][1].tolist(), [int(x) for x in p.wait()[1]]))
    # the synchronous form on a fresh stand-in is the model
    st2 = OracleStages()
    is_new = torch.zeros(n, dtype=torch.uint8)
    _, slen = ShardedPacker(st2).pack(data, offs, lens, bufs[1][0], is_new, bufs[1][2], layout="slots")
    q.put((rank, got, is_new.tolist(), [int(x) for x in slen]))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_sharded_packer_two_in_flight():
    """ShardedPacker.submit / wait (the form bench.py --gpus N times): hash, exchange, global dedup + compression per
    batch with two batches in flight; results equal the synchronous form's, and a batch submitted without a reset sees
    the set the batch before left."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, model_new, model_len in res:
        assert got[0] == (model_new, model_len)
        assert got[1][0] == [0] * len(model_new) and got[1][1] == [0] * len(model_new)
        assert got[2] == (model_new, model_len)


def test_pwritev_all_loops_over_short_writes_and_refuses_zero(tmp_path):
    """ADVICE r3 (sharded_pack.py:190): a record run must reach the file whole even when the kernel writes short (Linux
    caps one write at 0x7ffff000 bytes); a write of 0 bytes is an error, not a silent truncation."""
    sys.path.insert(0, ROOT)
    from squishrs_amd.sharded_pack import pwritev_all
    path = tmp_path / "f.bin"
    path.write_bytes(b"\xee" * 10)
    fd = os.open(str(path), os.O_WRONLY)
    calls = []

    def short(fd_, bufs, at):   # writes at most 7 bytes per call, across iovec boundaries
        flat = b"".join(bytes(b) for b in bufs)[:7]
        calls.append((len(bufs), at, len(flat)))
        return os.pwrite(fd_, flat, at)

    bufs = [b"abc", memoryview(np.arange(40, dtype=np.uint8)), b"", b"tail-of-the-run"]
    n = pwritev_all(fd, bufs, 5, _pwritev=short)
    os.close(fd)
    want = b"abc" + bytes(range(40)) + b"tail-of-the-run"
    assert n == len(want) and len(calls) == (len(want) + 6) // 7
    assert path.read_bytes() == b"\xee" * 5 + want
    fd = os.open(str(path), os.O_WRONLY)
    with pytest.raises(OSError):
        pwritev_all(fd, [b"xyz"], 0, _pwritev=lambda *_: 0)
    os.close(fd)
    # more iovecs than one pwritev takes
    many = [bytes([i & 255]) * 3 for i in range(2500)]
    fd = os.open(str(path), os.O_WRONLY | os.O_TRUNC)
    assert pwritev_all(fd, many, 0) == 7500
    os.close(fd)
    assert path.read_bytes() == b"".join(many)


def test_sharded_pack_short_writes_small_runs_and_threaded_reader(tmp_path, monkeypatch):
    """The same archive through a writer that always writes short, with runs flushed every few hundred bytes, files read
    from disk by the threaded pread reader into the staging buffer, and the per-stage statistics filled in."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import squishrs_amd.sharded_pack as sp
    files = _sharded_files()
    root = tmp_path / "tree"
    for p, c in files:
        (root / p).parent.mkdir(parents=True, exist_ok=True)
        (root / p).write_bytes(c)
    walked = sp.walk_files(str(root))
    body = {p: c for p, c in files}
    want = oracle_lib.load().pack_archive([(r, body[r]) for r, _ in walked], timestamp=7)
    monkeypatch.setattr(sp, "WRITE_RUN_BYTES", 300)

    def short(fd_, bufs, at):
        flat = b"".join(bytes(b) for b in bufs)[:1001]
        return os.pwrite(fd_, flat, at)

    for threads in (1, 4):
        out = str(tmp_path / ("t%d.squish" % threads))
        stats = {}
        size = sp.pack_sharded(walked, out, OracleStages(), device="cpu", timestamp=7, batch_chunks=4, read_threads=threads,
                               stats=stats, _pwritev=short)
        assert open(out, "rb").read() == want and size == len(want)
        assert stats["chunks"] >= stats["compressed"] > 0 and stats["super_batches"] >= 2
        assert all(k in stats for k in ("read", "h2d", "submit", "wait", "d2h", "exchange2", "write"))
