"""GPU parity at the shapes of BASELINE configs[2] and [3], the ABI contract of sq_pack_batch (failure leaves
nothing committed; pointer sides; thread safety of the convenience forms, /root/reference/src/util/chunk.rs:80-100
is called from up to 25 Rayon workers) and the single-wave default-context path on incompressible chunks.
All through the C ABI (ctypes), checked against the CPU oracle."""
import ctypes
import threading

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu


def _first_occurrence(keys):
    seen, out = set(), np.zeros(len(keys), dtype=np.uint8)
    for i, k in enumerate(keys):
        if k not in seen:
            seen.add(k)
            out[i] = 1
    return out, len(seen)


def test_corpus_kernel_matches_numpy_generator():
    """tools/corpus_gen.hip is the device twin of corpus.random_bytes; gather_chunks makes exact duplicates."""
    import torch
    import corpus_gpu
    L = 4096
    d = corpus_gpu.random_chunks(9, 1000, 37, L)
    host = d.cpu().numpy().reshape(37, L)
    for i in (0, 1, 17, 36):
        assert host[i].tobytes() == corpus.random_bytes(9, 1000 + i, L)
    idx = torch.tensor([5, 5, 0, 36], dtype=torch.int64, device="cuda")
    g = torch.empty(4 * L, dtype=torch.uint8, device="cuda")
    corpus_gpu.gather_chunks(g, d, idx, L)
    gh = g.cpu().numpy().reshape(4, L)
    assert (gh[0] == host[5]).all() and (gh[1] == host[5]).all() and (gh[2] == host[0]).all() and (gh[3] == host[36]).all()


def test_config2_half_duplicates_pack_batch_and_stages(oracle):
    """BASELINE configs[2] in small: 2,400 x 256 KiB chunks, the second half exact duplicates of the first in a
    seeded permutation (plus a few duplicates INSIDE the first half), through (a) sq_pack_batch on device pointers
    and (b) the stage calls with select = is_new.  is_new against a first-occurrence model, unique count, every
    new frame's payload, duplicates' frames empty, a spread sample of frames against the oracle."""
    import torch
    import corpus_gpu
    import squishrs_amd
    n, L = 2400, 262144
    half = n // 2
    rng = np.random.default_rng(2)
    src = np.arange(n, dtype=np.int64)
    src[half:] = rng.permutation(half)
    src[100] = 7           # duplicates inside the first half too
    src[101] = 7
    src[half - 1] = 100    # base chunk 100's only copy in the first half sits at its end
    base = corpus_gpu.random_chunks(2, 0, half, L)
    data = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    corpus_gpu.gather_chunks(data, base, torch.from_numpy(src), L)
    content = src          # which base content chunk i carries
    exp_new, exp_unique = _first_occurrence([int(x) for x in content])
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=1 << 14)
    try:
        hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
        comp = torch.empty(exp_unique * (squishrs_amd.compress_bound(L) + 16), dtype=torch.uint8, device="cuda")
        # (a) sq_pack_batch, two batches (state persists), device pointers
        cut = 1000
        co1 = st.pack_device(data[:cut * L], offs[:cut], lens[:cut], hashes[:cut * 16], is_new[:cut], comp)
        used = int(co1[-1])
        co2 = st.pack_device(data[cut * L:], offs[:n - cut], lens[cut:], hashes[cut * 16:], is_new[cut:], comp[used:])
        got_new = is_new.cpu().numpy()
        assert (got_new == exp_new).all(), np.nonzero(got_new != exp_new)[0][:10]
        assert st.len() == exp_unique
        sizes = np.concatenate([np.diff(co1.astype(np.int64)), np.diff(co2.astype(np.int64))])
        assert ((sizes > 0) == (exp_new == 1)).all()
        assert (sizes[exp_new == 1] <= 262159).all() and (sizes[exp_new == 1] >= 262000).all()
        hh = hashes.cpu().numpy().reshape(n, 16)
        starts = np.concatenate([co1[:-1].astype(np.int64), co2[:-1].astype(np.int64) + used])
        host_base = base.cpu().numpy().reshape(half, L)
        sample = sorted(set([0, 7, 100, 101, 999, 1000, 1001, half - 1, half, n - 1] + [int(x) for x in rng.integers(0, n, size=20)]))
        for i in sample:
            c = host_base[int(content[i])].tobytes()
            assert hh[i].tobytes() == oracle.hash_chunk(c), i
            if exp_new[i]:
                f = comp[int(starts[i]):int(starts[i]) + int(sizes[i])].cpu().numpy().tobytes()
                assert f == oracle.compress_chunk(c), i
        # every raw-block frame carries its input (device-side, all new chunks)
        newi = np.nonzero((exp_new == 1) & (sizes == 262159))[0]
        assert len(newi) > 0.8 * exp_unique
        st_t = torch.from_numpy(starts[newi]).cuda()
        ar = torch.arange(131072, device="cuda")
        for k0 in range(0, len(newi), 256):
            s = st_t[k0:k0 + 256]
            ids = torch.from_numpy(newi[k0:k0 + 256]).cuda()
            blk0 = comp[(s[:, None] + 12 + ar[None, :]).reshape(-1)].view(len(s), -1)
            blk1 = comp[(s[:, None] + 12 + 131072 + 3 + ar[None, :]).reshape(-1)].view(len(s), -1)
            inp = data.view(n, L)[ids]
            assert torch.equal(blk0, inp[:, :131072]) and torch.equal(blk1, inp[:, 131072:])
        # (b) the stage calls with select = is_new on a fresh set: same flags, same frames
        st.reset()
        h2 = torch.zeros_like(hashes)
        n2 = torch.zeros_like(is_new)
        st.hash_device(data, offs, lens, h2)
        st.dedup_device(h2, n, 0, n, n2)
        assert torch.equal(h2, hashes) and torch.equal(n2, is_new) and st.len() == exp_unique
        slots = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
        so, sl = st.compress_device_slots(data, offs, lens, n2, slots)
        assert (sl.astype(np.int64) == sizes).all()
        for i in sample:
            if exp_new[i]:
                a = slots[int(so[i]):int(so[i]) + int(sl[i])]
                b = comp[int(starts[i]):int(starts[i]) + int(sizes[i])]
                assert torch.equal(a, b), i
    finally:
        st.close()


def test_config3_mixed_2m_chunks_full_machine(oracle):
    """BASELINE configs[3] in small: 320 Silesia-style mixed-entropy chunks of 2 MiB (40 distinct contents x 8,
    so a spread of waves sees each) + ragged tails of all three strategies in ONE batch through sq_pack_batch
    semantics (hash -> dedup -> compress new).  Distinct contents against the oracle frame for frame; the copies
    must be flagged duplicates."""
    import torch
    import squishrs_amd
    distinct = [corpus.mixed(7000 + i, 2097152) for i in range(40)]
    tails = [corpus.mixed(7100, 1234567), corpus.text_like(7101, 300001), corpus.records(7102, 200000),
             corpus.text_like(7103, 15000), corpus.make("runs", 7104, 2097152), corpus.skewed(7105, 524288), b"tail"]
    chunks = distinct + tails + distinct * 7
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    try:
        res = st.insert_batch(chunks)
        assert st.len() == len(distinct) + len(tails)
        ref = []
        for i, (c, r) in enumerate(zip(chunks, res)):
            if i < len(distinct) + len(tails):
                assert r.hash == oracle.hash_chunk(c), i
                ref.append(oracle.compress_chunk(c))
                assert r.compressed_data == ref[i], (i, len(c))
            else:
                assert r.compressed_data is None and r.hash == res[(i - len(tails)) % len(distinct)].hash
        # all 320 copies through the compress stage alone (no dedup): 320 waves busy with a 2 MiB chunk each
        from test_compress_gpu import _batch_compress
        frames = _batch_compress(st, distinct * 8)
        for i, f in enumerate(frames):
            assert f == ref[i % len(distinct)], i
    finally:
        st.close()


def test_compress_chunk_default_ctx_single_wave_incompressible(oracle):
    """The stateless form runs ONE compressor wave (and one or two stage-1 waves) on its private context.  Round 1
    saw this path hang during development on a random 70,000-byte chunk (DESIGN.md "Incidents"); pin it: random
    chunks of the btlazy2 size class, a planted one, and the 16 KiB boundary, against the oracle."""
    import squishrs_amd
    cases = [corpus.random_bytes(41, 0, 70000), corpus.random_bytes(41, 1, 262144), corpus.planted(5, 262144, 0),
             corpus.planted(6, 70000, 3), corpus.random_bytes(41, 2, 16385), corpus.random_bytes(41, 3, 16384),
             corpus.random_bytes(41, 4, 131073)]
    for c in cases:
        assert squishrs_amd.compress_chunk(c) == oracle.compress_chunk(c), len(c)
        assert squishrs_amd.hash_chunk(c) == oracle.hash_chunk(c)


def test_convenience_forms_from_many_threads(oracle):
    """hash_chunk / compress(.,12) are called from up to 25 Rayon workers in the reference (chunk.rs:80,
    cmd/mod.rs:16): sq_hash_chunk / sq_compress_chunk share one context behind a mutex."""
    import squishrs_amd
    work = [(corpus.random_bytes(51, i, 3000 + 977 * i), corpus.text_like(60 + i, 20000 + 1000 * i)) for i in range(12)]
    want = [(oracle.hash_chunk(a), oracle.compress_chunk(b)) for a, b in work]
    errs = []

    def run(tid):
        try:
            for rep in range(3):
                for k in range(tid, len(work), 4):
                    a, b = work[k]
                    if squishrs_amd.hash_chunk(a) != want[k][0] or squishrs_amd.compress_chunk(b) != want[k][1]:
                        errs.append((tid, k))
        except Exception as e:  # noqa: BLE001
            errs.append((tid, repr(e)))

    th = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[:5]


def test_pack_batch_failure_commits_nothing_then_retry(oracle):
    """squish_gpu.h: when sq_pack_batch fails (here SQ_E_CAP: frames do not fit comp_cap) the dedup set, the unique
    count and the chunk index are as before the call; the same batch retried with enough room reports the flags a
    fresh context reports, and an earlier batch's entries survive the roll-back."""
    import torch
    import squishrs_amd
    from squishrs_amd import _lib
    first = [corpus.random_bytes(61, i, 50000) for i in range(5)]
    batch = [corpus.random_bytes(61, 2, 50000), corpus.text_like(70, 80000), corpus.random_bytes(62, 0, 262144),
             corpus.text_like(70, 80000), corpus.random_bytes(62, 1, 100000), corpus.random_bytes(61, 4, 50000)]
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    try:
        r0 = st.insert_batch(first)
        assert all(r.compressed_data is not None for r in r0) and st.len() == 5
        lens = np.array([len(c) for c in batch], dtype=np.uint32)
        offs = np.zeros(len(batch), dtype=np.uint64)
        offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
        data = np.frombuffer(b"".join(batch), dtype=np.uint8)
        hashes = np.zeros(len(batch) * 16, dtype=np.uint8)
        is_new = np.zeros(len(batch), dtype=np.uint8)
        small = np.zeros(100000, dtype=np.uint8)           # far too small for the three new frames
        coff = np.zeros(len(batch) + 1, dtype=np.uint64)
        rc = st._L.sq_pack_batch(st._h, data.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(batch),
                                 hashes.ctypes.data, is_new.ctypes.data, small.ctypes.data, small.size, coff.ctypes.data)
        assert rc == _lib.SQ_E_CAP
        assert st.len() == 5                                # nothing committed
        res = st.insert_batch(batch)                        # the retry
        flags = [r.compressed_data is not None for r in res]
        assert flags == [False, True, True, False, True, False]
        assert st.len() == 8
        for c, r in zip(batch, res):
            assert r.hash == oracle.hash_chunk(c)
            if r.compressed_data is not None:
                assert r.compressed_data == oracle.compress_chunk(c)
        # the same on device pointers
        st.reset()
        d = torch.from_numpy(np.concatenate([data, np.zeros(64, dtype=np.uint8)])).cuda()
        dh = torch.zeros(len(batch) * 16, dtype=torch.uint8, device="cuda")
        dn = torch.zeros(len(batch), dtype=torch.uint8, device="cuda")
        dc = torch.zeros(600000, dtype=torch.uint8, device="cuda")
        with pytest.raises(squishrs_amd.SquishError) as ei:
            st.pack_device(d, offs, lens, dh, dn, dc[:1000])
        assert ei.value.code == _lib.SQ_E_CAP and st.len() == 0
        co = st.pack_device(d, offs, lens, dh, dn, dc)
        assert dn.cpu().numpy().tolist() == [1, 1, 1, 0, 1, 1] and st.len() == 5
        assert int(co[-1]) == sum(len(oracle.compress_chunk(batch[i])) for i in (0, 1, 2, 4, 5))
    finally:
        st.close()


def test_pack_batch_rejects_mixed_pointer_sides():
    import torch
    import squishrs_amd
    from squishrs_amd import _lib
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 10)
    try:
        c = corpus.text_like(1, 5000)
        data = np.frombuffer(c, dtype=np.uint8)
        d = torch.from_numpy(data.copy()).cuda()
        offs = np.zeros(1, dtype=np.uint64)
        lens = np.array([len(c)], dtype=np.uint32)
        hashes_h = np.zeros(16, dtype=np.uint8)
        new_d = torch.zeros(1, dtype=torch.uint8, device="cuda")
        comp_d = torch.zeros(8192, dtype=torch.uint8, device="cuda")
        coff = np.zeros(2, dtype=np.uint64)
        rc = st._L.sq_pack_batch(st._h, d.data_ptr(), offs.ctypes.data, lens.ctypes.data, 1, hashes_h.ctypes.data,
                                 new_d.data_ptr(), comp_d.data_ptr(), comp_d.numel(), coff.ctypes.data)
        assert rc == _lib.SQ_E_INVAL and st.len() == 0
        rc = st._L.sq_pack_batch(st._h, data.ctypes.data, offs.ctypes.data, lens.ctypes.data, 1, hashes_h.ctypes.data,
                                 new_d.data_ptr(), comp_d.data_ptr(), comp_d.numel(), coff.ctypes.data)
        assert rc == _lib.SQ_E_INVAL and st.len() == 0
        # a chunk longer than the context's max_chunk is refused before anything is committed
        st2 = squishrs_amd.ChunkStore(device=0, max_chunk=4096, dedup_capacity=1 << 10)
        with pytest.raises(squishrs_amd.SquishError):
            st2.insert_batch([c])
        assert st2.len() == 0
        st2.close()
    finally:
        st.close()


def test_pack_submit_wait_two_batches_in_flight(oracle):
    """sq_pack_submit / sq_pack_wait: two batches in flight on one context's two lanes.  Four batches of mixed
    content with duplicates inside and ACROSS batches (the dedup set is shared and stays in batch order), submitted
    back to back with at most two in flight: flags, unique count and every frame equal the synchronous path's and the
    oracle's; a third submit without a wait is refused; the synchronous calls are refused while batches are in flight."""
    import torch
    import squishrs_amd
    from squishrs_amd import _lib
    L = 262144
    base = [corpus.random_bytes(81, i, L) for i in range(40)] + [corpus.planted(7, L, 0), corpus.text_like(9, L), corpus.planted(8, L, 9)]
    plans = [list(range(0, 20)) + [3, 40, 41], [5, 20, 21, 22, 41, 23, 42], list(range(24, 40)) + [0, 42, 24], [41, 1, 2]]
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=1 << 12)
    try:
        bufs = []
        for plan in plans:
            n = len(plan)
            data = torch.frombuffer(bytearray(b"".join(base[i] for i in plan)), dtype=torch.uint8).cuda()
            offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
            lens = np.full(n, L, dtype=np.uint32)
            bufs.append((data, offs, lens, torch.zeros(n * 16, dtype=torch.uint8, device="cuda"),
                         torch.zeros(n, dtype=torch.uint8, device="cuda"),
                         torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")))
        results = []
        torch.cuda.synchronize()   # (torch's fills of the output tensors run on torch's stream, the lanes on their own)
        st.pack_submit(*bufs[0], reset=True)
        st.pack_submit(*bufs[1])
        with pytest.raises(squishrs_amd.SquishError):       # both lanes busy
            st.pack_submit(*bufs[2])
        with pytest.raises(squishrs_amd.SquishError):       # synchronous calls wait their turn
            st.reset()
        results.append(st.pack_wait(len(plans[0])))
        st.pack_submit(*bufs[2])
        results.append(st.pack_wait(len(plans[1])))
        st.pack_submit(*bufs[3])
        results.append(st.pack_wait(len(plans[2])))
        results.append(st.pack_wait(len(plans[3])))
        seen = set()
        for plan, (data, offs, lens, hashes, is_new, comp), (coff, clen) in zip(plans, bufs, results):
            nn = is_new.cpu().numpy()
            hh = hashes.cpu().numpy().reshape(-1, 16)
            for j, i in enumerate(plan):
                assert hh[j].tobytes() == oracle.hash_chunk(base[i])
                new = i not in seen
                seen.add(i)
                assert bool(nn[j]) == new, (plan, j)
                if new:
                    f = comp[int(coff[j]):int(coff[j]) + int(clen[j])].cpu().numpy().tobytes()
                    assert f == oracle.compress_chunk(base[i]), (plan, j)
                else:
                    assert clen[j] == 0
        assert st.len() == len(seen)
        # and the synchronous path still works on the same context afterwards
        st.reset()
        r = st.insert_batch([base[41], base[0], base[41]])
        assert [x.compressed_data is not None for x in r] == [True, True, False]
        assert r[0].compressed_data == oracle.compress_chunk(base[41])
    finally:
        st.close()


def test_pack_submit_wait_host_buffers(oracle):
    """The pipelined form with HOST buffers (what the C++ driver's GPU stage does): three ragged batches, two in
    flight, results copied back by sq_pack_wait: hashes, flags, frames (packed back to back) against the oracle."""
    import squishrs_amd
    kinds = ["text", "random", "mixed", "records", "planted3", "runs", "zeros", "skewed"]
    batches = []
    for b in range(3):
        chunks = [corpus.make(kinds[(3 * b + i) % len(kinds)], 300 + 10 * b + i, [70000, 262144, 5000, 400000, 131072][(b + i) % 5])
                  for i in range(6)]
        chunks.append(chunks[1])                      # a duplicate inside the batch
        if b:
            chunks.append(batches[0][0][2])           # ... and one of an earlier batch
        lens = np.array([len(c) for c in chunks], dtype=np.uint32)
        offs = np.zeros(len(chunks), dtype=np.uint64)
        pos = 0
        for i, c in enumerate(chunks):
            offs[i] = pos
            pos += (len(c) + 255) & ~255
        data = np.zeros(pos + 256, dtype=np.uint8)
        for i, c in enumerate(chunks):
            data[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
        cap = int(sum(squishrs_amd.compress_bound(len(c)) + 64 for c in chunks))
        batches.append((chunks, data, offs, lens, np.zeros(len(chunks) * 16, dtype=np.uint8), np.zeros(len(chunks), dtype=np.uint8),
                        np.zeros(cap, dtype=np.uint8)))
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    L = st._L
    try:
        def submit(b, reset):
            chunks, data, offs, lens, hashes, is_new, comp = b
            squishrs_amd.chunk.check(L.sq_pack_submit(st._h, data.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(chunks),
                                                      hashes.ctypes.data, is_new.ctypes.data, comp.ctypes.data, comp.size, 1 if reset else 0))
        submit(batches[0], True)
        submit(batches[1], False)
        r0 = st.pack_wait(len(batches[0][0]))
        submit(batches[2], False)
        r1 = st.pack_wait(len(batches[1][0]))
        r2 = st.pack_wait(len(batches[2][0]))
        seen = set()
        for (chunks, data, offs, lens, hashes, is_new, comp), (coff, clen) in zip(batches, (r0, r1, r2)):
            for j, c in enumerate(chunks):
                assert hashes[16 * j:16 * j + 16].tobytes() == oracle.hash_chunk(c)
                new = c not in seen
                seen.add(c)
                assert bool(is_new[j]) == new
                if new:
                    assert comp[int(coff[j]):int(coff[j]) + int(clen[j])].tobytes() == oracle.compress_chunk(c), (j, len(c))
                else:
                    assert clen[j] == 0
        assert st.len() == len(seen)
    finally:
        st.close()


def test_rowbatch_every_kind_and_size_vs_oracle(gpu_store, oracle):
    """The lazy2 / row-hash strategy (chunks > 256 KiB) over speculative search batches (sq_zstd_rowbatch.h): every
    corpus kind at ragged sizes of that strategy -- long runs (every lane of a batch in one row: in-batch candidates
    and delegation to the exact finder), records (matches at fixed distances: repcode cache and refresh), skewed and
    random stretches (no-match fast-forward, lazy skipping), text (dozens of true candidates per search: windowed
    evaluation), planted repeats -- every frame against the oracle."""
    from test_compress_gpu import _batch_compress
    rng = np.random.default_rng(77)
    kinds = ["text", "records", "runs", "skewed", "mixed", "random", "zeros", "planted2", "planted4"]
    chunks = []
    for i in range(54):
        n = int(rng.choice([262145, 262200, 300000, 524288, 524289, 1048576, 1500001, 2097151, 2097152,
                            int(rng.integers(262145, 2097152))]))
        chunks.append(corpus.make(kinds[i % len(kinds)], 5000 + i, n))
    # two chunks made of alternating short stretches: the parser changes mode every few KiB
    parts = [corpus.make(kinds[j % 6], 6000 + j, 3000 + 700 * (j % 5)) for j in range(300)]
    chunks.append(b"".join(parts)[:2097152])
    chunks.append((b"".join(parts[::-1]) * 2)[:1234567])
    frames = _batch_compress(gpu_store, chunks)
    bad = [(i, len(c)) for i, (c, f) in enumerate(zip(chunks, frames)) if f != oracle.compress_chunk(c)]
    assert not bad, bad[:10]


def _stage1_decides(c: bytes) -> bool:
    """Model of k_incompressible_batch's decision (sq_zstd_fast.h header): the no-match certificate over the whole
    predicted schedule + HUF's sampled pre-check.  (Planted chunks whose plant makes a search succeed fail it here as
    they do on the device; this is the statement of the certificate, not of what libzstd finds.)"""
    n = len(c)
    if not (16384 < n <= 262144):
        return False
    len0 = min(n, 131072)
    len1 = n - len0
    if len0 < 40960 or (len1 and len1 < 40960):
        return False
    a = np.frombuffer(c, dtype=np.uint8)
    v = (a[:-3].astype(np.uint32) | (a[1:-2].astype(np.uint32) << 8) | (a[2:-1].astype(np.uint32) << 16) | (a[3:].astype(np.uint32) << 24))
    _, first, inv = np.unique(v, return_index=True, return_inverse=True)
    first_of = first[inv]                        # first position that holds position p's 4 bytes
    for s_ in corpus.lazy_schedule(n):
        if first_of[s_] < s_:                    # an earlier position with the same 4 bytes
            return False
        if v[s_] == v[s_ + 1]:                   # the repcode test (offset 1) of the search at s fires
            return False
    for b0, bl in ((0, len0), (len0, len1)):
        if bl == 0:
            continue
        blk = a[b0:b0 + bl]
        l1 = np.bincount(blk[:4096], minlength=256).max()
        l2 = np.bincount(blk[-4096:], minlength=256).max()
        if l1 + l2 > ((2 * 4096) >> 7) + 4:
            return False
    return True


def test_stage1_workgroup_edges_sizes_and_alignments(gpu_store, oracle):
    """Stage 1 (one chunk per workgroup of four waves, sq_zstd_fast.h): every way the work is dealt out.  Sizes at the
    strategy's and the path's boundaries (16 KiB + 1, 40,960 = the smallest block HUF's sampled pre-check accepts, one
    block / two blocks, a second block too short to be searched, fewer rounds than waves), at every start alignment
    modulo 16 (an aligned chunk takes the bitmap variant of the scan and lets the scan write the raw blocks, any
    other start takes the level-word variant and the tail copy), with contents that end certified (random), are handed
    over (planted repeats in either block) or are not this path's at all (text, zeros) -- every frame against the
    oracle."""
    import torch
    import squishrs_amd
    rng = np.random.default_rng(2024)
    sizes = [16385, 16400, 17408, 20000, 40959, 40960, 40961, 65535, 65536, 65537, 100000, 131071, 131072, 131073,
             131080, 131081, 135000, 172032, 172033, 200000, 262143, 262144]
    kinds = ["random", "random", "planted0", "planted3", "planted6", "planted9", "random", "text", "zeros", "skewed"]
    chunks, starts = [], []
    pos = 0
    for i in range(160):
        n = sizes[i % len(sizes)] if i < 132 else int(rng.integers(16385, 262145))
        k = kinds[i % len(kinds)]
        if k.startswith("planted") and n < 20000:
            k = "random"
        chunks.append(corpus.make(k, 9000 + i, n))
        pos = (pos + 63) // 64 * 64 + (i % 16)          # start alignment i mod 16
        starts.append(pos)
        pos += n
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for s, c in zip(starts, chunks):
        buf[s:s + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    offs = np.array(starts, dtype=np.uint64)
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    comp = torch.zeros(int(sum(squishrs_amd.compress_bound(len(c)) + 64 for c in chunks)), dtype=torch.uint8, device="cuda")
    for _ in range(2):                                  # the second pass runs on persisted table generations
        coff = gpu_store.compress_device(d, offs, lens, None, comp)
        host = comp.cpu().numpy()
        bad = [(i, len(c), starts[i] % 16) for i, c in enumerate(chunks)
               if host[int(coff[i]):int(coff[i + 1])].tobytes() != oracle.compress_chunk(c)]
        assert not bad, bad[:10]
    # How many of them stage 1 decides is not a guess: a chunk is decided there iff it is of this strategy with blocks of
    # >= 40,960 bytes, no scheduled search has an earlier position with its 4 bytes (or the offset-1 repcode test at
    # it fires) and HUF's sampled pre-check calls each block uncompressible -- computed here from the bytes.  The
    # planted chunks are built to be handed over (the ones that plant BESIDE a searched position may still certify).
    model = [_stage1_decides(c) for c in chunks]
    is_random = [(kinds[i % len(kinds)] == "random") or (kinds[i % len(kinds)].startswith("planted") and lens[i] < 20000) for i in range(160)]
    lo = sum(1 for m, r in zip(model, is_random) if m and r)
    hi = sum(1 for m in model if m)
    decided = gpu_store.timings().fast_decided
    assert lo <= decided <= hi, f"stage 1 decided {decided} chunks; the model says {lo} random ones (at most {hi} with planted ones that certify)"
    assert lo >= 25


def test_shard_split_form_two_batches_in_flight(oracle):
    """sq_shard_submit_hash / sq_shard_wait_hash / sq_shard_submit_rest (the sharded pipelined form; here one rank, so
    the "gathered" hashes are the batch's own): two batches in flight with cross-batch duplicates give the flags, the
    unique count and the frames of sq_pack_submit; calls out of order are refused."""
    import torch
    import squishrs_amd
    from squishrs_amd import SquishError
    rng = np.random.default_rng(31)
    L, n = 65536, 96
    base = [corpus.make("random" if i % 3 else "text", 700 + i, L) for i in range(n)]
    b0 = base[:64]
    b1 = base[32:]                                   # 32 duplicates of batch 0, then 32 new
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=4096)

    def dev(chunks):
        d = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
        k = len(chunks)
        offs = np.arange(k, dtype=np.uint64) * np.uint64(L)
        lens = np.full(k, L, dtype=np.uint32)
        return d, offs, lens, torch.zeros(k * 16, dtype=torch.uint8, device="cuda"), \
            torch.zeros(k, dtype=torch.uint8, device="cuda"), torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")

    A, B = dev(b0), dev(b1)
    with pytest.raises(SquishError):
        st.shard_wait_hash()                         # nothing submitted
    st.shard_submit_hash(A[0], A[1], A[2], A[3])
    with pytest.raises(SquishError):
        st.shard_submit_hash(B[0], B[1], B[2], B[3])  # the first batch's rest is still owed
    st.shard_wait_hash()
    st.shard_submit_rest(A[3], 64, 0, A[4], A[5], reset=True)
    st.shard_submit_hash(B[0], B[1], B[2], B[3])
    st.shard_wait_hash()
    st.shard_submit_rest(B[3], 64, 0, B[4], B[5])
    offA, lenA = st.pack_wait(64)
    offB, lenB = st.pack_wait(64)
    assert A[4].cpu().tolist() == [1] * 64
    assert B[4].cpu().tolist() == [0] * 32 + [1] * 32
    assert st.len() == 96
    for (d, offs, lens, h, isn, comp), off, ln, chunks in ((A, offA, lenA, b0), (B, offB, lenB, b1)):
        hh = h.cpu().numpy().reshape(-1, 16)
        for i in (0, 31, 32, 63):
            assert hh[i].tobytes() == oracle.hash_chunk(chunks[i])
            fr = comp[int(off[i]):int(off[i]) + int(ln[i])].cpu().numpy().tobytes()
            assert fr == (oracle.compress_chunk(chunks[i]) if isn[i] else b"")
    st.close()


def test_failing_shard_rest_consumes_the_hash_step(oracle):
    """include/squish_gpu.h: a sq_shard_submit_rest[_idx] that finds a pending hash step consumes it whatever it returns
    (SQ_E_CAP for a small comp buffer, SQ_E_INVAL for a missing index array) and commits nothing: a second rest call is
    out of order, the batch starts again with sq_shard_submit_hash and then gives the flags of a fresh set."""
    import torch
    import squishrs_amd
    from squishrs_amd import SquishError
    L, n = 65536, 16
    base = [corpus.random_bytes(91, i, L) for i in range(n)]
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=1024)
    try:
        d = torch.frombuffer(bytearray(b"".join(base)), dtype=torch.uint8).cuda()
        offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
        lens = np.full(n, L, dtype=np.uint32)
        h = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
        f = torch.zeros(n, dtype=torch.uint8, device="cuda")
        comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
        idx = torch.arange(n, dtype=torch.int32, device="cuda")
        # (a) comp_cap too small: SQ_E_CAP, the hash step is gone
        st.shard_submit_hash(d, offs, lens, h)
        st.shard_wait_hash()
        with pytest.raises(SquishError):
            st.shard_submit_rest(h, n, 0, f, comp[:1024])
        with pytest.raises(SquishError):
            st.shard_submit_rest(h, n, 0, f, comp)           # nothing pending any more
        assert st.len() == 0                                  # nothing was committed
        # (b) the idx form without an index array: SQ_E_INVAL, consumed as well
        st.shard_submit_hash(d, offs, lens, h)
        st.shard_wait_hash()

        class _Null:
            @staticmethod
            def data_ptr():
                return 0
        with pytest.raises(SquishError):
            st.shard_submit_rest_idx(h, n, _Null, f, comp)
        with pytest.raises(SquishError):
            st.shard_wait_hash()
        assert st.len() == 0
        # (c) the batch again, from the hash: all first occurrences
        st.shard_submit_hash(d, offs, lens, h)
        st.shard_wait_hash()
        st.shard_submit_rest_idx(h, n, idx, f, comp)
        co, cl = st.pack_wait(n)
        assert f.cpu().tolist() == [1] * n and st.len() == n
        assert comp[int(co[5]):int(co[5]) + int(cl[5])].cpu().numpy().tobytes() == oracle.compress_chunk(base[5])
    finally:
        st.close()


def test_empty_shard_batch_still_orders_the_shared_dedup_set(oracle):
    """A rank may hold 0 chunks of a super-batch while the other ranks' hashes (n_total > 0) must be committed to its
    replicated set: sq_shard_submit_rest queues that copy + insert on the lane, and sq_pack_wait of such an EMPTY batch
    must not return before they ran -- the next batch (other lane) orders its own dedup only behind a lane that is
    still busy.  Then: a reset batch behind it.  Flags against the first-occurrence model throughout."""
    import torch
    import squishrs_amd
    L = 65536
    base = [corpus.random_bytes(81, i, L) for i in range(48)]
    st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=4096)
    try:
        d = torch.frombuffer(bytearray(b"".join(base)), dtype=torch.uint8).cuda()
        offs = np.arange(48, dtype=np.uint64) * np.uint64(L)
        lens = np.full(48, L, dtype=np.uint32)
        others = torch.zeros(48 * 16, dtype=torch.uint8, device="cuda")
        st.hash_device(d, offs, lens, others)                       # "the other ranks'" hashes
        e_u8 = torch.zeros(16, dtype=torch.uint8, device="cuda")    # (non-null pointers for the empty batch)
        none_o, none_l = np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32)
        for rep in range(3):
            # empty batch of this rank, 48 hashes of the others; reset first
            st.shard_submit_hash(e_u8, none_o, none_l, e_u8)
            st.shard_wait_hash()
            st.shard_submit_rest(others, 48, 0, e_u8, e_u8, reset=True)
            co, cl = st.pack_wait(0)
            assert len(co) == 0
            # the same chunks as an ordinary batch on the OTHER lane: every one a duplicate of what was just committed
            h = torch.zeros(48 * 16, dtype=torch.uint8, device="cuda")
            f = torch.ones(48, dtype=torch.uint8, device="cuda")
            c = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
            st.pack_submit(d, offs, lens, h, f, c)
            co, cl = st.pack_wait(48)
            assert f.cpu().tolist() == [0] * 48 and (cl == 0).all(), rep
            assert torch.equal(h, others) and st.len() == 48
            # a reset batch behind it: first occurrences only
            st.pack_submit(d[:8 * L], offs[:8], lens[:8], h[:128], f[:8], c, reset=True)
            co, cl = st.pack_wait(8)
            assert f[:8].cpu().tolist() == [1] * 8 and st.len() == 8
            assert c[int(co[3]):int(co[3]) + int(cl[3])].cpu().numpy().tobytes() == oracle.compress_chunk(base[3])
    finally:
        st.close()
