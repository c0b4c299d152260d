"""GPU parity of the FUSED form of a pack batch (DESIGN.md 4.2 "fused hash"): compress stage 1 runs before the dedup
decision on every chunk and computes XXH3-128 beside its scan; the hash kernel only fills in the chunks stage 1 did
not stream.  Results must not depend on the form: hashes (/root/reference/src/util/chunk.rs:46-49), is_new flags
(chunk.rs:80-100, first occurrence in index order), frames (chunk.rs:89-90) are compared between the two forms, with
the oracle and with the first-occurrence model.  All through the C ABI."""
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


def _layout(chunks, starts_mod16):
    """Chunks laid into one buffer; chunk i starts at an offset that is starts_mod16[i] modulo 16."""
    offs, pos = [], 0
    for c, m in zip(chunks, starts_mod16):
        pos = (pos + 255) & ~255
        pos += m
        offs.append(pos)
        pos += len(c)
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for c, o in zip(chunks, offs):
        buf[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    return buf, np.array(offs, dtype=np.uint64), np.array([len(c) for c in chunks], dtype=np.uint32)


def _pack(store, data, offs, lens, pipelined):
    import torch
    n = len(lens)
    hashes = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
    is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
    comp = torch.zeros(store.slots_bound(lens) + 64, dtype=torch.uint8, device="cuda")
    if pipelined:
        store.pack_submit(data, offs, lens, hashes, is_new, comp, reset=True)
        co, cl = store.pack_wait(n)
        frames = [comp[int(co[i]):int(co[i]) + int(cl[i])].cpu().numpy().tobytes() for i in range(n)]
    else:
        store.reset()
        co = store.pack_device(data, offs, lens, hashes, is_new, comp)
        frames = [comp[int(co[i]):int(co[i + 1])].cpu().numpy().tobytes() for i in range(n)]
    return hashes.cpu().numpy().reshape(n, 16), is_new.cpu().numpy(), frames, store.len()


def test_fused_form_ragged_batch_matches_split_form_and_oracle(oracle):
    """Every kind of chunk the fused kernel can meet in one batch: random chunks of stage 1's size class at aligned
    and unaligned starts (only aligned ones are hashed by stage 1), every boundary length of XXH3's long path around
    a 1 KiB block, compressible chunks (stage 1 gives up before its scan: hash kernel), chunks of the other two
    strategies, tiny chunks, planted repeats (hand-over to stage 2), exact duplicates of all of these."""
    import torch
    import squishrs_amd
    rng = np.random.default_rng(77)
    chunks, align = [], []
    for i, n in enumerate([262144, 262143, 262145 - 2, 131072, 131073, 200000, 65536, 16385, 16400, 17408, 17409, 17407, 18432,
                           40960, 99999, 100352, 100353, 100351, 100415, 100416, 100417, 262144 - 63, 262144 - 64, 262144 - 65]):
        chunks.append(corpus.random_bytes(50, i, n))
        align.append(0)
    for i, n in enumerate([262144, 70000, 131072, 20000]):           # unaligned starts: no fused hash, same results
        chunks.append(corpus.random_bytes(51, i, n))
        align.append([1, 4, 8, 15][i])
    for kind, n in (("text", 262144), ("text", 70000), ("records", 131073), ("zeros", 65536), ("runs", 200000), ("skewed", 262144),
                    ("mixed", 262144)):
        chunks.append(corpus.make(kind, 5, n))
        align.append(0)
    for n in (1, 12, 240, 241, 3000, 16384):                         # btopt class and XXH3's short paths
        chunks.append(corpus.make("text", 9, n))
        align.append(0)
    chunks.append(corpus.make("mixed", 3, 300000))                   # lazy2 class
    align.append(0)
    chunks.append(corpus.make("random", 4, 2 << 20))
    align.append(0)
    for k in range(6):                                               # planted repeats: stage 1 hands over to stage 2
        chunks.append(corpus.make("planted%d" % (k % 5), 60 + k, 262144))
        align.append(0)
    base_n = len(chunks)
    for j in (0, 3, 5, 24, 28, 30, 36, base_n - 1, 0):                # exact duplicates (the last one twice)
        chunks.append(chunks[j])
        align.append(0 if j != 24 else 3)
    exp_new, exp_unique = _first_occurrence(chunks)
    buf, offs, lens = _layout(chunks, align)
    data = torch.from_numpy(buf).cuda()
    res = {}
    for fuse in (True, False):
        st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12, fuse_hash=fuse)
        try:
            for pipelined in (True, False):
                res[(fuse, pipelined)] = _pack(st, data, offs, lens, pipelined)
        finally:
            st.close()
    ref = res[(False, False)]
    for key, r in res.items():
        assert (r[0] == ref[0]).all(), ("hashes differ between the forms", key, np.nonzero((r[0] != ref[0]).any(axis=1))[0][:10])
        assert (r[1] == exp_new).all(), ("is_new", key, np.nonzero(r[1] != exp_new)[0][:10])
        assert r[3] == exp_unique, key
        for i, (a, b) in enumerate(zip(r[2], ref[2])):
            assert a == b, ("frame differs between the forms", key, i, len(a), len(b))
    for i, c in enumerate(chunks):
        assert ref[0][i].tobytes() == oracle.hash_chunk(c), i
        if exp_new[i]:
            assert ref[2][i] == oracle.compress_chunk(c), i
        else:
            assert ref[2][i] == b""


def test_fused_form_many_chunks_hashes_and_flags(oracle):
    """3,000 x 256 KiB + 2,000 x 64 KiB random chunks with 20 % duplicates, more chunks than resident workgroups
    (the per-workgroup pool of block sums is reused from chunk to chunk): every hash against the hash kernel's, every
    flag against the model, every raw-block frame's payload on the device, two batches in flight."""
    import torch
    import corpus_gpu
    import squishrs_amd
    for L, n in ((262144, 3000), (65536, 2000)):
        rng = np.random.default_rng(L)
        src = np.arange(n, dtype=np.int64)
        dup = rng.random(n) < 0.2
        dup[0] = False
        src[dup] = rng.integers(0, n, size=int(dup.sum()))
        src = np.where(dup, np.minimum(src, np.arange(n) - 1), src)   # a duplicate copies an EARLIER index's content
        for i in range(n):                                            # resolve chains: content id = root
            src[i] = src[src[i]] if dup[i] else i
        base = corpus_gpu.random_chunks(61, 0, n, L)
        data = torch.empty(n * L, dtype=torch.uint8, device="cuda")
        corpus_gpu.gather_chunks(data, base, torch.from_numpy(src), L)
        exp_new, exp_unique = _first_occurrence([int(x) for x in src])
        offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
        lens = np.full(n, L, dtype=np.uint32)
        st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=1 << 14, fuse_hash=True)
        try:
            want = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
            st.hash_device(data, offs, lens, want)                     # the hash kernel (pinned by test_hash_gpu.py)
            bufs = []
            for k in range(3):                                         # two in flight, three batches: both lanes, pool reuse
                h = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
                f = torch.zeros(n, dtype=torch.uint8, device="cuda")
                c = torch.zeros(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
                bufs.append((h, f, c))
            st.pack_submit(data, offs, lens, *bufs[0], reset=True)
            st.pack_submit(data, offs, lens, *bufs[1], reset=True)
            out0 = st.pack_wait(n)
            st.pack_submit(data, offs, lens, *bufs[2], reset=True)
            out1 = st.pack_wait(n)
            out2 = st.pack_wait(n)
            assert st.timings().fast_ms > 0
            frame = 9 + 3 * (2 if L > 131072 else 1) + L if L != 65536 else 65536 + 10
            for (h, f, c), (co, cl) in zip(bufs, (out0, out1, out2)):
                assert torch.equal(h, want)
                assert (f.cpu().numpy() == exp_new).all()
                assert ((cl > 0) == (exp_new == 1)).all()
                raw = np.nonzero(cl == frame)[0]
                assert len(raw) > 0.85 * exp_unique
                hdr = 9 if L > 65535 + 256 else 7
                dv = data.view(n, L)
                stride = int(co[1] - co[0])
                fr = c[:n * stride].view(n, stride)
                rows = torch.from_numpy(raw).cuda()
                for k0 in range(0, len(raw), 512):
                    r = rows[k0:k0 + 512]
                    blk0 = min(L, 131072)
                    assert torch.equal(fr[r][:, hdr + 3:hdr + 3 + blk0], dv[r][:, :blk0])
                    if L > 131072:
                        assert torch.equal(fr[r][:, hdr + 6 + blk0:hdr + 6 + L], dv[r][:, blk0:])
            assert st.len() == exp_unique
            hh = want.cpu().numpy().reshape(n, 16)
            hb = base.cpu().numpy().reshape(n, L)
            for i in (0, 1, n // 2, n - 1):
                assert hh[i].tobytes() == oracle.hash_chunk(hb[int(src[i])].tobytes())
        finally:
            st.close()


def test_form_is_chosen_from_the_previous_batch(oracle):
    """Default configuration: a batch after one with many duplicates takes the split form, a batch after one with few
    takes the fused form; the results are the same.  (What can be observed from outside: both produce the oracle's
    hashes / flags / frames whatever came before.)"""
    import torch
    import squishrs_amd
    uniq = [corpus.random_bytes(70, i, 65536) for i in range(40)]
    dups = [uniq[i % 4] for i in range(40)]
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    try:
        for batch in (uniq, dups, dups, uniq, uniq):
            buf, offs, lens = _layout(batch, [0] * len(batch))
            h, f, fr, u = _pack(st, torch.from_numpy(buf).cuda(), offs, lens, True)
            exp_new, exp_unique = _first_occurrence(batch)
            assert (f == exp_new).all() and u == exp_unique
            for i, c in enumerate(batch):
                assert h[i].tobytes() == oracle.hash_chunk(c)
                assert fr[i] == (oracle.compress_chunk(c) if exp_new[i] else b"")
    finally:
        st.close()


def _far_repeats(seed, n, first, count, spacing, block1_random=True):
    """Random bytes with `count` planted 4-byte repeats of far-away earlier positions, `spacing` bytes apart from position `first`
    on: a block that has hundreds of short matches and is STILL emitted raw (each saves less than it costs), i.e. a block
    whose compression attempt writes a long sequences section before it is thrown away."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=n, dtype=np.uint8)
    for k in range(count):
        p, q = first + spacing * k, 500 + 13 * k
        if p + 4 <= n and q + 4 < p:
            a[p:p + 4] = a[q:q + 4]
    return a.tobytes()


def test_rejected_attempt_of_one_block_leaves_the_next_blocks_raw_payload_alone(oracle):
    """ADVICE r3 (sq_zstd_kernel.h, the in-place raw-block shortcut): stage 1's scan writes the raw payload of EVERY block of a
    chunk into the slot; when stage 2 then attempts block 0 -- here with 600 / 2,000 / 6,000 short far repeats: raw literals plus
    a long sequences section, rejected as raw in the end -- that attempt must not run over block 1's payload, which is emitted
    without another copy.  Fused and split form, 2 and 1.5 blocks; every frame against the oracle."""
    import torch
    import squishrs_amd
    chunks = []
    for seed, (cnt, sp) in enumerate(((600, 100), (2000, 40), (6000, 16), (1500, 60))):
        chunks.append(_far_repeats(40 + seed, 262144, 30000, cnt, sp))
        chunks.append(_far_repeats(50 + seed, 131072 + 65536, 20000, cnt, sp))
    chunks.append(corpus.random_bytes(77, 1, 262144))
    buf, offs, lens = _layout(chunks, [0] * len(chunks))
    data = torch.from_numpy(buf).cuda()
    want = [oracle.compress_chunk(c) for c in chunks]
    for fuse in (True, False):
        st = squishrs_amd.ChunkStore(device=0, max_chunk=262144, dedup_capacity=1024, fuse_hash=fuse)
        try:
            _h, new, frames, _u = _pack(st, data, offs, lens, True)
            assert new.tolist() == [1] * len(chunks)
            for i, (f, w) in enumerate(zip(frames, want)):
                assert f == w, (fuse, i, len(f), len(w))
        finally:
            st.close()
