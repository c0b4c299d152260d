"""GPU parity of stage 1's no-match certificate on chunks above 256 KiB -- the lazy2 / row-hash size class, i.e. the
reference's own 2 MiB chunks (/root/reference/src/util/chunk.rs:11, src/archive/writer.rs:238-246) and the tails of
large files: one job per pair of 128 KiB blocks, combined per chunk (sq_zstd_fast.h fastpath_blocks).  Every frame
against the oracle (whose frames for the planted variants are pinned by real libzstd 1.5.7 goldens,
tests/golden/zstd12_frames.json "plantedL*"); at scale, every raw-block frame's payload on the device."""
import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu


def _compress(store, chunks, starts_mod16=None):
    import torch
    import squishrs_amd
    offs, pos = [], 0
    for i, c in enumerate(chunks):
        pos = (pos + 255) & ~255
        pos += 0 if starts_mod16 is None else starts_mod16[i]
        offs.append(pos)
        pos += len(c)
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for c, o in zip(chunks, offs):
        buf[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    offs = np.array(offs, dtype=np.uint64)
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    comp = torch.zeros(int(sum(squishrs_amd.compress_bound(len(c)) + 64 for c in chunks)), dtype=torch.uint8, device="cuda")
    coff = store.compress_device(d, offs, lens, None, comp)
    host = comp.cpu().numpy()
    return [host[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(chunks))]


def test_large_chunks_every_variant_and_size(oracle):
    import squishrs_amd
    rng = np.random.default_rng(8)
    chunks, align = [], []
    sizes = [2097152, 2097151, 2097152 - 4096, 1048576, 1048577, 786432, 262145, 262144 + 40960, 262144 + 40959, 393216, 393216 + 50000,
             524288, 524289, 600000, 1300000, 1966080 + 40960, 1966080 + 100]
    for i, n in enumerate(sizes):                                     # random: certified unless a tail is too short
        chunks.append(corpus.random_bytes(91, i, n))
        align.append(0)
    for v in range(10):                                               # the planted variants (goldens of real libzstd)
        for seed, n in ((1, 2097152), (2, 1148576), (3, 600000)):
            chunks.append(corpus.make("plantedL%d" % v, seed, n))
            align.append(0)
    for i, a in enumerate([1, 3, 8, 15]):                             # unaligned starts: left to stage 2, same frames
        chunks.append(corpus.random_bytes(92, i, 2097152 if i % 2 else 700000))
        align.append(a)
    for kind, n in (("mixed", 2097152), ("text", 400000), ("zeros", 2097152), ("runs", 1000000), ("records", 300000), ("skewed", 2097152)):
        chunks.append(corpus.make(kind, 3, n))                        # not this path's: bucket check / certificate fail early
        align.append(0)
    for i in range(6):                                                # and the other two classes in the same batch
        chunks.append(corpus.random_bytes(93, i, [262144, 100000, 16384, 5000, 131073, 40960][i]))
        align.append(0)
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    try:
        for _ in range(2):
            frames = _compress(st, chunks, align)
            bad = [(i, len(c), align[i]) for i, (c, f) in enumerate(zip(chunks, frames)) if f != oracle.compress_chunk(c)]
            assert not bad, bad[:10]
        decided = st.timings().fast_decided
        # certified here: the random chunks whose blocks are all >= 40,960 bytes (a five-byte coincidence at a searched
        # position is a 3 % event per 2 MiB), planted variants 0 (four equal bytes only) and 4 (beyond the finder's reach)
        assert decided >= 10 + 6 + 3, decided
    finally:
        st.close()


def test_large_random_chunks_at_scale_through_pack_submit(oracle):
    """640 x 2 MiB + 384 x 1 MiB random chunks through sq_pack_submit (two batches in flight): every frame is the raw-block
    frame of its chunk (payload compared on the device), all but a few decided by stage 1, hashes equal the hash
    kernel's, samples against the oracle."""
    import torch
    import corpus_gpu
    import squishrs_amd
    for L, n in ((2 << 20, 640), (1 << 20, 384)):
        data = corpus_gpu.random_chunks(94, 0, n, L)
        offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
        lens = np.full(n, L, dtype=np.uint32)
        st = squishrs_amd.ChunkStore(device=0, max_chunk=2 << 20, dedup_capacity=1 << 13)
        try:
            want = torch.zeros(n * 16, dtype=torch.uint8, device="cuda")
            st.hash_device(data, offs, lens, want)
            bufs = []
            for _ in range(2):
                bufs.append((torch.zeros(n * 16, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda"),
                             torch.zeros(st.slots_bound(lens), dtype=torch.uint8, device="cuda")))
            st.pack_submit(data, offs, lens, *bufs[0], reset=True)
            st.pack_submit(data, offs, lens, *bufs[1], reset=True)
            outs = [st.pack_wait(n), st.pack_wait(n)]
            decided = st.timings().fast_decided
            assert decided >= 0.93 * n, decided
            nb = L // 131072
            frame = 9 + nb * 3 + L
            for (h, f, c), (co, cl) in zip(bufs, outs):
                assert torch.equal(h, want) and int(f.sum().item()) == n
                assert (cl == frame).all()
                stride = int(co[1] - co[0])
                fr = c[:n * stride].view(n, stride)
                dv = data.view(n, L)
                for k0 in range(0, n, 64):
                    for b in range(nb):
                        o = 9 + b * 131075
                        assert torch.equal(fr[k0:k0 + 64, o + 3:o + 3 + 131072], dv[k0:k0 + 64, b * 131072:(b + 1) * 131072])
                        hdr = fr[k0:k0 + 64, o:o + 3].cpu().numpy()
                        want_h = ((1 if b + 1 == nb else 0) + (131072 << 3)).to_bytes(3, "little")
                        assert (hdr == np.frombuffer(want_h, dtype=np.uint8)).all()
            host = data.cpu().numpy().reshape(n, L)
            for i in (0, n // 2, n - 1):
                fr_i = bufs[1][2][int(outs[1][0][i]):int(outs[1][0][i]) + frame].cpu().numpy().tobytes()
                assert fr_i == oracle.compress_chunk(host[i].tobytes())
        finally:
            st.close()
