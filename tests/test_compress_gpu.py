"""GPU parity for the zstd level-12 frame compressor kernel, through the C ABI:
bit-exact vs the CPU oracle and vs the golden vectors produced by real libzstd 1.5.7; plus
size-independent properties (decode round trip, determinism) at full configuration sizes.
Where the GPU box image still carries libzstd 1.5.7 it is used as one more, direct check."""
import hashlib
import json
import os

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ZZ = json.load(open(os.path.join(G, "zstd12_frames.json")))


def _batch_compress(store, chunks):
    """sq_compress_batch over device-resident chunks; returns list of frames."""
    import torch
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    offs = np.zeros(len(chunks), dtype=np.uint64)
    pos = 0
    for i, c in enumerate(chunks):
        offs[i] = pos
        pos += (len(c) + 255) // 256 * 256
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for i, c in enumerate(chunks):
        buf[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    import squishrs_amd
    cap = int(sum(squishrs_amd.compress_bound(len(c)) + 64 for c in chunks))
    comp = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    coff = store.compress_device(d, offs, lens, None, comp)
    host = comp.cpu().numpy()
    return [host[int(coff[i]):int(coff[i + 1])].tobytes() for i in range(len(chunks))]


def test_known_answers_single_chunk_api():
    import squishrs_amd
    # SURVEY.md Appendix B (real libzstd 1.5.7 outputs)
    assert squishrs_amd.compress_chunk(b"hello squish").hex() == "28b52ffd200c61000068656c6c6f20737175697368"
    assert squishrs_amd.compress_chunk(b"").hex() == "28b52ffd2000010000"
    assert squishrs_amd.compress_chunk(b"\x01" * 1024).hex() == "28b52ffd60000345000008010100fc2b2004"
    assert squishrs_amd.compress_chunk(b"abc" * 1000).hex() == "28b52ffd60b80a550000186162630100b2d37743"
    assert squishrs_amd.compress_chunk(bytes(262144)).hex() == "28b52ffda0000004005400001000000100fbff39c00203001000"


def test_golden_literals(gpu_store):
    names = sorted(ZZ["literals"])
    frames = _batch_compress(gpu_store, [bytes.fromhex(ZZ["literals"][k]["hex"]) for k in names])
    for k, f in zip(names, frames):
        assert f.hex() == ZZ["literals"][k]["frame"], k


def test_golden_generated_all_strategies(gpu_store, oracle):
    cases = ZZ["generated"]
    chunks = [corpus.make(e["kind"], e["seed"], e["len"]) for e in cases]
    frames = _batch_compress(gpu_store, chunks)
    for e, c, f in zip(cases, chunks, frames):
        tag = (e["kind"], e["seed"], e["len"])
        assert len(f) == e["clen"], tag
        assert hashlib.sha256(f).hexdigest() == e["sha256"], tag
        if "frame" in e:
            assert f.hex() == e["frame"], tag


def test_golden_special_presplit_rle_raw(gpu_store, oracle):
    import gen_golden
    sp = gen_golden.special_inputs()
    names = sorted(ZZ["special"])
    frames = _batch_compress(gpu_store, [sp[k] for k in names])
    for k, f in zip(names, frames):
        assert len(f) == ZZ["special"][k]["clen"], k
        assert hashlib.sha256(f).hexdigest() == ZZ["special"][k]["sha256"], k
        assert oracle.decompress(f, len(sp[k])) == sp[k]


def test_vs_oracle_ragged_batch(gpu_store, oracle):
    rng = np.random.default_rng(11)
    chunks = []
    kinds = ["text", "records", "runs", "skewed", "mixed", "random"]
    for i in range(60):
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 17000), rng.integers(17000, 270000),
                            rng.integers(270000, 1200000)]))
        chunks.append(corpus.make(kinds[i % len(kinds)], 100 + i, n))
    frames = _batch_compress(gpu_store, chunks)
    for c, f in zip(chunks, frames):
        assert f == oracle.compress_chunk(c), len(c)


def test_store_insert_semantics(oracle):
    # src/util/tests.rs:106-166 restated against the HIP-backed ChunkStore
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    data = b"\x2a" * 2048
    r1 = st.insert(data)
    r2 = st.insert(data)
    assert r1.compressed_data is not None and r2.compressed_data is None
    assert r1.hash == r2.hash and st.len() == 1 and not st.is_empty()
    assert len(r1.compressed_data) < len(data)
    assert oracle.decompress(r1.compressed_data, 2048) == data
    r3 = st.insert(b"other chunk")
    assert r3.compressed_data is not None and r3.hash != r1.hash and st.len() == 2
    # batch form == n x insert in index order, bit-exact vs the oracle store
    import oracle_lib
    ost = oracle_lib.OracleStore(oracle)
    chunks = [corpus.text_like(5, 40000), corpus.random_bytes(9, 1, 70000), corpus.text_like(5, 40000),
              corpus.mixed(2, 300000), corpus.random_bytes(9, 1, 70000), b"x"]
    st2 = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    got = st2.insert_batch(chunks)
    for c, g in zip(chunks, got):
        h, comp = ost.insert(c)
        assert g.hash == h and g.compressed_data == comp
    assert st2.len() == ost.len() == 4
    st.close()
    st2.close()


def test_full_size_properties_256k(gpu_store, oracle):
    """BASELINE configs 1/2 shape: 256 KiB random chunks -> raw-block frames of 262,159 bytes that
    decode back; sampled bit-exact vs oracle; deterministic across runs."""
    chunks = [corpus.random_bytes(21, i, 262144) for i in range(96)]
    f1 = _batch_compress(gpu_store, chunks)
    f2 = _batch_compress(gpu_store, chunks)
    assert f1 == f2
    for c, f in zip(chunks, f1):
        assert len(f) == 262159
        assert f[9 + 3:9 + 3 + 131072] == c[:131072]
    for i in (0, 17, 95):
        assert f1[i] == oracle.compress_chunk(chunks[i])


def test_full_size_properties_2m(gpu_store, oracle):
    chunks = [corpus.mixed(40 + i, 2097152) for i in range(6)] + [corpus.text_like(3, 2097152)]
    frames = _batch_compress(gpu_store, chunks)
    for c, f in zip(chunks, frames):
        assert oracle.decompress(f, len(c)) == c
    assert frames[0] == oracle.compress_chunk(chunks[0])
    assert frames[-1] == oracle.compress_chunk(chunks[-1])


def test_direct_vs_libzstd_when_present(gpu_store):
    import zref
    if not zref.available():
        pytest.skip("libzstd 1.5.7 not in this image")
    chunks = [corpus.text_like(77, 100000), corpus.mixed(78, 500000), corpus.records(79, 12000)]
    for c, f in zip(chunks, _batch_compress(gpu_store, chunks)):
        assert f == zref.compress(c)


def test_scale_no_corruption_random_256k(gpu_store, oracle):
    """30,000 random 256 KiB chunks through the persistent kernel (many chunks per wave, so every
    workspace is reused dozens of times): every frame must be the 262,159-byte two-raw-block frame
    whose payload is the input; a sample is compared with the oracle bit for bit.  Guards against
    stale table words leaking from one chunk into the next."""
    import torch
    import corpus_gpu
    import squishrs_amd
    n, L = 30000, 262144
    data = corpus_gpu.random_chunks(5, 0, n, L, device="cuda", group=8192)
    offs = np.arange(n, dtype=np.uint64) * np.uint64(L)
    lens = np.full(n, L, dtype=np.uint32)
    comp = torch.empty(n * (squishrs_amd.compress_bound(L) + 16), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        coff = gpu_store.compress_device(data, offs, lens, None, comp)
        sizes = np.diff(coff.astype(np.int64))
        assert (sizes == 262159).all(), np.nonzero(sizes != 262159)[0][:10]
        frames = comp[:n * 262159].view(n, 262159)
        assert torch.equal(frames[:, 12:12 + 131072], data.view(n, L)[:, :131072])
        assert torch.equal(frames[:, 12 + 131072 + 3:], data.view(n, L)[:, 131072:])
    for i in (0, 12345, n - 1):
        c = corpus.random_bytes(5, i, L)
        assert frames[i].cpu().numpy().tobytes() == oracle.compress_chunk(c)


def test_scale_compressible_sampled_decode(gpu_store, oracle):
    """1,024 compressible 256 KiB chunks + 96 mixed 2 MiB chunks: every frame of a spread sample
    decodes back to its input with the oracle's independent decoder and equals the oracle frame."""
    rng = np.random.default_rng(3)
    kinds = ["text", "records", "mixed", "skewed"]
    chunks = [corpus.make(kinds[i % 4], 1000 + (i % 37), 262144) for i in range(1024)]
    frames = _batch_compress(gpu_store, chunks)
    ref = {}
    for i in rng.choice(len(chunks), size=48, replace=False):
        key = (kinds[i % 4], 1000 + (i % 37))
        if key not in ref:
            ref[key] = oracle.compress_chunk(chunks[i])
        assert frames[i] == ref[key], i
    big = [corpus.mixed(500 + (i % 12), 2097152) for i in range(96)]
    fb = _batch_compress(gpu_store, big)
    refb = {}
    for i in range(0, 96, 5):
        k = i % 12
        if k not in refb:
            refb[k] = oracle.compress_chunk(big[i])
        assert fb[i] == refb[k], i
    for i in range(96):
        assert fb[i] == fb[i % 12]          # identical inputs -> identical frames, whichever wave


def test_persistent_waves_many_chunks_per_wave(oracle):
    """4 compressor waves, 60 chunks of mixed classes: every wave keeps ONE workspace across ~15
    chunks, so hash-table generation tags (no clear between chunks of one size class), the forced
    clear on a class / strategy change and the no-match certificate's own generation are all
    exercised on the device.  Every frame must equal the oracle's."""
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12, compress_waves=4)
    kinds = ["random", "text", "planted0", "mixed", "planted3", "records", "random", "planted6", "skewed", "runs"]
    sizes = [262144, 262144, 200000, 262144, 70000, 131072, 262144, 262144, 40000, 262144, 9000, 600000]
    chunks = [corpus.make(kinds[i % len(kinds)], 100 + i, sizes[(i * 7) % len(sizes)]) for i in range(60)]
    try:
        frames = _batch_compress(st, chunks)
    finally:
        st.close()
    for i, (c, f) in enumerate(zip(chunks, frames)):
        assert f == oracle.compress_chunk(c), (i, kinds[i % len(kinds)], len(c))


def test_slots_layout_equals_packed(gpu_store, oracle):
    """sq_compress_batch_slots leaves every frame in its slot: same frames as the packed layout,
    length 0 for chunks that were not selected, slots disjoint and ascending."""
    import torch
    import squishrs_amd
    chunks = [corpus.make(k, 40 + i, n) for i, (k, n) in enumerate(
        [("random", 262144), ("text", 50000), ("planted1", 200000), ("zeros", 4096), ("mixed", 300000),
         ("random", 262144), ("records", 17000), ("text", 11)])]
    packed = _batch_compress(gpu_store, chunks)
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    offs = np.zeros(len(chunks), dtype=np.uint64)
    pos = 0
    for i, c in enumerate(chunks):
        offs[i] = pos
        pos += (len(c) + 255) // 256 * 256
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for i, c in enumerate(chunks):
        buf[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    select = torch.tensor([1, 1, 0, 1, 1, 1, 0, 1], dtype=torch.uint8, device="cuda")
    comp = torch.zeros(gpu_store.slots_bound(lens), dtype=torch.uint8, device="cuda")
    coff, clen = gpu_store.compress_device_slots(d, offs, lens, select, comp)
    host = comp.cpu().numpy()
    end = 0
    for i, c in enumerate(chunks):
        assert int(coff[i]) >= end and int(coff[i]) % 16 == 0
        end = int(coff[i]) + squishrs_amd.compress_bound(len(c))
        if int(select[i]):
            assert host[int(coff[i]):int(coff[i]) + int(clen[i])].tobytes() == packed[i] == oracle.compress_chunk(c)
        else:
            assert clen[i] == 0
    # too small an output buffer is refused, not overrun
    with pytest.raises(squishrs_amd.SquishError):
        gpu_store.compress_device_slots(d, offs, lens, select, comp[:1000])


def test_table_generations_persist_across_launches(oracle):
    """The two table pools keep their generation tags from launch to launch while their shapes do not
    change (no clearing), and start over when they do: several launches on one ctx, alternating
    between batches of the same and of different size classes / wave counts; every frame every time
    must equal the oracle's."""
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12, compress_waves=3)
    a = [corpus.make(k, 300 + i, 262144) for i, k in enumerate(["random", "planted0", "random", "planted4", "text",
                                                               "random", "planted9", "random", "planted8", "random",
                                                               "planted1", "random"])]
    b = [corpus.make(k, 400 + i, 70000) for i, k in enumerate(["random", "text", "planted3", "random", "mixed"])]
    c = [corpus.make("random", 500 + i, 262144) for i in range(2)]      # fewer chunks than waves: other wave count
    ref = {id(x): [oracle.compress_chunk(ch) for ch in x] for x in (a, b, c)}
    try:
        for batch in (a, a, b, a, c, a, a, b, b):
            assert _batch_compress(st, batch) == ref[id(batch)]
    finally:
        st.close()


def test_unaligned_chunk_starts(gpu_store, oracle):
    """Chunks may start at any byte offset of the batch buffer.  The incompressible-chunk kernel reads its
    input in aligned 16-byte windows and recognises the scheduled search positions arithmetically, so every
    residue of the start address modulo 16 (and chunks that end right at the buffer's last bytes) must give
    the oracle's frames."""
    import torch
    import squishrs_amd
    kinds = ["random", "planted0", "random", "planted3", "planted6", "random", "planted9", "text", "planted4",
             "random", "planted1", "random", "mixed", "planted8", "random", "random", "planted2"]
    sizes = [262144, 200000, 70000, 131072 + 40961, 262144, 40960, 262144, 60000, 150000,
             16385, 262143, 131081, 100000, 262144, 131072, 262137, 99999]
    chunks = [corpus.make(kinds[i], 700 + i, sizes[i]) for i in range(len(kinds))]
    offs = np.zeros(len(chunks), dtype=np.uint64)
    pos = 3
    for i, c in enumerate(chunks):
        offs[i] = pos                      # residues 3, 4+, ... : every chunk start lands on another residue
        pos += len(c) + 1 + (i % 5)
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    buf = np.zeros(pos, dtype=np.uint8)    # the last chunk ends at the very end of the buffer (+ its gap)
    for i, c in enumerate(chunks):
        buf[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    assert len({int(o) % 16 for o in offs}) >= 8
    d = torch.from_numpy(buf).cuda()
    comp = torch.zeros(gpu_store.slots_bound(lens), dtype=torch.uint8, device="cuda")
    coff, clen = gpu_store.compress_device_slots(d, offs, lens, None, comp)
    host = comp.cpu().numpy()
    for i, c in enumerate(chunks):
        assert host[int(coff[i]):int(coff[i]) + int(clen[i])].tobytes() == oracle.compress_chunk(c), (i, kinds[i], sizes[i])


def test_stage1_stress_random_sizes_and_alignments(gpu_store, oracle):
    """600 chunks of the btlazy2 size range with arbitrary lengths, arbitrary start alignments and a
    sprinkling of planted repeats, in one batch: every frame against the oracle.  (Lengths around the
    block boundary, the 40,960-byte threshold of the literal pre-check and the ends of the range
    included.)"""
    import torch
    rng = np.random.default_rng(20240607)
    special = [16385, 16400, 40959, 40960, 40961, 131064, 131071, 131072, 131073, 131080, 131081, 131090,
               131072 + 40959, 131072 + 40960, 131072 + 40961, 262129, 262143, 262144]
    chunks = []
    for i in range(600):
        n = special[i] if i < len(special) else int(rng.integers(16385, 262145))
        kind = "random" if i % 4 else "planted%d" % int(rng.integers(0, 10))
        chunks.append(corpus.make(kind, 9000 + i, n))
    offs = np.zeros(len(chunks), dtype=np.uint64)
    pos = 1
    for i, c in enumerate(chunks):
        offs[i] = pos
        pos += len(c) + int(rng.integers(0, 37))
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, c in enumerate(chunks):
        buf[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    comp = torch.zeros(gpu_store.slots_bound(lens), dtype=torch.uint8, device="cuda")
    coff, clen = gpu_store.compress_device_slots(d, offs, lens, None, comp)
    host = comp.cpu().numpy()
    bad = [i for i, c in enumerate(chunks)
           if host[int(coff[i]):int(coff[i]) + int(clen[i])].tobytes() != oracle.compress_chunk(c)]
    assert not bad, [(i, len(chunks[i])) for i in bad[:10]]


def test_two_contexts_in_flight(oracle):
    """Two sq_ctx driven by two host threads at the same time (own streams, buffers, dedup sets): what a packer
    does to hide the tail of one batch's compress stage under the next batch's hash and stage 1
    (DESIGN.md section 6).  Every frame of both must still equal the oracle's."""
    import threading
    import squishrs_amd
    chunks_a = [corpus.random_bytes(31, i, 262144) for i in range(40)] + [corpus.text_like(5, 90000), corpus.planted(3, 262144, 4)]
    chunks_b = [corpus.random_bytes(32, i, 262144) for i in range(40)] + [corpus.mixed(6, 200000), corpus.planted(4, 262144, 7)]
    out = {}

    def run(tag, chunks):
        st = squishrs_amd.ChunkStore(device=0, max_chunk=262144, dedup_capacity=1 << 12)
        try:
            res = None
            for _ in range(3):
                st.reset()
                res = st.insert_batch(chunks)
            out[tag] = res
        except Exception as e:  # surfaced by the assertions below
            out[tag] = e
        finally:
            st.close()

    th = [threading.Thread(target=run, args=("a", chunks_a)), threading.Thread(target=run, args=("b", chunks_b))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for tag, chunks in (("a", chunks_a), ("b", chunks_b)):
        res = out[tag]
        assert not isinstance(res, Exception), res
        for c, r in zip(chunks, res):
            assert r.hash == oracle.hash_chunk(c)
            assert r.compressed_data == oracle.compress_chunk(c)
