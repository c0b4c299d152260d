"""GPU parity: HIP XXH3-128 + device dedup set through the C ABI vs the CPU oracle and the golden
vectors (bit-exact)."""
import json
import os

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _pack(chunks, align=256):
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    offs = np.zeros(len(chunks), dtype=np.uint64)
    pos = 0
    for i, c in enumerate(chunks):
        offs[i] = pos
        pos += (len(c) + align - 1) // align * align if align > 1 else len(c)
    buf = np.zeros(pos + 256, dtype=np.uint8)
    for i, c in enumerate(chunks):
        buf[int(offs[i]):int(offs[i]) + len(c)] = np.frombuffer(c, dtype=np.uint8)
    return buf, offs, lens


def _gpu_hashes(store, chunks, align=256):
    import torch
    buf, offs, lens = _pack(chunks, align)
    d = torch.from_numpy(buf).cuda()
    h = torch.zeros(len(chunks) * 16, dtype=torch.uint8, device="cuda")
    store.hash_device(d, offs, lens, h)
    return h.cpu().numpy().reshape(-1, 16)


def test_hash_golden_all_length_classes(gpu_store, oracle):
    xx = json.load(open(os.path.join(G, "xxh3_128.json")))
    chunks = [corpus.random_bytes(1, e["len"], e["len"]) for e in xx["random"]]
    got = _gpu_hashes(gpu_store, chunks)
    for e, g, c in zip(xx["random"], got, chunks):
        assert g.tobytes().hex() == e["hash_le"], e["len"]
        assert g.tobytes() == oracle.hash_chunk(c)


def test_hash_unaligned_and_ragged(gpu_store, oracle):
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(1, 300000, size=200)] + [0, 1, 240, 241, 1024, 1025, 2097152]
    chunks = [corpus.random_bytes(2, i, n) for i, n in enumerate(lens)]
    got = _gpu_hashes(gpu_store, chunks, align=1)      # byte-packed: arbitrary alignment
    for g, c in zip(got, chunks):
        assert g.tobytes() == oracle.hash_chunk(c), len(c)


def test_hash_chunk_api(oracle):
    import squishrs_amd
    assert squishrs_amd.hash_chunk(b"hello squish").hex() == "04d876a05811253b5c518b1a786dad1f"
    assert squishrs_amd.hash_chunk(b"some test data").hex() == "8bb490f540ec66cf9f9ea5e575cc3ff2"
    # src/util/tests.rs:88-104: determinism and inequality
    assert squishrs_amd.hash_chunk(b"abc") == squishrs_amd.hash_chunk(b"abc")
    assert squishrs_amd.hash_chunk(b"abc") != squishrs_amd.hash_chunk(b"abd")


def test_dedup_first_occurrence_in_index_order(oracle):
    import torch
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 18)
    rng = np.random.default_rng(9)
    n = 50000
    ids = rng.integers(0, 20000, size=n)
    # 16-byte keys derived from ids (some ids repeat => duplicates)
    keys = np.zeros((n, 16), dtype=np.uint8)
    for i, v in enumerate(ids):
        keys[i] = np.frombuffer(oracle.hash_chunk(int(v).to_bytes(8, "little")), dtype=np.uint8)
    d = torch.from_numpy(keys.reshape(-1)).cuda()
    is_new = torch.zeros(n, dtype=torch.uint8, device="cuda")
    half = n // 2
    st.dedup_device(d[:half * 16], half, 0, half, is_new[:half])          # batch 1
    st.dedup_device(d[half * 16:], n - half, 0, n - half, is_new[half:])  # batch 2 (state persists)
    got = is_new.cpu().numpy()
    seen = set()
    exp = np.zeros(n, dtype=np.uint8)
    for i, v in enumerate(ids):
        if int(v) not in seen:
            seen.add(int(v))
            exp[i] = 1
    assert (got == exp).all()
    assert st.len() == len(seen)
    st.close()


def test_hash_large_batch_property(gpu_store, oracle):
    """Full-size config shape (256 KiB chunks): checksum-of-checksums vs oracle on a sample, and
    duplicates hash equal."""
    import torch
    n = 512
    chunks = [corpus.random_bytes(3, i % 400, 262144) for i in range(n)]   # 112 duplicates
    got = _gpu_hashes(gpu_store, chunks)
    for i in (0, 1, 77, 399, 400, 511):
        assert got[i].tobytes() == oracle.hash_chunk(chunks[i])
    for i in range(400, n):
        assert (got[i] == got[i - 400]).all()
