"""Stage 1's second pass on the device ("one short match", squishrs_amd/csrc/sq_zstd_fast.h fastpath_blocks): a block that
ONE short match keeps from being certified is settled by the workgroup itself -- the match by byte comparisons, the rest of the
block by a second certificate laid at the match's end -- instead of being handed to k_compress_batch.  Every frame against
the oracle, in the split and in the fused form of a batch; reference behaviour: zstd::encode_all(level 12) of
/root/reference/src/fsutil/writer.rs:23-62."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_one_short_match_chunks_split_and_fused_forms(oracle):
    import fuzz_one_match_sim as f   # (its make_case; the simulator library it loads is not used here)
    import squishrs_amd
    from test_compress_gpu import _batch_compress
    rng = random.Random(4242)
    chunks, kinds = [], []
    while len(chunks) < 360:
        d, k = f.make_case(rng)
        if d is not None:
            chunks.append(d)
            kinds.append(k)
    want = [oracle.compress_chunk(c) for c in chunks]
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 14)
    frames = _batch_compress(st, chunks)
    decided = st.timings().fast_decided
    bad = [(i, kinds[i], len(chunks[i])) for i in range(len(chunks)) if frames[i] != want[i]]
    assert not bad, bad[:10]
    # the pass decided a good part of the plain cases (without it stage 1 decides none of these chunks: each has a coincidence)
    assert decided >= 40, decided
    st.reset()
    res = st.insert_batch(chunks)
    bad2 = [(i, kinds[i]) for i, r in enumerate(res) if r.compressed_data != want[i] or r.hash != oracle.hash_chunk(chunks[i])]
    assert not bad2, bad2[:10]
    st.close()


def test_compressible_batches_take_the_second_build_of_the_compressor_kernel(oracle):
    """k_compress_batch exists at 2 and at 3 waves per SIMD (squish_s2_w3.co); a batch runs the second when the batch before it
    left more than a quarter of its chunks to stage 2 (sq_api.hip, sq_ctx::f_s2w3).  Both must give the oracle's frames."""
    import corpus
    import squishrs_amd
    from test_compress_gpu import _batch_compress
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    loaded, _ = st.stage2_kernel()
    assert loaded, "squish_s2_w3.co was not loaded"
    text = [corpus.make("text", 900 + i, n) for i, n in enumerate([70000, 262144, 20000, 131073, 9000, 200000] * 4)]
    rnd = [corpus.make("random", 950 + i, 262144) for i in range(24)]
    want_text = [oracle.compress_chunk(c) for c in text]
    want_rnd = [oracle.compress_chunk(c) for c in rnd]
    assert _batch_compress(st, rnd) == want_rnd          # incompressible: stage 1 decides (nearly) all
    assert _batch_compress(st, text) == want_text        # first compressible batch: still the first build
    assert st.stage2_kernel()[1] is False
    assert _batch_compress(st, text) == want_text        # the batch before was compressible: the second build
    assert st.stage2_kernel()[1] is True
    assert _batch_compress(st, rnd) == want_rnd          # (history says compressible: the second build, on random chunks)
    assert _batch_compress(st, rnd) == want_rnd
    assert st.stage2_kernel()[1] is False
    st.close()


def test_trim_gives_the_pools_back_and_the_next_batch_is_unchanged(oracle):
    """sq_ctx_trim (include/squish_gpu.h): cached device buffers released, dedup set kept, frames of the next batch unchanged."""
    import corpus
    import squishrs_amd
    import torch
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12)
    chunks = [corpus.make(k, 1300 + i, n) for i, (k, n) in enumerate([("text", 100000), ("random", 262144), ("mixed", 300000), ("records", 50000)] * 3)]
    want = [oracle.compress_chunk(c) for c in chunks]
    res = st.insert_batch(chunks)
    assert [r.compressed_data for r in res[:4]] == want[:4]
    free0 = torch.cuda.mem_get_info()[0]
    st.trim()
    assert torch.cuda.mem_get_info()[0] > free0 + (64 << 20)     # (12 chunks: a pool of 12 workspaces of ~14 MB)
    assert st.len() == 12
    res2 = st.insert_batch(chunks)                                   # all duplicates now: the set survived
    assert all(r.compressed_data is None for r in res2)
    st.reset()
    res3 = st.insert_batch(chunks)
    assert [r.compressed_data for r in res3[:4]] == want[:4]
    st.close()
