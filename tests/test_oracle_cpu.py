"""CPU suite: the oracle (oracle/*.c) against the committed golden vectors, which were produced by
the real libzstd 1.5.7 / xxHash (tools/gen_golden.py).  No GPU, no /root/reference."""
import hashlib
import json
import os

import pytest

import corpus

G = os.path.join(os.path.dirname(__file__), "golden")
XX = json.load(open(os.path.join(G, "xxh3_128.json")))
ZZ = json.load(open(os.path.join(G, "zstd12_frames.json")))


def test_hash_literals(oracle):
    for name, e in XX["literals"].items():
        assert oracle.hash_chunk(bytes.fromhex(e["hex"])).hex() == e["hash_le"], name


def test_hash_random_lengths(oracle):
    for e in XX["random"]:
        d = corpus.random_bytes(1, e["len"], e["len"])
        assert oracle.hash_chunk(d).hex() == e["hash_le"], e["len"]


def test_survey_known_answers(oracle):
    # SURVEY.md Appendix B
    assert oracle.hash_chunk(b"hello squish").hex() == "04d876a05811253b5c518b1a786dad1f"
    assert oracle.hash_chunk(b"").hex() == "7f498d4624c30160d8984701d306aa99"
    assert oracle.compress_chunk(b"hello squish").hex() == "28b52ffd200c61000068656c6c6f20737175697368"
    assert oracle.compress_chunk(b"").hex() == "28b52ffd2000010000"
    assert oracle.compress_chunk(b"\x01" * 1024).hex() == "28b52ffd60000345000008010100fc2b2004"
    assert oracle.compress_chunk(b"abc" * 1000).hex() == "28b52ffd60b80a550000186162630100b2d37743"
    assert oracle.compress_chunk(bytes(262144)).hex() == "28b52ffda0000004005400001000000100fbff39c00203001000"


def test_frames_literals(oracle):
    for name, e in ZZ["literals"].items():
        assert oracle.compress_chunk(bytes.fromhex(e["hex"])).hex() == e["frame"], name


@pytest.mark.parametrize("idx", range(len(ZZ["generated"])))
def test_frames_generated(oracle, idx):
    e = ZZ["generated"][idx]
    d = corpus.make(e["kind"], e["seed"], e["len"])
    fr = oracle.compress_chunk(d)
    assert len(fr) == e["clen"], (e["kind"], e["len"])
    assert hashlib.sha256(fr).hexdigest() == e["sha256"]
    if "frame" in e:
        assert fr.hex() == e["frame"]
    assert oracle.hash_chunk(d).hex() == e["hash_le"]
    assert oracle.decompress(fr, len(d)) == d   # independent decoder round trip


def test_frames_special(oracle):
    import gen_golden
    sp = gen_golden.special_inputs()
    for name, e in ZZ["special"].items():
        fr = oracle.compress_chunk(sp[name])
        assert len(fr) == e["clen"], name
        assert hashlib.sha256(fr).hexdigest() == e["sha256"], name
        assert oracle.decompress(fr, e["len"]) == sp[name]


def test_archive_roundtrip_fixture(oracle):
    # tests/roundtrip.rs of the reference: one file "file.txt" = "hello squish" 
    fx = json.load(open(os.path.join(G, "roundtrip_hello_squish.json")))
    files = [(p, bytes.fromhex(h)) for p, h in fx["files"].items()]
    assert oracle.pack_archive(files, fx["timestamp"]).hex() == fx["archive"]
    assert len(bytes.fromhex(fx["archive"])) == 124


def test_store_semantics(oracle):
    # src/util/tests.rs:106-166: first insert Some, duplicate None + same hash, len counts uniques,
    # compressed smaller than 2048 x 42 and decodable
    import oracle_lib
    st = oracle_lib.OracleStore(oracle)
    data = b"\x2a" * 2048
    h1, c1 = st.insert(data)
    h2, c2 = st.insert(data)
    assert c1 is not None and c2 is None and h1 == h2 and st.len() == 1
    assert len(c1) < len(data) and oracle.decompress(c1, 2048) == data
    h3, c3 = st.insert(b"other chunk")
    assert c3 is not None and h3 != h1 and st.len() == 2


def test_cross_shard_plan_for_config4():
    """bench.py --gpus N (BASELINE configs[4] shape): 30 % of all chunks copy an original on another shard."""
    import numpy as np
    for world, n in ((2, 1000), (8, 500)):
        plan = corpus.cross_shard_plan(world, n)
        N = world * n
        dup = plan != np.arange(N)
        assert dup.sum() == int(0.3 * N)
        assert (plan[plan[dup]] == plan[dup]).all()                  # sources are originals
        assert ((plan[dup] // n) != (np.nonzero(dup)[0] // n)).all()  # ... on another shard
        assert len(np.unique(plan)) == N - int(0.3 * N)
    assert (corpus.cross_shard_plan(1, 10) == np.arange(10)).all()


def test_oracle_is_reentrant(oracle):
    """The checker is called from thread pools (oracle/bench_threads.c, tools/gpu_fuzz.py): frames made on 16 threads
    must equal the ones made one at a time.  (Round 2: the pre-block splitter's fingerprints were one static pair
    shared by all threads, and a GPU fuzz run blamed the device for the oracle's own corruption.)"""
    from concurrent.futures import ThreadPoolExecutor
    import corpus
    chunks = [corpus.make(["text", "records", "mixed", "skewed", "zeros"][i % 5], i, 300_000 + 37_001 * i)
              for i in range(20)]
    single = [oracle.compress_chunk(c) for c in chunks]
    for _ in range(2):
        with ThreadPoolExecutor(16) as ex:
            assert list(ex.map(oracle.compress_chunk, chunks)) == single
