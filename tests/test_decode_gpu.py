"""GPU parity of the zstd frame DECODER (sq_decompress_batch, SURVEY.md 8(f) row f2: the counterpart of
zstd::bulk::decompress in ArchiveReader::read_chunks, /root/reference/src/archive/reader.rs:259-314), all through the C ABI:
  * every frame of tests/golden/zstd_decode_frames.json -- frames of the REAL libzstd 1.5.7 at levels -5 .. 19, with and
    without content checksum / content size, concatenated and skippable frames -- decodes to its content;
  * every golden level-12 frame of the pack path (tests/golden/zstd12_frames.json) whose input the corpus can rebuild;
  * GPU pack -> GPU unpack round trips over every corpus kind and size class (2 MiB row-hash chunks included), device
    pointers and host pointers;
  * damaged frames: an error status (libzstd's wording) or -- where the damage leaves a valid stream -- some output, never
    a fault, never a write outside the record's capacity; a capacity that is too small is reported as such."""
import hashlib
import json
import os

import numpy as np
import pytest

import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def store():
    import squishrs_amd
    st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 14)
    yield st
    st.close()


def _expected(case):
    if case["kind"] == "concat":
        return b"".join(corpus.make(k, s, n) for k, s, n in case["parts"])
    return corpus.make(case["kind"], case["seed"], case["nbytes"])


def test_real_libzstd_frames_of_every_level_decode(store):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd_decode_frames.json")))
    assert len(fx["cases"]) >= 60
    want = [_expected(c) for c in fx["cases"]]
    for c, w in zip(fx["cases"], want):
        assert hashlib.sha256(w).hexdigest() == c["sha256"], c["name"]
    got = store.decompress_batch([bytes.fromhex(c["frame"]) for c in fx["cases"]], [2 << 20] * len(want))
    for c, w, g in zip(fx["cases"], want, got):
        assert g == w, c["name"]
    assert store.last_decode_errors == [None] * len(want)


def test_level12_goldens_of_the_pack_path_decode(store):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd12_frames.json")))
    frames, want = [], []
    for name, c in sorted(fx["literals"].items()):   # frame AND input as literal bytes (real libzstd 1.5.7, level 12)
        if isinstance(c, dict) and "frame" in c and "hex" in c:
            frames.append(bytes.fromhex(c["frame"]))
            want.append(bytes.fromhex(c["hex"]))
    assert len(frames) >= 6
    got = store.decompress_batch(frames, [max(len(w), 1) for w in want])
    assert got == want


@pytest.mark.parametrize("device_side", [False, True])
def test_gpu_pack_then_gpu_unpack_roundtrip(store, oracle, device_side):
    import torch
    chunks = []
    for kind in ("text", "records", "runs", "skewed", "mixed", "random", "zeros"):
        for seed, n in ((1, 1), (2, 300), (3, 9000), (4, 16385), (5, 70000), (6, 262144), (7, 300001), (8, 2 << 20)):
            chunks.append(corpus.make(kind, 100 + seed, n))
    store.reset()
    res = store.insert_batch(chunks)
    frames = [r.compressed_data if r.compressed_data is not None else oracle.compress_chunk(c) for r, c in zip(res, chunks)]
    if not device_side:
        got = store.decompress_batch(frames, [2 << 20] * len(frames))
        assert got == chunks
        return
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    offs = np.zeros(len(frames), dtype=np.uint64)
    offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]                  # back to back, no alignment at all
    src = torch.frombuffer(bytearray(b"".join(frames)), dtype=torch.uint8).cuda()
    caps = np.array([len(c) for c in chunks], dtype=np.uint32)         # exact capacities
    ooff = np.zeros(len(frames), dtype=np.uint64)
    ooff[1:] = np.cumsum(caps.astype(np.uint64))[:-1]
    out = torch.full((int(caps.sum()) + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    out_len, status = store.decompress_device(src, offs, lens, out, ooff, caps)
    assert (status == 0).all() and (out_len == caps).all()
    host = out.cpu().numpy().tobytes()
    assert host[:int(caps.sum())] == b"".join(chunks)
    assert host[int(caps.sum()):] == b"\xee" * 64                      # nothing written behind the last record


def test_damaged_frames_report_and_stay_inside_their_capacity(store, oracle):
    import random
    rnd = random.Random(9)
    base = [corpus.make("text", 7, 50000), corpus.make("mixed", 8, 300000), corpus.make("runs", 9, 20000)]
    frames, caps, origs = [], [], []
    for d in base:
        fr = oracle.compress_chunk(d)
        for _ in range(60):
            f2 = bytearray(fr)
            for _k in range(rnd.randint(1, 4)):
                f2[rnd.randrange(len(f2))] = rnd.randrange(256)
            if rnd.random() < 0.3:
                f2 = f2[:rnd.randrange(1, len(f2))]
            frames.append(bytes(f2))
            caps.append(len(d))
            origs.append(d)
    # plus: intact frames whose capacity is one byte short, an empty record, garbage
    frames += [oracle.compress_chunk(base[0]), b"", b"not a zstd frame at all"]
    caps += [len(base[0]) - 1, 16, 64]
    origs += [None, None, None]
    got = store.decompress_batch(frames, caps)
    errs = store.last_decode_errors
    host_model = []
    for f, cap in zip(frames, caps):
        try:
            host_model.append(oracle.decompress(f, cap))
        except Exception:  # noqa: BLE001 -- the oracle's decoder raises on damage
            host_model.append(None)
    n_err = 0
    for g, e, m, cap in zip(got, errs, host_model, caps):
        if g is None:
            n_err += 1
            assert e in ("Data corruption detected", "Src size is incorrect", "Unknown frame descriptor", "Unsupported frame parameter",
                         "Dictionary mismatch", "Destination buffer is too small", "Restored data doesn't match checksum"), e
        else:
            assert len(g) <= cap
        # the two decoders agree on what decodes and to what
        assert (g is None) == (m is None) and (g is None or g == m)
    assert n_err > 100
    assert errs[-3] == "Destination buffer is too small" and errs[-2] == "Src size is incorrect" and errs[-1] == "Unknown frame descriptor"
