"""CPU: the GPU zstd decoder's source (squishrs_amd/csrc/sq_zstd_decode.h) compiled with -DSQ_SIM (scalar bodies for the
wave primitives, LDS image and literal buffer poisoned): frames of the real libzstd 1.5.7 decode to their content, frames
of the oracle's level-12 encoder round-trip, damaged frames are judged exactly as the oracle's independent decoder judges
them.  Test build only (tests/sim/sim_decode.cpp); the device code path is covered by tests/test_decode_gpu.py."""
import ctypes
import hashlib
import json
import os
import random
import subprocess

import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "sim", "sim_decode.cpp")
SO = os.path.join(ROOT, "tests", "sim", "libsq_simdec.so")


@pytest.fixture(scope="module")
def simdec():
    deps = [SRC, os.path.join(ROOT, "squishrs_amd", "csrc", "sq_zstd_decode.h"), os.path.join(ROOT, "squishrs_amd", "csrc", "sq_wave.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", SO + ".tmp%d" % os.getpid(), SRC])
        os.replace(SO + ".tmp%d" % os.getpid(), SO)
    L = ctypes.CDLL(SO)
    L.sim_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int]
    L.sim_decompress.restype = ctypes.c_uint

    def dec(frame, cap, poison=1):
        out = ctypes.create_string_buffer(cap + 64)
        n = ctypes.c_size_t(0)
        st = L.sim_decompress(frame, len(frame), out, cap, ctypes.byref(n), poison)
        assert out.raw[cap:] == b"\0" * 64, "the decoder wrote behind its capacity"
        return st, out.raw[:n.value]
    dec.lib = L
    return dec


def _expected(case):
    if case["kind"] == "concat":
        return b"".join(corpus.make(k, s, n) for k, s, n in case["parts"])
    return corpus.make(case["kind"], case["seed"], case["nbytes"])


def test_sim_decodes_real_libzstd_frames(simdec):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd_decode_frames.json")))
    assert len(fx["cases"]) >= 60
    for c in fx["cases"]:
        want = _expected(c)
        assert hashlib.sha256(want).hexdigest() == c["sha256"], c["name"]
        st, got = simdec(bytes.fromhex(c["frame"]), 2 << 20)
        assert st == 0 and got == want, c["name"]
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd12_frames.json")))["literals"]
    k = 0
    for name, c in sorted(lit.items()):
        if isinstance(c, dict) and "frame" in c and "hex" in c:
            st, got = simdec(bytes.fromhex(c["frame"]), max(len(c["hex"]) // 2, 1))
            assert st == 0 and got == bytes.fromhex(c["hex"]), name
            k += 1
    assert k >= 6


@pytest.mark.parametrize("kind", ["text", "records", "runs", "skewed", "mixed", "random", "zeros"])
def test_sim_roundtrips_level12_frames(simdec, oracle, kind):
    for seed, n in ((1, 1), (2, 100), (3, 5000), (4, 12288), (5, 70000), (6, 262144), (7, 300000), (8, 2 << 20)):
        d = corpus.make(kind, seed, n)
        st, got = simdec(oracle.compress_chunk(d), len(d))   # exact capacity, as the reference's orig_size would be for a full chunk
        assert st == 0 and got == d, (kind, seed, n)


def test_sim_and_oracle_decoder_agree_on_damaged_frames(simdec, oracle):
    rnd = random.Random(9)
    seen = {}
    for d in (corpus.make("text", 7, 50000), corpus.make("mixed", 8, 300000), corpus.make("runs", 9, 20000)):
        fr = oracle.compress_chunk(d)
        for _ in range(120):
            f2 = bytearray(fr)
            for _k in range(rnd.randint(1, 4)):
                f2[rnd.randrange(len(f2))] = rnd.randrange(256)
            if rnd.random() < 0.3:
                f2 = f2[:rnd.randrange(1, len(f2))]
            f2 = bytes(f2)
            st, got = simdec(f2, len(d))
            try:
                m = oracle.decompress(f2, len(d))
            except AssertionError:
                m = None
            assert (st != 0) == (m is None) and (st != 0 or got == m)
            seen[st] = seen.get(st, 0) + 1
    assert seen.get(1, 0) > 100 and seen.get(0, 0) > 10      # corrupt ones and survivors both occur
    # capacity one byte short / empty input / not a frame
    d = corpus.make("text", 7, 50000)
    assert simdec(oracle.compress_chunk(d), len(d) - 1)[0] == 6
    assert simdec(b"", 16)[0] == 2
    assert simdec(b"not a zstd frame at all", 64)[0] == 3


def test_decoder_lds_allows_12_waves_per_cu(simdec):
    assert simdec.lib.sim_decode_lds_bytes() * 12 <= 160 * 1024
