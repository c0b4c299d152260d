"""CPU suite: the PRODUCT's compressor source (squishrs_amd/csrc/sq_zstd_*.h) compiled with
-DSQ_SIM -- every wave primitive replaced by a scalar body, workspace and LDS image poisoned -- and
diffed against the oracle.  This exercises the wave-uniform control logic that cannot run here
without a GPU: frame/block loop, pre-split, the three parsers, dead-insert elision, speculative
search epochs with deferred-search replay, rollback, eager fallback.  Test build only; never linked
into libsquish_gpu.so."""
import ctypes
import os
import subprocess

import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_SRC = os.path.join(ROOT, "tests", "sim", "sim_main.cpp")
SIM_SO = os.path.join(ROOT, "tests", "sim", "libsq_sim.so")


@pytest.fixture(scope="module")
def sim():
    srcs = [SIM_SRC] + [os.path.join(ROOT, "squishrs_amd", "csrc", f)
                        for f in os.listdir(os.path.join(ROOT, "squishrs_amd", "csrc")) if f.endswith(".h")]
    if not os.path.exists(SIM_SO) or any(os.path.getmtime(s) > os.path.getmtime(SIM_SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", SIM_SO, SIM_SRC])
    lib = ctypes.CDLL(SIM_SO)
    lib.sim_compress_chunk.restype = ctypes.c_size_t
    lib.sim_compress_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.sim_lds_bytes.restype = ctypes.c_size_t
    return lib


def _sim_compress(lib, data):
    cap = len(data) + (len(data) >> 8) + 1024
    dst = ctypes.create_string_buffer(cap)
    n = lib.sim_compress_chunk(data, len(data), dst, cap, 1)
    return dst.raw[:n]


CASES = [(k, s, n) for k in ("random", "text", "mixed", "runs", "records", "skewed", "zeros")
         for s, n in ((0, 3000), (1, 16384), (0, 20000), (1, 70000), (0, 131073), (1, 262144))]
CASES += [("mixed", 2, 600000), ("text", 3, 1048577), ("random", 4, 2097152), ("mixed", 5, 2097152)]


@pytest.mark.parametrize("kind,seed,n", CASES)
def test_sim_matches_oracle(sim, oracle, kind, seed, n):
    d = corpus.make(kind, seed, n)
    assert _sim_compress(sim, d) == oracle.compress_chunk(d)


def test_lds_budget_allows_8_waves_per_cu(sim):
    # 160 KiB of LDS per CU, one wave per workgroup
    assert sim.sim_lds_bytes() * 8 <= 160 * 1024
