"""Measurement for the round-3 review's item 3 (configs[3]): would ONE fetch per diagonal (owner - candidate) instead of one
per (owner, candidate) pair pay in the row-hash search batches?  Runs the SQ_SIM build of the product's compressor (CPU) over
2 MiB chunks of the bench corpus kinds and prints, per kind: pairs per evaluation round, pairs with a live candidate,
distinct diagonals per round, and the share of pairs that lie inside the match of an EARLIER owner's pair on the same
diagonal (what a per-diagonal evaluation would get for free).  usage: python tools/diag_stats.py [kinds...]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus

SO = "/tmp/libsq_sim_diag.so"
subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", SO, os.path.join(ROOT, "tests", "sim", "sim_main.cpp")])
L = ctypes.CDLL(SO)
L.sim_compress_chunk.restype = ctypes.c_size_t
L.sim_compress_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
st = (ctypes.c_ulonglong * 5)()
for kind in (sys.argv[1:] or ["mixed", "text", "records", "skewed"]):
    tot = [0] * 5
    for seed in range(3):
        d = corpus.make(kind, 31000 + seed, 2 << 20)
        dst = ctypes.create_string_buffer(len(d) + 65536)
        n = L.sim_compress_chunk(d, len(d), dst, len(dst), 0)
        assert n
        L.sim_rb_stats(st)
        for k in range(5):
            tot[k] += st[k]
    r, p, live, nd, cov = tot
    print(f"{kind:8s} rounds {r:8d}  pairs/round {p / max(r, 1):6.1f}  live/round {live / max(r, 1):6.1f}  diagonals/round {nd / max(r, 1):6.1f}  "
          f"live pairs per diagonal {live / max(nd, 1):5.2f}  pairs inside an earlier pair's match on their diagonal {100.0 * cov / max(live, 1):5.1f} %")
