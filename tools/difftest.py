"""Differential test of the CPU oracle against real libzstd 1.5.7 / python-xxhash (build
container only).  Usage: python tools/difftest.py [--kinds text,random] [--sizes 70000,300000]
[--seeds 3] [--hash]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import corpus  # noqa: E402
import zref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Seq(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint), ("litLength", ctypes.c_uint),
                ("matchLength", ctypes.c_uint), ("rep", ctypes.c_uint)]


def load_oracle():
    o = ctypes.CDLL(os.environ.get("SQO_SO", os.path.join(ROOT, "oracle", "_build", "libsq_oracle.so")))
    o.sqo_compress_chunk.restype = ctypes.c_size_t
    o.sqo_compress_chunk.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    o.sqo_compress_bound.restype = ctypes.c_size_t
    o.sqo_compress_bound.argtypes = [ctypes.c_size_t]
    o.sqo_generate_sequences.restype = ctypes.c_size_t
    o.sqo_generate_sequences.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    o.sqo_hash_chunk.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return o


def oracle_compress(o, data):
    cap = o.sqo_compress_bound(len(data)) + 64
    dst = ctypes.create_string_buffer(cap)
    n = o.sqo_compress_chunk(data, len(data), dst, cap)
    return dst.raw[:n]


def oracle_seqs(o, data):
    cap = len(data) // 3 + 1024
    arr = (Seq * cap)()
    n = o.sqo_generate_sequences(data, len(data), arr, cap)
    return [(arr[i].offset, arr[i].litLength, arr[i].matchLength, arr[i].rep) for i in range(n)]


def oracle_hash(o, data):
    out = ctypes.create_string_buffer(16)
    o.sqo_hash_chunk(data, len(data), out)
    return out.raw


def first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return min(len(a), len(b)) if len(a) != len(b) else -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="text,random,records,runs,skewed,mixed,zeros")
    ap.add_argument("--sizes", default="20000,70000,140000,262144,300000,700000,2097152")
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--hash", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    o = load_oracle()
    nfail = 0
    ntot = 0
    if a.hash:
        lens = list(range(0, 300)) + [1023, 1024, 1025, 2047, 2048, 2049, 4096, 65536, 100000, 262144, 2097152]
        for ln in lens:
            d = corpus.random_bytes(1, ln, ln)
            ntot += 1
            if oracle_hash(o, d) != zref.xxh3_128_le(d):
                print("HASH MISMATCH len", ln)
                nfail += 1
        print(f"hash: {ntot - nfail}/{ntot} ok")
    for kind in a.kinds.split(","):
        for size in [int(s) for s in a.sizes.split(",")]:
            for seed in range(a.seeds):
                data = corpus.make(kind, seed, size)
                t0 = time.time()
                ref = zref.compress(data)
                t1 = time.time()
                mine = oracle_compress(o, data)
                t2 = time.time()
                ntot += 1
                ok = ref == mine
                if not ok:
                    nfail += 1
                    rs = zref.generate_sequences(data)
                    ms = oracle_seqs(o, data)
                    sd = first_diff(rs, ms)
                    _, rb = zref.parse_frame(ref)
                    try:
                        _, mb = zref.parse_frame(mine)
                        mbs = [(t, s, len(b)) for t, s, b in mb]
                    except Exception as e:  # noqa: BLE001
                        mbs = f"unparseable ({e})"
                    print(f"MISMATCH {kind} n={size} seed={seed}: ref {len(ref)} B, oracle {len(mine)} B,"
                          f" first byte diff {first_diff(ref, mine)}; seq diff at {sd} "
                          f"(ref nseq {len(rs)}, mine {len(ms)})")
                    if sd >= 0:
                        print("   ref :", rs[max(0, sd - 2):sd + 3])
                        print("   mine:", ms[max(0, sd - 2):sd + 3])
                    print("   ref blocks :", [(t, s, len(b)) for t, s, b in rb][:12])
                    print("   mine blocks:", mbs if isinstance(mbs, str) else mbs[:12])
                elif a.verbose:
                    print(f"ok {kind} n={size} seed={seed} clen={len(ref)} ref {t1 - t0:.3f}s oracle {t2 - t1:.3f}s")
    print(f"{ntot - nfail}/{ntot} ok")
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
