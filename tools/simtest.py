"""Diff the host wave-simulator build of the device compressor (tests/sim/libsq_sim.so) against
the oracle (and real libzstd when present).  Build container helper."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import corpus  # noqa: E402
import difftest  # noqa: E402
import zref  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_sim():
    s = ctypes.CDLL(os.path.join(ROOT, "tests", "sim", "libsq_sim.so"))
    s.sim_compress_chunk.restype = ctypes.c_size_t
    s.sim_compress_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    s.sim_workspace_bytes.restype = ctypes.c_size_t
    s.sim_workspace_bytes.argtypes = [ctypes.c_size_t]
    s.sim_lds_bytes.restype = ctypes.c_size_t
    return s


def sim_compress(s, data, poison=1):
    cap = len(data) + (len(data) >> 8) + 1024
    dst = ctypes.create_string_buffer(cap)
    n = s.sim_compress_chunk(data, len(data), dst, cap, poison)
    return dst.raw[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="text,random,records,runs,skewed,mixed,zeros")
    ap.add_argument("--sizes", default="7,100,1000,5000,16384,20000,70000,140000,262144,300000,700000,2097152")
    ap.add_argument("--seeds", type=int, default=2)
    a = ap.parse_args()
    s = load_sim()
    o = difftest.load_oracle()
    print("LDS bytes per wave:", s.sim_lds_bytes(), " workspace(256K):", s.sim_workspace_bytes(262144),
          " workspace(2M):", s.sim_workspace_bytes(2097152))
    bad = tot = 0
    for kind in a.kinds.split(","):
        for size in [int(x) for x in a.sizes.split(",")]:
            for seed in range(a.seeds):
                d = corpus.make(kind, seed, size)
                ref = difftest.oracle_compress(o, d)
                got = sim_compress(s, d)
                tot += 1
                if got != ref:
                    bad += 1
                    print(f"MISMATCH {kind} n={size} seed={seed}: oracle {len(ref)} sim {len(got)} first diff {difftest.first_diff(ref, got)}")
    print(f"{tot - bad}/{tot} ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
