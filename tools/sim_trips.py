"""Where a compressible chunk's dependent trips to memory go (experiment tooling).
usage: python tools/sim_trips.py [kind] [chunk_bytes] [seed]
Builds the wave simulator (tests/sim/sim_main.cpp) with -DSQ_SIM_TRIPS: every uniform load the wave waits for and every
drain (wv_mem_sync) is counted under the phase the match finder is in.  One wave's time on this class is trips x ~1 us
(DESIGN 4.1 "the tree step"), so the table is the profile of stage 2 on such a chunk."""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 41000
so = os.path.join(tempfile.gettempdir(), "libsq_sim_trips.so")
subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DSQ_SIM_TRIPS", "-o", so, os.path.join(ROOT, "tests", "sim", "sim_main.cpp")])
lib = ctypes.CDLL(so)
lib.sim_compress_chunk.restype = ctypes.c_size_t
lib.sim_compress_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
d = corpus.make(kind, seed, n)
cap = n + (n >> 8) + 1024
dst = ctypes.create_string_buffer(cap)
out = (ctypes.c_ulonglong * 16)()
lib.sim_trip_counts(out)
m = lib.sim_compress_chunk(d, n, dst, cap, 1)
lib.sim_trip_counts(out)
names = ["parser and the rest", "search: hash + bucket head", "search: deferred-search check", "search: candidate scan",
         "search: candidate insertions", "search: final walk", "insertion of skipped positions", "tree steps: second trip (> 8 equal)"]
tot = sum(out)
print(f"{kind} {n} B seed {seed}: {m} B out, {tot} trips ({tot / n:.2f} per input byte)")
for i, nm in enumerate(names):
    if out[i] or out[8 + i]:
        print(f"  {nm:34s} loads {out[i]:9d}  drains {out[8 + i]:8d}  {100.0 * (out[i] + out[8 + i]) / tot:5.1f} %")
