"""How often stage 1's second pass ("one short match", sq_zstd_fast.h fastpath_blocks) applies to random 256 KiB chunks and why
not: N random chunks through a TRACING simulator build of the two kernels (CPU).  Prints: chunks stage 1 decided, chunks with a
hand-over in block 0 / block 1 / both, second passes run / held, and the source lines of the refusals.
usage: python tools/one_match_stats.py [chunks] [seed]"""
import collections
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
SO = "/tmp/libsq_sim_trace.so"
subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-DSQ_SIM_TRACE", "-include", "cstdio", "-o", SO,
                       os.path.join(ROOT, "tests", "sim", "sim_main.cpp")])
s = ctypes.CDLL(SO)
s.sim_compress_two_stage.restype = ctypes.c_size_t
s.sim_compress_two_stage.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
n = 262144
log = tempfile.TemporaryFile()
saved = os.dup(2)
os.dup2(log.fileno(), 2)
decided = 0
for it in range(N):
    d = corpus.random_bytes(78, seed * 1000003 + it, n)
    dst = ctypes.create_string_buffer(n + 2048)
    dec = ctypes.c_int(0)
    s.sim_compress_two_stage(d, n, dst, n + 2048, ctypes.byref(dec))
    decided += dec.value
os.dup2(saved, 2)
log.seek(0)
txt = log.read().decode()
f0 = f1 = both = 0
for m in re.finditer(r"\[turn0\].*fail0 (\d) fail1 (\d)", txt):
    a, b = int(m.group(1)), int(m.group(2))
    f0 += a and not b
    f1 += b and not a
    both += a and b
held = sum(1 for m in re.finditer(r"second pass: certEnd (\d+) ilim (\d+) wakeBad (\d) nOvf (\d+)", txt)
           if int(m.group(1)) >= int(m.group(2)) and m.group(3) == "0" and m.group(4) == "0")
ran = len(re.findall(r"second pass:", txt))
why = collections.Counter(re.findall(r"refused at line (\d+)", txt))
print(f"{N} random chunks of 256 KiB: stage 1 decided {decided}; hand-over in block 0 only {f0}, block 1 only {f1}, both {both}; "
      f"second passes run {ran}, held {held}; refusals by source line (the first is 'certified anyway / not this case'): {dict(why)}")
