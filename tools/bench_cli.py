"""End-to-end timing of the C++ host driver on the GPU box (SURVEY.md 8(d) "product number"): a synthetic tree
of N files x 256 KiB of random bytes (BASELINE configs[0]/[1] shape) is packed from disk through sq_pack_batch
(host buffers: H2D of the chunks, D2H of hashes / flags / frames, the single writer), listed, unpacked on the host
and compared.  Prints one JSON line.  usage: python tools/bench_cli.py [files] [file_bytes] [dir]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "squishrs_amd", "host", "squishrs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fb = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
EXTRA = {"SQUISH_PINNED": os.environ["SQUISH_PINNED"]} if "SQUISH_PINNED" in os.environ else {}
base = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="sqcli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
tree = os.path.join(base, "in")
os.makedirs(tree, exist_ok=True)
rng = np.random.default_rng(12345)
t0 = time.time()
for i in range(n):
    d = os.path.join(tree, f"d{i // 256:03d}")
    os.makedirs(d, exist_ok=True)
    rng.integers(0, 256, fb, dtype=np.uint8).tofile(os.path.join(d, f"f{i:06d}.bin"))
gen_s = time.time() - t0
arch = os.path.join(base, "a.squish")
out = os.path.join(base, "out")


def run(*a):
    t = time.perf_counter()
    r = subprocess.run([CLI, *a], capture_output=True, text=True, env=dict(os.environ, SQUISH_TIMING="1", **EXTRA))
    assert r.returncode == 0, r.stderr
    if "timing]" in r.stderr:
        sys.stderr.write(r.stderr)
    return time.perf_counter() - t


run("pack", tree, "-o", arch)            # warm-up: library load, context creation, page cache
os.remove(arch)
pack_s = run("pack", tree, "-o", arch)
list_s = run("list", arch, "--simple")
unpack_s = run("unpack", arch, "-o", out)
ok = subprocess.run(["diff", "-rq", tree, out], capture_output=True).returncode == 0
total = n * fb
print(json.dumps({"files": n, "file_bytes": fb, "input_MiB": total / 2**20, "archive_bytes": os.path.getsize(arch),
                  "pack_s": pack_s, "pack_MiBps": total / 2**20 / pack_s, "list_s": list_s,
                  "unpack_s": unpack_s, "unpack_MiBps": total / 2**20 / unpack_s, "roundtrip_identical": ok,
                  "where": base, "gen_s": gen_s}), flush=True)
shutil.rmtree(base, ignore_errors=True)
