"""Soak of the C++ driver on compressible data (GPU box): a tree of mixed files (multi-chunk files, tails of every
strategy, exact duplicates, tiny and empty files), packed with small batches so that dozens of batches go through
sq_pack_submit / sq_pack_wait with two in flight; the archive is compared byte for byte (timestamp masked) with the
oracle's .squish restatement and unpacked and diffed.  usage: python tools/soak_cli.py [total_MiB]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus
import oracle_lib

CLI = os.path.join(ROOT, "squishrs_amd", "host", "squishrs")
total = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 256 << 20
BATCH = os.environ.get("SOAK_BATCH", "default")
rng = np.random.default_rng(99)
base = tempfile.mkdtemp(prefix="sqsoak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
tree = os.path.join(base, "in")
files = {}
kinds = ["mixed", "text", "records", "runs", "skewed", "random"]
made = 0
i = 0
while made < total:
    k = kinds[i % len(kinds)]
    n = int(rng.choice([0, 11, 3000, 15000, 20000, 70000, 262144, 300000, 2097152, 2097153, 5000000, int(rng.integers(1, 9000000))]))
    data = corpus.make(k, 8000 + i, n) if n else b""
    files[f"d{i % 7}/s{i % 3}/f{i:04d}.{k}"] = data
    if i % 9 == 4:
        files[f"dup/f{i:04d}.copy"] = data        # exact duplicate of a whole file
    made += n
    i += 1
for rel, data in files.items():
    p = os.path.join(tree, rel)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    with open(p, "wb") as f:
        f.write(data)
arch = os.path.join(base, "a.squish")
t0 = time.perf_counter()
r = subprocess.run([CLI, "pack", tree, "-o", arch], capture_output=True, text=True,
                   env=dict(os.environ, SQUISH_TIMING="1", **({"SQUISH_BATCH_BYTES": os.environ["SOAK_BATCH"]} if "SOAK_BATCH" in os.environ else {})))
pack_s = time.perf_counter() - t0
assert r.returncode == 0, r.stderr
a = open(arch, "rb").read()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cli_gpu import _manifest_order
order = _manifest_order(a)
assert sorted(order) == sorted(files)
t0 = time.perf_counter()
want = oracle_lib.load().pack_archive([(rel, files[rel]) for rel in order], timestamp=0)
oracle_s = time.perf_counter() - t0
same = a[19:] == want[19:]
out = os.path.join(base, "out")
r2 = subprocess.run([CLI, "unpack", arch, "-o", out], capture_output=True, text=True)
ok = r2.returncode == 0 and subprocess.run(["diff", "-rq", tree, out], capture_output=True).returncode == 0
print(f"soak: {len(files)} files, {made / 2**20:.0f} MiB, archive {len(a)} B; pack {pack_s:.2f} s (batch bytes: {BATCH}); archive == oracle: {same} "
      f"(oracle, one thread: {oracle_s:.1f} s); unpack + diff identical: {ok}")
print(r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")
shutil.rmtree(base, ignore_errors=True)
assert same and ok
