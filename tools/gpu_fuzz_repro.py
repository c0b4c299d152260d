"""Narrowing a gpu_fuzz mismatch: the first 512 fuzz chunks as one batch, then every mismatching chunk alone and in
small groups (experiment tooling)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, oracle_lib
sys.argv = ["x", "1500", "1"]
src = open(os.path.join(ROOT, "tools", "gpu_fuzz.py")).read().split("t0 = time.time()")[0]
exec(src)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    chunks = list(ex.map(make, range(512)))
import torch, squishrs_amd
from test_compress_gpu import _batch_compress
o = oracle_lib.load()
with ThreadPoolExecutor(16) as ex:
    want = list(ex.map(o.compress_chunk, chunks))
st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 16)
fr = _batch_compress(st, chunks)
bad = [i for i in range(512) if fr[i] != want[i]]
print("batch of 512: bad", len(bad), bad[:20], flush=True)
for i in bad[:6]:
    f1 = _batch_compress(st, [chunks[i]])[0]
    a, b = fr[i], want[i]
    k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), -1)
    dec_ok = None
    try:
        dec_ok = o.decompress(a, len(chunks[i])) == chunks[i]
    except AssertionError:
        dec_ok = False
    print(i, len(chunks[i]), "alone ok:", f1 == want[i], "len gpu/oracle", len(a), len(b), "first diff at", k, "decodes to input:", dec_ok, flush=True)
# only the row-range chunks, in one batch
rows = [i for i in range(512) if len(chunks[i]) > 262144]
fr2 = _batch_compress(st, [chunks[i] for i in rows])
bad2 = [rows[j] for j in range(len(rows)) if fr2[j] != want[rows[j]]]
print("row-range chunks only (", len(rows), "): bad", len(bad2), bad2[:10], flush=True)
# small chunks first then row chunks, with few waves: many chunks per wave
st4 = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12, compress_waves=8)
sub = list(range(0, 64))
fr3 = _batch_compress(st4, [chunks[i] for i in sub])
bad3 = [sub[j] for j in range(len(sub)) if fr3[j] != want[sub[j]]]
print("first 64 chunks on 8 waves: bad", len(bad3), bad3[:10], flush=True)
