"""GPU bring-up probe: compress a list of cases one at a time through the C ABI, logging progress
and timing to gpurun_out/probe.log before/after each case (so a hang is attributable)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus  # noqa: E402
import oracle_lib  # noqa: E402
import squishrs_amd  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "probe.log"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()
    os.fsync(LOG.fileno())


cases = []
for spec in sys.argv[1:]:
    kind, seed, n, rep = spec.split(":")
    cases.append((kind, int(seed), int(n), int(rep)))

o = oracle_lib.load()
st = squishrs_amd.ChunkStore(device=0, dedup_capacity=1 << 12, compress_waves=int(os.environ.get("SQ_WAVES", "0")))
for kind, seed, n, rep in cases:
    chunks = [corpus.make(kind, seed + i, n) for i in range(rep)]
    lens = np.array([len(c) for c in chunks], dtype=np.uint32)
    stride = (n + 255) // 256 * 256
    offs = np.arange(rep, dtype=np.uint64) * stride
    buf = np.zeros(rep * stride + 256, dtype=np.uint8)
    for i, c in enumerate(chunks):
        buf[i * stride:i * stride + n] = np.frombuffer(c, dtype=np.uint8)
    d = torch.from_numpy(buf).cuda()
    comp = torch.zeros(rep * (squishrs_amd.compress_bound(n) + 64), dtype=torch.uint8, device="cuda")
    log(f"start {kind} seed={seed} n={n} x{rep}")
    t0 = time.time()
    coff = st.compress_device(d, offs, lens, None, comp)
    dt = time.time() - t0
    host = comp.cpu().numpy()
    ok = 0
    t1 = time.time()
    ncheck = rep if os.environ.get("SQ_PROBE_CHECK") == "all" else min(4, rep)
    for i, c in enumerate(chunks[:ncheck]):
        f = host[int(coff[i]):int(coff[i + 1])].tobytes()
        ok += (f == o.compress_chunk(c))
    cpu = (time.time() - t1) / ncheck
    log(f"  done {dt * 1e3:.1f} ms wall, kernel {st.timings().compress_ms:.1f} ms; parity {ok}/{ncheck};"
        f" clen0={int(coff[1] - coff[0])}; oracle cpu {cpu * 1e3:.1f} ms/chunk")
st.close()
