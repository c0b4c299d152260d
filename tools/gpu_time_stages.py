"""Timing helper: N random 256 KiB chunks generated in HBM, compress stage only (slots layout);
prints the stage timers.  Used with the SQ_FAST_DBG / SQ_NO_FASTPATH experiment switches, so it
does not check the frames."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus_gpu, squishrs_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 262144
st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
data = corpus_gpu.random_chunks(1, 0, n, L, device="cuda", group=16384)
offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, dtype=np.uint32)
comp = torch.empty(st.slots_bound(lens), dtype=torch.uint8, device="cuda")
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    try:
        st.compress_device_slots(data, offs, lens, None, comp)
    except squishrs_amd.SquishError as e:
        print("error", e)
    t = st.timings()
    print(f"rep {rep}: fast {t.fast_ms:.2f} ms  compress {t.compress_ms:.2f} ms  dbg={os.environ.get('SQ_FAST_DBG', '0')}", flush=True)
