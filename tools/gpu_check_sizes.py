"""Debug: pack N random 256 KiB chunks on the GPU and report every frame whose length differs
from the expected raw-block frame (262,159 B); compare those against the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, corpus_gpu, oracle_lib, squishrs_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
L = 262144
o = oracle_lib.load()
st = squishrs_amd.ChunkStore(device=0, max_chunk=L, dedup_capacity=n + 1024)
data = corpus_gpu.random_chunks(1, 0, n, L, device="cuda", group=16384)
offs = np.arange(n, dtype=np.uint64) * np.uint64(L); lens = np.full(n, L, dtype=np.uint32)
comp = torch.empty(n * (squishrs_amd.compress_bound(L) + 16), dtype=torch.uint8, device="cuda")
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    coff = st.compress_device(data, offs, lens, None, comp)
    sizes = np.diff(coff.astype(np.int64))
    bad = np.nonzero(sizes != 262159)[0]
    print(f"rep {rep}: total {int(coff[-1])} bad frames: {len(bad)}", flush=True)
    for i in bad[:10]:
        c = corpus.random_bytes(1, int(i), L)
        fr = comp[int(coff[i]):int(coff[i + 1])].cpu().numpy().tobytes()
        ref = o.compress_chunk(c)
        print(f"  chunk {i}: gpu len {len(fr)} oracle len {len(ref)} equal={fr == ref} gpu_chunk_ok={bytes(data[i*L:(i+1)*L].cpu().numpy()) == c}", flush=True)
        if fr != ref:
            import zref
            try:
                print("   gpu blocks", [(t, s) for t, s, _ in zref.parse_frame(fr)[1]], " oracle blocks", [(t, s) for t, s, _ in zref.parse_frame(ref)[1]])
            except Exception as e:
                print("   parse error", e)
