"""Device-side twin of tools/corpus.py:random_bytes -- the same xoshiro256** streams (64 interleaved lanes
per chunk, seeded through splitmix64), produced by ONE launch of a small HIP kernel of our own
(tools/corpus_gen.hip -> tools/libsq_corpus.so, built by __graft_entry__.build()) so that BASELINE-sized
workloads (tens of GiB) are generated directly in HBM.  chunk i of `random_chunks(config_id, first, n, L)`
equals corpus.random_bytes(config_id, first + i, L).  Measurement tooling, not part of the product.

(The first version evaluated the generator with torch integer ops: ~10^5 tiny elementwise launches, 2 s, and
the trigger of a profiler crash under `rocprofv3 --pmc` -- DESIGN.md "Incidents".)"""
import ctypes
import os

import numpy as np
import torch

import corpus

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsq_corpus.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built; run __graft_entry__.build()")
        L = ctypes.CDLL(path)
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        L.sqgen_random_chunks.argtypes = [vp, u64, u64, u64, u64, vp]
        L.sqgen_gather_chunks.argtypes = [vp, vp, vp, u64, u64, vp]
        L.sqgen_random_chunks_indexed.argtypes = [vp, u64, vp, u64, u64, vp]
        _LIB = L
    return _LIB


def random_chunks(config_id: int, first: int, n: int, L: int, device="cuda", out=None, group=None):
    """Returns a uint8 tensor of n*L bytes (L multiple of 512) holding n chunks back to back."""
    assert L % 512 == 0
    if out is None:
        out = torch.empty(n * L, dtype=torch.uint8, device=device)
    assert out.is_cuda and out.numel() >= n * L
    seed_base = (corpus.BASE_SEED ^ (config_id << 32)) & ((1 << 64) - 1)
    r = _lib().sqgen_random_chunks(out.data_ptr(), seed_base, first, n, L, torch.cuda.current_stream().cuda_stream)
    assert r == 0, f"sqgen_random_chunks failed ({r})"
    return out


def random_chunks_indexed(config_id: int, index, L: int, device="cuda", out=None):
    """chunk i = corpus.random_bytes(config_id, index[i], L); equal index entries are exact duplicates.
    index: int64 numpy array or tensor."""
    assert L % 512 == 0
    if not torch.is_tensor(index):
        index = torch.from_numpy(np.ascontiguousarray(index, dtype=np.int64))
    index = index.to(device=device, dtype=torch.int64).contiguous()
    n = index.numel()
    if out is None:
        out = torch.empty(n * L, dtype=torch.uint8, device=device)
    assert out.is_cuda and out.numel() >= n * L
    seed_base = (corpus.BASE_SEED ^ (config_id << 32)) & ((1 << 64) - 1)
    r = _lib().sqgen_random_chunks_indexed(out.data_ptr(), seed_base, index.data_ptr(), n, L,
                                           torch.cuda.current_stream().cuda_stream)
    assert r == 0, f"sqgen_random_chunks_indexed failed ({r})"
    torch.cuda.current_stream().synchronize()   # `index` may be a temporary
    return out


def gather_chunks(dst, src, index, L: int):
    """dst chunk i = src chunk index[i] (index: int64 tensor on the device); exact duplicates."""
    index = index.to(device=dst.device, dtype=torch.int64).contiguous()
    n = index.numel()
    assert dst.numel() >= n * L and L % 16 == 0
    r = _lib().sqgen_gather_chunks(dst.data_ptr(), src.data_ptr(), index.data_ptr(), n, L,
                                   torch.cuda.current_stream().cuda_stream)
    assert r == 0, f"sqgen_gather_chunks failed ({r})"
    torch.cuda.current_stream().synchronize()   # `index` may be a temporary
    return dst
