"""Device-side twin of tools/corpus.py:random_bytes -- the same xoshiro256** streams (64
interleaved lanes per chunk, seeded through splitmix64), evaluated with torch integer ops so
BASELINE-sized workloads (tens of GiB) are generated directly in HBM.  chunk i of
`random_chunks(config_id, first, n, L)` equals corpus.random_bytes(config_id, first + i, L)."""
import numpy as np
import torch

import corpus

_M64 = (1 << 64) - 1


def _seeds(config_id: int, first: int, n: int) -> np.ndarray:
    """(n, 4, 64) uint64 initial states, same splitmix64 chain as corpus.xoshiro_bytes."""
    x = (np.uint64(corpus.BASE_SEED ^ (config_id << 32)) ^ np.arange(first, first + n, dtype=np.uint64))
    out = np.empty((n, 64, 4), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(64):
            for i in range(4):
                x = x + np.uint64(0x9E3779B97F4A7C15)
                z = x.copy()
                z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
                z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
                out[:, j, i] = z ^ (z >> np.uint64(31))
    return out.transpose(0, 2, 1).copy()  # (n, 4, 64)


def _rotl(x, k):
    # int64 tensors: logical right shift emulated by masking the sign extension
    return (x << k) | ((x >> (64 - k)) & ((1 << k) - 1))


def random_chunks(config_id: int, first: int, n: int, L: int, device="cuda", out=None, group=2048):
    """Returns a uint8 tensor of n*L bytes (L multiple of 512) holding n chunks back to back."""
    assert L % 512 == 0
    rounds = L // 512
    if out is None:
        out = torch.empty(n * L, dtype=torch.uint8, device=device)
    o64 = out.view(torch.int64).view(n, rounds, 64)
    for g0 in range(0, n, group):
        g = min(group, n - g0)
        st = torch.from_numpy(_seeds(config_id, first + g0, g).view(np.int64)).to(device)  # (g,4,64)
        s0, s1, s2, s3 = st[:, 0].clone(), st[:, 1].clone(), st[:, 2].clone(), st[:, 3].clone()
        for r in range(rounds):
            o64[g0:g0 + g, r] = _rotl(s1 * 5, 7) * 9
            t = s1 << 17
            s2 ^= s0
            s3 ^= s1
            s1 ^= s2
            s0 ^= s3
            s2 ^= t
            s3 = _rotl(s3, 45)
    return out
