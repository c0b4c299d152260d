"""Deterministic synthetic inputs shared by the oracle diff tests, the golden-vector generator,
the GPU parity tests and bench.py.  Pure numpy; nothing here reads
/root/reference or any file outside the repo, so it runs unchanged on the GPU box.

PRNG: xoshiro256** seeded by splitmix64(0x7371756973685253 ^ config_id ^ index); bytes are the
little-endian u64 stream."""
import numpy as np

_M64 = (1 << 64) - 1
BASE_SEED = 0x7371756973685253


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31), x


def _rotl(x, k):
    return ((x << np.uint64(k)) | (x >> np.uint64(64 - k)))


def xoshiro_bytes(seed: int, nbytes: int) -> bytes:
    """nbytes of xoshiro256** output.  Vectorised over 64 interleaved streams (stream j is seeded
    from splitmix64 chain element 4j..4j+3), so it is fast in numpy and trivially reproducible
    on a GPU with one stream per lane."""
    lanes = 64
    s = np.zeros((4, lanes), dtype=np.uint64)
    x = seed & _M64
    for j in range(lanes):
        for i in range(4):
            v, x = splitmix64(x)
            s[i, j] = v
    nwords = (nbytes + 7) // 8
    rounds = (nwords + lanes - 1) // lanes
    out = np.empty((rounds, lanes), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for r in range(rounds):
            out[r] = _rotl(s[1] * np.uint64(5), 7) * np.uint64(9)
            t = s[1] << np.uint64(17)
            s[2] ^= s[0]
            s[3] ^= s[1]
            s[1] ^= s[2]
            s[0] ^= s[3]
            s[2] ^= t
            s[3] = _rotl(s[3], 45)
    return out.reshape(-1).tobytes()[:nbytes]


def random_bytes(config_id: int, index: int, nbytes: int) -> bytes:
    return xoshiro_bytes(BASE_SEED ^ (config_id << 32) ^ index, nbytes)


_WORDS = None


def _wordlist():
    """A fixed pseudo-vocabulary (no external corpus): 4096 'words' built from syllables."""
    global _WORDS
    if _WORDS is None:
        syl = ["ka", "to", "ri", "mo", "se", "lu", "an", "de", "pi", "vo", "ne", "sha", "qu", "ex",
               "ing", "er", "ti", "on", "al", "st", "re", "un", "co", "mp", "th", "ou", "gh", "ea"]
        rng = np.random.default_rng(12345)
        words = []
        for _ in range(4096):
            k = int(rng.integers(1, 5))
            words.append("".join(syl[int(i)] for i in rng.integers(0, len(syl), size=k)))
        _WORDS = words
    return _WORDS


def text_like(seed: int, nbytes: int) -> bytes:
    """Zipf-distributed words with punctuation/newlines/indentation: compresses to roughly
    25-35 % at zstd level 12, with a realistic mix of short and long matches."""
    rng = np.random.default_rng(seed)
    words = _wordlist()
    out = bytearray()
    ranks = rng.zipf(1.3, size=nbytes // 3 + 16)
    ranks = np.minimum(ranks - 1, len(words) - 1)
    puncts = rng.integers(0, 40, size=len(ranks))
    i = 0
    while len(out) < nbytes:
        w = words[int(ranks[i])]
        p = int(puncts[i])
        i += 1
        out += w.encode()
        if p == 0:
            out += b".\n"
        elif p == 1:
            out += b",\n    "
        elif p == 2:
            out += b" = "
        elif p == 3:
            out += b"(" + str(int(ranks[i % len(ranks)])).encode() + b")"
        else:
            out += b" "
    return bytes(out[:nbytes])


def records(seed: int, nbytes: int) -> bytes:
    """Structured binary records: 32-byte rows with slowly varying fields."""
    rng = np.random.default_rng(seed)
    n = nbytes // 32 + 1
    rec = np.zeros((n, 32), dtype=np.uint8)
    ids = np.arange(n, dtype=np.uint32) + np.uint32(rng.integers(0, 1 << 20))
    rec[:, 0:4] = ids.view(np.uint8).reshape(n, 4)
    ts = (np.cumsum(rng.integers(0, 50, size=n)).astype(np.uint64) + np.uint64(1_700_000_000))
    rec[:, 4:12] = ts.view(np.uint8).reshape(n, 8)
    rec[:, 12] = rng.integers(0, 4, size=n)
    rec[:, 13] = rng.integers(0, 16, size=n)
    val = rng.normal(1000, 30, size=n).astype(np.float32)
    rec[:, 16:20] = val.view(np.uint8).reshape(n, 4)
    rec[:, 20:24] = rng.integers(0, 256, size=(n, 4))
    return rec.reshape(-1).tobytes()[:nbytes]


def runs(seed: int, nbytes: int) -> bytes:
    """Long zero / RLE runs with occasional noise."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    while len(out) < nbytes:
        kind = int(rng.integers(0, 4))
        ln = int(rng.integers(64, 40000))
        if kind == 0:
            out += bytes(ln)
        elif kind == 1:
            out += bytes([int(rng.integers(0, 256))]) * ln
        elif kind == 2:
            pat = bytes(rng.integers(0, 256, size=int(rng.integers(2, 9)), dtype=np.uint8))
            out += (pat * (ln // len(pat) + 1))[:ln]
        else:
            out += bytes(rng.integers(0, 256, size=min(ln, 512), dtype=np.uint8))
    return bytes(out[:nbytes])


def skewed(seed: int, nbytes: int, alpha: float = 1.5) -> bytes:
    """i.i.d. bytes from a skewed distribution: few LZ matches, Huffman-compressible."""
    rng = np.random.default_rng(seed)
    v = np.minimum(rng.zipf(alpha, size=nbytes) - 1, 255).astype(np.uint8)
    perm = np.random.default_rng(seed ^ 0x55).permutation(256).astype(np.uint8)
    return perm[v].tobytes()


def mixed(seed: int, nbytes: int, region: int = 65536) -> bytes:
    """Silesia-style mixed entropy: regions of text / records / runs / random."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    k = 0
    while len(out) < nbytes:
        kind = int(rng.integers(0, 5))
        ln = min(int(rng.integers(region // 4, region * 2)), nbytes - len(out))
        s = seed * 1000003 + k
        if kind == 0:
            out += text_like(s, ln)
        elif kind == 1:
            out += records(s, ln)
        elif kind == 2:
            out += runs(s, ln)
        elif kind == 3:
            out += random_bytes(99, s & 0xFFFFFFFF, ln)
        else:
            out += skewed(s, ln)
        k += 1
    return bytes(out[:nbytes])


GENERATORS = {
    "random": lambda seed, n: random_bytes(7, seed, n),
    "text": text_like,
    "records": records,
    "runs": runs,
    "skewed": skewed,
    "mixed": mixed,
    "zeros": lambda seed, n: bytes(n),
}


def lazy_schedule(nbytes: int, block: int = 128 << 10):
    """Positions (offsets into the chunk) the zstd lazy parser searches while it finds nothing:
    step = ((ip - anchor) >> 8) + 1, restarted at every 128 KiB block; the first block starts at 1."""
    out = []
    start = 0
    while start < nbytes:
        end = min(start + block, nbytes)
        p = start + (1 if start == 0 else 0)
        while p < end - 8:
            out.append(p)
            p += ((p - start) >> 8) + 1
        start = end
    return out


def planted(seed: int, nbytes: int, variant: int = 0) -> bytes:
    """Incompressible bytes with a few planted repeats, aimed at (or deliberately beside) the
    positions the lazy parser searches on incompressible data.  Exercises the compressor's
    no-match certificate: where it must stop, the hand-over to the exact search code, and the
    deferred tree updates of the searches it skipped."""
    b = bytearray(random_bytes(77, seed, nbytes))
    sched = lazy_schedule(nbytes)
    rnd = xoshiro_bytes(seed * 7919 + variant, 64)
    pick = lambda k, lo, hi: lo + int.from_bytes(rnd[4 * k:4 * k + 4], "little") % max(1, hi - lo)

    def copy(dst, src, ln):
        b[dst:dst + ln] = b[src:src + ln]

    if not sched:
        return bytes(b)
    mid = sched[pick(0, len(sched) // 3, 2 * len(sched) // 3)]
    late = sched[pick(1, 3 * len(sched) // 4, len(sched) - 1)]
    if variant == 0:      # a 40-byte repeat exactly at a searched position
        copy(mid, pick(2, 0, mid - 64), 40)
    elif variant == 1:    # only 4 equal bytes at a searched position (a length-4 match)
        copy(mid, pick(2, 0, mid - 64), 4)
    elif variant == 2:    # a repeat that starts one byte after a searched position (found at depth 0 only via s+1..)
        copy(mid + 1, pick(2, 0, mid - 64), 24)
    elif variant == 3:    # a run of equal bytes over a searched position: the repcode (offset 1) test fires
        b[mid - 2:mid + 30] = bytes([b[mid]]) * 32
    elif variant == 4:    # two plants, the later one first in memory order of the source
        copy(late, pick(2, 0, mid - 64), 300)
        copy(mid, pick(3, 0, mid - 64), 17)
    elif variant == 5:    # repeat of a LATER region at an earlier searched position: nothing to find there,
        src = min(nbytes - 64, late + 100)   # but the later region then matches the earlier one
        copy(mid, src, 33)
    elif variant == 6:    # three searched positions share their 4 bytes (bucket shared inside the schedule)
        a, c, d = sched[len(sched) // 5], sched[len(sched) // 2], sched[len(sched) - 3]
        copy(c, a, 4)
        copy(d, a, 9)
    elif variant in (8, 9):  # a match in block 0 only: block 1 keeps its own certificate, but starts with
        first = [x for x in sched if x < (128 << 10)]   # the repcode offset that block 0 left behind
        t0 = first[3 * len(first) // 4]
        # 8: too short a match to pay for a compressed block -> block 0 is emitted raw and its repcodes
        # are dropped; 9: long enough -> block 0 is compressed and block 1 inherits its repcode offset
        ln = 50 if variant == 8 else 6000
        src = pick(2, 0, max(65, t0 - ln - 64) - 64)
        copy(t0, src, ln)
        if variant == 9 and nbytes > (128 << 10) + 4096:  # ... which then fires at a searched position of block 1
            second = [x for x in sched if x >= (128 << 10)]
            t1 = second[len(second) // 2]
            off = t0 - src
            b[t1 + 1:t1 + 9] = b[t1 + 1 - off:t1 + 9 - off]
    elif variant == 7:    # plant in the last block only, source in the first block
        last_start = ((nbytes - 1) // (128 << 10)) * (128 << 10)
        tail = [x for x in sched if x >= last_start]
        t = tail[len(tail) // 2] if tail else mid
        copy(t, pick(2, 0, max(65, min(t, 128 << 10)) - 64), 64)
    return bytes(b)


def row_hash24(five: bytes, bits: int = 24) -> int:
    """ZSTD_hash5 of the row-hash finder (salt 0): ((read64LE << 24) * prime5) >> (64 - bits); five = the 5 bytes."""
    v = int.from_bytes(five[:5] + bytes(3), "little")
    return (((v << 24) & ((1 << 64) - 1)) * 889523592379 & ((1 << 64) - 1)) >> (64 - bits)


def planted_large(seed: int, nbytes: int, variant: int = 0) -> bytes:
    """Incompressible bytes of the lazy2 / row-hash size class (> 256 KiB: the reference's 2 MiB chunks and the tails
    above 256 KiB) with plants aimed at what is particular to that finder: candidates come from the row and tag of the
    5-byte hash (a 4-byte coincidence whose fifth byte differs is harmless unless the hashes collide), searches stop
    16 bytes before a block's end, the certificate of stage 1 works pair of blocks by pair of blocks."""
    b = bytearray(random_bytes(78, seed * 16 + variant, nbytes))
    block = 128 << 10
    sched = []
    start = 0
    while start < nbytes:
        end = min(start + block, nbytes)
        p = start + (1 if start == 0 else 0)
        while p < end - 16:
            sched.append(p)
            p += ((p - start) >> 8) + 1
        start = end
    rnd = xoshiro_bytes(seed * 104729 + variant, 64)
    pick = lambda k, lo, hi: lo + int.from_bytes(rnd[4 * k:4 * k + 4], "little") % max(1, hi - lo)
    nblk = (nbytes + block - 1) // block

    def in_block(bi, frac=0.6):
        xs = [x for x in sched if bi * block <= x < (bi + 1) * block]
        return xs[int(len(xs) * frac)]

    def copy(dst, src, ln):
        b[dst:dst + ln] = b[src:src + ln]

    mid_b = nblk // 2
    if variant == 0:      # 4 equal bytes at a searched position, the fifth differs: no candidate (unless the hashes collide)
        s = in_block(mid_b)
        src = pick(0, 64, s - 64)
        copy(s, src, 4)
        if b[s + 4] == b[src + 4]:
            b[s + 4] ^= 0x55
    elif variant == 1:    # 5 equal bytes at a searched position of a middle block: a match of length >= 5
        s = in_block(mid_b)
        copy(s, pick(0, 64, s - 64), 5)
    elif variant == 2:    # 40 bytes at a searched position of the last block
        s = in_block(nblk - 1, 0.4)
        copy(s, pick(0, 64, s - 64), 40)
    elif variant == 3:    # 8 bytes in block 0 and 8 bytes in a late block
        s0 = in_block(0, 0.8)
        copy(s0, pick(0, 16, s0 - 16), 8)
        s1 = in_block(min(nblk - 1, 5), 0.3)
        copy(s1, pick(1, 64, s1 - 64), 8)
    elif variant == 4:    # a repeat at a position the tree finders would search but the row-hash finder does not
        e = (mid_b + 1) * block      # (the last positions before end - 8 of a block: it stops at end - 16)
        copy(e - 12, pick(0, 64, e - 200), 9)
    elif variant == 5:    # a run of equal bytes over a searched position: the repcode (offset 1) test fires
        s = in_block(mid_b, 0.7)
        b[s - 1:s + 7] = bytes([b[s]]) * 8
    elif variant == 6:    # two searched positions of different pairs of blocks share 6 bytes
        a = in_block(1, 0.5)
        c = in_block(min(nblk - 1, 2 * ((nblk - 1) // 2)), 0.5)
        copy(c, a, 6)
    elif variant == 7:    # the earlier copy lies in the FIRST block, the search in the last pair of blocks
        s = in_block(nblk - 1, 0.2)
        copy(s, pick(0, 64, block - 64), 7)
    elif variant == 8:    # 5 bytes in the dense start of a block (every position is searched there)
        s = mid_b * block + 100
        copy(s, pick(0, 64, s - 64), 5)
    elif variant == 9:    # several: 5 bytes in a middle block, 7 bytes in the last pair with the copy in block 0
        s = in_block(mid_b, 0.35)
        copy(s, pick(0, 64, s - 64), 5)
        s2 = in_block(nblk - 1, 0.55)
        copy(s2, pick(1, 64, block - 64), 7)
    # (A 4-byte coincidence with a different fifth byte can never meet its twin in the finder: of two 5-byte strings
    # that share their first four bytes, (v << 24) * prime differs in the top 8 bits of the product by x * (prime & 0xFF),
    # and that prime's low byte is odd -- the 256 fifth bytes give 256 different rows.)
    return bytes(b)


def cross_shard_plan(world: int, n: int, frac: float = 0.3, seed: int = 4):
    """BASELINE configs[4]: `world` contiguous shards of n chunks; int(frac * world * n) of ALL chunks are exact
    copies of an "original" (never itself a copy) that lives on ANOTHER shard.  Returns the content index of every
    global chunk position (int64[world * n]; plan[g] == g for originals).  Distinct contents = world*n - copies."""
    N = world * n
    rng = np.random.default_rng(seed)
    plan = np.arange(N, dtype=np.int64)
    if world < 2:
        return plan
    is_dup = np.zeros(N, dtype=bool)
    is_dup[rng.choice(N, size=int(frac * N), replace=False)] = True
    orig = np.nonzero(~is_dup)[0]
    dups = np.nonzero(is_dup)[0]
    pick = orig[rng.integers(0, len(orig), size=len(dups))]
    same = (pick // n) == (dups // n)
    while same.any():                                            # same shard: draw again
        pick[same] = orig[rng.integers(0, len(orig), size=int(same.sum()))]
        same = (pick // n) == (dups // n)
    plan[dups] = pick
    return plan


def make(kind: str, seed: int, nbytes: int) -> bytes:
    if kind.startswith("plantedL"):
        return planted_large(seed, nbytes, int(kind[8:] or 0))
    if kind.startswith("planted"):
        return planted(seed, nbytes, int(kind[7:] or 0))
    return GENERATORS[kind](seed, nbytes)
