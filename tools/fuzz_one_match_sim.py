"""Fuzz of stage 1's second pass ("one short match", fastpath_blocks) on the CPU: random chunks of the btlazy2 size class
with ONE planted coincidence on a searched position (4..12 bytes, sometimes extended backwards, sometimes with what the
second pass must refuse beside it: a second copy, a repeat behind the match, a repeat at the depth searches, a member of the
search's bucket that shares its first byte) through the simulator build of the product's two kernels against the oracle.
Prints how many blocks the second pass decided.  usage: python tools/fuzz_one_match_sim.py [seed] [cases]"""
import ctypes, os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/tools'); sys.path.insert(0, ROOT + '/tests')
import corpus, oracle_lib
o = oracle_lib.load()
_sim = None


def sim():
    """the simulator build of the product's kernels (tests/test_sim_cpu.py builds it), loaded at first use: make_case is
    also imported by tests/test_one_match_gpu.py on the GPU box, which needs no simulator"""
    global _sim
    if _sim is None:
        _sim = ctypes.CDLL(ROOT + '/tests/sim/libsq_sim.so')
        _sim.sim_compress_two_stage.restype = ctypes.c_size_t
        _sim.sim_compress_two_stage.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.POINTER(ctypes.c_int)]
    return _sim


def two(d):
    s = sim()
    cap = len(d) + (len(d) >> 8) + 1024
    dst = ctypes.create_string_buffer(cap)
    dec = ctypes.c_int(0)
    n = s.sim_compress_two_stage(d, len(d), dst, cap, ctypes.byref(dec))
    return dst.raw[:n], dec.value


def hash4(v, bits):
    return ((v * 2654435761) & 0xFFFFFFFF) >> (32 - bits)


def make_case(rng):
    """one chunk: (bytes, kind) -- or (None, kind) when the draw does not fit"""
    n = rng.choice([262144, 262144, 262144, 200000, 140000, 131072 + 45000, 131072, 90000])
    b = bytearray(corpus.random_bytes(78, rng.randrange(1 << 30), n))
    sched = corpus.lazy_schedule(n)
    t = rng.choice(sched[len(sched) // 16:])
    ln = rng.choice([4, 4, 4, 4, 4, 4, 5, 6, 8, 12])
    kind = rng.choice(["plain", "plain", "plain", "back", "second", "behind", "behind4", "behind4", "depth", "bucket", "rep1", "late"])
    if t < 200 or t + ln + 16 > n:
        return None, kind
    src = rng.randrange(2, t - ln - 8)
    b[t:t + ln] = b[src:src + ln]
    if kind == "back":       # the match extends backwards
        k = rng.randrange(1, 6)
        if src - k > 2 and t - k > 2:
            b[t - k:t] = b[src - k:src]
    elif kind == "second":   # a second earlier copy of the 4 bytes
        s2 = rng.randrange(2, t - 8)
        b[s2:s2 + 4] = b[t:t + 4]
    elif kind == "behind":   # something to find right behind the match (the new schedule is dense there)
        u = t + ln + rng.randrange(0, 40)
        s3 = rng.randrange(2, t - 8)
        if u + 6 < n:
            b[u:u + 5] = b[s3:s3 + 5]
    elif kind == "behind4":  # ONE more short match on the new schedule's dense start (stage 1's third pass), sometimes a third
        u = t + ln + rng.randrange(0, 200)
        s3 = rng.randrange(2, t - 8)
        if u + 8 < n:
            b[u:u + 4] = b[s3:s3 + 4]
            if rng.random() < 0.25:
                u2 = u + 4 + rng.randrange(0, 100)
                s7 = rng.randrange(2, t - 8)
                if u2 + 8 < n:
                    b[u2:u2 + 4] = b[s7:s7 + 4]
    elif kind == "depth":    # the depth searches at t + 1 / t + 2 find something
        s4 = rng.randrange(2, t - 16)
        d = rng.choice([1, 2])
        b[s4:s4 + 6] = b[t + d:t + d + 6]
    elif kind == "bucket":   # an earlier position in t's bucket that shares its first byte only
        v = int.from_bytes(b[t:t + 4], "little")
        w_ = 15
        while (1 << w_) < n:
            w_ += 1
        bits = 17 if w_ == 17 else w_ + 1   # get_cparams: hashLog of the btlazy2 class
        for _ in range(400000):
            w = (v & 0xFF) | (rng.randrange(1 << 24) << 8)
            if w != v and hash4(w, bits) == hash4(v, bits):
                s5 = rng.randrange(2, t - 8)
                if abs(s5 - src) > 8:
                    b[s5:s5 + 4] = w.to_bytes(4, "little")
                break
    elif kind == "rep1":     # a run of equal bytes at the match (repcode 1 fires)
        b[t:t + 6] = bytes([b[t]]) * 6
    elif kind == "late":     # a second coincidence far behind the match, on the NEW schedule
        lim = min((t // 131072 + 1) * 131072, n)
        if t + ln < lim - 16:
            u = rng.randrange(t + ln, lim - 16)
            s6 = rng.randrange(2, t - 8)
            b[u:u + 4] = b[s6:s6 + 4]
    return bytes(b), kind


def run(seed, cases, verbose=True):
    rng = random.Random(seed)
    bad = 0
    decided = 0
    per = {}
    t0 = time.time()
    s = sim()
    before = s.sim_one_match_blocks()
    for it in range(cases):
        d, kind = make_case(rng)
        if d is None:
            continue
        exp = o.compress_chunk(d)
        got, dec = two(d)
        decided += dec
        c = per.setdefault(kind, [0, 0])
        c[0] += 1
        c[1] += dec
        if got != exp:
            bad += 1
            print("MISMATCH case", it, "n", len(d), "kind", kind)
            open(f'/tmp/fuzz1m_bad_{seed}_{it}.bin', 'wb').write(d)
    took = s.sim_one_match_blocks() - before
    if verbose:
        print("cases", cases, "bad", bad, "decided by stage 1", decided, "blocks decided by the second pass", took, "time", round(time.time() - t0, 1))
        print("  per kind (cases, decided by stage 1):", {k: tuple(v) for k, v in sorted(per.items())})
    return bad, took


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 200)
