"""Fuzz of the hand-over / tail-certificate logic on the CPU: random chunks of the btlazy2 size class with a few
planted coincidences (3..20 bytes copied onto or beside searched positions, many of them in the dense start of
block 1) through the simulator build of the product's compressor (tests/sim/libsq_sim.so: one-stage and
stage 1 + stage 2) against the oracle.  usage: python tools/fuzz_planted_sim.py [seed] [cases]"""
import ctypes, os, sys, random, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT+'/tools'); sys.path.insert(0, ROOT+'/tests')
import corpus, oracle_lib
o=oracle_lib.load()
s=ctypes.CDLL(ROOT+'/tests/sim/libsq_sim.so')
s.sim_compress_two_stage.restype=ctypes.c_size_t
s.sim_compress_two_stage.argtypes=[ctypes.c_char_p,ctypes.c_size_t,ctypes.c_void_p,ctypes.c_size_t,ctypes.POINTER(ctypes.c_int)]
s.sim_compress_chunk.restype=ctypes.c_size_t
s.sim_compress_chunk.argtypes=[ctypes.c_char_p,ctypes.c_size_t,ctypes.c_void_p,ctypes.c_size_t,ctypes.c_int]
def two(d):
    cap=len(d)+(len(d)>>8)+1024; dst=ctypes.create_string_buffer(cap); dec=ctypes.c_int(0)
    n=s.sim_compress_two_stage(d,len(d),dst,cap,ctypes.byref(dec)); return dst.raw[:n]
def one(d):
    cap=len(d)+(len(d)>>8)+1024; dst=ctypes.create_string_buffer(cap)
    n=s.sim_compress_chunk(d,len(d),dst,cap,1); return dst.raw[:n]
rng=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
N=int(sys.argv[2]) if len(sys.argv)>2 else 60
bad=0; t0=time.time()
for it in range(N):
    n=rng.choice([262144,262144,200000,140000,131072+600,90000])
    b=bytearray(corpus.random_bytes(77, rng.randrange(1<<30), n))
    sched=corpus.lazy_schedule(n)
    # several planted coincidences: short ones (4 bytes: "maybe" that may come to nothing or to a 4-byte match),
    # longer ones, near block starts and elsewhere; some one byte off a searched position
    for k in range(rng.randrange(1,7)):
        t=rng.choice(sched[len(sched)//8:]) + rng.choice([0,0,0,1,-1])
        if rng.random()<0.5:
            second=[x for x in sched if x>=131072 and x<131072+4096]
            if second: t=rng.choice(second)+rng.choice([0,0,1])
        ln=rng.choice([3,4,4,4,5,6,8,20])
        if t<100 or t+ln>n: continue
        src=rng.randrange(2,t-ln-1)
        b[t:t+ln]=b[src:src+ln]
    d=bytes(b)
    exp=o.compress_chunk(d)
    if two(d)!=exp or one(d)!=exp:
        bad+=1; print("MISMATCH case",it,"n",n); open(f'/tmp/fuzz_bad_{it}.bin','wb').write(d)
print("cases",N,"bad",bad,"time",round(time.time()-t0,1))
