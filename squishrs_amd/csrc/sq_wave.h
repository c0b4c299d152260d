// sq_wave.h -- wavefront-level primitives for the one-wavefront-per-chunk zstd compressor.
//
// Execution model: a workgroup is ONE wave of 64 lanes and owns one chunk.  Control flow is
// wave-uniform: every lane executes the same "scalar" program on the same values (uniform loads
// broadcast one address; single-lane stores publish), and the 64 lanes are used as a vector unit
// by the primitives in this file: match-length extension, 64-entry tag-row compare, batched hash
// insertion with in-wave conflict ranking, LDS histograms, coalesced copies/fills; the prefix scan
// (wv_excl_scan) that places every lane's bits in a zstd bit stream (sq_zstd_entropy.h, wvb_tile)
// and the candidate pairs of a row-hash search batch (sq_zstd_rowbatch.h).
//
// The same source also builds with -DSQ_SIM for the host: every primitive then has a plain
// scalar body with identical semantics.  That build exists ONLY for tests/ (a CPU-side simulator
// to debug the uniform control logic where there is no GPU); it is never linked into
// libsquish_gpu.so and is not a fallback.
#pragma once
#include "sq_common.h"

#if SQ_ON_DEVICE
#define SQ_LDS __shared__
#define SQ_NOINLINE __device__ __noinline__
SQ_DEV u32 wv_lane() { return __lane_id(); }
SQ_DEV u32 wv_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
SQ_DEV u64 wv_uni64(u64 v) {
    u32 lo = wv_uni((u32)v), hi = wv_uni((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
SQ_DEV u64 wv_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
SQ_DEV u32 wv_readlane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
SQ_DEV u64 wv_readlane64(u64 v, u32 l) {
    return ((u64)wv_readlane((u32)(v >> 32), l) << 32) | wv_readlane((u32)v, l);
}
// Order LDS / global accesses made by different lanes of this wave.  One wave issues its LDS
// and its vector-memory instructions in program order, so all that is needed is to stop the
// COMPILER from moving accesses across this point; a __builtin_amdgcn_fence (even at wavefront
// scope) makes hipcc drain vmcnt/lgkmcnt to 0, which serialises every pipelined load behind it.
SQ_DEV void wv_fence() { asm volatile("" ::: "memory"); }
// Full drain: every earlier vector-memory / LDS operation of this wave has COMPLETED.  Needed
// where a later instruction reads what an earlier instruction (possibly of another lane, possibly
// a flat_* one from a serial region) wrote: gfx950 issues a wave's memory instructions in order
// but lets a load be performed before an older store has landed.
SQ_DEV void wv_mem_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// Drain + drop this CU's vector-L1 lines (buffer_inv sc1): used once per chunk / block so no line
// cached while the workspace held another chunk's tables can be hit again.
SQ_DEV void wv_mem_sync_inv() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#else
#define SQ_LDS static thread_local
#define SQ_NOINLINE static __attribute__((noinline))
SQ_DEV u32 wv_uni(u32 v) { return v; }
SQ_DEV u64 wv_uni64(u64 v) { return v; }
SQ_DEV void wv_fence() {}
#ifdef SQ_SIM_TRIPS
SQ_DEV void wv_mem_sync();
#else
SQ_DEV void wv_mem_sync() {}
#endif
SQ_DEV void wv_mem_sync_inv() {}
#endif

// A serial region: plain scalar code run by lane 0 only while the other lanes idle at the
// reconvergence point.  No wave primitive may be called inside.  Results leave through LDS /
// global memory (followed by wv_fence()) or through wv_uni() of a lane-0 register.
#if SQ_ON_DEVICE
#define WV_SERIAL if (wv_lane() == 0)
#else
#define WV_SERIAL if (1)
#endif

// (simulator build with -DSQ_SIM_TRIPS: count the dependent trips to memory -- a uniform load the wave waits for, a drain --
// by the phase the match finder says it is in: tools/sim_trips.py)
#if !SQ_ON_DEVICE && defined(SQ_SIM_TRIPS)
static thread_local unsigned long long g_sim_trips[16];
static thread_local int g_sim_phase;
struct SimPhase {
    int old;
    explicit SimPhase(int p) : old(g_sim_phase) { g_sim_phase = p; }
    ~SimPhase() { g_sim_phase = old; }
};
#define SQ_TRIP() (g_sim_trips[g_sim_phase]++)
#define SQ_TRIP_EXT() (g_sim_trips[7]++) // (a tree step's second trip: the strings agree in the eight bytes read with the node)
#define SQ_TRIP_PHASE(p) SimPhase sq_trip_phase_##p(p)
SQ_DEV void wv_mem_sync() { g_sim_trips[8 + (g_sim_phase & 7)]++; } // (drains: counted apart, 8 + phase)
#else
#define SQ_TRIP() ((void)0)
#define SQ_TRIP_EXT() ((void)0)
#define SQ_TRIP_PHASE(p) ((void)0)
#endif

// ---- uniform global memory access: all lanes read one address; lane 0 alone writes ------------
// The loads are relaxed agent-scope ATOMIC loads on purpose: a plain load from a wave-uniform
// address is turned into s_load_dword by hipcc, and the scalar data cache is not coherent with
// this wave's own vector stores (observed: ~1 stale table word per 10^4 chunks -> impossible
// match indices).  Atomic loads stay on the vector path (global_load ... sc1).
SQ_DEV u32 gld32(const u32* p) {
#if SQ_ON_DEVICE
    return wv_uni(__hip_atomic_load((const SQ_G u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
    SQ_TRIP();
    return *p;
#endif
}
SQ_DEV u32 gld8(const u8* p) {
#if SQ_ON_DEVICE
    return wv_uni((u32)__hip_atomic_load((const SQ_G u8*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
    return *p;
#endif
}
SQ_DEV u32 gld16(const u16* p) {
#if SQ_ON_DEVICE
    return wv_uni((u32)__hip_atomic_load((const SQ_G u16*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#else
    return *p;
#endif
}
SQ_DEV void gst32(u32* p, u32 v) {
#if SQ_ON_DEVICE
    if (wv_lane() == 0) sq_gst32(p, v);
#else
    *p = v;
#endif
}
SQ_DEV void gst16(u16* p, u32 v) {
#if SQ_ON_DEVICE
    if (wv_lane() == 0) *(SQ_G u16*)p = (u16)v;
#else
    *p = (u16)v;
#endif
}
SQ_DEV void gst8(u8* p, u32 v) {
#if SQ_ON_DEVICE
    if (wv_lane() == 0) sq_gwr8(p, v);
#else
    *p = (u8)v;
#endif
}
// uniform unaligned source reads (chunk bytes are never written by the kernel)
SQ_DEV u32 srd32(const u8* p) { SQ_TRIP(); return wv_uni(sq_grd32(p)); }
SQ_DEV u64 srd64(const u8* p) { SQ_TRIP(); return wv_uni64(sq_grd64(p)); }
SQ_DEV u32 srd16(const u8* p) { return wv_uni(sq_grd16(p)); }
SQ_DEV u32 srd8(const u8* p) { return wv_uni(sq_grd8(p)); }

// ---- wv_count: length of the common prefix of a[0..maxLen) and b[..] (ZSTD_count) -------------
SQ_DEV u32 wv_count(const u8* a, const u8* b, u32 maxLen) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    // fast uniform probe of the first 8 bytes: most candidates die here
    if (maxLen >= 8) {
        u64 x = srd64(a) ^ srd64(b);
        if (x) return (u32)(__builtin_ctzll(x) >> 3);
    }
    for (u32 base = 0; base < maxLen; base += 512) {
        const u32 o = base + lane * 8;
        u64 x;
        if (o + 8 <= maxLen) {
            x = sq_grd64(a + o) ^ sq_grd64(b + o);
        } else if (o < maxLen) {
            const u32 valid = maxLen - o; // 1..7
            x = 0;
            for (u32 k = 0; k < valid; k++) x |= (u64)(u8)(sq_grd8(a + o + k) ^ sq_grd8(b + o + k)) << (8 * k);
            x |= 1ull << (8 * valid); // stop at the limit
        } else {
            x = 1; // beyond the limit
        }
        const u64 m = wv_ballot(x != 0);
        if (m) {
            const u32 first = (u32)__builtin_ctzll(m);
            const u64 xv = wv_readlane64(x, first);
            u32 r = base + first * 8 + (u32)(__builtin_ctzll(xv) >> 3);
            return r < maxLen ? r : maxLen;
        }
    }
    return maxLen;
#else
    u32 n = 0;
    while (n < maxLen && a[n] == b[n]) n++;
    SQ_TRIP();
    if (maxLen >= 8 && n >= 8)
        for (u32 base = 0; base <= n && base < maxLen; base += 512) SQ_TRIP();
    return n;
#endif
}

// uniform 8-byte load of two adjacent table words (p 8-byte aligned)
SQ_DEV void gld_pair(const u32* p, u32* a, u32* b) {
#if SQ_ON_DEVICE
    u64 const v = __hip_atomic_load((const SQ_G u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *a = wv_uni((u32)v);
    *b = wv_uni((u32)(v >> 32));
#else
    SQ_TRIP();
    *a = p[0];
    *b = p[1];
#endif
}

// One step of a binary-tree walk: the node's two child words (p, 8-byte aligned) AND the first eight bytes of the two strings
// it compares, issued together -- one trip to memory per step instead of two (the child words used to be waited for before
// the bytes were even asked for: profiles/round4f_tree_step_loads.txt).  have8: the strings have eight bytes left.
SQ_DEV void wv_tree_step_loads(const u32* p, const u8* a, const u8* b, bool have8, u32* c0, u32* c1, u64* wa, u64* wb) {
#if SQ_ON_DEVICE
    u64 const v = __hip_atomic_load((const SQ_G u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 x = 0, y = 0;
    if (have8) {
        x = sq_grd64(a);
        y = sq_grd64(b);
    }
    // (all three are in flight before the first of them is read back)
    asm volatile("" : "+v"(x), "+v"(y));
    *c0 = wv_uni((u32)v);
    *c1 = wv_uni((u32)(v >> 32));
    *wa = wv_uni64(x);
    *wb = wv_uni64(y);
#else
    SQ_TRIP();
    *c0 = p[0];
    *c1 = p[1];
    *wa = *wb = 0;
    (void)a; (void)b; (void)have8;
#endif
}

// wv_count plus the ordering of the first differing byte: *bLess = (b[len] < a[len]) when
// len < maxLen (undefined otherwise).  Saves the separate byte-compare round trip of the tree walks.
// (pre: the strings' first eight bytes are in wa0 / wb0 already -- wv_tree_step_loads -- when maxLen >= 8)
SQ_DEV u32 wv_count_cmp(const u8* a, const u8* b, u32 maxLen, u32* bLess, bool pre = false, u64 wa0 = 0, u64 wb0 = 0) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    if (maxLen >= 8) {
        u64 const wa = pre ? wa0 : srd64(a), wb = pre ? wb0 : srd64(b);
        u64 const x = wa ^ wb;
        if (x) {
            u32 const k = (u32)(__builtin_ctzll(x) >> 3);
            *bLess = (u32)((wb >> (8 * k)) & 0xFF) < (u32)((wa >> (8 * k)) & 0xFF); // (32-bit: a scalar compare)
            return k;
        }
    }
    for (u32 base = 0; base < maxLen; base += 512) {
        const u32 o = base + lane * 8;
        u64 x, wa = 0, wb = 0;
        if (o + 8 <= maxLen) {
            wa = sq_grd64(a + o);
            wb = sq_grd64(b + o);
            x = wa ^ wb;
        } else if (o < maxLen) {
            const u32 valid = maxLen - o; // 1..7
            for (u32 k = 0; k < valid; k++) {
                wa |= (u64)sq_grd8(a + o + k) << (8 * k);
                wb |= (u64)sq_grd8(b + o + k) << (8 * k);
            }
            x = (wa ^ wb) | (1ull << (8 * valid)); // stop at the limit
        } else {
            x = 1; // beyond the limit
        }
        const u64 m = wv_ballot(x != 0);
        if (m) {
            const u32 first = (u32)__builtin_ctzll(m);
            const u64 xv = wv_readlane64(x, first);
            const u32 k = (u32)(__builtin_ctzll(xv) >> 3);
            u32 r = base + first * 8 + k;
            if (r >= maxLen) return maxLen;
            const u64 av = wv_readlane64(wa, first), bv = wv_readlane64(wb, first);
            *bLess = (u32)((bv >> (8 * k)) & 0xFF) < (u32)((av >> (8 * k)) & 0xFF);
            return r;
        }
    }
    return maxLen;
#else
    u32 n = 0;
    while (n < maxLen && a[n] == b[n]) n++;
    if (n < maxLen) *bLess = b[n] < a[n];
    if (!pre && maxLen >= 8) SQ_TRIP();
    if (n >= 8 || maxLen < 8)
        for (u32 base = 0; base <= n && base < maxLen; base += 512) SQ_TRIP_EXT();
    (void)wa0; (void)wb0;
    return n;
#endif
}

// ---- coalesced fill / copy / compare ---------------------------------------------------------
SQ_DEV void wv_fill32(u32* dst, u32 v, size_t nwords) { // dst 16-byte aligned, nwords % 4 == 0
#if SQ_ON_DEVICE
    sq_v4u q = (sq_v4u){v, v, v, v};
    u8* d = (u8*)dst;
    const size_t n16 = nwords / 4;
    for (size_t i = wv_lane(); i < n16; i += 64) sq_gwr128_s(d + 16 * i, q);
    wv_mem_sync();
#else
    for (size_t i = 0; i < nwords; i++) dst[i] = v;
#endif
}
SQ_DEV void wv_memcpy(u8* dst, const u8* src, u32 n) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    u32 pos = lane * 16;
    // 8 KiB per iteration: eight independent 16 B/lane loads in flight before the stores
    for (; pos + 7 * 1024 + 16 <= n; pos += 8192) {
        sq_v4u v[8];
#pragma unroll
        for (u32 k = 0; k < 8; k++) v[k] = sq_grd128_s(src + pos + 1024 * k);
#pragma unroll
        for (u32 k = 0; k < 8; k++) sq_gwr128_s(dst + pos + 1024 * k, v[k]);
    }
    for (; pos + 16 <= n; pos += 1024) sq_gwr128_s(dst + pos, sq_grd128_s(src + pos));
    if (pos < n)
        for (u32 k = pos; k < n; k++) sq_gwr8(dst + k, sq_grd8(src + k)); // the single partial lane
    wv_mem_sync();
#else
    memcpy(dst, src, n);
#endif
}
// 1 if all n bytes equal src[0] (ZSTD_isRLE); n >= 1
SQ_DEV u32 wv_is_rle(const u8* src, u32 n) {
#if SQ_ON_DEVICE
    const u32 b0 = srd8(src);
    const u64 pat = 0x0101010101010101ull * b0;
    bool bad = false;
    u32 pos = wv_lane() * 8;
    // like ZSTD_isRLE, give up at the first difference: look at the first 512 bytes alone, then
    // 2 KiB per step
    if (pos + 8 <= n) {
        bad = sq_grd64(src + pos) != pat;
        pos += 512;
    }
    if (wv_ballot(bad)) return 0;
    while (wv_ballot(pos + 8 <= n)) {
#pragma unroll
        for (u32 k = 0; k < 4; k++)
            if (pos + 8 <= n) {
                bad |= (sq_grd64(src + pos) != pat);
                pos += 512;
            }
        if (wv_ballot(bad)) return 0;
    }
    if (pos < n)
        for (u32 k = pos; k < n; k++) bad |= (sq_grd8(src + k) != b0);
    return wv_ballot(bad) == 0;
#else
    for (u32 i = 1; i < n; i++)
        if (src[i] != src[0]) return 0;
    return 1;
#endif
}

// ---- byte histogram into count[0..255] (u32, LDS) ----------------------------------------------
// Returns nothing; count must be zeroed by the caller via wv_zero_lds.
SQ_DEV void wv_zero_lds(u32* p, u32 nwords) {
#if SQ_ON_DEVICE
    SQ_L u32* const lp = (SQ_L u32*)p;
    for (u32 i = wv_lane(); i < nwords; i += 64) lp[i] = 0;
    wv_fence();
#else
    for (u32 i = 0; i < nwords; i++) p[i] = 0;
#endif
}
SQ_DEV void wv_hist_bytes(u32* count /*LDS*/, const u8* src, u32 n) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    SQ_L u32* const lc = (SQ_L u32*)count;
    u32 pos = lane * 16;
    // 16 B per lane per load; the next two loads are issued before the current two are counted
    auto count16 = [&](sq_v4u q) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 w = q[k];
            __hip_atomic_fetch_add(&lc[w & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[(w >> 8) & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[(w >> 16) & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[w >> 24], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if (pos + 1024 + 16 <= n) {
        sq_v4u a = sq_grd128_s(src + pos), b = sq_grd128_s(src + pos + 1024);
        pos += 2048;
        for (; pos + 1024 + 16 <= n; pos += 2048) {
            sq_v4u const na = sq_grd128_s(src + pos), nb = sq_grd128_s(src + pos + 1024);
            count16(a);
            count16(b);
            a = na;
            b = nb;
        }
        count16(a);
        count16(b);
    }
    for (; pos + 16 <= n; pos += 1024) {
        sq_v4u a = sq_grd128_s(src + pos);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 w = a[k];
            __hip_atomic_fetch_add(&lc[w & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[(w >> 8) & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[(w >> 16) & 0xFF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lc[w >> 24], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (pos < n)
        for (u32 k = pos; k < n; k++) __hip_atomic_fetch_add(&lc[sq_grd8(src + k)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    wv_fence();
#else
    for (u32 i = 0; i < n; i++) count[src[i]]++;
#endif
}
// max over p[0..n) and index of the last nonzero entry (n <= 256); uniform results
SQ_DEV void wv_max_and_last(const u32* p /*LDS*/, u32 n, u32* maxOut, u32* lastNonZero) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    u32 mx = 0, last = 0;
    const SQ_L u32* const lp = (const SQ_L u32*)p;
    for (u32 i = lane; i < n; i += 64) {
        u32 v = lp[i];
        mx = v > mx ? v : mx;
        if (v) last = i;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        u32 o = (u32)__shfl_xor((int)mx, m, 64);
        mx = o > mx ? o : mx;
        u32 l2 = (u32)__shfl_xor((int)last, m, 64);
        last = l2 > last ? l2 : last;
    }
    *maxOut = wv_uni(mx);
    *lastNonZero = wv_uni(last);
#else
    u32 mx = 0, last = 0;
    for (u32 i = 0; i < n; i++) {
        if (p[i] > mx) mx = p[i];
        if (p[i]) last = i;
    }
    *maxOut = mx;
    *lastNonZero = last;
#endif
}

// ---- in-wave grouping of equal keys -------------------------------------------------------------
// Each active lane holds `key`.  Returns for the calling lane: rank = number of lower active lanes
// with the same key, gsize = group size, prevLane = closest lower lane of the group (or 64).
// `scratch` = 2048-byte LDS table private to the wave.
#if SQ_ON_DEVICE
struct WvGroup {
    u32 rank, gsize, prevLane;
};
SQ_DEV WvGroup wv_group(u32 key, bool active, u8* scratch /*LDS 2048 B*/) {
    const u32 lane = wv_lane();
    WvGroup g;
    g.rank = 0;
    g.gsize = 1;
    g.prevLane = 64;
    const u32 slot = (key * 2654435761u) >> 21; // 11 bits
    SQ_L u8* const ls = (SQ_L u8*)scratch;
    if (active) ls[slot] = (u8)lane;
    wv_fence();
    bool loser = active && (ls[slot] != (u8)lane);
    u64 todo = wv_ballot(loser);
    const u64 act = wv_ballot(active);
    const u64 below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    while (todo) {
        const u32 l = (u32)__builtin_ctzll(todo);
        const u32 k = wv_readlane(key, l);
        const u64 m = wv_ballot(active && key == k) & act;
        if (active && key == k) {
            const u64 lower = m & below;
            g.rank = (u32)__popcll(lower);
            g.gsize = (u32)__popcll(m);
            g.prevLane = lower ? (63u - (u32)__builtin_clzll(lower)) : 64u;
        }
        todo &= ~m;
    }
    wv_fence();
    return g;
}
#endif

// exclusive prefix sum over lanes of a u32 (device only); returns (exclusive, total)
#if SQ_ON_DEVICE
SQ_DEV u32 wv_excl_scan(u32 v, u32* total) {
    // Hillis-Steele inside every row of 16 lanes with DPP row shifts (zero fill), then the two row broadcasts of
    // GCN / CDNA: lane 15 of a row into the next row (rows 1 and 3), lane 31 into rows 2 and 3.  Six v_add_u32 with
    // a DPP operand instead of six trips through the LDS crossbar (ds_bpermute).
    u32 x = v;
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    *total = wv_readlane(x, 63);
    return x - v;
}
#endif
