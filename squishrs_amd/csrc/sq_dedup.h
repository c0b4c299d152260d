// sq_dedup.h -- device dedup set + frame compaction kernels (gfx950).
//
// Replaces ChunkStore.primary_store: Arc<DashMap<ChunkHash, ()>> and the entry() test-and-set of
// /root/reference/src/util/chunk.rs:20-24,80-100, with the deterministic rule "the first
// occurrence in global chunk-index order is the Vacant one" (`-j 1` order).
//
// Data layout in HBM:
//   keys  : 16 B per chunk ever seen, indexed by GLOBAL chunk index g (append-only)
//   table : u32 per slot, open addressing / linear probing, value = (representative g) + 1,
//           0 = empty.  Slot index = low bits of the hash's first 8 bytes (already uniform).
// Two kernels per batch, no locks, no spinning on other threads:
//   insert : CAS(0 -> g+1); on collision compare the 16-byte keys; equal => atomicMin(g+1)
//   resolve: is_new[i] = (table[slot of key i] == g_i + 1)
// Exactly one successful CAS(0 -> x) happens per distinct new key, which is what the unique
// counter (ChunkStore::len, chunk.rs:116-118) accumulates.
#pragma once
#include "sq_common.h"

namespace sqd {

__device__ __forceinline__ bool key_eq(const u64* __restrict__ keys, u64 a, u64 klo, u64 khi) {
    return keys[2 * a] == klo && keys[2 * a + 1] == khi;
}

__global__ void __launch_bounds__(256)
k_dedup_insert(const u64* __restrict__ keys, u64 g_first, u64 n, u32* __restrict__ table, u64 mask,
               unsigned long long* __restrict__ unique_count) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 added = 0;
    if (i < n) {
        const u64 g = g_first + i;
        const u64 klo = keys[2 * g], khi = keys[2 * g + 1];
        const u32 mine = (u32)(g + 1);
        u64 slot = klo & mask;
        for (;;) {
            u32 cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0) {
                cur = atomicCAS(&table[slot], 0u, mine);
                if (cur == 0) {
                    added = 1;
                    break;
                }
            }
            if (key_eq(keys, (u64)cur - 1, klo, khi)) {
                atomicMin(&table[slot], mine);
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
    // one atomic per wave
    const u64 b = __ballot(added);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(unique_count, (unsigned long long)__popcll(b));
}

__global__ void __launch_bounds__(256)
k_dedup_resolve(const u64* __restrict__ keys, u64 g_first, u32 n, const u32* __restrict__ table,
                u64 mask, u8* __restrict__ is_new) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 g = g_first + i;
    const u64 klo = keys[2 * g], khi = keys[2 * g + 1];
    u64 slot = klo & mask;
    for (;;) {
        u32 cur = table[slot];
        if (cur == 0) { // cannot happen after insert; be safe
            is_new[i] = 0;
            return;
        }
        if (key_eq(keys, (u64)cur - 1, klo, khi)) {
            is_new[i] = (cur == (u32)(g + 1)) ? 1 : 0;
            return;
        }
        slot = (slot + 1) & mask;
    }
}

// Fused form of the compress stage (sq_api.hip compress_stage2): stage 1 ran on every chunk BEFORE the dedup decision;
// the frames it produced for chunks that turned out to be duplicates are dropped (ChunkStore::insert returns
// compressed_data: None for them, /root/reference/src/util/chunk.rs:84-87), and stage 1's count of decided chunks is
// corrected.
__global__ void __launch_bounds__(256)
k_drop_unselected(u32* __restrict__ frame_len, const u8* __restrict__ select, u32 n, u32* __restrict__ decided) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    bool drop = false;
    if (i < n && select[i] == 0 && frame_len[i] != 0) {
        frame_len[i] = 0;
        drop = true;
    }
    const u64 b = __ballot(drop);
    if ((threadIdx.x & 63) == 0 && b) atomicSub(decided, (u32)__popcll(b));
}

// Fused form with stage 2 launched before the dedup decision (sq_api.hip pack_enqueue): the chunks on stage 2's two work lists
// (work[0] / work[1] = their lengths, entries at work + 2 and work + 2 + n) were compressed whatever their flag; the frames of
// the duplicates among them are dropped here, BEFORE k_drop_unselected looks at what is left (stage 1's frames, whose count
// it corrects).  grid.y = the list.
__global__ void __launch_bounds__(256)
k_drop_listed(u32* __restrict__ frame_len, const u8* __restrict__ select, u32 n, const u32* __restrict__ work) {
    u32 const q = blockIdx.x * blockDim.x + threadIdx.x, pass = blockIdx.y;
    if (q < work[pass] && q < n) {
        u32 const idx = work[2 + (pass ? n : 0u) + q];
        if (idx < n && select[idx] == 0) frame_len[idx] = 0;
    }
}

// Sharded form with any assignment of chunks to ranks (sq_shard_submit_rest_idx): the dedup stage has resolved the flags
// of ALL chunks of the global order; this rank's chunk i is entry idx[i] of it.
__global__ void __launch_bounds__(256)
k_pick_flags(const u8* __restrict__ all_flags, const u32* __restrict__ idx, u32 n, u32 n_total, u8* __restrict__ out) {
    u32 const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        u32 const g = idx[i];
        out[i] = g < n_total ? all_flags[g] : 0;
    }
}

// Undo of one batch's k_dedup_insert (sq_pack_batch failing after the commit): empty every slot whose
// representative index is >= g_first.  Entries of earlier batches were placed before any of these existed, so
// their probe paths never run across the slots emptied here -- no tombstones are needed.
__global__ void __launch_bounds__(256)
k_dedup_rollback(u32* __restrict__ table, u64 nslots, u64 g_first) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i < nslots; i += stride) {
        const u32 v = table[i];
        if (v != 0 && (u64)v - 1 >= g_first) table[i] = 0;
    }
}

// Frames were written into per-chunk slots (slot_off[i], 16-byte aligned); pack them contiguously
// in index order.  One wave per frame, 16 B per lane per step.
__global__ void __launch_bounds__(256)
k_compact_frames(const u8* __restrict__ slots, const u64* __restrict__ slot_off,
                 const u32* __restrict__ frame_len, const u64* __restrict__ out_off, u32 n,
                 u8* __restrict__ out) {
    const u32 lane = threadIdx.x & 63;
    const u32 c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= n) return;
    const u32 L = frame_len[c];
    if (L == 0) return;
    const u8* src = slots + slot_off[c];
    u8* dst = out + out_off[c];
    u32 pos = lane * 16;
    for (; pos + 16 <= L; pos += 64 * 16) {
        uint4 v;
        __builtin_memcpy(&v, src + pos, 16);
        __builtin_memcpy(dst + pos, &v, 16);
    }
    // tail (< 16 bytes) belongs to exactly one lane
    if (pos < L) {
        for (u32 k = pos; k < L; k++) dst[k] = src[k];
    }
}

// Single-block exclusive scan of u32 lengths into u64 offsets (n+1 entries).  n <= a few 100k.
__global__ void __launch_bounds__(1024)
k_scan_lengths(const u32* __restrict__ len, u32 n, u64* __restrict__ off) {
    __shared__ u64 part[1024];
    const u32 t = threadIdx.x;
    const u32 per = (n + 1023) / 1024;
    const u32 lo = t * per, hi = (lo + per < n) ? lo + per : n;
    u64 s = 0;
    for (u32 i = lo; i < hi; i++) s += len[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (u32 d = 1; d < 1024; d <<= 1) {
        u64 v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 base = (t == 0) ? 0 : part[t - 1];
    for (u32 i = lo; i < hi; i++) {
        off[i] = base;
        base += len[i];
    }
    if (t == 1023) off[n] = part[1023];
}

// HBM read-bandwidth calibration: sums 16 B/lane streams (roofline reference for bench.py).
__global__ void __launch_bounds__(256)
k_read_bw(const uint4* __restrict__ p, u64 n16, u32* __restrict__ sink) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (; i < n16; i += stride) {
        // (non-temporal, as the hash kernel's and stage 1's streams are: the roof these are compared with)
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        v4u const v = __builtin_nontemporal_load((const __attribute__((address_space(1))) v4u*)(p + i));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc; // never true in practice; keeps the loads alive
}

// Random-access calibration: every lane reads `per_lane` pieces of REQ x 16 bytes at pseudo-random REQ x 16-byte aligned
// places of the buffer (an LCG per lane; DEPTH independent pieces in flight per lane, like the candidate evaluation of
// the row-hash search batches).  The rate of such requests, not the streaming bandwidth, is what the compressor runs
// against on compressible data; bench.py reports it beside the configs[3] leg, tools/gather_sweep.py
// sweeps request size x depth x buffer size.  The index is a multiply-shift of the generator's upper bits -- NOT a
// 64-bit modulo, which is a ~40-instruction software division per request and was what the first version of this
// probe measured (the same 50 G/s whether or not the buffer fitted the Infinity Cache).
template <int REQ, int DEPTH>
__global__ void __launch_bounds__(256)
k_gather_bw(const uint4* __restrict__ p, u64 n16, u32 per_lane, u32* __restrict__ sink) {
    u64 x = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    u32 const nPieces = (u32)(n16 / REQ); // (< 2^32: buffers up to 64 GiB)
    u32 acc = 0;
    for (u32 i = 0; i < per_lane; i += DEPTH) {
        uint4 v[DEPTH][REQ];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            u64 const at = (u64)__umulhi((u32)(x >> 32), nPieces) * REQ;
#pragma unroll
            for (int r = 0; r < REQ; r++) v[k][r] = p[at + r];
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++)
#pragma unroll
            for (int r = 0; r < REQ; r++) acc ^= v[k][r].x ^ v[k][r].y ^ v[k][r].z ^ v[k][r].w;
    }
    if (acc == 0x12345678u) *sink = acc; // never true in practice; keeps the loads alive
}

} // namespace sqd
