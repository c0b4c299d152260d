// sq_xxh3.h -- batched XXH3-128 (seed 0, default secret) for gfx950: k_xxh3_128_rows (16 lanes per chunk,
// the default) and k_xxh3_128_batch (one wavefront per chunk, the first mapping).
//
// Replaces hash_chunk, /root/reference/src/util/chunk.rs:46-49
//     xxh3_128(chunk).to_le_bytes()          (xxhash-rust 0.8.15, Cargo.lock:1593-1596)
//
// Layout of the long path (len > 240): one XXH3 "block" is 1 KiB = 16 stripes of 64 B.  A wave
// reads a block with ONE coalesced 16 B/lane load (global_load_dwordx4, 1 KiB per instruction):
// lane l holds stripe s = l>>2, accumulator words i0 = 2*(l&3) and i0+1.  Per-stripe
// contributions are wrapping adds, so every lane computes its two 64-bit contributions
// independently and a 4-step butterfly (xor 4, 8, 16, 32) sums the 16 stripes; only the per-block
// scramble is serial.  The secret words each lane needs are loop constants held in registers.
// HBM traffic = the chunk bytes once + 16 B out: the kernel is HBM-bound by construction.
#pragma once
#include "sq_common.h"

namespace sqx {

__device__ __constant__ const u8 kSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

constexpr u32 P32_1 = 0x9E3779B1U, P32_2 = 0x85EBCA77U, P32_3 = 0xC2B2AE3DU;
constexpr u64 P64_1 = 0x9E3779B185EBCA87ULL, P64_2 = 0xC2B2AE3D27D4EB4FULL,
              P64_3 = 0x165667B19E3779F9ULL, P64_4 = 0x85EBCA77C2B2AE63ULL,
              P64_5 = 0x27D4EB2F165667C5ULL, PMX1 = 0x165667919E3779F9ULL,
              PMX2 = 0x9FB21C651E98DF25ULL;

__device__ __forceinline__ u64 sec64(u32 o) { return sq_rd64(kSecret + o); }
__device__ __forceinline__ u32 sec32(u32 o) { return sq_rd32(kSecret + o); }
__device__ __forceinline__ u64 avalanche3(u64 h) {
    h ^= h >> 37;
    h *= PMX1;
    h ^= h >> 32;
    return h;
}
__device__ __forceinline__ u64 avalanche64(u64 h) {
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    h ^= h >> 32;
    return h;
}
__device__ __forceinline__ u64 mul128_fold64(u64 a, u64 b) { return (a * b) ^ __umul64hi(a, b); }
__device__ __forceinline__ u64 mix16B(const u8* in, u32 so) {
    return mul128_fold64(sq_rd64(in) ^ sec64(so), sq_rd64(in + 8) ^ sec64(so + 8));
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int m) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__shfl_xor((int)lo, m, 64);
    hi = (u32)__shfl_xor((int)hi, m, 64);
    return ((u64)hi << 32) | lo;
}

struct H128 {
    u64 lo, hi;
};

// ---- inputs of 0..240 bytes: scalar code, executed by lane 0's values (all lanes redundantly)
__device__ inline H128 hash_short(const u8* in, u32 len) {
    H128 r;
    if (len == 0) {
        r.lo = avalanche64(sec64(64) ^ sec64(72));
        r.hi = avalanche64(sec64(80) ^ sec64(88));
    } else if (len <= 3) {
        u8 c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        u32 cl = ((u32)c1 << 16) | ((u32)c2 << 24) | (u32)c3 | (len << 8);
        u32 sw = __builtin_bswap32(cl);
        u32 ch = (sw << 13) | (sw >> 19);
        r.lo = avalanche64((u64)cl ^ (u64)(sec32(0) ^ sec32(4)));
        r.hi = avalanche64((u64)ch ^ (u64)(sec32(8) ^ sec32(12)));
    } else if (len <= 8) {
        u32 ilo = sq_rd32(in), ihi = sq_rd32(in + len - 4);
        u64 keyed = (ilo + ((u64)ihi << 32)) ^ (sec64(16) ^ sec64(24));
        u64 m = P64_1 + ((u64)len << 2);
        u64 lo = keyed * m, hi = __umul64hi(keyed, m);
        hi += lo << 1;
        lo ^= hi >> 3;
        lo ^= lo >> 35;
        lo *= PMX2;
        lo ^= lo >> 28;
        r.lo = lo;
        r.hi = avalanche3(hi);
    } else if (len <= 16) {
        u64 bl = sec64(32) ^ sec64(40), bh = sec64(48) ^ sec64(56);
        u64 ilo = sq_rd64(in), ihi = sq_rd64(in + len - 8);
        u64 x = ilo ^ ihi ^ bl;
        u64 lo = x * P64_1, hi = __umul64hi(x, P64_1);
        lo += (u64)(len - 1) << 54;
        ihi ^= bh;
        hi += ihi + (u64)(u32)ihi * (u64)(P32_2 - 1);
        lo ^= __builtin_bswap64(hi);
        u64 hlo = lo * P64_2, hhi = __umul64hi(lo, P64_2);
        hhi += hi * P64_2;
        r.lo = avalanche3(hlo);
        r.hi = avalanche3(hhi);
    } else {
        u64 alo = (u64)len * P64_1, ahi = 0;
        auto mix32B = [&](const u8* i1, const u8* i2, u32 so) {
            alo += mix16B(i1, so);
            alo ^= sq_rd64(i2) + sq_rd64(i2 + 8);
            ahi += mix16B(i2, so + 16);
            ahi ^= sq_rd64(i1) + sq_rd64(i1 + 8);
        };
        if (len <= 128) {
            for (int i = (int)((len - 1) / 32); i >= 0; i--)
                mix32B(in + 16 * i, in + len - 16 * (i + 1), 32 * i);
        } else {
            for (u32 i = 32; i < 160; i += 32) mix32B(in + i - 32, in + i - 16, i - 32);
            alo = avalanche3(alo);
            ahi = avalanche3(ahi);
            for (u32 i = 160; i <= len; i += 32) mix32B(in + i - 32, in + i - 16, 3 + i - 160);
            mix32B(in + len - 16, in + len - 32, 136 - 17 - 16);
        }
        r.lo = avalanche3(alo + ahi);
        r.hi = (u64)0 - avalanche3(alo * P64_1 + ahi * P64_4 + (u64)len * P64_2);
    }
    return r;
}

__device__ __forceinline__ void contrib(const uint4& v, u64 sk0, u64 sk1, u64& c0, u64& c1) {
    u64 d0 = ((u64)v.y << 32) | v.x, d1 = ((u64)v.w << 32) | v.z;
    u64 k0 = d0 ^ sk0, k1 = d1 ^ sk1;
    c0 += (u64)(u32)k0 * (u64)(u32)(k0 >> 32) + d1; // acc[i0]   += mul(k0) ; acc[i0] (= (i0+1)^1) += d1
    c1 += (u64)(u32)k1 * (u64)(u32)(k1 >> 32) + d0; // acc[i0+1] += mul(k1) ; acc[i0+1] += d0
}
#ifndef SQX_NT
#define SQX_NT 1 // non-temporal chunk loads: hash kernel 4.30 -> 3.95 ms (6.7 TB/s), step with two batches in flight 23.5 -> 22.5 ms
#endif
__device__ __forceinline__ uint4 ld16(const u8* p) {
#if SQX_NT
    // every chunk is read exactly once here: non-temporal loads keep the stream from displacing the L2 lines the
    // compress stage of the other batch in flight is gathering
    typedef unsigned int v4u_ua __attribute__((ext_vector_type(4), aligned(1)));
    v4u_ua const w = __builtin_nontemporal_load((const __attribute__((address_space(1))) v4u_ua*)p);
    return make_uint4(w.x, w.y, w.z, w.w);
#else
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
#endif
}

// The end of the long path for one wave: lane l holds the accumulator pair (acc[2j], acc[2j+1]), j = l & 3, as it
// stands after the last FULL 1 KiB block (all 16 lanes of a class hold the same values).  Adds the stripes of the
// last partial block and the final 64 bytes (secret offset 192-64-7), merges.  Every lane returns the hash.
// Also the tail of the hash that compress stage 1 computes beside its scan (sq_zstd_fast.h, fast_hash_chain).
__device__ inline H128 hash_finish_wave(const u8* __restrict__ p, u64 len, u32 lane, u64 a0, u64 a1) {
    const u32 s = lane >> 2, i0 = 2 * (lane & 3);
    const u64 sk0 = sec64(8 * s + 8 * i0), sk1 = sec64(8 * s + 8 * i0 + 8);
    const u64 nbBlocks = (len - 1) >> 10;
    const u8* q = p + lane * 16;
    {   // last partial block: full stripes, then the final 64 bytes with secret offset 192-64-7
        const u64 done = nbBlocks << 10;
        const u32 nbStripes = (u32)(((len - 1) - done) >> 6);
        u64 c0 = 0, c1 = 0;
        if (s < nbStripes) {
            uint4 v = ld16(q + done);
            contrib(v, sk0, sk1, c0, c1);
        }
        if (lane < 4) {
            uint4 v = ld16(p + len - 64 + lane * 16);
            contrib(v, sec64(121 + 8 * i0), sec64(121 + 8 * i0 + 8), c0, c1);
        }
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) {
            c0 += shfl_xor64(c0, m);
            c1 += shfl_xor64(c1, m);
        }
        a0 += c0;
        a1 += c1;
    }
    // mergeAccs: lanes with (lane&3)==i hold acc[2i], acc[2i+1]
    const u32 i = lane & 3;
    u64 lo = mul128_fold64(a0 ^ sec64(11 + 16 * i), a1 ^ sec64(11 + 16 * i + 8));
    u64 hi = mul128_fold64(a0 ^ sec64(117 + 16 * i), a1 ^ sec64(117 + 16 * i + 8));
    lo += shfl_xor64(lo, 1);
    lo += shfl_xor64(lo, 2);
    hi += shfl_xor64(hi, 1);
    hi += shfl_xor64(hi, 2);
    H128 r;
    r.lo = avalanche3(lo + len * P64_1);
    r.hi = avalanche3(hi + ~(len * P64_2));
    return r;
}

// One wave, one chunk.  Every lane returns the full hash.
__device__ inline H128 hash_long_wave(const u8* __restrict__ p, u64 len, u32 lane) {
    const u32 s = lane >> 2, i0 = 2 * (lane & 3);
    const u64 sk0 = sec64(8 * s + 8 * i0), sk1 = sec64(8 * s + 8 * i0 + 8);
    const u64 sc0 = sec64(128 + 8 * i0), sc1 = sec64(128 + 8 * i0 + 8);
    const u64 init[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    u64 a0 = init[i0], a1 = init[i0 + 1];
    const u64 nbBlocks = (len - 1) >> 10;
    const u8* q = p + lane * 16;

    auto finish_block = [&](u64 c0, u64 c1, bool scramble) {
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) {
            c0 += shfl_xor64(c0, m);
            c1 += shfl_xor64(c1, m);
        }
        a0 += c0;
        a1 += c1;
        if (scramble) {
            a0 = (a0 ^ (a0 >> 47) ^ sc0) * P32_1;
            a1 = (a1 ^ (a1 >> 47) ^ sc1) * P32_1;
        }
    };

    u64 b = 0;
    // 4 blocks (4 KiB per wave) in flight per iteration
    for (; b + 4 <= nbBlocks; b += 4) {
        uint4 v0 = ld16(q + (b << 10)), v1 = ld16(q + ((b + 1) << 10)), v2 = ld16(q + ((b + 2) << 10)),
              v3 = ld16(q + ((b + 3) << 10));
        u64 c0, c1;
        c0 = c1 = 0;
        contrib(v0, sk0, sk1, c0, c1);
        finish_block(c0, c1, true);
        c0 = c1 = 0;
        contrib(v1, sk0, sk1, c0, c1);
        finish_block(c0, c1, true);
        c0 = c1 = 0;
        contrib(v2, sk0, sk1, c0, c1);
        finish_block(c0, c1, true);
        c0 = c1 = 0;
        contrib(v3, sk0, sk1, c0, c1);
        finish_block(c0, c1, true);
    }
    for (; b < nbBlocks; b++) {
        uint4 v = ld16(q + (b << 10));
        u64 c0 = 0, c1 = 0;
        contrib(v, sk0, sk1, c0, c1);
        finish_block(c0, c1, true);
    }
    return hash_finish_wave(p, len, lane, a0, a1);
}

// grid: ceil(n / wavesPerBlock) blocks of 64*wavesPerBlock threads
// ---- 16 lanes per chunk, four chunks per wave -------------------------------------------------
// The wave-per-chunk mapping above spends most of its instructions on the 4-step butterfly (16
// ds_bpermute per 1 KiB block) that sums the 16 stripes.  Here a row of 16 lanes owns a chunk: lane
// 4*g + j holds the accumulator pair j (words 2j, 2j+1) for the stripes g, g+4, g+8, g+12 of every block,
// sums its four stripes privately and the four groups are then summed with two DPP row rotations (no
// LDS traffic): ~70 instructions per block and row instead of ~76 per block and wave.  A row's 16-byte
// loads cover 256 contiguous bytes; the four rows of a wave stream four different chunks.
__device__ __forceinline__ u64 row_ror64(u64 v, int ctrl) { // DPP row rotation of a 64-bit value
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    if (ctrl == 4) {
        lo = (u32)__builtin_amdgcn_update_dpp(0, (int)lo, 0x124, 0xF, 0xF, false);
        hi = (u32)__builtin_amdgcn_update_dpp(0, (int)hi, 0x124, 0xF, 0xF, false);
    } else {
        lo = (u32)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xF, 0xF, false);
        hi = (u32)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xF, 0xF, false);
    }
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 quad_xor64(u64 v, int m) { // value of lane ^ m inside the quad
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    if (m == 1) {
        lo = (u32)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, false); // quad_perm [1,0,3,2]
        hi = (u32)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
    } else {
        lo = (u32)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, false); // quad_perm [2,3,0,1]
        hi = (u32)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
    }
    return ((u64)hi << 32) | lo;
}

// One 16-lane row, one chunk of `len` > 240 bytes.  Every lane of the row returns the hash.
// sums (or null): the eight 64-bit sums of every FULL 1 KiB block of the chunk (what the stripes of a block add to the
// accumulators: 64 B per block, accumulator order), left by the scan of compress stage 1 (sq_zstd_fast.h): the blocks
// are then not read at all, only the scramble chain and the tail run here.
__device__ inline H128 hash_long_row(const u8* __restrict__ p, u64 len, u32 lane, const u64* __restrict__ sums = nullptr) {
    const u32 rl = lane & 15, g = rl >> 2, j = rl & 3, i0 = 2 * j;
    const u64 sc0 = sec64(128 + 8 * i0), sc1 = sec64(128 + 8 * i0 + 8);
    // secret words of the four stripes this lane owns (stripe s uses secret offset 8*s)
    u64 k0[4], k1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        k0[t] = sec64(8 * (g + 4 * t) + 8 * i0);
        k1[t] = sec64(8 * (g + 4 * t) + 8 * i0 + 8);
    }
    const u64 init[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    u64 a0 = init[i0], a1 = init[i0 + 1];
    const u64 nbBlocks = (len - 1) >> 10;
    const u8* q = p + 64 * g + 16 * j; // this lane's 16 bytes of stripe g of block 0

    auto fold = [&](u64 c0, u64 c1, bool scramble) { // sum the four groups of the row, then the block's scramble
        c0 += row_ror64(c0, 4);
        c1 += row_ror64(c1, 4);
        c0 += row_ror64(c0, 8);
        c1 += row_ror64(c1, 8);
        a0 += c0;
        a1 += c1;
        if (scramble) {
            a0 = (a0 ^ (a0 >> 47) ^ sc0) * P32_1;
            a1 = (a1 ^ (a1 >> 47) ^ sc1) * P32_1;
        }
    };
    u64 b = 0;
    if (sums) { // lane (g, j) takes pair j of every block (the four g of a row load the same 16 bytes)
        const u8* const sp = (const u8*)sums + 16 * j;
        auto step = [&](const uint4& v) {
            a0 += ((u64)v.y << 32) | v.x;
            a1 += ((u64)v.w << 32) | v.z;
            a0 = (a0 ^ (a0 >> 47) ^ sc0) * P32_1;
            a1 = (a1 ^ (a1 >> 47) ^ sc1) * P32_1;
        };
        for (; b + 8 <= nbBlocks; b += 8) {
            uint4 v[8];
#pragma unroll
            for (int t = 0; t < 8; t++) __builtin_memcpy(&v[t], sp + ((b + t) << 6), 16);
#pragma unroll
            for (int t = 0; t < 8; t++) step(v[t]);
        }
        for (; b < nbBlocks; b++) {
            uint4 v;
            __builtin_memcpy(&v, sp + (b << 6), 16);
            step(v);
        }
    }
    // two blocks (8 loads per lane) in flight
    for (; b + 2 <= nbBlocks; b += 2) {
        uint4 v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = ld16(q + (b << 10) + 256 * t);
        u64 c0 = 0, c1 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) contrib(v[t], k0[t], k1[t], c0, c1);
        fold(c0, c1, true);
        c0 = c1 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) contrib(v[4 + t], k0[t], k1[t], c0, c1);
        fold(c0, c1, true);
    }
    for (; b < nbBlocks; b++) {
        u64 c0 = 0, c1 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) contrib(ld16(q + (b << 10) + 256 * t), k0[t], k1[t], c0, c1);
        fold(c0, c1, true);
    }
    { // last partial block: full stripes, then the final 64 bytes with secret offset 192-64-7
        const u64 done = nbBlocks << 10;
        const u32 nbStripes = (u32)(((len - 1) - done) >> 6);
        u64 c0 = 0, c1 = 0;
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (g + 4 * t < nbStripes) contrib(ld16(q + done + 256 * t), k0[t], k1[t], c0, c1);
        if (g == 0) {
            uint4 v = ld16(p + len - 64 + j * 16);
            contrib(v, sec64(121 + 8 * i0), sec64(121 + 8 * i0 + 8), c0, c1);
        }
        fold(c0, c1, false);
    }
    // mergeAccs: lanes with (lane & 3) == j hold acc[2j], acc[2j+1]
    u64 lo = mul128_fold64(a0 ^ sec64(11 + 16 * j), a1 ^ sec64(11 + 16 * j + 8));
    u64 hi = mul128_fold64(a0 ^ sec64(117 + 16 * j), a1 ^ sec64(117 + 16 * j + 8));
    lo += quad_xor64(lo, 1);
    lo += quad_xor64(lo, 2);
    hi += quad_xor64(hi, 1);
    hi += quad_xor64(hi, 2);
    H128 r;
    r.lo = avalanche3(lo + len * P64_1);
    r.hi = avalanche3(hi + ~(len * P64_2));
    return r;
}

// grid = ceil(n / (4 * waves per workgroup)), one row of 16 lanes per chunk
__global__ void __launch_bounds__(256)
k_xxh3_128_rows(const u8* __restrict__ data, const u64* __restrict__ off, const u32* __restrict__ len, u32 n,
                u8* __restrict__ hashes, const u8* __restrict__ summed, const u8* __restrict__ sums, u64 sums_stride) {
    const u32 lane = threadIdx.x & 63;
    const u32 c = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    // Rows are independent: the DPP rotations / quad permutes of hash_long_row only read lanes of the same
    // row (quad), and a row's 16 lanes always take the same path, so rows may diverge freely (other
    // lengths, the short path, no chunk at all).
    // summed / sums (or null): summed[c] != 0 says that compress stage 1 has streamed chunk c and left the sums of
    // its full blocks at sums + c * sums_stride (the fused form of a pack batch): those chunks are not read again
    const bool live = c < n;
    H128 h = {0, 0};
    if (live) {
        const u8* p = data + off[c];
        const u32 L = len[c];
        if (L <= 240)
            h = hash_short(p, L);
        else
            h = hash_long_row(p, L, lane, (summed != nullptr && summed[c] != 0) ? (const u64*)(sums + (u64)c * sums_stride) : nullptr);
    }
    if (live && (lane & 15) == 0) {
        u64* o = (u64*)(hashes + 16ull * c);
        o[0] = h.lo; // u128.to_le_bytes(): low64 LE || high64 LE
        o[1] = h.hi;
    }
}

__global__ void __launch_bounds__(256)
k_xxh3_128_batch(const u8* __restrict__ data, const u64* __restrict__ off,
                 const u32* __restrict__ len, u32 n, u8* __restrict__ hashes) {
    const u32 lane = threadIdx.x & 63;
    const u32 c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= n) return;
    const u8* p = data + off[c];
    const u32 L = len[c];
    H128 h;
    if (L <= 240)
        h = hash_short(p, L);
    else
        h = hash_long_wave(p, L, lane);
    if (lane == 0) {
        u64* o = (u64*)(hashes + 16ull * c);
        o[0] = h.lo; // u128.to_le_bytes(): low64 LE || high64 LE
        o[1] = h.hi;
    }
}

} // namespace sqx
