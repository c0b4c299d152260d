/*
 * xxh3_oracle.c -- CPU ORACLE (test infrastructure, see sq_oracle.h): XXH3-128, seed 0,
 * default 192-byte secret, restated from the published XXH3 algorithm (xxHash 0.8.x spec).
 *
 * Follows /root/reference/src/util/chunk.rs:46-49:
 *     pub fn hash_chunk(chunk: &[u8]) -> ChunkHash { xxh3_128(chunk).to_le_bytes() }
 * The arithmetic is in the third-party crate xxhash-rust 0.8.15 (Cargo.lock:1593-1596), which
 * implements the same XXH3 spec.  Pinned by tests/golden/xxh3_*.json (python-xxhash outputs).
 */
#include "sq_oracle.h"
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

static const u8 kSecret[192] = {
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

#define P32_1 0x9E3779B1U
#define P32_2 0x85EBCA77U
#define P32_3 0xC2B2AE3DU
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
#define PMX1 0x165667919E3779F9ULL
#define PMX2 0x9FB21C651E98DF25ULL

static u64 rd64(const u8* p) {
    u64 v;
    memcpy(&v, p, 8);
    return v; /* little-endian host */
}
static u32 rd32(const u8* p) {
    u32 v;
    memcpy(&v, p, 4);
    return v;
}
static u32 swap32(u32 x) { return __builtin_bswap32(x); }
static u64 swap64(u64 x) { return __builtin_bswap64(x); }
static u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }

static u64 avalanche3(u64 h) {
    h ^= h >> 37;
    h *= PMX1;
    h ^= h >> 32;
    return h;
}
static u64 avalanche64(u64 h) {
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    h ^= h >> 32;
    return h;
}
static u64 mul128_fold64(u64 a, u64 b) {
    __uint128_t p = (__uint128_t)a * b;
    return (u64)p ^ (u64)(p >> 64);
}
static u64 mix16B(const u8* in, const u8* sec) { /* seed 0 */
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}

typedef struct {
    u64 lo, hi;
} h128;

static h128 len_1to3(const u8* in, size_t len) {
    u8 c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
    u32 cl = ((u32)c1 << 16) | ((u32)c2 << 24) | (u32)c3 | ((u32)len << 8);
    u32 ch = rotl32(swap32(cl), 13);
    u64 bl = (u64)(rd32(kSecret) ^ rd32(kSecret + 4));
    u64 bh = (u64)(rd32(kSecret + 8) ^ rd32(kSecret + 12));
    h128 r = {avalanche64((u64)cl ^ bl), avalanche64((u64)ch ^ bh)};
    return r;
}
static h128 len_4to8(const u8* in, size_t len) {
    u32 ilo = rd32(in), ihi = rd32(in + len - 4);
    u64 i64 = ilo + ((u64)ihi << 32);
    u64 bitflip = rd64(kSecret + 16) ^ rd64(kSecret + 24);
    u64 keyed = i64 ^ bitflip;
    __uint128_t m = (__uint128_t)keyed * (P64_1 + ((u64)len << 2));
    u64 lo = (u64)m, hi = (u64)(m >> 64);
    hi += lo << 1;
    lo ^= hi >> 3;
    lo ^= lo >> 35;
    lo *= PMX2;
    lo ^= lo >> 28;
    hi = avalanche3(hi);
    h128 r = {lo, hi};
    return r;
}
static h128 len_9to16(const u8* in, size_t len) {
    u64 bl = rd64(kSecret + 32) ^ rd64(kSecret + 40);
    u64 bh = rd64(kSecret + 48) ^ rd64(kSecret + 56);
    u64 ilo = rd64(in), ihi = rd64(in + len - 8);
    __uint128_t m = (__uint128_t)(ilo ^ ihi ^ bl) * P64_1;
    u64 lo = (u64)m, hi = (u64)(m >> 64);
    lo += (u64)(len - 1) << 54;
    ihi ^= bh;
    hi += ihi + (u64)(u32)ihi * (u64)(P32_2 - 1);
    lo ^= swap64(hi);
    {
        __uint128_t h = (__uint128_t)lo * P64_2;
        u64 hlo = (u64)h, hhi = (u64)(h >> 64);
        hhi += hi * P64_2;
        h128 r = {avalanche3(hlo), avalanche3(hhi)};
        return r;
    }
}
static h128 mix32B(h128 acc, const u8* i1, const u8* i2, const u8* sec) {
    acc.lo += mix16B(i1, sec);
    acc.lo ^= rd64(i2) + rd64(i2 + 8);
    acc.hi += mix16B(i2, sec + 16);
    acc.hi ^= rd64(i1) + rd64(i1 + 8);
    return acc;
}
static h128 finish_mid(h128 acc, size_t len) {
    h128 r;
    r.lo = acc.lo + acc.hi;
    r.hi = acc.lo * P64_1 + acc.hi * P64_4 + (u64)len * P64_2;
    r.lo = avalanche3(r.lo);
    r.hi = (u64)0 - avalanche3(r.hi);
    return r;
}
static h128 len_17to128(const u8* in, size_t len) {
    h128 acc = {(u64)len * P64_1, 0};
    int i = (int)((len - 1) / 32);
    for (; i >= 0; i--) acc = mix32B(acc, in + 16 * i, in + len - 16 * (i + 1), kSecret + 32 * i);
    return finish_mid(acc, len);
}
static h128 len_129to240(const u8* in, size_t len) {
    h128 acc = {(u64)len * P64_1, 0};
    unsigned i;
    for (i = 32; i < 160; i += 32) acc = mix32B(acc, in + i - 32, in + i - 16, kSecret + i - 32);
    acc.lo = avalanche3(acc.lo);
    acc.hi = avalanche3(acc.hi);
    for (i = 160; i <= len; i += 32)
        acc = mix32B(acc, in + i - 32, in + i - 16, kSecret + 3 + i - 160);
    acc = mix32B(acc, in + len - 16, in + len - 32, kSecret + 136 - 17 - 16);
    return finish_mid(acc, len);
}

static void accumulate_stripe(u64 acc[8], const u8* in, const u8* sec) {
    for (int i = 0; i < 8; i++) {
        u64 d = rd64(in + 8 * i);
        u64 k = d ^ rd64(sec + 8 * i);
        acc[i ^ 1] += d;
        acc[i] += (u64)(u32)k * (u64)(k >> 32);
    }
}
static void scramble(u64 acc[8], const u8* sec) {
    for (int i = 0; i < 8; i++) {
        u64 a = acc[i];
        a ^= a >> 47;
        a ^= rd64(sec + 8 * i);
        a *= P32_1;
        acc[i] = a;
    }
}
static u64 merge_accs(const u64 acc[8], const u8* sec, u64 start) {
    u64 r = start;
    for (int i = 0; i < 4; i++)
        r += mul128_fold64(acc[2 * i] ^ rd64(sec + 16 * i), acc[2 * i + 1] ^ rd64(sec + 16 * i + 8));
    return avalanche3(r);
}
static h128 hash_long(const u8* in, size_t len) {
    u64 acc[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    const size_t nbStripesPerBlock = (192 - 64) / 8; /* 16 */
    const size_t blockLen = 64 * nbStripesPerBlock;  /* 1024 */
    size_t nbBlocks = (len - 1) / blockLen;
    for (size_t n = 0; n < nbBlocks; n++) {
        for (size_t s = 0; s < nbStripesPerBlock; s++)
            accumulate_stripe(acc, in + n * blockLen + s * 64, kSecret + s * 8);
        scramble(acc, kSecret + 192 - 64);
    }
    {
        size_t nbStripes = ((len - 1) - blockLen * nbBlocks) / 64;
        for (size_t s = 0; s < nbStripes; s++)
            accumulate_stripe(acc, in + nbBlocks * blockLen + s * 64, kSecret + s * 8);
        accumulate_stripe(acc, in + len - 64, kSecret + 192 - 64 - 7);
    }
    h128 r;
    r.lo = merge_accs(acc, kSecret + 11, (u64)len * P64_1);
    r.hi = merge_accs(acc, kSecret + 192 - 64 - 11, ~((u64)len * P64_2));
    return r;
}

void sqo_hash_chunk(const uint8_t* chunk, size_t len, uint8_t out[16]) {
    h128 h;
    if (len == 0) {
        h.lo = avalanche64(rd64(kSecret + 64) ^ rd64(kSecret + 72));
        h.hi = avalanche64(rd64(kSecret + 80) ^ rd64(kSecret + 88));
    } else if (len <= 3)
        h = len_1to3(chunk, len);
    else if (len <= 8)
        h = len_4to8(chunk, len);
    else if (len <= 16)
        h = len_9to16(chunk, len);
    else if (len <= 128)
        h = len_17to128(chunk, len);
    else if (len <= 240)
        h = len_129to240(chunk, len);
    else
        h = hash_long(chunk, len);
    memcpy(out, &h.lo, 8); /* u128.to_le_bytes(): low64 LE || high64 LE */
    memcpy(out + 8, &h.hi, 8);
}
