// sq_common.h -- shared scalar types and small helpers for the gfx950 pack path.
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;
typedef int32_t s32;
typedef int64_t s64;

#if defined(__HIPCC__) && !defined(SQ_SIM)
#include <hip/hip_runtime.h>
#define SQ_DEV __device__ __forceinline__
#define SQ_DEVFN __device__
#define SQ_ON_DEVICE 1
#else
#include <string.h>
#define SQ_DEV static inline
#define SQ_DEVFN static
#define SQ_ON_DEVICE 0
#endif

#define SQ_WAVE 64

SQ_DEV u32 sq_highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }

// Unaligned little-endian loads (gfx950 runs with unaligned access mode on: these compile to
// single global_load_dword/dwordx2/dwordx4 instructions).
SQ_DEV u16 sq_rd16(const u8* p) {
    u16 v;
    __builtin_memcpy(&v, p, 2);
    return v;
}
SQ_DEV u32 sq_rd32(const u8* p) {
    u32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
SQ_DEV u64 sq_rd64(const u8* p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
