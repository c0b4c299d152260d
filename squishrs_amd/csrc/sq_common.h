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

// ---- explicit address spaces -------------------------------------------------------------------
// Pointers that travel through structs / non-inlined calls degrade to generic ("flat") pointers;
// flat_load/flat_store count on BOTH vmcnt and lgkmcnt and complete out of order, so hipcc drains
// every counter to 0 before using a flat load's result -- which serialises all pipelined loads.
// Every table / chunk / output byte of the compressor lives in global memory (HBM) and every
// statistic in LDS, so the hot primitives assert the address space at the access:
//   sq_g*  : global (address_space 1)  -> global_load_* / global_store_*
//   SQ_L   : LDS    (address_space 3)  -> ds_*
#if SQ_ON_DEVICE
#define SQ_G __attribute__((address_space(1)))
#define SQ_L __attribute__((address_space(3)))
typedef u16 __attribute__((aligned(1))) sq_u16_ua;
typedef u32 __attribute__((aligned(1))) sq_u32_ua;
typedef u64 __attribute__((aligned(1))) sq_u64_ua;
typedef u32 sq_v4u __attribute__((ext_vector_type(4)));
typedef u32 sq_v2u __attribute__((ext_vector_type(2)));
typedef sq_v4u __attribute__((aligned(1))) sq_v4u_ua;
SQ_DEV u32 sq_grd8(const u8* p) { return *(const SQ_G u8*)p; }
SQ_DEV u32 sq_grd16(const u8* p) { return *(const SQ_G sq_u16_ua*)p; }
SQ_DEV u32 sq_grd32(const u8* p) { return *(const SQ_G sq_u32_ua*)p; }
SQ_DEV u64 sq_grd64(const u8* p) { return *(const SQ_G sq_u64_ua*)p; }
SQ_DEV sq_v4u sq_grd128(const u8* p) { return *(const SQ_G sq_v4u_ua*)p; }
SQ_DEV void sq_gwr128(u8* p, sq_v4u v) { *(SQ_G sq_v4u_ua*)p = v; }
// streaming variants (nt): data that is read or written once and should not displace what the gathers brought into L2
SQ_DEV sq_v4u sq_grd128_nt(const u8* p) { return __builtin_nontemporal_load((const SQ_G sq_v4u_ua*)p); }
SQ_DEV void sq_gwr128_nt(u8* p, sq_v4u v) { __builtin_nontemporal_store(v, (SQ_G sq_v4u_ua*)p); }
// the streaming passes of the compressor proper (copies, histograms, certificate scans, table clears); 0 for A/B runs.
// Headline step 22.3 -> 21.6 ms, configs[3] sample unchanged within 2 % (profiles/round2b_stream_nontemporal.txt).
#ifndef SQ_STREAM_NT
#define SQ_STREAM_NT 1
#endif
#if SQ_STREAM_NT
SQ_DEV sq_v4u sq_grd128_s(const u8* p) { return sq_grd128_nt(p); }
SQ_DEV void sq_gwr128_s(u8* p, sq_v4u v) { sq_gwr128_nt(p, v); }
#else
SQ_DEV sq_v4u sq_grd128_s(const u8* p) { return sq_grd128(p); }
SQ_DEV void sq_gwr128_s(u8* p, sq_v4u v) { sq_gwr128(p, v); }
#endif
SQ_DEV void sq_gwr8(u8* p, u32 v) { *(SQ_G u8*)p = (u8)v; }
SQ_DEV u32 sq_gld32(const u32* p) { return *(const SQ_G u32*)p; }          // aligned table word
SQ_DEV void sq_gst32(u32* p, u32 v) { *(SQ_G u32*)p = v; }
// table loads: words this kernel itself rewrites (hash / chain / tag tables).  Relaxed agent-scope
// atomic loads = global_load ... sc1: served by L2, never by a possibly stale L1 line or by the
// scalar cache.  (Random 4-byte gathers gain nothing from the 32 KiB L1 anyway.)
SQ_DEV u32 sq_tld32(const u32* p) { return __hip_atomic_load((const SQ_G u32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SQ_DEV u32 sq_tld8(const u8* p) { return (u32)__hip_atomic_load((const SQ_G u8*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SQ_DEV sq_v2u sq_gld64v(const u32* p) { return *(const SQ_G sq_v2u*)p; }   // 8-byte aligned pair
SQ_DEV void sq_gst64v(u32* p, u32 a, u32 b) { *(SQ_G sq_v2u*)p = (sq_v2u){a, b}; }
#else
#define SQ_G
#define SQ_L
SQ_DEV u32 sq_grd8(const u8* p) { return *p; }
SQ_DEV u32 sq_grd16(const u8* p) { return sq_rd16(p); }
SQ_DEV u32 sq_grd32(const u8* p) { return sq_rd32(p); }
SQ_DEV u64 sq_grd64(const u8* p) { return sq_rd64(p); }
SQ_DEV void sq_gwr8(u8* p, u32 v) { *p = (u8)v; }
SQ_DEV u32 sq_gld32(const u32* p) { return *p; }
SQ_DEV u32 sq_tld32(const u32* p) { return *p; }
SQ_DEV u32 sq_tld8(const u8* p) { return *p; }
SQ_DEV void sq_gst32(u32* p, u32 v) { *p = v; }
#endif
