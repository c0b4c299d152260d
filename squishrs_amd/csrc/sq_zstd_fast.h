// sq_zstd_fast.h -- the incompressible-chunk kernel of the zstd level-12 pack path.
//
// For a chunk on which zstd-12 (btlazy2: 16 KiB < len <= 256 KiB, /root/reference/src/util/chunk.rs:89-90
// -> libzstd 1.5.7) provably finds nothing, the frame is known without running the compressor:
//   * the no-match certificate (cert_compute_core, sq_zstd_match.h) proves that every search of the
//     lazy parser fails in every block: no sequences;
//   * a block without sequences goes to ZSTD_compressLiterals with "suspect uncompressible" set;
//     for srcSize >= 40960 HUF_compress first looks at the byte histograms of the block's first and
//     last 4 KiB: largest1 + largest2 <= (8192 >> 7) + 4 = 68 means "not compressible" -> raw
//     literals -> section + 1 > block -> ZSTD_noCompressBlock: the block is emitted raw
//     (and it is not an RLE block, since no byte fills 4 KiB);
// so the frame is header + raw blocks.  Everything else is left to k_compress_batch (frame length 0),
// together with the certificate ends already computed here.
//
// Why a separate kernel: this path is a few streaming passes with all of its state in LDS (the
// certificate's set S, 29 KB per chunk, shared by the four waves of a workgroup at 128 VGPRs: 16 waves
// per CU); k_compress_batch carries the whole zstd state machine (248 VGPRs, 20 KB of LDS of entropy
// tables per wave) and keeps the certificate's table in HBM.
#pragma once
#include "sq_zstd_match.h"
#if SQ_ON_DEVICE
#include "sq_xxh3.h" // the fused hash: stage 1 computes XXH3-128 beside its scan (one read of the chunk instead of two)
#endif
#include <type_traits>
#if !SQ_ON_DEVICE
#include <algorithm>
#include <utility>
#include <vector>
#endif

namespace sqz {

#define SQZ_FAST_VALS 3584u  // >= scheduled searches of a chunk of this strategy (2 x 1748)
#define SQZ_FAST_KEYS 1024u  // value buckets: top 10 bits
#define SQZ_FAST_HITS 32u
#ifndef SQZ_FAST_LOOKUP_BATCH
#define SQZ_FAST_LOOKUP_BATCH 64u // Bloom positives collected before their buckets are looked up
#endif
#define SQZ_FAST_BUCKET_MAX 24u // a bucket holds 3.4 values on average; beyond this the data is not random

// One chunk is certified by a WORKGROUP of SQZ_FAST_WG_WAVES waves that share the chunk's set S in LDS: the
// filter, the bucketed values and the hit list exist once per chunk, each wave scans its own contiguous part of
// the chunk with a candidate list of its own.  Two waves per chunk halve the LDS per wave (8 workgroups = 16 waves
// per CU instead of 8 waves) and the registers the build needs per lane (each wave gathers half of S), which is
// what lets the scan run at 4 waves per SIMD instead of 2 ("workgroup edition").
#ifndef SQZ_FAST_WG_WAVES
#define SQZ_FAST_WG_WAVES 4u
#endif
#ifndef SQZ_FAST_NT
#define SQZ_FAST_NT 3 // 1: the scan's stream loads are non-temporal, 2: its raw-block stores, 3: both (the lines the build gathered stay in L2 for the stream's pass over the same bytes: stage 1 13.46 -> 13.25 ms)
#endif
#ifndef SQZ_FAST_DEPTH
#define SQZ_FAST_DEPTH 5u // rounds of the scan's stream in flight per wave
#endif
#ifndef SQZ_FAST_DEPTH_HS
#define SQZ_FAST_DEPTH_HS 4u // ... in the fused-hash form of the scan (register pressure: one round's 5 registers less)
#endif
#ifndef SQZ_FAST_WAVES_PER_SIMD
#define SQZ_FAST_WAVES_PER_SIMD 4
#endif
#if SQ_ON_DEVICE
#define FW_NW SQZ_FAST_WG_WAVES
SQ_DEV u32 fw_wave() { return wv_uni((u32)threadIdx.x >> 6); } // this wave's number inside its workgroup
SQ_DEV void fw_sync() { __syncthreads(); }                      // every wave of the workgroup reaches every fw_sync()
SQ_DEV bool fw_first() { return threadIdx.x == 0; }
#else
#define FW_NW 1u
SQ_DEV u32 fw_wave() { return 0; }
SQ_DEV void fw_sync() {}
SQ_DEV bool fw_first() { return true; }
#endif
// f(integral_constant<I>) for I = I0 .. N - 1, stopping at the first call that returns false (returns false then)
template <u32 I, u32 N, class F>
SQ_DEV bool fw_static_for(F&& f) {
    if constexpr (I < N) {
        if (!f(std::integral_constant<u32, I>{})) return false;
        return fw_static_for<I + 1, N>(f);
    } else
        return true;
}
// words of FastLds::sh
#define FW_SH_NHITS 0u  // hits listed so far (may run past SQZ_FAST_HITS: the list is then full)
#define FW_SH_CERT0 1u  // certEnd  (atomic minimum over the waves)
#define FW_SH_CERT1 2u  // certEnd1
#define FW_SH_FLAGS 3u  // 1: the repeats inside S do not fit the hit list, 2: a bucket too large for this path, 8: second pass, the bucket check failed
#define FW_SH_WAKE 4u   // second pass, the bucket check: 4 = the search position it belongs to, 6 = the match's position, 7 = that
                        // search's bucket hash | its first byte << 24
#define FW_SH_IDX 5u    // the chunk index the workgroup pulled from the queue
#define FW_SH_FIN 8u    // [FW_NW] position each wave's scan ended at
#define FW_SH_SAMPLE 12u // [4] largest count of each 4 KiB sample of the HUF pre-check
// (FW_SH_FLAGS bit 2 (4): some wave did not stream its whole run -- the fused hash is not available)

// The fused hash ("fused hash").  With a 16-byte aligned chunk a round of the scan IS one 1 KiB block of
// XXH3's long path (lane l holds stripe l >> 2, accumulator pair l & 3: the layout of sqx::hash_long_wave), and a
// block's contribution to the accumulators does not depend on them (acc[i] += mul(data ^ secret), acc[i ^ 1] += data;
// only the scramble BETWEEN blocks is a recurrence).  So each round also computes its block's eight 64-bit sums
// (2 per lane, reduced over the row's four stripes with two DPP rotations, over the four rows through LDS) and
// every few rounds the wave stores them (64 B per block) to the chunk's slice of a pool in global memory.  The serial
// part -- 255 scrambles per 256 KiB chunk, a few microseconds of dependent arithmetic -- is NOT done here (run by one
// wave at the end of a chunk it held the workgroup's slot for a fifth of the chunk's time: stage 1 13.9 -> 18.4 ms):
// sqx::k_xxh3_128_rows, launched behind this kernel, runs it for all chunks at once, 16 lanes per chunk, and adds the
// tail of the long path.  Cost here: ~20 VALU instructions per round and 64 B written per KiB read; saved: the hash
// kernel's read of every chunk.
#define SQZ_FAST_HPOOL_BLOCKS 260u                           // 256 blocks + a dump area for the idle lanes of a flush
#define SQZ_FAST_HPOOL_BYTES (SQZ_FAST_HPOOL_BLOCKS * 64u)   // per chunk

// (a, b) += row_ror:4 (a, b); (a, b) += row_ror:8 (a, b) for two 64-bit values a = a1:a0, b = b1:b0 -- each lane ends
// with the sum over the four lanes of its row that share its (lane & 3).  64-bit adds whose first operand comes
// through DPP, two instructions each, in place (a row rotation reads its 16 lanes before it writes them; as C++ the
// compiler spends five instructions per add: two DPP moves, their zeroed destinations, one 64-bit add).  The two chains
// alternate, so no DPP instruction reads a register written less than three instructions before it (DPP hazard: 2 wait
// states); the s_nop covers whatever VALU instruction wrote the inputs.
#define SQZ_ROW_SUM4(a0, a1, b0, b1)                                                                   \
    asm("s_nop 1\n\t"                                                                                  \
        "v_add_co_u32_dpp %0, vcc, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"                    \
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"              \
        "v_add_co_u32_dpp %2, vcc, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"                    \
        "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"              \
        "v_add_co_u32_dpp %0, vcc, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                    \
        "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"              \
        "v_add_co_u32_dpp %2, vcc, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                    \
        "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc row_ror:8 row_mask:0xf bank_mask:0xf"                    \
        : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1)                                                         \
        :                                                                                                \
        : "vcc")

// LDS of one workgroup of k_incompressible_batch (34.1 KB at four waves: the 4 workgroups = 16 waves per CU that the
// 128 registers per lane allow use 137 of the CU's 160 KB)
struct FastLds {
    u32 bloom[SQZ_BLOOM_WORDS];      // certificate Bloom filter; afterwards the 256-bin sample histograms (256 words per wave)
    u16 vals[SQZ_FAST_VALS];         // per scheduled position: bits 6..21 of its 4-byte value, grouped by bits 22..31
    u32 ends[SQZ_FAST_KEYS / 2 + 2]; // per bucket, packed 2 x u16: first counts, then starts, finally END offsets
    u16 idx[SQZ_FAST_VALS];          // beside vals: which scheduled search (its number in the schedule) the value belongs to
    u32 cand[FW_NW * 256];           // per wave, a few rounds' Bloom positives: 128 x (position, bytes)
    u32 hits[2 * SQZ_FAST_HITS];     // (position, bytes) of scanned positions whose bytes equal a scheduled position's;
                                     // position 0 = two scheduled positions with equal bytes
    u32 sh[16];                      // words shared by the waves (FW_SH_*)
    u32 keep[8];                     // fastpath_blocks: the first pass's answers while the second pass runs (registers are short there)
    u32 hsec[256];                   // fused hash: lane l's two secret words (dword k at 64 k + l), the same for every chunk
    u32 hstage[FW_NW * SQZ_FAST_DEPTH_HS * 64]; // fused hash: per wave, the (round, row, pair) sums of a loop step, dword k at 64 k + 16 round + 4 row + pair
#ifdef SQ_PROF_S1
    unsigned long long prof[8]; // 100 MHz ticks: 0 build, 1 scan loop, 2 look-ups (inside 1), 3 hit resolution, 4 pre-check + frame, 5 whole chunk
#endif
};

#define SQZ_FAST_MIN_BLOCK (4096u * 10u) // HUF's sampled pre-check needs srcSize >= 10 samples
#define SQZ_CERT_HINT_VALID 0x80000000u   // flag in certOut[1]: the two words are a certificate
// certOut[0]: the certificate end of block 0 (a position: < 2^19) in the low bits; above it, in units of 64 bytes, how many
// bytes of the chunk the scan has ALREADY written as raw-block payload at their place in a frame of full raw blocks
// (dst + header + 3 (b + 1) + offset): the compressor skips its own copy of a raw block that is in place.
#define SQZ_CERT_HINT_POS_BITS 19u
#define SQZ_CERT_HINT_POS_MASK ((1u << SQZ_CERT_HINT_POS_BITS) - 1u)
#define SQZ_CERT_HINT_RAW_LOG 6u
#define SQZ_CERT_HINT_RAW_MAX ((1u << (32u - SQZ_CERT_HINT_POS_BITS)) - 1u)

// ZSTD_writeFrameHeader for a single-segment frame with content size n; returns the header size.
SQ_DEV u32 fast_frame_header(u8* dst, u32 n) {
    u32 const fcsCode = (n >= 256) + (n >= 65536 + 256);
    WV_SERIAL {
        dst[0] = 0x28;
        dst[1] = 0xB5;
        dst[2] = 0x2F;
        dst[3] = 0xFD;
        dst[4] = (u8)((fcsCode << 6) + (1 << 5));
        if (fcsCode == 0)
            dst[5] = (u8)n;
        else if (fcsCode == 1) {
            u32 v = n - 256;
            dst[5] = (u8)v;
            dst[6] = (u8)(v >> 8);
        } else {
            dst[5] = (u8)n;
            dst[6] = (u8)(n >> 8);
            dst[7] = (u8)(n >> 16);
            dst[8] = (u8)(n >> 24);
        }
    }
    return 5 + (fcsCode == 0 ? 1 : (fcsCode == 1 ? 2 : 4));
}

// the same from a workgroup: its first wave writes, every wave gets the size
SQ_DEV u32 fast_frame_header_all(u8* dst, u32 n) {
    if (fw_wave() == 0) return fast_frame_header(dst, n);
    return 5 + (n < 256 ? 1u : (n < 65536 + 256 ? 2u : 4u));
}

#if defined(SQ_PROF_S1) && SQ_ON_DEVICE
#define FPROF_T0(v) unsigned long long v = __builtin_amdgcn_s_memrealtime()
#define FPROF_ADD(L, slot, v)                                              \
    do {                                                                   \
        unsigned long long e_ = __builtin_amdgcn_s_memrealtime();          \
        if (threadIdx.x == 0) (L)->prof[slot] += e_ - (v);                 \
    } while (0)
#else
#define FPROF_T0(v) do { } while (0)
#define FPROF_ADD(L, slot, v) do { } while (0)
#endif

// ---------------------------------------------------------------- the certificate, LDS edition
// Same statement as cert_compute_core (sq_zstd_match.h) -- for each block, the first scheduled
// search position s for which an earlier position holds the same 4 bytes, or whose repcode test
// fires -- but the set S lives in LDS instead of a 2 MiB table in HBM (whose 3,500 random 8-byte
// writes and ~6,000 random reads per chunk were 60 % of this kernel's time):
//   build : the values at the scheduled positions are bucketed by their top 10 bits with a counting
//           sort in LDS (count, prefix sum, scatter: the bucket "end" array doubles as the cursor);
//           two scheduled positions with equal values show up as equal neighbours in a bucket;
//   scan  : as before (aligned 16-byte stream loads, value-keyed Bloom filter, scheduled positions
//           left out -- here in EVERY round, so that a Bloom positive that is found in its bucket is
//           a genuine repeat); positives are looked up in their bucket (~3.4 values);
//   hits  : the rare genuine repeats (0.1 per random chunk) are resolved at the end by walking S.

// scheduled positions among [rlo, rhi) of level k (rlo, rhi relative to the block start, inside
// level k), as bits relative to r; wd = the level word of k
SQ_DEV u32 fast_sched_bits(u32 wd, u32 k, u32 rlo, u32 rhi, u32 r) {
    u32 const first = wd & 0x1FFFFu, step = k + 1;
    if (first == 0x1FFFFu || rlo >= rhi) return 0;
    u32 nxt = first;
    if (rlo > first) {
        u32 const x = rlo - first;
        u32 const m = (step >= 17) ? x - ((x * (wd >> 17)) >> 16) * step : x % step;
        nxt = rlo + (m ? step - m : 0u);
    }
    u32 bits = 0;
    for (; nxt < rhi; nxt += step) bits |= 1u << (nxt - r);
    return bits;
}
// scheduled positions of ONE block (start bs, end blkEnd, level words lv) among the 16 positions that
// start at lo, as bits relative to lo
SQ_DEV u32 fast_sched_mask_blk(const u32* lv, s32 lo, u32 bs, u32 blkEnd, u32 wdA, u32 wdB, bool haveWords) {
    if (blkEnd - bs < 9) return 0; // too small a block to be searched at all
    s32 const limPos = (s32)(blkEnd - 8); // a block's searches stop 8 bytes before its end
    s32 const a = lo < (s32)bs ? (s32)bs : lo, e = (lo + 16 < limPos) ? lo + 16 : limPos;
    if (a >= e) return 0;
    u32 const rlo = (u32)(a - (s32)bs), rhi = (u32)(e - (s32)bs);
    u32 const r = (u32)(lo - (s32)bs); // may wrap when lo < bs: bits are taken relative to it modulo 2^32
    u32 const kA = rlo >> 8, kB = (rhi - 1) >> 8;
    u32 const wa = haveWords ? wdA : sq_gld32(lv + (kA < SQZ_SCHEDLVL_N ? kA : SQZ_SCHEDLVL_N - 1));
    u32 const wb = haveWords ? wdB : sq_gld32(lv + (kB < SQZ_SCHEDLVL_N ? kB : SQZ_SCHEDLVL_N - 1));
    u32 const split = (kA + 1) << 8;
    u32 bits = fast_sched_bits(wa, kA, rlo, rhi < split ? rhi : split, r);
    if (kB != kA) bits |= fast_sched_bits(wb, kB, split, rhi, r);
    return bits;
}
// all scheduled positions among the 16 that start at position lo (blocks start at SQZ_WINDOW_START and
// blk1); wdA / wdB: the level words loadLevels() fetched for this window, valid unless it straddles blk1
SQ_DEV u32 fast_sched_mask(const u32* lvlWords /*global*/, s32 lo, u32 blk1, u32 chunkEndPos, u32 wdA, u32 wdB,
                           bool haveWords) {
    bool const two = blk1 != 0xFFFFFFFFu;
    u32 const end0 = two ? blk1 : chunkEndPos;
    if (two && lo < (s32)blk1 && lo + 16 > (s32)blk1) // straddles the block boundary (one lane of one round)
        return fast_sched_mask_blk(lvlWords, lo, SQZ_WINDOW_START, end0, 0, 0, false) |
               fast_sched_mask_blk(lvlWords + SQZ_SCHEDLVL_N, lo, blk1, chunkEndPos, 0, 0, false);
    if (two && lo >= (s32)blk1) return fast_sched_mask_blk(lvlWords + SQZ_SCHEDLVL_N, lo, blk1, chunkEndPos, wdA, wdB, haveWords);
    return fast_sched_mask_blk(lvlWords, lo, SQZ_WINDOW_START, end0, wdA, wdB, haveWords);
}

#if SQ_ON_DEVICE
// the scan's copy of it, kept out of line: it runs in ~10 of a chunk's 256 rounds, and inlined into each of the
// five unrolled rounds it was most of the loop's code (the loop no longer fitted the instruction cache)
__device__ __noinline__ u32 fast_sched_mask_rare(const u32* lvlWords, s32 lo, u32 blk1, u32 chunkEndPos, u32 wdA, u32 wdB,
                                                 u32 haveWords) {
    return fast_sched_mask(lvlWords, lo, blk1, chunkEndPos, wdA, wdB, haveWords != 0);
}
#endif

// The same for a window [ur, ur + 16) (relative to its block's start) that lies in levels >= 16, inside
// the block's searched range and inside one block: at most one scheduled position per level part, no
// loop, no division.  wa / wb: level words of the levels of ur and ur + 15.
SQ_DEV u32 fast_sched_mask_plain(u32 ur, u32 wa, u32 wb) {
    u32 const kA = ur >> 8, kB = (ur + 15) >> 8, split = (kA + 1) << 8;
    u32 bits = 0;
    { // part in level kA: [ur, min(ur + 16, split))
        u32 const first = wa & 0x1FFFFu, step = kA + 1, hi = ur + 16 < split ? ur + 16 : split;
        u32 const x = ur > first ? ur - first : 0u;
        u32 const m = x - ((x * (wa >> 17)) >> 16) * step;
        u32 const nxt = (ur > first ? ur : first) + (m ? step - m : 0u);
        if (first != 0x1FFFFu && nxt < hi) bits |= 1u << (nxt - ur);
    }
    if (kB != kA) { // part in level kB: [split, ur + 16); that level's first scheduled position is >= split
        u32 const first = wb & 0x1FFFFu;
        if (first != 0x1FFFFu && first < ur + 16) bits |= 1u << (first - ur);
    }
    return bits;
}

SQ_DEV u32 fast_cnt16_add(SQ_L u32* arr, u32 key) { // packed 16-bit counters; returns the old count
#if SQ_ON_DEVICE
    u32 const old = __hip_atomic_fetch_add(&arr[key >> 1], 1u << (16 * (key & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    u32 const old = arr[key >> 1];
    arr[key >> 1] = old + (1u << (16 * (key & 1)));
#endif
    return (old >> (16 * (key & 1))) & 0xFFFFu;
}
SQ_DEV u32 fast_get16(const SQ_L u32* arr, u32 key) { return (arr[key >> 1] >> (16 * (key & 1))) & 0xFFFFu; }

// positions of the scheduled searches i, i + 64, ..., i + 448 (clamped to the last one): the eight table
// loads are issued together (left to itself hipcc waits for each before it computes the next address)
template <bool TAIL>
SQ_DEV void cert_pos8(const CertCtx& cx, u32 i, u32 nCert, u32 (&sp)[8]) {
    u32 t[8], ad[8];
#pragma unroll
    for (u32 g = 0; g < 8; g++) {
        u32 const j0 = (i + 64 * g < nCert) ? i + 64 * g : nCert - 1;
        bool const x = TAIL && j0 + 2 >= nCert; // (second pass: the two searches that are not on the schedule; as cert_pos_g)
        u32 const j = x ? 0u : j0;
        bool const first = j < cx.certCnt0;
        u32 const slot = first ? (cx.tab0 ? SQZ_SCHEDTAB_N : 0u) + j : SQZ_SCHEDTAB_N + cx.tabOff1 + (j - cx.certCnt0);
        t[g] = sq_gld32(cx.schedTab + 2 + slot);
        ad[g] = first ? cx.base0Pos : cx.base1Pos;
        if (x) ad[g] = 0x80000000u | (3u - (nCert - j0));
    }
    // all eight words pass through one empty asm: nothing that uses one of them can be scheduled before
    // the last load has been issued
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
    for (u32 g = 0; g < 8; g++) sp[g] = (ad[g] & 0x80000000u) ? cx.wakeLim + (ad[g] & 3u) : t[g] + ad[g];
}

// the three 2-byte windows of a scheduled value as keys of the scan's filter (see the scan).  A key's cell is the
// BYTE at (key & 0x1FFF) and its bit there is key >> 13: address and bit index are then one v_bfe each on the
// scanning side (no shift-and-mask pair), whatever byte offset the window sits at in its register.
SQ_DEV void vbloom_add3(SQ_L u32* const B, u32 v) {
#pragma unroll
    for (u32 k = 0; k < 3; k++) {
        u32 const key = (v >> (8 * k)) & 0xFFFFu;
        u32 const word = (key & 0x1FFFu) >> 2, bit = 8 * (key & 3u) + (key >> 13);
#if SQ_ON_DEVICE
        __hip_atomic_fetch_or(&B[word], 1u << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        B[word] |= 1u << bit;
#endif
    }
}

// cx: base, schedTab, certCnt0, nCert, base1Pos, tabOff1 (0), repOff, chunkEndPos are inputs;
// certEnd / certEnd1 the results.
// rawOut (or null): where the frame's first block header goes.  While the scan streams the chunk it also
// writes the raw blocks' payload there (3 bytes of header before each block), so a certified chunk is read
// twice (values at S, stream) instead of three times; cx.rawEnd = the position up to which that was done
// (SQZ_WINDOW_START = nothing).  Only for 16-byte aligned chunks: a lane's 16 source bytes then never
// straddle the chunk start or the block boundary.
// TAIL: the instantiation of stage 1's SECOND pass (fastpath_blocks; cx.tab0 == 3, see CertCtx): a copy of its own of the
// aligned scan without the fused hash, so that nothing of it costs the first pass's loops a register or a branch.
template <bool TAIL>
SQ_DEV void cert_lds_core(CertCtx& cx, FastLds* L, u8* rawOut) {
    cx.rawEnd = SQZ_WINDOW_START;
    cx.hashDone = 0;
    cx.wakeBad = 0;
    const u8* const base = cx.base;
    u32 const nCert = cx.nCert;
    u32 const blk1 = wv_uni((nCert > cx.certCnt0) ? cx.base1Pos : 0xFFFFFFFFu);
    u32 certEnd = cx.chunkEndPos, certEnd1 = cx.chunkEndPos;
    u32 const pEnd = wv_uni(cx.chunkEndPos - 3); // one past the last p with p + 4 <= chunkEnd
    const u32* const lvlWords = cx.schedTab + SQZ_SCHEDLVL_OFF;
    SQ_L u32* const bloom = (SQ_L u32*)L->bloom;
    SQ_L u16* const vals = (SQ_L u16*)L->vals; // 16-bit tags: with the bucket's 10 bits, 26 bits of the value
    SQ_L u32* const ends = (SQ_L u32*)L->ends;
    SQ_L u16* const idxs = (SQ_L u16*)L->idx;
    SQ_L u32* const hits = (SQ_L u32*)L->hits;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u32 const wv = fw_wave();
    u32 const tid = 64 * wv + lane;
    constexpr u32 NW = FW_NW, NT = 64 * FW_NW;
    constexpr u32 ITS_ALL = SQZ_FAST_VALS / 512;      // batches of 512 scheduled positions
    constexpr u32 ITS = (ITS_ALL + NW - 1) / NW;      // ... of which this wave takes numbers wv, wv + NW, ...
    // set bits of m below this lane (v_mbcnt: no per-lane mask register to keep alive through the scan)
    auto rankIn = [&](u64 m) -> u32 { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); };
    SQ_L u32* const cand = (SQ_L u32*)L->cand + 256 * wv;
    SQ_L u32* const sh = (SQ_L u32*)L->sh;
    for (u32 i = tid; i < SQZ_BLOOM_WORDS; i += NT) bloom[i] = 0;
    for (u32 i = tid; i < SQZ_FAST_KEYS / 2 + 2; i += NT) ends[i] = 0;
    if (tid == 0) {
        sh[FW_SH_NHITS] = 0;
        sh[FW_SH_CERT0] = certEnd;
        sh[FW_SH_CERT1] = certEnd1;
        sh[FW_SH_FLAGS] = 0;
    }
    fw_sync();
    u32 vmin = 0xFFFFFFFFu, vmin1 = 0xFFFFFFFFu; // lane-local minima of the violating scheduled positions
    auto note = [&](u32 sviol, u32& v0, u32& v1) {
        if (sviol < blk1)
            v0 = sviol < v0 ? sviol : v0;
        else
            v1 = sviol < v1 ? sviol : v1;
    };
    auto settle = [&](u32 v0, u32 v1) { // wave minima of the violations -> the workgroup's certEnd / certEnd1
        if (wv_ballot((v0 & v1) != 0xFFFFFFFFu)) {
            u32 x = v0, z = v1;
            for (int sft = 32; sft >= 1; sft >>= 1) {
                u32 const y = (u32)__shfl_xor((int)x, sft, 64), w = (u32)__shfl_xor((int)z, sft, 64);
                x = y < x ? y : x;
                z = w < z ? w : z;
            }
            if (lane == 0) {
                __hip_atomic_fetch_min(&sh[FW_SH_CERT0], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(&sh[FW_SH_CERT1], z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    auto pushHit = [&](bool have, u32 p, u32 vp) -> bool { // false: the list is full
        u64 const m = wv_ballot(have);
        if (!m) return true;
        u32 const cnt = (u32)__popcll(m);
        u32 at = 0;
        if (lane == 0) at = __hip_atomic_fetch_add(&sh[FW_SH_NHITS], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        at = wv_uni(at);
        u32 const slot = at + rankIn(m);
        if (have && slot < SQZ_FAST_HITS) {
            hits[2 * slot] = p;
            hits[2 * slot + 1] = vp;
        }
        return at + cnt <= SQZ_FAST_HITS;
    };
    FPROF_T0(t_build);
    // ---- build pass 1: Bloom filter, bucket counts, repcode test.  The values at the scheduled positions are
    // gathered ONCE (an 8-byte load gives the bytes at s and at s + 1) and stay in registers for the scatter
    // below: every gather is a 64-byte sector of HBM, 3,500 of them per chunk and pass.
    u32 const nBuild = (cx.dbg & 1) ? 0u : nCert; // SQ_FAST_DBG timing experiments: 1 = no build, 2 = no scan
    u32 vv[8 * ITS];
    if (cx.repOff <= 1) {
        // all positions first (short trips to the L2-resident table), then ALL gathers in flight at
        // once: one trip to HBM per chunk
        u32 spA[8 * ITS];
        u64 q[8 * ITS];
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
            u32 sp[8];
#pragma unroll
            for (u32 g = 0; g < 8; g++) sp[g] = 0;
            if (i0 < nBuild) cert_pos8<TAIL>(cx, i0 + lane, nCert, sp);
#pragma unroll
            for (u32 g = 0; g < 8; g++) spA[8 * it + g] = sp[g];
        }
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
#pragma unroll
            for (u32 g = 0; g < 8; g++) q[8 * it + g] = 0;
            if (i0 < nBuild) {
#pragma unroll
                for (u32 g = 0; g < 8; g++) q[8 * it + g] = sq_grd64(base + spA[8 * it + g]); // s + 8 <= block end
            }
        }
        wv_fence();
        u64 repBits = 0; // (the upper halves of the 8-byte loads are dead after this loop)
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 lo32[8];
#pragma unroll
            for (u32 g = 0; g < 8; g++) lo32[g] = (u32)q[8 * it + g];
            asm volatile("" : "+v"(lo32[0]), "+v"(lo32[1]), "+v"(lo32[2]), "+v"(lo32[3]), "+v"(lo32[4]), "+v"(lo32[5]), "+v"(lo32[6]), "+v"(lo32[7]));
#pragma unroll
            for (u32 g = 0; g < 8; g++) {
                vv[8 * it + g] = lo32[g];
                // repcode test (offset 1) of the search at s: the bytes at s + 1 equal the bytes at s
                if ((u32)(q[8 * it + g] >> 8) == lo32[g]) repBits |= 1ull << (8 * it + g);
            }
        }
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
            if (i0 < nBuild) {
#pragma unroll
                for (u32 g = 0; g < 8; g++) {
                    u32 vg = vv[8 * it + g];
                    bool rep = (repBits >> (8 * it + g)) & 1;
                    if (cx.dbg & 32) { // timing experiment: made-up values instead of the gathered ones
                        vg = spA[8 * it + g] * 2654435761u;
                        rep = false;
                        vv[8 * it + g] = vg;
                    }
                    if (i0 + 64 * g + lane < nCert) {
                        vbloom_add3(bloom, vg);
                        (void)fast_cnt16_add(ends, vg >> 22);
                        if (cx.repOff && rep) note(spA[8 * it + g], vmin, vmin1); // repcode (offset 1) at s+1
                    }
                }
            }
        }
    } else {
        // (a tail certificate: repcode offset of a match.  As above -- positions, then all gathers at once -- with one more
        // gather per search: the four bytes its repcode test compares with)
        u32 spA[8 * ITS];
        u64 q[8 * ITS];
        u32 vr[8 * ITS];
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
            u32 sp[8];
#pragma unroll
            for (u32 g = 0; g < 8; g++) sp[g] = 0;
            if (i0 < nBuild) cert_pos8<TAIL>(cx, i0 + lane, nCert, sp);
#pragma unroll
            for (u32 g = 0; g < 8; g++) spA[8 * it + g] = sp[g];
        }
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
#pragma unroll
            for (u32 g = 0; g < 8; g++) q[8 * it + g] = 0, vr[8 * it + g] = 0;
            if (i0 < nBuild) {
#pragma unroll
                for (u32 g = 0; g < 8; g++) {
                    q[8 * it + g] = sq_grd64(base + spA[8 * it + g]); // s + 8 <= block end
                    vr[8 * it + g] = sq_grd32(base + spA[8 * it + g] + 1 - cx.repOff);
                }
            }
        }
        wv_fence();
#pragma unroll
        for (u32 it = 0; it < ITS; it++) {
            u32 const i0 = 512 * (it * NW + wv);
#pragma unroll
            for (u32 g = 0; g < 8; g++) {
                u32 const vg = (u32)q[8 * it + g];
                vv[8 * it + g] = vg;
                if (i0 < nBuild && i0 + 64 * g + lane < nCert) {
                    vbloom_add3(bloom, vg);
                    (void)fast_cnt16_add(ends, vg >> 22);
                    if ((u32)(q[8 * it + g] >> 8) == vr[8 * it + g]) note(spA[8 * it + g], vmin, vmin1); // repcode at s+1
                }
            }
        }
    }
    // second pass: the bucket check (CertCtx::wakeLim) as members of the set.  A position in front of the search that shares
    // the search's first byte AND its bucket holds one of the 2^(24 - hashLog) values  v = y * K^-1,  y = bucket << (32 - hashLog)
    // | t << 8 | (first byte * K & 0xFF)  (K: hash4's odd multiplier): they join S' as entries of their own (index 0xFF00 + t),
    // the scan then finds such a position like any other repeat and the hits' resolution tells it apart.
    static_assert((u32)(2654435761u * 0x0E8B2F51u) == 1u, "inverse of hash4's multiplier");
    u32 const nPseudo = (TAIL && cx.wakeLim) ? 1u << (24u - cx.wakeHashLog) : 0u;
    u32 const hlog = cx.wakeHashLog; // (the size class's, not a per-pass value)
    auto pseudoVal = [&](u32 t, u32 hashAndByte) -> u32 { // hashAndByte: bucket | first byte << 24
        u32 const y = ((hashAndByte & 0xFFFFFFu) << (32u - hlog)) | (t << 8) | (((hashAndByte >> 24) * 2654435761u) & 0xFFu);
        return y * 0x0E8B2F51u;
    };
    // (the check's parameters wait in LDS for the hits' resolution: registers are short during the scan)
    if (tid == 0) {
        sh[FW_SH_WAKE] = cx.wakeLim;
        sh[FW_SH_WAKE + 2] = cx.wakeSkip;
        sh[FW_SH_WAKE + 3] = cx.wakeHash | (cx.wakeByte << 24);
    }
    if (tid < nPseudo) {
        u32 const pv = pseudoVal(tid, cx.wakeHash | (cx.wakeByte << 24));
        vbloom_add3(bloom, pv);
        (void)fast_cnt16_add(ends, pv >> 22);
    }
    fw_sync(); // every wave's counts are in
    // ---- prefix sum: counts -> bucket starts (16 buckets = 8 words per lane; one wave does it)
    if (wv == 0) {
        u32 w[8], c[16], tot = 0;
#pragma unroll
        for (u32 k = 0; k < 8; k++) w[k] = ends[8 * lane + k];
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            c[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            tot += c[k];
        }
        // structured data piles its values into a few buckets: that is not what this path is for
        u32 mx = 0;
#pragma unroll
        for (u32 k = 0; k < 16; k++) mx = c[k] > mx ? c[k] : mx;
        if (wv_ballot(mx > SQZ_FAST_BUCKET_MAX) != 0 && lane == 0) sh[FW_SH_FLAGS] = 2;
        u32 all = 0;
        u32 run = wv_excl_scan(tot, &all);
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            u32 const s0 = run;
            run += c[2 * k];
            u32 const s1 = run;
            run += c[2 * k + 1];
            ends[8 * lane + k] = s0 | (s1 << 16);
        }
    }
    fw_sync();
    if (sh[FW_SH_FLAGS] & 2) { // no certificate (every wave takes this exit)
        cx.certEnd = SQZ_WINDOW_START;
        cx.certEnd1 = (blk1 != 0xFFFFFFFFu) ? SQZ_WINDOW_START : 0u;
        return;
    }
    // ---- build pass 2: scatter the values (from registers); a bucket's start counts up to its end
#pragma unroll
    for (u32 it = 0; it < ITS; it++) {
        u32 const i0 = 512 * (it * NW + wv);
        if (i0 < nBuild) {
#pragma unroll
            for (u32 g = 0; g < 8; g++)
                if (i0 + 64 * g + lane < nCert) {
                    u32 const vg = vv[8 * it + g];
                    u32 const slot = fast_cnt16_add(ends, vg >> 22);
                    if (slot < SQZ_FAST_VALS) {
                        vals[slot] = (u16)(vg >> 6);
                        idxs[slot] = (u16)(i0 + 64 * g + lane);
                    }
                }
        }
    }
    if (tid < nPseudo) {
        u32 const pv = pseudoVal(tid, cx.wakeHash | (cx.wakeByte << 24));
        u32 const slot = fast_cnt16_add(ends, pv >> 22);
        if (slot < SQZ_FAST_VALS) {
            vals[slot] = (u16)(pv >> 6);
            idxs[slot] = (u16)(0xFF00u + tid);
        }
    }
    fw_sync();
    // ---- scheduled positions with equal values: equal neighbours inside a bucket
    for (u32 k0 = 64 * wv; k0 < ((cx.dbg & 128) ? 0u : SQZ_FAST_KEYS); k0 += NT) { // 128: timing experiment
        u32 const k = k0 + lane;
        u32 const lo = k ? fast_get16(ends, k - 1) : 0u, hi = fast_get16(ends, k);
        {   // quick look first: a bucket holds 3.4 values on average; its first 12 are read with independent
            // loads and compared pairwise in registers.  Only when some bucket of the 64 has equal tags (or
            // more than 12 values) does the exact loop below run.
            u32 t[12];
#pragma unroll
            for (u32 a = 0; a < 12; a++) {
                u32 const x = vals[(lo + a < SQZ_FAST_VALS) ? lo + a : SQZ_FAST_VALS - 1];
                t[a] = (lo + a < hi) ? x : 0x10000u + a; // beyond the bucket: a value no tag (and no other filler) equals
            }
            bool any = hi > lo + 12;
#pragma unroll
            for (u32 a = 1; a < 12; a++)
#pragma unroll
                for (u32 c = 0; c < a; c++) any = any || (t[a] == t[c]);
            if (!wv_ballot(any)) continue;
        }
        // every value that occurs twice in the bucket is reported once (at its second occurrence)
        u32 b = lo + 1;
        for (;;) {
            bool dup = false;
            u32 dupv = 0;
            for (; b < hi && !dup; b++) {
                u32 const vb = vals[b];
                u32 seen = 0;
                for (u32 a = lo; a < b; a++) seen += (vals[a] == vb) ? 1u : 0u;
                if (seen == 1) { // exactly one earlier copy: first time these 26 bits repeat
                    dup = true;
                    dupv = (k << 22) | (vb << 6); // the bits 6..31 they share; resolved on full values below
                }
            }
            if (!wv_ballot(dup)) break;
            if (!pushHit(dup, 0u, dupv) && lane == 0)
                __hip_atomic_fetch_or(&sh[FW_SH_FLAGS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    settle(vmin, vmin1);
    fw_sync();
    certEnd = sh[FW_SH_CERT0];
    certEnd1 = sh[FW_SH_CERT1];
    if (sh[FW_SH_FLAGS] & 1) { // cannot even list the repeats inside the schedule: no certificate
        certEnd = certEnd1 = SQZ_WINDOW_START;
    }
    FPROF_ADD(L, 0, t_build);
    // ---- scan
    const SQ_L u32* const B = bloom;
    // (wv_uni: the chunk's address and length reach this function through vector loads of off[] / len[]; values derived
    // from them are wave-uniform but sit in vector registers, and hipcc then computes every address of the scan with
    // 64-bit VALU adds per lane.  One v_readfirstlane each puts them where the scalar unit can do that arithmetic.)
    const u8* const src0 = (const u8*)wv_uni64((u64)(size_t)(base + SQZ_WINDOW_START));
    const u8* const a0 = src0 - ((size_t)src0 & 15);
    s32 const p0 = (s32)SQZ_WINDOW_START - (s32)(src0 - a0); // position of a0 (may be negative)
    u32 const chunkEndU = wv_uni(cx.chunkEndPos), base0U = wv_uni(cx.base0Pos), tailU = wv_uni(cx.tail), tab0U = wv_uni(cx.tab0);
    const u8* const lastByte = src0 - SQZ_WINDOW_START + chunkEndU - 1;
    const u8* const aLast = lastByte - ((size_t)lastByte & 15);
    rawOut = (u8*)wv_uni64((u64)(size_t)rawOut);
    bool const rawCopy = !TAIL && rawOut != nullptr && src0 == a0;
    // the fused hash needs the whole chunk streamed (a scan that would stop at an early certificate end goes on:
    // what it finds beyond that end changes nothing) and rounds that are XXH3 blocks (aligned chunk)
    bool const hashOn = !TAIL && cx.hashPool != nullptr && src0 == a0 && !(cx.dbg & 2); // (not in the second pass)
    auto scanStop = [&]() -> u32 {
        u32 const ce = (blk1 != 0xFFFFFFFFu) ? certEnd1 : certEnd;
        return (ce < pEnd && !hashOn) ? ce : pEnd;
    };
    // the rounds of 1 KiB from p0 up to the stop are dealt out in contiguous runs, one per wave
    u32 const nRounds = wv_uni(p0 < (s32)scanStop() ? ((u32)((s32)scanStop() - p0) + 1023u) >> 10 : 0u);
    // (fused hash: runs of a multiple of four rounds, so that only the last wave's last loop step can be short)
    u32 const perWave = hashOn ? ((nRounds + NW - 1) / NW + SQZ_FAST_DEPTH_HS - 1) / SQZ_FAST_DEPTH_HS * SQZ_FAST_DEPTH_HS : (nRounds + NW - 1) / NW;
    auto runStart = [&](u32 w) -> s32 { return p0 + (s32)(1024u * (perWave * w < nRounds ? perWave * w : nRounds)); };
    s32 pos = (s32)wv_uni((u32)runStart(wv));
    s32 const myStop = (s32)wv_uni((u32)((runStart(wv + 1) < (s32)scanStop()) ? runStart(wv + 1) : (s32)scanStop()));
    u32 const lastOff = (u32)(aLast - a0);
    auto loadRound = [&](s32 rp) -> sq_v4u { // uniform base + 32-bit lane offset, clamped to the last window
        u32 const o = (u32)(rp - p0) + 16 * lane;
#if SQZ_FAST_NT & 1
        return sq_grd128_nt(a0 + (o < lastOff ? o : lastOff));
#else
        return sq_grd128(a0 + (o < lastOff ? o : lastOff));
#endif
    };
    // level words of the lane's 16 positions in the round at rp (two, for a window across two levels)
    // Level words for the round at rp.  A round spans 1024 bytes = at most 5 levels of one block (rounds
    // that touch the block boundary or the dense start take the general routine, which loads its own):
    // lane l loads word (k0 + (l & 7)), k0 = level of the round's first byte; pickLevels() hands every
    // lane the two words of its own 16 positions.
    // AL (std::true_type): the chunk starts on a 16-byte boundary, so a lane's 16 positions are one window
    // of the schedule's bitmap (SQZ_SCHEDBITS_OFF): `wl` is then that window's mask itself, loaded with the
    // stream, and none of the level arithmetic below runs.
    const u16* const schedBits = (const u16*)wv_uni64((u64)(size_t)(cx.schedTab + SQZ_SCHEDBITS_OFF));
    auto loadLevels = [&](auto AL, auto HS, s32 rp, u32& wl) {
        if constexpr (decltype(AL)::value) {
            // (rounds in front of the first certified block -- the earlier blocks of a large chunk, base0Pos > the
            // chunk's start -- hold no scheduled position of THIS certificate: their word is loaded like any other, from
            // a clamped index, and dropped in round())
            bool const second = (blk1 != 0xFFFFFFFFu) && rp >= (s32)blk1;
            u32 const tabSel = second ? SQZ_SCHEDBITS_N : (tab0U ? SQZ_SCHEDBITS_N : 0u);
            // (per lane and signed: in the second pass base0U is an anchor at any byte offset, and the round that contains it
            // has lanes on both sides -- those behind it need their own word, r >> 4 = the word around the lane's first position)
            s32 const r = rp + (s32)(16 * lane) - (s32)(second ? blk1 : base0U);
            u32 const wi = r > 0 ? (u32)r >> 4 : 0u;
            // (a 4-byte load of the 2-byte word: the upper half -- the next window's word -- never reaches the result,
            // the masks it is combined with have 16 bits; as a 2-byte load the value's zero extension became an
            // instruction of its own that hipcc placed at the loop's back edge, behind a wait for ALL loads in flight)
            wl = sq_grd32((const u8*)(schedBits + tabSel + (wi < SQZ_SCHEDBITS_N ? wi : SQZ_SCHEDBITS_N - 1)));
            return;
        }
        bool const second = (blk1 != 0xFFFFFFFFu) && rp >= (s32)blk1;
        u32 const bs = second ? blk1 : SQZ_WINDOW_START;
        s32 const r0 = rp - (s32)bs;
        u32 const k0 = r0 > 0 ? (u32)r0 >> 8 : 0u;
        u32 const k = k0 + (lane & 7);
        wl = sq_gld32(lvlWords + (second ? SQZ_SCHEDLVL_N : 0u) + (k < SQZ_SCHEDLVL_N ? k : SQZ_SCHEDLVL_N - 1));
    };
    auto pickLevels = [&](s32 rp, u32 wl, u32& wa, u32& wb) {
        bool const second = (blk1 != 0xFFFFFFFFu) && rp >= (s32)blk1;
        u32 const bs = second ? blk1 : SQZ_WINDOW_START;
        s32 const r0 = rp - (s32)bs;
        u32 const k0 = r0 > 0 ? (u32)r0 >> 8 : 0u;
        s32 const r = r0 + (s32)(16 * lane);
        u32 const kA = r > 0 ? (u32)r >> 8 : 0u, kB = r + 15 > 0 ? (u32)(r + 15) >> 8 : 0u;
        wa = (u32)__shfl((int)wl, (int)((kA - k0) & 7), 64);
        wb = (u32)__shfl((int)wl, (int)((kB - k0) & 7), 64);
    };
    auto valAt = [&](sq_v4u w, u32 d4, u32 j) -> u32 {
        u32 const a = j >> 2;
        u32 const lo = a == 0 ? w.x : a == 1 ? w.y : a == 2 ? w.z : w.w;
        u32 const hi = a == 0 ? w.y : a == 1 ? w.z : a == 2 ? w.w : d4;
        return __builtin_amdgcn_alignbyte(hi, lo, j & 3);
    };
    // is vp among the scheduled values?  (its bucket: ~3.4 values)
    auto inSet = [&](u32 vp) -> bool {
        u32 const k = vp >> 22;
        u32 const lo = k ? fast_get16(ends, k - 1) : 0u, hi = fast_get16(ends, k);
        // the first 8 values of the bucket with independent reads (one LDS latency), the rest in a loop
        u32 b[8];
#pragma unroll
        for (u32 a = 0; a < 8; a++) b[a] = vals[(lo + a < SQZ_FAST_VALS) ? lo + a : SQZ_FAST_VALS - 1];
        u32 const tp = (vp >> 6) & 0xFFFFu;
        bool f = false;
#pragma unroll
        for (u32 a = 0; a < 8; a++) f = f || (lo + a < hi && b[a] == tp);
        for (u32 a = lo + 8; a < hi; a++) f = f || (vals[a] == tp);
        return f;
    };
    // Scheduled positions among a lane's 16, for windows whose levels are all >= 16 (one candidate per
    // level part, no loop, no division); anything else (first 4 KiB of a block, the window across the
    // block boundary, the last windows of a block) takes the general routine.
    auto schedMaskFast = [&](s32 lo, u32 wa, u32 wb) -> u32 {
        bool const second = (blk1 != 0xFFFFFFFFu) && lo >= (s32)blk1;
        u32 const bs = second ? blk1 : SQZ_WINDOW_START;
        u32 const blkEnd = (!second && blk1 != 0xFFFFFFFFu) ? blk1 : chunkEndU;
        s32 const r = lo - (s32)bs;
        bool const plain = r >= 4096 && (u32)(lo + 16) <= blkEnd - 8 && !((blk1 != 0xFFFFFFFFu) && lo < (s32)blk1 && lo + 16 > (s32)blk1);
        if (wv_ballot(!plain)) {
            u32 const m = fast_sched_mask_rare(lvlWords, lo, blk1, chunkEndU, wa, wb, plain ? 1u : 0u);
            // This rare path loads level words itself.  Left pending at the join, such a load makes hipcc
            // guard the main path's registers with s_waitcnt vmcnt(0) in EVERY round, which drains the four
            // rounds of stream loads in flight (measured: the scan ran at one memory latency per round).
            // A real s_waitcnt here (vmcnt(0), expcnt/lgkmcnt untouched) closes the path with nothing pending.
            __builtin_amdgcn_s_waitcnt(0x0F70);
            return m;
        }
        return fast_sched_mask_plain((u32)r, wa, wb);
    };
    // look the collected positives up, 64 at a time, while at least `atLeast` are waiting; false = list of hits full
    u32 nCand = 0;
    s32 pendFrom = pos; // first position whose positives have not all been looked up yet
    auto lookUp = [&](u32 atLeast) -> bool {
        bool ok = true;
        FPROF_T0(t_look);
        if (cx.dbg & 256) nCand = 0; // timing experiment: positives dropped
        while (nCand >= atLeast && nCand) {
            u32 const take = nCand < 64 ? nCand : 64, j0 = nCand - take; // from the end of the list
            bool const have = lane < take;
            u32 const hp = have ? cand[2 * (j0 + lane)] : 0u, hv = have ? cand[2 * (j0 + lane) + 1] : 0u;
            bool const hit = have && inSet(hv);
            if (!pushHit(hit, hp, hv)) ok = false;
            nCand = j0;
        }
        FPROF_ADD(L, 2, t_look);
        return ok;
    };
    // one round: Bloom tests, scheduled positions removed, positives collected; false = list of hits full
    // fused hash: this lane's slot of the wave's staging area (rows of 64 B: lanes 0..3 of each row write)
    SQ_L u32* const hstWave = (SQ_L u32*)L->hstage + wv * (SQZ_FAST_DEPTH_HS * 64u);
    // byte address (LDS) of this lane's staging slot for round 0 of a step: rows of 16 lanes, lanes 0..3 of a row hold its sums
    u32 const hstLaneAddr = (u32)(size_t)hstWave + 4u * (((lane & 0x30u) >> 2) | (lane & 3u));
    const SQ_L u32* const hsecLane = (const SQ_L u32*)L->hsec + lane; // word k of the lane's secret at [64 k + lane]: four
                                                                      // conflict-free 4-byte reads (a 16-byte read per lane costs ~40 LDS cycles)
    // fused hash: the loop always runs whole steps of SQZ_FAST_DEPTH_HS rounds with ONE exit test per step (exits
    // after single rounds made hipcc merge them with the loop's latch in a block of register copies of the stream
    // loads in flight, i.e. a wait for all of them in every step).  Rounds beyond the wave's run, and all rounds
    // once the filter part has given up (hsDead: the list of hits is full), only hash and copy.
    bool hsDead = false;
    u32 const lane16 = 16u * lane, lane2 = 2u * lane; // the constant per-lane offsets behind the scan's wave-uniform pointers
    // ---- the pieces of a round (every lambda is inlined at each of its uses)
    // the dword behind the lane's 16 bytes: lane l + 1's first, for lane 63 the next round's lane 0
    auto nextDword = [&](sq_v4u w, sq_v4u wn) __attribute__((always_inline)) -> u32 {
        // lane 63 continues in lane 0 of the next round's load: a wave rotation, no trip through an SGPR
        // (a v_readfirstlane of that vector gets hoisted into earlier rounds, where it waits for a load
        // that is meant to stay in flight)
        u32 const n63 = (u32)__builtin_amdgcn_update_dpp(0, (int)wn.x, 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
        // wave_shl:1 has no source for lane 63, which therefore keeps `old` (bound_ctrl off): the rotated word -- two DPP
        // moves, no select, no lane mask to keep in scalar registers
        return (u32)__builtin_amdgcn_update_dpp((int)n63, (int)w.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
    };
    // The filter's probes of the value at p are keyed by its three 2-byte windows (at p, p + 1, p + 2) in
    // ONE table: a probe of position p is also a probe of its neighbours.  So each of the lane's 18
    // windows (16 of its own bytes, 2 reaching into the next lane's) is looked up once, and a position
    // passes when the bits of its own window and of the next two are set: three probes (0.3 % false
    // positives at 10,500 keys in 64 Kibit) for the price of one.  Returns the 16 positions' verdicts.
    // The 18 answers are collected Horner-wise (one v_lshl_or each), i.e. in REVERSE order: window j ends up
    // at bit 17 - j, the three-window AND works the same on the mirrored word, and one v_bfrev turns it back.
    // Order of issue: all 18 addresses, all 18 reads, then the 18 bit indices (VALU work that covers the reads'
    // latency), then the answers.  (Compiled as written above -- four reads, their answers, the next four --
    // a round paid five LDS round trips one after the other.)
    auto filterBits = [&](sq_v4u w, u32 d4) __attribute__((always_inline)) -> u32 {
        u32 const d[5] = {w.x, w.y, w.z, w.w, d4};
        u32 src[18], pw[18];
#pragma unroll
        for (u32 q = 0; q < 5; q++) {
            u32 const y = q < 4 ? __builtin_amdgcn_alignbyte(d[q < 4 ? q + 1 : q], d[q], 3) : 0u; // the window across two words
#pragma unroll
            for (u32 k = 0; k < (q < 4 ? 4u : 2u); k++) src[4 * q + k] = k < 3 ? d[q] : y;
        }
        // (the reads and their one wait are spelled out: hipcc's scheduler keeps at most eight of them in flight and
        // puts the waits where it likes; lgkmcnt(0) is right whatever else is outstanding)
        u32 const ldsB = (u32)(size_t)B;
#pragma unroll
        for (u32 j = 0; j < 18; j++) {
            u32 const ad = __builtin_amdgcn_ubfe(src[j], (j & 3) < 3 ? 8 * (j & 3) : 0u, 13u) + ldsB;
            asm volatile("ds_read_u8 %0, %1" : "=v"(pw[j]) : "v"(ad));
        }
        u32 bi[18];
#pragma unroll
        for (u32 j = 0; j < 18; j++) bi[j] = __builtin_amdgcn_ubfe(src[j], ((j & 3) < 3 ? 8 * (j & 3) : 0u) + 13u, 3u);
        asm volatile("" : "+v"(bi[0]), "+v"(bi[1]), "+v"(bi[2]), "+v"(bi[3]), "+v"(bi[4]), "+v"(bi[5]), "+v"(bi[6]), "+v"(bi[7]), "+v"(bi[8]),
                          "+v"(bi[9]), "+v"(bi[10]), "+v"(bi[11]), "+v"(bi[12]), "+v"(bi[13]), "+v"(bi[14]), "+v"(bi[15]), "+v"(bi[16]), "+v"(bi[17]));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(pw[0]), "+v"(pw[1]), "+v"(pw[2]), "+v"(pw[3]), "+v"(pw[4]), "+v"(pw[5]), "+v"(pw[6]), "+v"(pw[7]), "+v"(pw[8]),
                       "+v"(pw[9]), "+v"(pw[10]), "+v"(pw[11]), "+v"(pw[12]), "+v"(pw[13]), "+v"(pw[14]), "+v"(pw[15]), "+v"(pw[16]), "+v"(pw[17]));
        u32 bm = 0;
#pragma unroll
        for (u32 j = 0; j < 18; j++) {
            u32 const bit = __builtin_amdgcn_ubfe(pw[j], bi[j], 1u);
            // (spelled out: left to itself hipcc turns the chain into shifts and three-way ORs, 7 instead of 4 per word)
            asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(bm) : "v"(bm), "v"(bit));
        }
        return __builtin_bitreverse32(bm & (bm << 1) & (bm << 2)) >> 14; // (16 bits: windows j, j + 1, j + 2 of position j)
    };
    // compaction (one positive per lane and turn) into the list that collects the positives of several
    // rounds: a look-up is a chain of dependent LDS reads, paid once per 64 positives instead of once
    // per round.  false = the list of hits is full
    auto collect = [&](u32 cm, s32 lo, sq_v4u w, u32 d4) __attribute__((always_inline)) -> bool {
        bool ok = true;
        for (;;) {
            u64 const m = wv_ballot(cm != 0);
            if (!m) break;
            // no room: look the waiting ones up now.  (Rare: the caller empties the list every five rounds;
            // only where every 2nd / 3rd position is a scheduled one -- bytes 256..1023 of a block -- do all
            // the positions between them pass the three-window filter, ~300 in one round.)
            if (nCand + (u32)__popcll(m) > 128 && !lookUp(1)) ok = false;
            if (cm) {
                u32 const j = (u32)__builtin_ctz(cm);
                cm &= cm - 1;
                u32 const slot = nCand + rankIn(m);
                if (slot < 128) {
                    cand[2 * slot] = (u32)(lo + (s32)j);
                    cand[2 * slot + 1] = valAt(w, d4, j);
                }
            }
            nCand += (u32)__popcll(m);
        }
        wv_fence();
        return ok;
    };
    // XXH3: the eight sums of this 1 KiB block (XXH3_accumulate_512 over its 16 stripes, without the
    // accumulators): two per lane, then over the four stripes of the row; lanes 0..3 of each row stage the row's four
    // pairs of sums (64 B per row) for the step's flush.  (The predicate and the slot's offset are recomputed each
    // round -- four instructions -- rather than kept in registers through the loop, which is at its limit of 128
    // vector and ~100 scalar registers: hence the laundered lane number.)
    auto hashRound = [&](auto IR, sq_v4u w) __attribute__((always_inline)) {
        sq_v4u const ks = (sq_v4u){hsecLane[0], hsecLane[64], hsecLane[128], hsecLane[192]};
        u64 const d0 = ((u64)w.y << 32) | w.x, d1 = ((u64)w.w << 32) | w.z;
        u64 const k0 = d0 ^ (((u64)ks.y << 32) | ks.x), k1 = d1 ^ (((u64)ks.w << 32) | ks.z);
        u64 c0 = (u64)(u32)k0 * (u64)(u32)(k0 >> 32) + d1;
        u64 c1 = (u64)(u32)k1 * (u64)(u32)(k1 >> 32) + d0;
        u32 a0_ = (u32)c0, a1_ = (u32)(c0 >> 32), b0_ = (u32)c1, b1_ = (u32)(c1 >> 32);
        SQZ_ROW_SUM4(a0_, a1_, b0_, b1_);
        // lanes 0..3 of each row -- exec = 0x000F000F000F000F, a literal: no compare, no mask register to keep -- store
        // word k of (round, slot) at [64 k + 16 round + slot]: the flush reads [64 k + lane].  (Wave-uniform code: exec is
        // all ones here and is put back; LDS executes a wave's accesses in order, the flush's reads come behind these.)
        {
            u32 const hwA = hstLaneAddr + 64u * decltype(IR)::value; // byte address of this lane's slot in round IR's row group
            u64 keep;
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b32 exec_lo, 0x000f000f\n\t"
                         "s_mov_b32 exec_hi, 0x000f000f\n\t"
                         "ds_write2st64_b32 %1, %2, %3 offset1:1\n\t"
                         "ds_write2st64_b32 %1, %4, %5 offset0:2 offset1:3\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(keep)
                         : "v"(hwA), "v"(a0_), "v"(a1_), "v"(b0_), "v"(b1_)
                         : "memory");
        }
    };
    // where the raw-block payload of the round at rp goes (block b's payload sits 3 (b + 1) bytes behind its place in
    // the chunk: every block but the last is full)
    auto rawDst = [&](u32 rp) __attribute__((always_inline)) -> u8* {
        return rawOut + (rp - SQZ_WINDOW_START) + 3u * (((rp - SQZ_WINDOW_START) >> 17) + 1u);
    };
    // one round: filter tests, scheduled positions removed, positives collected; false = list of hits full
    auto round = [&](auto AL, auto HS, auto IR, sq_v4u w, sq_v4u wn, u32 wl) -> bool {
        bool const active = !decltype(HS)::value || (!hsDead && pos < myStop);
        u32 wa = 0, wb = 0;
        if constexpr (!decltype(AL)::value) pickLevels(pos, wl, wa, wb);
        u32 const d4 = nextDword(w, wn);
        // (the fused-hash form always streams the whole chunk: its stop is the chunk's last position, no certificate end to keep)
        u32 const stop = decltype(HS)::value ? pEnd : scanStop();
        s32 const lo = pos + (s32)(16 * lane);
        u32 okMask = 0;
        if constexpr (decltype(AL)::value) {
            // aligned chunk: the round starts at or after the chunk start and lies in one block, so the
            // destination of the raw copy is a wave-uniform base plus 16 bytes per lane
            // (of a large chunk only the certified blocks are copied, the ones in front belong to other workgroups)
            if (rawCopy && (u32)pos >= base0U && (u32)lo + 16 <= chunkEndU) // (clamped loads beyond the end fail this test)
#if SQZ_FAST_NT & 2
                sq_gwr128_nt(rawDst((u32)pos) + 16 * lane, w);
#else
                sq_gwr128(rawDst((u32)pos) + 16 * lane, w);
#endif
        } else {
            s32 f = (s32)SQZ_WINDOW_START - lo, l = (s32)stop - lo;
            f = f < 0 ? 0 : (f > 16 ? 16 : f);
            l = l < 0 ? 0 : (l > 16 ? 16 : l);
            okMask = ((1u << l) - 1u) & ~((1u << f) - 1u);
        }
        u32 cm = filterBits(w, d4);
        if constexpr (decltype(AL)::value) {
            // the window's bitmap word; a block's searches stop 8 bytes before its end (a short last block
            // uses a prefix of the table)
            bool const second = (blk1 != 0xFFFFFFFFu) && pos >= (s32)blk1;
            u32 const lim = ((!second && blk1 != 0xFFFFFFFFu) ? blk1 : chunkEndU) - tailU;
            // An INTERIOR round -- all of its 1,024 positions inside the scanned range, the certified blocks and one
            // block's searched range: 250 of a 256 KiB chunk's 256 -- needs no per-lane clipping: the scheduled positions
            // are the bitmap word as loaded (its upper half, the next window's word, lies above cm's 16 bits).  The test
            // is wave-uniform and the edge rounds are kept behind a real branch (the asm statement: hipcc would otherwise
            // turn the clipping into twelve selects that every round executes).
            constexpr bool tailPass = TAIL; // (second pass: every general round is an edge round)
            bool const interior = active && !tailPass && (u32)pos + 1024 <= stop && (u32)pos + 1024 <= lim && (u32)pos >= base0U;
            if (interior)
                cm &= ~wl;
            else {
                asm volatile("; edge round");
                okMask = active ? 0xFFFFu : 0u;
                if ((u32)pos + 1024 > stop) {
                    s32 l = (s32)stop - lo;
                    l = l < 0 ? 0 : (l > 16 ? 16 : l);
                    okMask &= (1u << l) - 1u;
                }
                u32 sm = (u32)pos >= base0U ? wl : 0u;
                if constexpr (!decltype(HS)::value) {
                    if constexpr (tailPass) {
                        // the schedule's bitmap starts at base0U, which is no multiple of 16 away from the lane's window:
                        // bit k of the window = bit k + r of the bitmap, r = lo - base0U (word r >> 4 and the one behind it
                        // are in wl; in front of base0U nothing is scheduled)
                        // (a lane that straddles base0U: the schedule starts dense, its first 16 positions are all searched)
                        s32 const r = lo - (s32)base0U;
                        sm = r >= 0 ? (wl >> (r & 15)) & 0xFFFFu : (r > -16 ? (0xFFFFu << (u32)(-r)) & 0xFFFFu : 0u);
                    }
                }
                if ((u32)pos + 1024 > lim) {
                    s32 nb = (s32)lim - lo;
                    nb = nb < 0 ? 0 : (nb > 16 ? 16 : nb);
                    sm &= (1u << nb) - 1u;
                }
                cm &= okMask & ~sm;
            }
        } else
            cm &= okMask & ~schedMaskFast(lo, wa, wb);
        if (nCand == 0 && active) pendFrom = pos; // nothing older than this round is waiting for its look-up
        bool ok = collect(cm, lo, w, d4);
        if constexpr (decltype(HS)::value) hashRound(IR, w); // (here, after the compaction loop: the filter's 54 temporaries are dead)
        if (nCand > 128) { // not with a working filter on random data: treat like a full list of hits
            ok = false;
            cx.nOvf = 1u | (nCand << 8);
            nCand = 128;
        }
        pos += 1024;
        return ok;
    };
    // A PLAIN round of the fused-hash scan: all of its 1,024 positions lie inside the wave's run, inside the chunk and
    // inside the searched range of one block (fastStep() below vouches for four of them at a time), so nothing is
    // clipped per lane or tested per round: filter, the bitmap word as loaded, positives, hash, raw copy.
    auto roundPlain = [&](auto IR, sq_v4u w, sq_v4u wn, u32 wl, u64 dstp) __attribute__((always_inline)) -> bool {
        u32 const d4 = nextDword(w, wn);
        s32 const lo = pos + (s32)(16 * lane);
        if (rawCopy) // (dstp: the round's place in the frame, a scalar; 16 bytes per lane behind it)
#if SQZ_FAST_NT & 2
            sq_gwr128_nt((u8*)(size_t)dstp + lane16, w);
#else
            sq_gwr128((u8*)(size_t)dstp + lane16, w);
#endif
        u32 const cm = filterBits(w, d4) & ~wl;
        if (nCand == 0 && !hsDead) pendFrom = pos;
        bool ok = collect(cm, lo, w, d4);
        hashRound(IR, w);
        if (nCand > 128) {
            ok = false;
            cx.nOvf = 1u | (nCand << 8);
            nCand = 128;
        }
        pos += 1024;
        return ok;
    };
    // The same for the scan WITHOUT the fused hash (split form of a batch, sharded batches, the block pairs of large chunks).
    // INB (std::true_type): a plain round inside the certified blocks -- bitmap word, raw copy; false: a round in FRONT of
    // them (the earlier blocks of a large chunk, three quarters of the rounds a 2 MiB chunk's jobs scan): no scheduled
    // position, nothing to copy, no bitmap word loaded at all.
    auto roundPlainN = [&](auto INB, sq_v4u w, sq_v4u wn, u32 wl) __attribute__((always_inline)) -> bool {
        u32 const d4 = nextDword(w, wn);
        s32 const lo = pos + (s32)(16 * lane);
        u32 cm = filterBits(w, d4);
        if constexpr (decltype(INB)::value) {
            if (rawCopy) {
                u64 dstp = (u64)(size_t)rawDst((u32)pos);
                asm volatile("" : "+s"(dstp));
#if SQZ_FAST_NT & 2
                sq_gwr128_nt((u8*)(size_t)dstp + lane16, w);
#else
                sq_gwr128((u8*)(size_t)dstp + lane16, w);
#endif
            }
            // (second pass: the schedule's bitmap starts at base0U, no multiple of 16 away from the lane's window -- the same
            // shift for every lane and round; loadBitsU's four bytes hold the window's word and the one behind it)
            if constexpr (TAIL) wl >>= ((u32)pos - base0U) & 15u;
            cm &= ~wl;
        }
        if (nCand == 0) pendFrom = pos;
        bool ok = collect(cm, lo, w, d4);
        if (nCand > 128) {
            ok = false;
            cx.nOvf = 1u | (nCand << 8);
            nCand = 128;
        }
        pos += 1024;
        return ok;
    };
    FPROF_T0(t_scan);
    bool full = false, scanned = false;
    // fused hash: the sums of the (four) rounds of one loop step, staged in LDS as [round][row][pair] x 16 B = 1 KiB
    // -> the pool.  Lane 16 q + 4 row + j reads its 16 bytes, the four rows of a round are added up with the same two
    // DPP rotations as the stripes of a row, and the lanes of row 0 store round q's 64 bytes.  The other lanes (and the
    // rounds a short last step did not run) store into the pool's dump area instead: nothing here is conditional, so no
    // store is pending on one path only (hipcc would guard the stream loads of the next rounds conservatively).
    u64* const hpool = (u64*)wv_uni64((u64)(size_t)cx.hashPool);
    auto hflush = [&](u32 r, u32 blk0) {
        static_assert(SQZ_FAST_DEPTH_HS == 4, "one lane per (round, row, pair) of a step");
        wv_fence();
        u32 a0 = hstWave[lane], a1 = hstWave[64 + lane], b0 = hstWave[128 + lane], b1 = hstWave[192 + lane];
        SQZ_ROW_SUM4(a0, a1, b0, b1);
        u32 const q = lane >> 4, j = lane & 3;
        bool const live = q < r && (lane & 12) == 0;
        u32 const blk = live ? blk0 + q : 256u + ((lane >> 2) & 3);
        sq_gwr128((u8*)(hpool + 8 * blk + 2 * j), (sq_v4u){a0, a1, b0, b1});
        wv_fence();
    };
    // ---- the fused-hash scan's fast steps (aligned chunk, HS): is the step of four rounds at pos made of plain rounds,
    // and are the loads it issues for the three rounds behind it inside the chunk?  One wave-uniform test per step.
    u32 const loadEnd = (u32)(p0 + (s32)lastOff) + 16u; // one past the last 16-byte window the stream may load
    auto fastStep = [&]() -> bool {
        bool const second = (blk1 != 0xFFFFFFFFu) && pos >= (s32)blk1;
        u32 const lim = ((!second && blk1 != 0xFFFFFFFFu) ? blk1 : chunkEndU) - tailU; // end of the block's searched range
        u32 const e = lim < (u32)myStop ? lim : (u32)myStop;
        return !hsDead && !(cx.dbg & 1024) /* timing experiment: general steps only */ && pos >= (s32)base0U && !tab0U &&
               (u32)pos + 1024u * SQZ_FAST_DEPTH_HS <= e && (u32)pos + 1024u * (2u * SQZ_FAST_DEPTH_HS - 1u) <= loadEnd;
    };
    // ... and of the scan without the hash (steps of SQZ_FAST_DEPTH rounds): 1 = plain rounds inside the certified blocks,
    // 2 = rounds in front of them, with every round the step prefetches in front of them too (their bitmap words are not
    // loaded), 0 = neither (a general step)
    auto fastKindN = [&]() -> u32 {
        if ((cx.dbg & 1024) || (u32)pos + 1024u * (2u * SQZ_FAST_DEPTH - 1u) > loadEnd) return 0u;
        u32 const stepEnd = (u32)pos + 1024u * SQZ_FAST_DEPTH;
        u32 const stp = scanStop();
        if (stepEnd > (u32)myStop || stepEnd > stp) return 0u;
        if (pos >= (s32)base0U) {
            bool const second = (blk1 != 0xFFFFFFFFu) && pos >= (s32)blk1;
            u32 const lim = ((!second && blk1 != 0xFFFFFFFFu) ? blk1 : chunkEndU) - tailU;
            return stepEnd <= lim ? 1u : 0u;
        }
        return (u32)pos + 1024u * (2u * SQZ_FAST_DEPTH - 1u) <= base0U ? 2u : 0u;
    };
    auto loadStreamU = [&](s32 rp) __attribute__((always_inline)) -> sq_v4u { // (unclamped: the round lies inside the chunk)
        u64 sp = (u64)(size_t)a0 + (u32)(rp - p0);
        asm volatile("" : "+s"(sp));
#if SQZ_FAST_NT & 1
        return sq_grd128_nt((const u8*)(size_t)sp + lane16);
#else
        return sq_grd128((const u8*)(size_t)sp + lane16);
#endif
    };
    auto loadBitsU = [&](s32 rp) __attribute__((always_inline)) -> u32 { // (as loadLevels(AL), index unclamped: inside a block)
        bool const second = (blk1 != 0xFFFFFFFFu) && rp >= (s32)blk1;
        u32 const tabSel = second ? SQZ_SCHEDBITS_N : (tab0U ? SQZ_SCHEDBITS_N : 0u);
        u64 bp = (u64)(size_t)(schedBits + tabSel + (((u32)rp - (second ? blk1 : base0U)) >> 4));
        asm volatile("" : "+s"(bp));
        return sq_grd32((const u8*)(size_t)bp + lane2);
    };
    auto scan = [&](auto AL, auto HS) {
        scanned = true;
        // SQZ_FAST_DEPTH rounds of 1 KiB in flight per wave: w[i] is used by one round of the unrolled body and
        // loaded again right after it
        constexpr u32 D = decltype(HS)::value ? SQZ_FAST_DEPTH_HS : SQZ_FAST_DEPTH;
        sq_v4u w[D];
        u32 lv[D];
#pragma unroll
        for (u32 i = 0; i + 1 < D; i++) {
            w[i] = loadRound(pos + (s32)(1024 * i));
            loadLevels(AL, HS, pos + (s32)(1024 * i), lv[i]);
        }
        for (;;) {
            w[D - 1] = loadRound(pos + (s32)(1024 * (D - 1)));
            loadLevels(AL, HS, pos + (s32)(1024 * (D - 1)), lv[D - 1]);
            // ONE look-up site for the D rounds below (inlined into each of them, the look-up code made
            // the loop several times the size of the instruction cache's share of a wave)
            if constexpr (decltype(HS)::value) {
                if (!hsDead && nCand >= SQZ_FAST_LOOKUP_BATCH && !lookUp(1)) {
                    cx.nOvf = 2u | (sh[FW_SH_NHITS] << 8);
                    hsDead = true;
                }
                // (a compile-time loop: the rotation w[i] / w[(i + 1) % D] must be register names, not indices)
                if (fastStep()) {
                    // four PLAIN rounds: no clipping, no per-round tests.  The loads for the rounds behind them and the raw
                    // copy go through three RUNNING scalar pointers (stream, schedule bitmap, frame) + one constant offset per
                    // lane: the scalar unit adds 1,024 / 128 / 1,024 per round, no vector instruction computes an address.
                    // (Laundered through "+s": hipcc otherwise re-associates base + lane offset + position into 64-bit vector
                    // adds of a per-lane pointer it keeps in two registers.)  Without a second schedule table in front
                    // (tab0U: fastStep) the bitmaps of block 0 and block 1 are one array: word (rp - base0) / 16 + lane.
                    u64 sp = (u64)(size_t)a0 + (u32)(pos - p0) + 1024u * D;
                    u64 bp = (u64)(size_t)schedBits + 2u * ((((u32)pos - base0U) >> 4) + 64u * D);
                    u64 dp = (u64)(size_t)rawDst((u32)pos);
                    asm volatile("" : "+s"(sp), "+s"(bp), "+s"(dp));
                    (void)fw_static_for<0, D>([&](auto I) -> bool {
                        constexpr u32 i = decltype(I)::value;
                        if (!roundPlain(I, w[i], w[(i + 1) % D], lv[i], dp)) hsDead = true;
                        dp += 1024;
                        if constexpr (i + 1 < D) {
#if SQZ_FAST_NT & 1
                            w[i] = sq_grd128_nt((const u8*)(size_t)sp + lane16);
#else
                            w[i] = sq_grd128((const u8*)(size_t)sp + lane16);
#endif
                            lv[i] = sq_grd32((const u8*)(size_t)bp + lane2);
                            sp += 1024;
                            bp += 128;
                        }
                        return true;
                    });
                } else {
                    asm volatile("; general step");
                    (void)fw_static_for<0, D>([&](auto I) -> bool {
                        constexpr u32 i = decltype(I)::value;
                        if (!round(AL, HS, I, w[i], w[(i + 1) % D], lv[i])) hsDead = true;
                        if constexpr (i + 1 < D) {
                            w[i] = loadRound(pos + (s32)(1024 * (D - 1)));
                            loadLevels(AL, HS, pos + (s32)(1024 * (D - 1)), lv[i]);
                        }
                        return true;
                    });
                }
                hflush(D, ((u32)(pos - p0) >> 10) - D); // the step's block sums (pos has advanced by 1 KiB per round)
                if (pos >= myStop) break;
            } else {
                if (nCand >= SQZ_FAST_LOOKUP_BATCH && !lookUp(1)) {
                    cx.nOvf = 2u | (sh[FW_SH_NHITS] << 8);
                    full = true;
                    break;
                }
                bool done;
                u32 const kind = decltype(AL)::value ? fastKindN() : 0u;
                if (kind == 1) { // plain rounds inside the certified blocks
                    done = !fw_static_for<0, D>([&](auto I) -> bool {
                        constexpr u32 i = decltype(I)::value;
                        if (!roundPlainN(std::true_type{}, w[i], w[(i + 1) % D], lv[i])) {
                            full = true;
                            return false;
                        }
                        if constexpr (i + 1 < D) {
                            w[i] = loadStreamU(pos + (s32)(1024 * (D - 1)));
                            lv[i] = loadBitsU(pos + (s32)(1024 * (D - 1)));
                        }
                        return true;
                    }) || pos >= myStop;
                } else if (kind == 2) { // rounds in front of the certified blocks
                    done = !fw_static_for<0, D>([&](auto I) -> bool {
                        constexpr u32 i = decltype(I)::value;
                        if (!roundPlainN(std::false_type{}, w[i], w[(i + 1) % D], 0u)) {
                            full = true;
                            return false;
                        }
                        if constexpr (i + 1 < D) w[i] = loadStreamU(pos + (s32)(1024 * (D - 1)));
                        return true;
                    }) || pos >= myStop;
                } else {
                    asm volatile("; general step");
                    done = !fw_static_for<0, D>([&](auto I) -> bool {
                        constexpr u32 i = decltype(I)::value;
                        if (!round(AL, HS, I, w[i], w[(i + 1) % D], lv[i])) {
                            full = true;
                            return false;
                        }
                        if (pos >= myStop) return false;
                        if constexpr (i + 1 < D) {
                            w[i] = loadRound(pos + (s32)(1024 * (D - 1)));
                            loadLevels(AL, HS, pos + (s32)(1024 * (D - 1)), lv[i]);
                        }
                        return true;
                    });
                }
                if (done) break;
            }
        }
        if constexpr (decltype(HS)::value) full = hsDead;
    };
    if (pos < myStop && !(cx.dbg & 2)) {
        if constexpr (TAIL) // (the caller only sends 16-byte aligned chunks here)
            scan(std::true_type{}, std::false_type{});
        else if (src0 == a0) {
            if (hashOn)
                scan(std::true_type{}, std::true_type{});
            else
                scan(std::true_type{}, std::false_type{});
        } else
            scan(std::false_type{}, std::false_type{});
    }
    if (scanned) {
        if (!full && !lookUp(1)) full = true; // the positives still waiting
        if (full && lane == 0) { // more genuine repeats than the list holds (compressible data): positions >=
            // pendFrom stay unverified, which only matters to searches beyond them
            u32 const cut = (u32)(pendFrom < (s32)SQZ_WINDOW_START ? (s32)SQZ_WINDOW_START : pendFrom) + 1;
            __hip_atomic_fetch_min(&sh[FW_SH_CERT0], cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(&sh[FW_SH_CERT1], cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (lane == 0) sh[FW_SH_FIN + wv] = (u32)pos;
    // a wave that gave up before the end of its run leaves holes in the pool: no fused hash for this chunk
    if (hashOn && pos < myStop && lane == 0) __hip_atomic_fetch_or(&sh[FW_SH_FLAGS], 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    fw_sync();
    if (hashOn && !(sh[FW_SH_FLAGS] & 4u)) {
        cx.hashDone = 1; // every full block's sums are in the pool: sqx::k_xxh3_128_rows runs the scramble chain over them
        if (tid == 0) sq_gwr8(cx.hashedFlag, 1u);
    }
    if (rawCopy) { // the chunk's prefix that is in place: the runs of the waves, up to the first one that stopped early
        s32 at = runStart(0);
        bool any = false;
        for (u32 w = 0; w < NW; w++) {
            s32 const fin = (s32)sh[FW_SH_FIN + w];
            any = any || fin > runStart(w);
            at = fin;
            if (fin < runStart(w + 1)) break;
        }
        if (any) cx.rawEnd = (u32)at < chunkEndU ? (u32)at : SQZ_WINDOW_START + ((chunkEndU - SQZ_WINDOW_START) & ~15u);
    }
    FPROF_ADD(L, 1, t_scan);
    FPROF_T0(t_res);
    // ---- resolve the hits by walking S: a scheduled s fails to be certified when an equal value sits at
    // an earlier position (a scanned position p, or for a pair inside the schedule the earliest member)
    u32 const nHits = sh[FW_SH_NHITS] < SQZ_FAST_HITS ? sh[FW_SH_NHITS] : SQZ_FAST_HITS;
    if (nHits && wv == 0) {
        // One hit per lane.  A hit names a class of scheduled values -- its bucket's entries with its 16-bit tag
        // (26 bits of the value) -- and idx[] says which searches those are: their positions come from the schedule
        // table and their 4 bytes from the chunk, a handful of loads per hit (S itself is not gathered again).
        static_assert(SQZ_FAST_HITS <= 64, "one hit per lane");
        bool const have = lane < nHits;
        u32 const hp = have ? hits[2 * lane] : 0u, hv = have ? hits[2 * lane + 1] : 0u;
        u32 const k = hv >> 22, tag = (hv >> 6) & 0xFFFFu;
        u32 const lo = (have && k) ? fast_get16(ends, k - 1) : 0u, hi = have ? fast_get16(ends, k) : 0u;
        auto posOf = [&](u32 i) -> u32 { return cert_pos_g(cx, i); }; // as cert_pos8
        // does an earlier position p with the 4 bytes of the search at s let that search succeed?  btlazy2: always (the
        // tree may reach it).  Row-hash finder (lazy2, minMatch 5): only from the row and with the tag of the search's
        // 5-byte hash -- two positions that share 4 bytes but not the fifth meet there only by a hash collision
        auto reaches = [&](u32 p, u32 s_) -> bool {
            if (cx.hashBits == 0) return true;
            return hash5(sq_grd64(base + p), cx.hashBits) == hash5(sq_grd64(base + s_), cx.hashBits);
        };
        u32 v0 = 0xFFFFFFFFu, v1 = 0xFFFFFFFFu;
        bool wbad = false;
        for (u32 a = lo; a < hi; a++) { // (a bucket holds 3.4 values on average, SQZ_FAST_BUCKET_MAX at most)
            if (vals[a] != tag) continue;
            if (idxs[a] >= 0xFF00u) { // second pass, a value of the bucket check: is p a position that check is about?
                if (hp != 0 && pseudoVal(idxs[a] - 0xFF00u, sh[FW_SH_WAKE + 3]) == hv && hp < sh[FW_SH_WAKE] && hp != sh[FW_SH_WAKE + 2]) wbad = true;
                continue;
            }
            u32 const sa = posOf(idxs[a]), va = sq_grd32(base + sa);
            if (hp != 0) { // a scanned position p with the bytes hv: every scheduled s > p with those bytes
                if (va == hv && sa > hp && reaches(hp, sa)) note(sa, v0, v1);
            } else { // two scheduled positions share bits 6..31: of two with equal bytes, the later search can succeed
                for (u32 b = a + 1; b < hi; b++) {
                    if (vals[b] != tag || idxs[b] >= 0xFF00u) continue;
                    u32 const sb = posOf(idxs[b]), vb = sq_grd32(base + sb);
                    if (va == vb && reaches(sa < sb ? sa : sb, sa > sb ? sa : sb)) note(sa > sb ? sa : sb, v0, v1);
                }
            }
        }
        settle(v0, v1);
        if (wv_ballot(wbad) && lane == 0) __hip_atomic_fetch_or(&sh[FW_SH_FLAGS], 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    fw_sync();
    certEnd = sh[FW_SH_CERT0];
    certEnd1 = sh[FW_SH_CERT1];
    if (sh[FW_SH_FLAGS] & 1) certEnd = certEnd1 = SQZ_WINDOW_START;
    cx.wakeBad = (sh[FW_SH_FLAGS] & 8u) ? 1u : 0u;
    wv_fence();
    FPROF_ADD(L, 3, t_res);
#else
    // scalar twin: the same statement, computed plainly
    (void)bloom; (void)vals; (void)ends; (void)lvlWords;
    u32 nH = 0; // the list of hits as the device leaves it: (p, bytes) of a scanned position, (0, bytes) of a pair inside S
    auto pushHit = [&](u32 p, u32 v) {
        if (nH < SQZ_FAST_HITS) {
            hits[2 * nH] = p;
            hits[2 * nH + 1] = v;
        }
        nH++;
    };
    auto note = [&](u32 sviol) {
        if (sviol < blk1)
            certEnd = sviol < certEnd ? sviol : certEnd;
        else
            certEnd1 = sviol < certEnd1 ? sviol : certEnd1;
    };
    // scheduled positions as a bitmap: for a chunk of at most two blocks through the same arithmetic the device uses,
    // for the last two blocks of a larger one (base0Pos beyond the chunk's start: the device takes the schedule's
    // bitmap table there) from the schedule tables themselves
    static thread_local u8 isSched[(2048u << 10) + 64];
    memset(isSched, 0, cx.chunkEndPos + 32);
    bool const general = cx.base0Pos != SQZ_WINDOW_START || cx.tab0 != 0 || cx.tail != 8;
    if (general)
        for (u32 i = 0; i < nCert; i++) isSched[cert_pos_g(cx, i)] = 1;
    else
        for (s32 lo = 0; lo < (s32)cx.chunkEndPos; lo += 16) {
            u32 const m = fast_sched_mask(lvlWords, lo, blk1, cx.chunkEndPos, 0, 0, false);
            for (u32 j = 0; j < 16; j++)
                if ((m >> j) & 1) isSched[lo + j] = 1;
        }
    auto reaches = [&](u32 p, u32 s_) -> bool { // (see the device twin)
        return cx.hashBits == 0 || hash5(sq_rd64(base + p), cx.hashBits) == hash5(sq_rd64(base + s_), cx.hashBits);
    };
    // (value, position) of every scheduled search, sorted by value then position
    std::vector<std::pair<u32, u32>> sv(nCert);
    for (u32 i = 0; i < nCert; i++) {
        u32 const sp = cert_pos_g(cx, i);
        sv[i] = std::make_pair(sq_rd32(base + sp), sp);
        if (cx.repOff && sq_rd32(base + sp + 1) == sq_rd32(base + sp + 1 - cx.repOff)) note(sp);
    }
    std::sort(sv.begin(), sv.end());
    for (u32 i = 1; i < nCert; i++) // an earlier scheduled position with the same bytes
        for (u32 j = i; j-- > 0 && sv[j].first == sv[i].first;)
            if (reaches(sv[j].second, sv[i].second)) {
                note(sv[i].second);
                pushHit(0u, sv[i].first);
            }
    // every scheduled position must be one the arithmetic knows (else the scan would skip a wrong set)
    for (u32 i = 0; i < nCert; i++)
        if (!isSched[sv[i].second]) certEnd = certEnd1 = SQZ_WINDOW_START;
    for (u32 p = SQZ_WINDOW_START; p < ((blk1 != 0xFFFFFFFFu) ? certEnd1 : certEnd) && p < pEnd; p++) {
        if (isSched[p]) continue;
        u32 const vp = sq_rd32(base + p);
        auto it = std::lower_bound(sv.begin(), sv.end(), std::make_pair(vp, p + 1));
        for (; it != sv.end() && it->first == vp; ++it) // scheduled positions beyond p with these bytes
            if (reaches(p, it->second)) {
                note(it->second);
                pushHit(p, vp);
            }
    }
    L->sh[FW_SH_NHITS] = nH;
    L->sh[FW_SH_FLAGS] = 0;
    if (cx.wakeLim) // second pass: the bucket check (see CertCtx)
        for (u32 p = SQZ_WINDOW_START; p < cx.wakeLim && p < pEnd; p++)
            if (p != cx.wakeSkip && base[p] == (u8)cx.wakeByte && hash4(sq_rd32(base + p), cx.wakeHashLog) == cx.wakeHash) cx.wakeBad = 1;
#endif
    cx.certEnd = certEnd;
    cx.certEnd1 = (blk1 != 0xFFFFFFFFu) ? certEnd1 : 0u;
}

// Blocks bA (even) and bA + 1 of a chunk of n bytes: the no-match certificate over their searches, with every earlier
// position of the chunk scanned; HUF's sampled pre-check of the two blocks; and, when both say "raw block", the two
// blocks' headers and payload in their place of the frame at dst (the frame header is the caller's).  Returns 1 then,
// 0 when at least one of the two blocks is not known to be raw.  *e0 / *e1: the certificate ends (positions).
//   n <= 256 KiB (btlazy2): bA = 0, the whole chunk -- what this path was built for.
//   n  > 256 KiB (lazy2 + row-hash finder, /root/reference/src/util/chunk.rs:11: the reference's own 2 MiB chunks):
//     one call per pair of blocks (the set S of a pair fills the workgroup's LDS), each scanning the chunk from its start
//     to the pair's end -- 4.5 passes over a 2 MiB chunk in total.  What differs from btlazy2: searches stop 16 bytes
//     before a block's end (the finder's hash cache), and an earlier position with a search's 4 bytes only matters when
//     it shares the search's row and tag, i.e. the (hashLog + 2)-bit hash of its FIVE bytes (minMatch 5): on random
//     data that is what makes the certificate hold at all (a 2 MiB chunk has ~6 four-byte coincidences at searched
//     positions, 0.03 five-byte ones).  Blocks: 128 KiB each while every block so far was emitted raw -- libzstd's
//     pre-block splitter only acts once `savings >= 3`, and a raw block contributes -3 (ZSTD_optimalBlockSize).
// The second pass (see fastpath_blocks, "one short match") as a function of its own: its copy of the scan gets registers and
// spill slots of its own instead of taking them from the first pass's loops, which are at the limit of both (inlined, its 70
// extra scalar spills cost the kernel a fourth spill register and the first pass's tightest round its last free one).
// Returns 1 when the tail certificate holds.  Every wave of the workgroup calls it with the same arguments.
SQ_DEV u32 tail_pass_body(const u8* base, const u32* schedTab, FastLds* L, u32 anc, u32 nEntries, u32 repOff, u32 blkEnd, u32 tail,
                          u32 wakeLim, u32 wakeSkip, u32 wakeHashByte, u32 hashLog, u32 eLog, u32 dbg) {
    CertCtx cx;
    cx.base = base;
    cx.E = nullptr;
    cx.eLog = eLog;
    cx.genBits = 0;
    cx.schedTab = schedTab;
    cx.base0Pos = anc;
    cx.tab0 = 3;
    cx.tail = tail;
    cx.hashBits = 0;
    cx.certCnt0 = cx.nCert = nEntries; // (no second block)
    cx.base1Pos = blkEnd;
    cx.tabOff1 = 0;
    cx.repOff = repOff;
    cx.chunkEndPos = blkEnd;
    cx.bloom = nullptr;
    cx.cand = nullptr;
    cx.ovf = nullptr;
    cx.lvl = nullptr;
    cx.certEnd = cx.certEnd1 = 0;
    cx.nOvf = 0;
    cx.hashPool = nullptr;
    cx.hashedFlag = nullptr;
    cx.hashDone = 0;
    cx.dbg = dbg;
    cx.wakeLim = wakeLim;
    cx.wakeSkip = wakeSkip;
    cx.wakeHash = wakeHashByte & 0xFFFFFFu;
    cx.wakeByte = wakeHashByte >> 24;
    cx.wakeHashLog = hashLog;
    cx.wakeBad = 0;
    cert_lds_core<true>(cx, L, nullptr);
#if !SQ_ON_DEVICE && defined(SQ_SIM_TRACE)
    fprintf(stderr, "[one match] second pass: certEnd %u ilim %u wakeBad %u nOvf %u anchor %u n %u\n", cx.certEnd, blkEnd - tail, cx.wakeBad, cx.nOvf, cx.base0Pos, cx.nCert);
#endif
    return (cx.certEnd >= blkEnd - tail && !cx.wakeBad && cx.nOvf == 0) ? 1u : 0u;
}
#if SQ_ON_DEVICE
__device__ __noinline__ u32 tail_pass(const u8* base, const u32* schedTab, FastLds* L, u32 anc, u32 nEntries, u32 repOff, u32 blkEnd,
                                      u32 tail, u32 wakeLim, u32 wakeSkip, u32 wakeHashByte, u32 hashLog, u32 eLog, u32 dbg) {
    // (a function's arguments arrive in vector registers: back into scalar ones, where the scan's address arithmetic wants them)
    return tail_pass_body((const u8*)wv_uni64((u64)(size_t)base), (const u32*)wv_uni64((u64)(size_t)schedTab), (FastLds*)wv_uni64((u64)(size_t)L),
                          wv_uni(anc), wv_uni(nEntries), wv_uni(repOff), wv_uni(blkEnd), wv_uni(tail), wv_uni(wakeLim), wv_uni(wakeSkip),
                          wv_uni(wakeHashByte), wv_uni(hashLog), wv_uni(eLog), wv_uni(dbg));
}
#else
#define tail_pass tail_pass_body
#endif

#if !SQ_ON_DEVICE && defined(SQ_SIM_TRACE) // (tools/one_match_stats.py: why the second pass was not tried / did not hold)
#define SQ_OM_BREAK { fprintf(stderr, "[one match] refused at line %d\n", __LINE__); break; }
#else
#define SQ_OM_BREAK break
#endif
#if !SQ_ON_DEVICE
static u32 g_sim_one_match_blocks = 0; // (simulator build: blocks decided by the second pass)
#endif
SQ_DEV u32 fastpath_blocks(const u8* src, u32 n, u8* dst, u32 cap, u32 bA, const CParams& cp, const u32* schedTab, FastLds* L,
                           u32* e0, u32* e1, u32 dbg, u64* hashPool, u8* hashedFlag, u32* rawDone = nullptr) {
    constexpr u32 BS = SQZ_BLOCKSIZE_MAX;
    bool const row = cp.strategy == STRAT_LAZY2;
    u32 const nBlocksAll = (n + BS - 1) / BS;
    u32 const vEnd = n < (bA + 2) * BS ? n : (bA + 2) * BS; // the part of the chunk this call looks at
    u32 const offA = bA * BS;
    u32 const lenA = vEnd - offA < BS ? vEnd - offA : BS, lenB = vEnd - offA - lenA;
    u32 const tail = row ? 16u : 8u;
    *e0 = *e1 = 0;
    if (row && ((size_t)src & 15)) return 0; // (the general form of the scan takes the schedule's bitmap: aligned chunks)
    CertCtx cx;
    cx.base = src - SQZ_WINDOW_START;
    cx.E = nullptr;
    cx.eLog = cp.chainLog - 1;
    cx.genBits = 0;
    cx.schedTab = schedTab;
    cx.base0Pos = SQZ_WINDOW_START + offA;
    cx.tab0 = bA ? 1u : 0u;
    cx.tail = tail;
    cx.hashBits = row ? cp.hashLog + 2 : 0u; // rowHashLog + 8 tag bits = (hashLog - 6) + 8
    u32 const full0 = gld32(schedTab), full1 = gld32(schedTab + 1);
    const u32* const tabA = schedTab + 2 + (cx.tab0 ? SQZ_SCHEDTAB_N : 0u);
    u32 const fullA = cx.tab0 ? full1 : full0;
    cx.certCnt0 = (lenA <= tail) ? 0u : (lenA == BS && tail == 8) ? fullA : sched_count(tabA, fullA, lenA - tail);
    u32 const cnt1 = (lenB <= tail) ? 0u : (lenB == BS && tail == 8) ? full1 : sched_count(schedTab + 2 + SQZ_SCHEDTAB_N, full1, lenB - tail);
    cx.nCert = cx.certCnt0 + cnt1;
    cx.base1Pos = cx.base0Pos + BS;
    cx.tabOff1 = 0;
    cx.repOff = 1;
    cx.chunkEndPos = vEnd + SQZ_WINDOW_START;
    cx.bloom = nullptr;
    cx.cand = nullptr;
    cx.ovf = nullptr;
    cx.lvl = nullptr;
    cx.certEnd = cx.certEnd1 = 0;
    cx.nOvf = 0;
    cx.hashPool = hashPool;
    cx.hashedFlag = hashedFlag;
    cx.hashDone = 0;
    cx.wakeLim = cx.wakeSkip = cx.wakeHash = cx.wakeByte = cx.wakeBad = 0;
    cx.wakeHashLog = 0;
    if (cx.nCert == 0 || cx.nCert > SQZ_FAST_VALS || lenA < 9) return 0;
    cx.dbg = dbg;
    u32 const nB = lenB ? 2u : 1u;
    bool const sized = lenA >= SQZ_FAST_MIN_BLOCK && (lenB == 0 || lenB >= SQZ_FAST_MIN_BLOCK);
    u32 const hdr = 5 + (n < 256 ? 1u : (n < 65536 + 256 ? 2u : 4u)); // = fast_frame_header()
    bool const room = cap >= n + hdr + 3 * nBlocksAll;
    // Two call sites of the scan: the first pass (its three instantiations: a second copy of THOSE no longer fits the
    // instruction cache) and the second pass of a chunk of the btlazy2 class, 16-byte aligned, that "one short match" below
    // keeps from being certified (cert_lds_core<true>: one instantiation of the aligned scan, run for ~7 % of random chunks).
    // What outlives the second pass is kept in LDS (L->keep: 0 / 1 the first pass's certificate ends, 2 its rawEnd, 3 the
    // block that ends up with ONE sequence (2: none), 4 / 5 the bytes its match covers, 6 the verdict), not in registers: the
    // scan is at its limit of 128 vector registers.
    SQ_L u32* const keep = (SQ_L u32*)L->keep;
    cert_lds_core<false>(cx, L, (sized && room && !(dbg & 4)) ? dst + hdr : nullptr);
    bool again = false;
    u32 tpAnc = 0, tpN = 0, tpOff = 0, tpEnd = 0, tpLim = 0, tpSkip = 0, tpHB = 0; // the second pass's arguments
    do {
        u32 const ce0 = cx.certEnd, ce1 = cx.certEnd1;
        // every search of the two blocks certainly fails?
        bool const fail0 = ce0 < cx.base0Pos + lenA - tail, fail1 = lenB && ce1 < cx.base1Pos + lenB - tail;
        bool const whole = !fail0 && !fail1;
        if (fw_first()) {
            keep[0] = ce0;
            keep[1] = ce1;
            keep[2] = cx.rawEnd;
            keep[3] = 2;
            keep[6] = whole ? 1u : 0u;
        }
        // ---- one short match.  ~7 % of random chunks have a searched position s* with the 4 bytes of ONE earlier position q
        // (the rest of the chunk certified).  What libzstd does there is a matter of a few byte comparisons, given that s*'s
        // bucket holds nothing that could disturb the tree walk (the second pass checks that): the search at s* finds q
        // (length = common prefix), the depth searches at s* + 1 and s* + 2 find nothing if no earlier position holds their 4
        // bytes, the match is extended backwards, and the parser goes on behind it with a NEW no-match schedule laid at the
        // match's end (repcode offset = the match's).  If that schedule's certificate -- the second pass, which also covers
        // the two depth searches -- holds as well, the block has exactly one sequence: its literals are raw (HUF's sampled
        // pre-check, on the literals), so its compressed form is larger than the block and libzstd emits it raw, like a
        // certified block, with the repcodes of the block's start handed on (the other block's certificate stands).
        // Anything else (a second coincidence, a repcode hit, a long match, a crowded bucket) is left to k_compress_batch.
#if !SQ_ON_DEVICE && defined(SQ_SIM_TRACE)
        fprintf(stderr, "[turn0] ce0 %u ce1 %u fail0 %d fail1 %d whole %d row %d sized %d room %d bA %u al %u nOvf %u\n", ce0, ce1, fail0, fail1, whole, row, sized, room, bA, (unsigned)((size_t)src & 15), cx.nOvf);
#endif
        if (whole || !(fail0 != fail1) || row || !sized || !room || bA != 0 || ((size_t)src & 15) || (dbg & (4 | 2048)) || cx.nOvf) SQ_OM_BREAK;
        {
            const u8* const base = cx.base;
            u32 const b = fail1 ? 1u : 0u;
            u32 const blk = b ? cx.base1Pos : cx.base0Pos, bs = b ? lenB : lenA, blkEnd = blk + bs, ilimPos = blkEnd - tail;
            u32 const sS = b ? ce1 : ce0;
            // s* is a scheduled search of the block (a full list of hits leaves another kind of end)
            const u32* const tabS = schedTab + 2 + (b ? SQZ_SCHEDTAB_N : 0u);
            u32 const nS = b ? cx.nCert - cx.certCnt0 : cx.certCnt0;
            u32 const jS = sched_count(tabS, nS, sS - blk);
            if (sS < blk || jS >= nS || gld32(tabS + jS) != sS - blk || sS + 3 >= ilimPos) SQ_OM_BREAK;
            SQ_L u32* const sh = (SQ_L u32*)L->sh;
            SQ_L u32* const hits = (SQ_L u32*)L->hits;
            u32 const nHitsAll = wv_uni(sh[FW_SH_NHITS]);
            if (nHitsAll > SQZ_FAST_HITS || (wv_uni(sh[FW_SH_FLAGS]) & 3u)) SQ_OM_BREAK;
            u32 const vS = srd32(base + sS), vS1 = srd32(base + sS + 1), vS2 = srd32(base + sS + 2);
            if (vS1 == vS || vS2 == vS1) SQ_OM_BREAK; // repcode (offset 1) at s* + 1 / s* + 2
            // the earlier positions with s*'s bytes: exactly one, a scanned one
            u32 q = 0, nq = 0;
            bool odd = false;
            for (u32 h = 0; h < nHitsAll; h++) {
                u32 const hp = wv_uni(hits[2 * h]), hv = wv_uni(hits[2 * h + 1]);
                if (hp == 0) {
                    if ((hv >> 6) == (vS >> 6)) odd = true; // two scheduled positions share (26 bits of) these bytes
                } else if (hv == vS && hp < sS) {
                    q = hp;
                    nq++;
                }
            }
            if (odd || nq != 1) SQ_OM_BREAK;
            u32 const off = sS - q;
            if (off < 2 || srd32(base + q) != vS) SQ_OM_BREAK;
            // the match: forwards to the block's end at most, backwards to the block's start / the window's first byte
            u32 ml = 4;
            while (ml < 64 && sS + ml < blkEnd && srd8(base + sS + ml) == srd8(base + q + ml)) ml++;
            if (ml >= 64) SQ_OM_BREAK;
            u32 start = sS;
            while (start > blk && start - off > SQZ_WINDOW_START && sS - start < 64 && srd8(base + start - 1) == srd8(base + start - off - 1)) start--;
            if (sS - start >= 64) SQ_OM_BREAK;
            u32 const anc = sS + ml, mlAll = anc - start;
            if (mlAll > (bs >> 6) || bs - mlAll < SQZ_FAST_MIN_BLOCK) SQ_OM_BREAK;
            if (anc <= ilimPos && srd32(base + anc) == srd32(base + anc - 1)) SQ_OM_BREAK; // immediate repcode (offset_2 = 1)
            // second pass: the "later" schedule laid at the match's end + the two depth searches, every position of the
            // block and of what lies in front of it scanned; the bucket check for s*
            u32 const nLater = gld32(schedTab + 1);
            u32 const j1 = anc < ilimPos ? sched_count(schedTab + 2 + SQZ_SCHEDTAB_N, nLater, ilimPos - anc) : 0u;
            tpAnc = anc;
            tpN = j1 + 2;
            tpOff = off;
            tpEnd = blkEnd;
            tpLim = sS;
            tpSkip = q;
            tpHB = hash4(vS, cp.hashLog) | ((vS & 0xFFu) << 24);
            if (fw_first()) {
                keep[3] = b;
                keep[4] = start - blk;
                keep[5] = anc - blk;
            }
        }
        again = true;
    } while (false);
    if (again) {
        fw_sync(); // (the shared words and the list of hits are written again by the second pass)
        bool const holds = tail_pass(cx.base, schedTab, L, tpAnc, tpN, tpOff, tpEnd, tail, tpLim, tpSkip, tpHB, cp.hashLog, cx.eLog, dbg) != 0;
        if (fw_first()) {
            keep[6] = holds ? 1u : 0u;
            if (!holds) keep[3] = 2;
        }
    }
    fw_sync();
    cx.certEnd = wv_uni(keep[0]); // what stage 2 is told is the first pass's answer
    cx.certEnd1 = wv_uni(keep[1]);
    cx.rawEnd = wv_uni(keep[2]);
    u32 const seqBlk = wv_uni(keep[3]), seqStart = wv_uni(keep[4]), seqEnd = wv_uni(keep[5]);
    bool whole = wv_uni(keep[6]) != 0;
    FPROF_T0(t_frame);
    *e0 = (dbg & 64) ? cx.nOvf : cx.certEnd; // (64: bring-up, why the scan gave up)
    *e1 = cx.certEnd1;
    if (rawDone) *rawDone = cx.rawEnd - SQZ_WINDOW_START; // chunk bytes whose raw-block payload the scan has put in place
    if (dbg & 4) whole = false; // timing experiment: certificate only
#if !SQ_ON_DEVICE
    if (whole && seqBlk < 2) g_sim_one_match_blocks++;
#endif
    u32 ok = 0;
    if (whole && sized && room) {
        // HUF's sampled pre-check: the 4 KiB samples (first and last of each block) are dealt out to the
        // waves; largest1 + largest2 of a block are added up through LDS
        u32 const wv = fw_wave();
        u32 const nSamples = 2 * nB;
        for (u32 j = wv; j < nSamples; j += FW_NW) {
            u32 const b = j >> 1, bs = b ? lenB : lenA;
            const u8* const blkAt = src + offA + (b ? lenA : 0u);
            u32 l = 0, dummy;
            wv_zero_lds(L->bloom + 256 * wv, 256);
            if (b != seqBlk)
                wv_hist_bytes(L->bloom + 256 * wv, blkAt + ((j & 1) ? bs - 4096 : 0u), 4096);
            else {
                // the block's literals are the block without [seqStart, seqEnd): literal i sits at block offset i (< seqStart)
                // or i + the match's length
                u32 const mlAll = seqEnd - seqStart, a = (j & 1) ? bs - mlAll - 4096 : 0u; // first literal of the sample
                u32 const n1 = a < seqStart ? (seqStart - a < 4096 ? seqStart - a : 4096u) : 0u; // ... of them in front of the match
                if (n1) wv_hist_bytes(L->bloom + 256 * wv, blkAt + a, n1);
                if (n1 < 4096) wv_hist_bytes(L->bloom + 256 * wv, blkAt + (a + n1) + mlAll, 4096 - n1);
            }
            wv_max_and_last(L->bloom + 256 * wv, 256, &l, &dummy);
            WV_SERIAL L->sh[FW_SH_SAMPLE + j] = l;
        }
        fw_sync();
        ok = 1; // (every wave reads the same four words)
        for (u32 b = 0; b < nB; b++)
            if (L->sh[FW_SH_SAMPLE + 2 * b] + L->sh[FW_SH_SAMPLE + 2 * b + 1] > ((2 * 4096) >> 7) + 4) ok = 0;
        if (ok) {
            u32 const done = cx.rawEnd - SQZ_WINDOW_START; // chunk bytes the scan has already put in place
            for (u32 b = 0; b < nB; b++) {
                u32 const bs = b ? lenB : lenA, b0 = offA + (b ? lenA : 0u), blk = bA + b;
                u8* const op = dst + hdr + (size_t)blk * (3u + BS);
                if (b % FW_NW == wv) {
                    u32 const h = ((blk + 1 == nBlocksAll) ? 1u : 0u) + (0u << 1) + (bs << 3); // ZSTD_noCompressBlock
                    WV_SERIAL {
                        op[0] = (u8)h;
                        op[1] = (u8)(h >> 8);
                        op[2] = (u8)(h >> 16);
                    }
                    u32 const have = done > b0 ? (done - b0 < bs ? done - b0 : bs) : 0u;
                    if (have < bs) wv_memcpy(op + 3 + have, src + b0 + have, bs - have);
                }
            }
        }
    }
    FPROF_ADD(L, 4, t_frame);
    return ok;
}

// Returns the frame length when the chunk is decided here, 0 when it is left to k_compress_batch.
// certOut[0..1]: the certificate ends for k_compress_batch (certOut[1] carries SQZ_CERT_HINT_VALID),
// or 0, 0 when no certificate was computed.
// hashPool / hashedFlag (device build, or null): the fused hash -- see SQZ_FAST_HPOOL_BLOCKS.
// (One call does the whole chunk: all pairs of blocks of a chunk above 256 KiB one after the other -- the form of the
// simulator build.  The kernel below deals the pairs out as jobs of their own and calls fastpath_blocks itself, at ONE
// call site: the scan is three instantiations of a long loop, inlined twice it no longer fits the instruction cache.)
#if !SQ_ON_DEVICE
SQ_DEV u32 fastpath_frame_wave(const u8* src, u32 n, u8* dst, u32 cap, u32* E, u32 genBits, const u32* schedTab,
                                 FastLds* L, u32* certOut, u32 dbg = 0, u64* hashPool = nullptr, u8* hashedFlag = nullptr, u32* pairInfo = nullptr) {
    (void)E; (void)genBits;
    if (pairInfo) *pairInfo = 0;
    CParams const cp = get_cparams(n);
    u32 c0 = 0, c1 = 0;
    u32 frame = 0;
    u32 const nBlocks = (n + SQZ_BLOCKSIZE_MAX - 1) / SQZ_BLOCKSIZE_MAX;
    if (cp.strategy == STRAT_BTLAZY2 && n >= 9) {
        u32 e0, e1;
        u32 const ok = fastpath_blocks(src, n, dst, cap, 0, cp, schedTab, L, &e0, &e1, dbg, hashPool, hashedFlag);
        if (e0 | e1) {
            c0 = e0;
            c1 = e1 | SQZ_CERT_HINT_VALID;
        }
        if (ok) frame = fast_frame_header_all(dst, n) + n + 3 * nBlocks;
    } else if (cp.strategy == STRAT_LAZY2 && !(dbg & 512)) { // (512: timing experiment, large chunks left to stage 2)
        u32 okAll = 1, info = 0;
        for (u32 bA = 0; bA < nBlocks; bA += 2) { // (every pair, as the kernel's jobs do: stage 2 wants each pair's verdict)
            u32 e0, e1;
            u32 const ok = fastpath_blocks(src, n, dst, cap, bA, cp, schedTab, L, &e0, &e1, dbg, nullptr, nullptr);
            fw_sync();
            info += 1u + (ok ? 0u : (0x100u | (0x10000u << (bA / 2))));
            okAll = okAll && ok;
        }
        if (pairInfo) *pairInfo = info;
        if (okAll) frame = fast_frame_header_all(dst, n) + n + 3 * nBlocks;
    }
    if (fw_wave() == 0) WV_SERIAL {
        certOut[0] = c0;
        certOut[1] = c1;
    }
    wv_mem_sync();
    return frame;
}
#endif

#if SQ_ON_DEVICE
// Persistent workgroups of FW_NW waves; every workgroup pulls JOBS from *queue until exhausted (every workgroup reaches
// j >= n_jobs).  A job is a chunk of at most two blocks, or one pair of blocks of a larger chunk (jobs[j] = chunk index |
// pair number << 28; jobs == null: job j = chunk j).  frame_len[idx] = frame size, or 0 = not decided here.
// pair_state[idx] (zeroed by the host; large chunks only): pairs finished so far (bits 0..7), how many of them
// failed (bits 8..15) and which (bit 16 + pair); the workgroup that finishes a chunk's last pair writes its frame header and
// its verdict; stage 2 reads the word (compress_frame_wave: certified pairs are replayed, not searched).
__global__ void __launch_bounds__(64 * SQZ_FAST_WG_WAVES, SQZ_FAST_WAVES_PER_SIMD)
k_incompressible_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* slots,
                       const u64* slot_off, u32* frame_len, u32* cert_out, u8* etab_pool, u64 etab_bytes,
                       u32* queue, const u32* schedTab, u32 dbg, u32 persist, u32* work, u8* hashed, u8* hpool,
                       const u32* jobs, u32 n_jobs, u32* pair_state) {
    // hashed / hpool (both or none): the fused form -- the kernel runs BEFORE the dedup decision on every chunk of the
    // batch (select == null) and, for each chunk it streams, leaves the sums of XXH3's full blocks in the chunk's slice
    // of hpool (SQZ_FAST_HPOOL_BYTES each) and hashed[idx] = 1 (hashed[] zeroed by the host); sqx::k_xxh3_128_rows
    // finishes those hashes and computes the others from the chunk bytes.
    // queue[1]: number of chunks decided here.  work: the chunks left to k_compress_batch, as two index
    // lists: [0] / [1] their lengths, then n slots for the "heavy" ones (certified in neither block: the
    // whole exact path) and n slots for the others.
    __shared__ FastLds lds;
    (void)etab_pool; (void)etab_bytes; (void)persist; // (the certificate's set lives in LDS: no table, no generations)
    bool const first = threadIdx.x == 0;
    if (threadIdx.x < 64) { // the secret words of XXH3's stripe loop as the scan's lanes need them
        u32 const l = threadIdx.x;
        u64 const k0 = sqx::sec64(8 * (l >> 2) + 16 * (l & 3)), k1 = sqx::sec64(8 * (l >> 2) + 16 * (l & 3) + 8);
        lds.hsec[l] = (u32)k0;
        lds.hsec[64 + l] = (u32)(k0 >> 32);
        lds.hsec[128 + l] = (u32)k1;
        lds.hsec[192 + l] = (u32)(k1 >> 32);
    }
#ifdef SQ_PROF_S1
    if (first)
        for (int k = 0; k < 8; k++) lds.prof[k] = 0;
#endif
    u32 const nj = jobs ? n_jobs : n;
    for (;;) {
        if (first) lds.sh[FW_SH_IDX] = atomicAdd(queue, 1u);
        fw_sync();
        u32 const j = wv_uni(lds.sh[FW_SH_IDX]);
        fw_sync(); // (the word is written again at the top of the next turn)
        if (j >= nj) break;
        u32 const job = jobs ? gld32(jobs + j) : j;
        u32 const idx = job & 0x0FFFFFFFu, pair = job >> 28;
        bool const taken = select == nullptr || select[idx] != 0;
        u32 const L = len[idx];
        bool const large = L > 2 * SQZ_BLOCKSIZE_MAX; // more than two blocks: one job per pair of blocks
        u32 r = 0;
        bool verdict = !large; // this job settles the chunk (the last pair of a large one does)
        if (taken) {
            u32 const cap = (u32)(slot_off[idx + 1] - slot_off[idx]);
            u8* const dst = slots + slot_off[idx];
            CParams const cp = get_cparams(L);
            u32 const nBlocks = (L + SQZ_BLOCKSIZE_MAX - 1) / SQZ_BLOCKSIZE_MAX, nPairs = (nBlocks + 1) / 2;
            wv_mem_sync_inv(); // no L1 line from the previous chunk's data survives
            FPROF_T0(t_chunk);
            bool run = (cp.strategy == STRAT_BTLAZY2 && L >= 9) || (large && !(dbg & 512)); // (512: timing experiment, large chunks left to stage 2)
            // (every pair of a large chunk is looked at even after another pair of it has failed: stage 2 replays the
            // certified pairs' table insertions instead of searching them, row_replay_nomatch)
            u32 ok = 0, e0 = 0, e1 = 0, rawDone = 0;
            if (run) // (the ONE call site, see fastpath_frame_wave)
                ok = fastpath_blocks(data + off[idx], L, dst, cap, large ? 2 * pair : 0u, cp, schedTab, &lds, &e0, &e1, dbg,
                                     (hpool && !large) ? (u64*)(hpool + (u64)idx * SQZ_FAST_HPOOL_BYTES) : nullptr,
                                     (hashed && !large) ? hashed + idx : nullptr, &rawDone);
            if (!large) {
                if (ok) r = fast_frame_header_all(dst, L) + L + 3 * nBlocks;
                if (first) { // the certificate ends for k_compress_batch, or 0, 0 when no certificate was computed; beside
                             // the first, how much raw-block payload is in place already (SQZ_CERT_HINT_RAW_*)
                    u32 const units = (rawDone >> SQZ_CERT_HINT_RAW_LOG) < SQZ_CERT_HINT_RAW_MAX ? (rawDone >> SQZ_CERT_HINT_RAW_LOG) : SQZ_CERT_HINT_RAW_MAX;
                    cert_out[2 * (size_t)idx] = e0 | (units << SQZ_CERT_HINT_POS_BITS);
                    cert_out[2 * (size_t)idx + 1] = (e0 | e1) ? (e1 | SQZ_CERT_HINT_VALID) : 0u;
                }
            } else {
                fw_sync();
                if (first) { // (one add: the fields do not overlap -- pairs finished, pairs failed, WHICH pairs failed)
                    u32 const inc = 1u + (ok ? 0u : (0x100u | (0x10000u << pair)));
                    lds.sh[FW_SH_IDX] = atomicAdd(pair_state + idx, inc) + inc;
                }
                fw_sync();
                u32 const now = wv_uni(lds.sh[FW_SH_IDX]);
                fw_sync();
                if ((now & 0xFFu) == nPairs) { // the chunk's last pair to finish
                    verdict = true;
                    if (!((now >> 8) & 0xFFu)) r = fast_frame_header_all(dst, L) + L + 3 * nBlocks;
                    if (first) {
                        cert_out[2 * (size_t)idx] = 0;
                        cert_out[2 * (size_t)idx + 1] = 0;
                    }
                }
            }
            wv_mem_sync();
            FPROF_ADD(&lds, 5, t_chunk);
        } else if (first && (!large || pair == 0)) {
            cert_out[2 * (size_t)idx] = 0;
            cert_out[2 * (size_t)idx + 1] = 0;
        }
        if (first && verdict) {
            frame_len[idx] = r;
            if (r)
                atomicAdd(queue + 1, 1u);
            else if (taken) {
                u32 const h0 = cert_out[2 * (size_t)idx] & SQZ_CERT_HINT_POS_MASK, h1 = cert_out[2 * (size_t)idx + 1];
                u32 const len0 = L < SQZ_BLOCKSIZE_MAX ? L : SQZ_BLOCKSIZE_MAX;
                bool const heavy = (h1 & SQZ_CERT_HINT_VALID) && L > SQZ_BLOCKSIZE_MAX + 8 && len0 >= 9 &&
                                   h0 < SQZ_WINDOW_START + len0 - 8 && (h1 & ~SQZ_CERT_HINT_VALID) < SQZ_WINDOW_START + L - 8;
                u32 const at = atomicAdd(work + (heavy ? 0 : 1), 1u);
                work[2 + (heavy ? 0u : n) + at] = idx;
            }
        }
    }
#ifdef SQ_PROF_S1
    fw_sync();
    if (first)
        for (int k = 0; k < 8; k++) ((unsigned long long*)(etab_pool + (u64)blockIdx.x * (etab_bytes + 256) + etab_bytes + 32))[k] = lds.prof[k];
#endif
}
#endif

} // namespace sqz
