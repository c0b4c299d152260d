// sq_zstd_decode.h -- Zstandard frame decoder (RFC 8878) on the GPU: the unpack side of SURVEY.md 8(f) row f2.
//
// Replaces, for a batch of chunk records, the reference's
//     zstd::bulk::decompress(&compressed_data, orig_size)          /root/reference/src/archive/reader.rs:259-314
// (zstd-sys 2.0.15 = C libzstd 1.5.7, not vendored under /root/reference).  One wavefront per frame (chunks are
// independent).  It decodes everything a conforming encoder may emit without a dictionary: raw / RLE / compressed
// blocks, raw / RLE / Huffman (1 or 4 streams, direct or FSE-coded weights) / treeless literals, predefined / RLE /
// FSE / repeat sequence tables, repeat offsets, several concatenated frames, skippable frames; a content checksum is
// verified (XXH64, as libzstd does).  Same statement as the host decoder squishrs_amd/host/zstd_frame_decoder.hpp,
// re-cut for a wave:
//   * everything that is a chain of dependent bit reads -- headers, table descriptions, the three interleaved FSE
//     states of the sequences -- runs as a serial region on lane 0 with its tables in LDS;
//   * the four Huffman streams of a literals section are decoded side by side by lanes 0..3 (64-bit bit windows,
//     4 KiB decode table in LDS) into a per-wave literal buffer in global memory;
//   * sequences are decoded a TILE of 64 at a time (lane 0) and executed by the whole wave: output positions by a
//     prefix sum over the tile, literal runs and non-overlapping short matches one sequence per lane with up to 64
//     bytes in flight per lane, long or self-overlapping matches by all 64 lanes together; a match that reads what an
//     earlier match of the same tile writes waits for it (rounds inside the tile, one memory round trip each).
// Under -DSQ_SIM (tests/sim) every wave primitive has a scalar body, so the control logic runs on a CPU against the
// golden frames of the real libzstd (tests/golden/zstd_decode_frames.json).
#pragma once
#include "sq_wave.h"

namespace sqdz {

// status of one input (a chunk record's compressed bytes); the C ABI reports them per chunk
enum : u32 {
    DZ_OK = 0,
    DZ_CORRUPT = 1,      // "Data corruption detected"
    DZ_SRCSIZE = 2,      // "Src size is incorrect"
    DZ_UNKNOWN = 3,      // "Unknown frame descriptor"
    DZ_UNSUPPORTED = 4,  // "Unsupported frame parameter"
    DZ_DICT = 5,         // "Dictionary mismatch"
    DZ_DSTSIZE = 6,      // "Destination buffer is too small"
    DZ_CHECKSUM = 7      // "Restored data doesn't match checksum"
};

#define SQDZ_BLOCK_MAX (128u << 10)
#define SQDZ_LITBUF_BYTES (SQDZ_BLOCK_MAX + 64u) // per wave, global memory
#define SQDZ_TILE 64u

// LDS of one decoding wave (12.1 KB: 13 waves per CU)
struct DecLds {
    u16 huf[2048];      // Huffman decode table: symbol | nbBits << 8
    u32 fse[3][512];    // LL / OF / ML decode tables: new-state base (16) | nbBits (8) << 16 | symbol (8) << 24
    u32 fseLog[3];
    u32 have[3];        // the table of LL / OF / ML exists (repeat mode may use it)
    u32 hufValid, hufBits;
    u32 tile[3][SQDZ_TILE]; // ll, ml, offset of the current tile of sequences
    u32 ret[16];        // results of the serial regions
    u8 weights[256];
    s16 norm[64];
    u16 next[64];
    // the sequence codes' base values and extra-bit counts (RFC 8878 3.1.1.3.2.1.1), copied from constant memory once per
    // kernel: indexed by lane 0 once per SEQUENCE -- from constant memory each look-up was a trip to L2 (measured: 15 ms per
    // 128 KiB block of text, four fifths of the decoder's time)
    u32 llBase[36], mlBase[53];
    u8 llBits[36], mlBits[53];
};
// fills the code tables of a wave's LDS (all lanes; call once before the first frame)
SQ_DEV void dec_lds_init(DecLds* L);

// ---- sequence code tables (RFC 8878 3.1.1.3.2.1)
#if SQ_ON_DEVICE
#define SQDZ_CONST __constant__ static const
#else
#define SQDZ_CONST static const
#endif
SQDZ_CONST u32 kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                             48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
SQDZ_CONST u8 kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
SQDZ_CONST u32 kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
                                             31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                             16387, 32771, 65539};
SQDZ_CONST u8 kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                            0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
SQDZ_CONST s16 kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
SQDZ_CONST s16 kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
SQDZ_CONST s16 kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

SQ_DEV void dec_lds_init(DecLds* L) {
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    if (lane < 36) {
        L->llBase[lane] = kLLBase[lane];
        L->llBits[lane] = kLLBits[lane];
    }
    if (lane < 53) {
        L->mlBase[lane] = kMLBase[lane];
        L->mlBits[lane] = kMLBits[lane];
    }
    wv_fence();
#else
    for (u32 i = 0; i < 36; i++) {
        L->llBase[i] = kLLBase[i];
        L->llBits[i] = kLLBits[i];
    }
    for (u32 i = 0; i < 53; i++) {
        L->mlBase[i] = kMLBase[i];
        L->mlBits[i] = kMLBits[i];
    }
#endif
}

// ---- forward (little-endian) bit reader over [p, p + n): table descriptions
struct FwdBits {
    const u8* p;
    u32 n, bit;
    bool bad;
};
SQ_DEV u32 fwd_read(FwdBits& b, u32 k) { // k <= 16
    u32 const byte = b.bit >> 3;
    if (((b.bit + k + 7) >> 3) > b.n) {
        b.bad = true;
        b.bit += k;
        return 0;
    }
    u32 w = 0;
    for (u32 i = 0; i < 3 && byte + i < b.n; i++) w |= sq_grd8(b.p + byte + i) << (8 * i);
    u32 const v = (w >> (b.bit & 7)) & ((1u << k) - 1u);
    b.bit += k;
    return v;
}

// ---- backward bit reader: Huffman and FSE payloads are written forward and read from their end; the last byte
// carries a 1 bit that marks the end.  Reading beyond the beginning yields zero bits (pos goes negative: the format
// lets the last reads run over, the caller checks).  A 64-bit window over bytes [wb, wb + 8) is kept in registers.
struct RevBits {
    const u8* p;
    u32 n;
    s32 pos;  // number of unread bits
    u64 w;    // bytes [wb, wb + 8) of the stream (zero beyond its end)
    s32 wb;   // byte index of the window, -1000: none loaded
};
SQ_DEV u64 rev_load(const u8* p, u32 n, s32 byte) { // bytes [byte, byte + 8) of the stream, zeros outside [0, n)
    if (byte >= 0 && (u32)byte + 8 <= n) return sq_grd64(p + byte);
    u64 w = 0;
    for (s32 i = 0; i < 8; i++) {
        s32 const b = byte + i;
        if (b >= 0 && (u32)b < n) w |= (u64)sq_grd8(p + b) << (8 * i);
    }
    return w;
}
SQ_DEV bool rev_init(RevBits& b, const u8* p, u32 n) {
    b.p = p;
    b.n = n;
    b.wb = -1000;
    b.w = 0;
    b.pos = 0;
    if (n == 0) return false;
    u32 const last = sq_grd8(p + n - 1);
    if (last == 0) return false;
    b.pos = (s32)((n - 1) * 8 + sq_highbit32(last));
    return true;
}
// the k bits below pos (k <= 32), without consuming them
SQ_DEV u32 rev_peek(RevBits& b, u32 k) {
    if (k == 0) return 0;
    s32 const lo = b.pos - (s32)k; // may be negative: those bits read as zero
    // the window must cover bits [max(lo, 0), pos)
    s32 const need = (lo < 0 ? 0 : lo) >> 3;
    if (need < b.wb || b.pos > 8 * (b.wb + 8) || b.wb == -1000) {
        s32 nb = ((b.pos + 7) >> 3) - 8; // window ends at the byte that holds bit pos - 1
        if (nb < 0) nb = 0;
        if (nb > need) nb = need;
        b.wb = nb;
        b.w = rev_load(b.p, b.n, nb);
    }
    if (lo >= 0) return (u32)(b.w >> (lo - 8 * b.wb)) & (u32)((1ull << k) - 1ull);
    if (b.pos <= 0) return 0;
    // pos bits are left: they form the TOP of the value, zeros below
    u32 const have = (u32)b.pos;
    u32 const v = (u32)(b.w >> (0 - 8 * b.wb)) & (u32)((1ull << have) - 1ull);
    return v << (k - have);
}
SQ_DEV u32 rev_read(RevBits& b, u32 k) {
    u32 const v = rev_peek(b, k);
    b.pos -= (s32)k;
    return v;
}

// ---- FSE decoding table from normalised counts (serial; table in LDS as base | nb << 16 | sym << 24)
SQ_DEV bool fse_build(SQ_L u32* tab, SQ_L u16* next, const SQ_L s16* norm, u32 nsym, u32 log) {
    u32 const size = 1u << log;
    u32 high = size;
    for (u32 s = 0; s < nsym; s++) {
        if (norm[s] == -1) {
            tab[--high] = s << 24;
            next[s] = 1;
        } else
            next[s] = (u16)norm[s];
    }
    u32 const step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    u32 pos = 0;
    for (u32 s = 0; s < nsym; s++) {
        for (s32 i = 0; i < norm[s]; i++) {
            tab[pos] = s << 24;
            do {
                pos = (pos + step) & mask;
            } while (pos >= high);
        }
    }
    if (pos != 0) return false;
    for (u32 i = 0; i < size; i++) {
        u32 const s = tab[i] >> 24;
        u32 const x = next[s]++;
        if (x == 0) return false;
        u32 const nb = log - sq_highbit32(x);
        tab[i] = (((x << nb) - size) & 0xFFFFu) | (nb << 16) | (s << 24);
    }
    return true;
}

// FSE table description ("NCount") at p[0 .. n); norm[] in LDS.  Returns the bytes it occupied, 0 = corrupt.
SQ_DEV u32 read_ncount(const u8* p, u32 n, u32 max_sym, u32 max_log, SQ_L s16* norm, u32* nsym, u32* log) {
    FwdBits b = {p, n, 0, false};
    u32 const al = 5 + fwd_read(b, 4);
    if (al > max_log || b.bad) return 0;
    s32 remaining = 1 << al;
    u32 s = 0;
    while (remaining > 0 && s <= max_sym) {
        u32 const bits = sq_highbit32((u32)remaining + 1) + 1;
        u32 val = fwd_read(b, bits);
        u32 const lower = (1u << (bits - 1)) - 1;
        u32 const threshold = (1u << bits) - 1 - ((u32)remaining + 1);
        if ((val & lower) < threshold) {
            b.bit -= 1;
            val &= lower;
        } else if (val > lower)
            val -= threshold;
        s32 const proba = (s32)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        norm[s++] = (s16)proba;
        if (proba == 0) {
            u32 rep = fwd_read(b, 2);
            for (;;) {
                for (u32 i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
                if (rep != 3 || b.bad) break;
                rep = fwd_read(b, 2);
            }
        }
        if (b.bad) return 0;
    }
    if (remaining != 0 || s > max_sym + 1 || b.bad) return 0;
    *nsym = s;
    *log = al;
    return (b.bit + 7) >> 3;
}

// Huffman decode table from the weights (weights[0 .. n), the deduced last one included).  Serial.
SQ_DEV bool huf_build(DecLds* L, u32 n) {
    SQ_L u8* const weights = (SQ_L u8*)L->weights;
    SQ_L u16* const tab = (SQ_L u16*)L->huf;
    u32 sum = 0;
    for (u32 i = 0; i < n; i++) {
        if (weights[i] > 12) return false;
        if (weights[i]) sum += 1u << (weights[i] - 1);
    }
    if (sum == 0) return false;
    u32 const maxBits = sq_highbit32(sum);
    if ((1u << maxBits) != sum || maxBits > 11 || maxBits == 0) return false;
    u32 const size = 1u << maxBits;
    u32 rankCount[13], rankIdx[13];
    for (u32 i = 0; i < 13; i++) rankCount[i] = 0;
    for (u32 i = 0; i < n; i++) {
        u32 const bits = weights[i] ? maxBits + 1 - weights[i] : 0;
        rankCount[bits]++;
    }
    rankIdx[maxBits] = 0;
    for (u32 i = maxBits; i >= 1; i--) rankIdx[i - 1] = rankIdx[i] + rankCount[i] * (1u << (maxBits - i));
    if (rankIdx[0] != size) return false;
    for (u32 i = 0; i < n; i++) {
        if (!weights[i]) continue;
        u32 const bits = maxBits + 1 - weights[i];
        u32 const len = 1u << (maxBits - bits);
        u32 const at = rankIdx[bits];
        for (u32 k = 0; k < len; k++) tab[at + k] = (u16)(i | (bits << 8));
        rankIdx[bits] = at + len;
    }
    L->hufBits = maxBits;
    L->hufValid = 1;
    return true;
}

// Huffman tree description at p[0 .. n): returns its size in bytes, 0 = corrupt.  Serial.
SQ_DEV u32 read_huf_tree(DecLds* L, const u8* p, u32 n) {
    if (n < 1) return 0;
    SQ_L u8* const weights = (SQ_L u8*)L->weights;
    u32 nw = 0, used;
    u32 const hb = sq_grd8(p);
    if (hb >= 128) { // 4-bit weights, two per byte
        nw = hb - 127;
        used = 1 + (nw + 1) / 2;
        if (used > n) return 0;
        for (u32 i = 0; i < nw; i++) {
            u32 const by = sq_grd8(p + 1 + i / 2);
            weights[i] = (u8)((i & 1) ? (by & 15) : (by >> 4));
        }
    } else { // FSE-coded weights: two interleaved states until the stream runs dry
        used = 1 + hb;
        if (hb == 0 || used > n) return 0;
        u32 ns, al;
        u32 const hdr = read_ncount(p + 1, hb, 12, 6, (SQ_L s16*)L->norm, &ns, &al);
        if (hdr == 0 || hdr >= hb) return 0;
        SQ_L u32* const wt = (SQ_L u32*)L->tile[0]; // (the weights' table, at most 2^6 entries, borrows the tile area: the sequence
                                                     // tables may have to survive for a block that repeats them)
        if (!fse_build(wt, (SQ_L u16*)L->next, (const SQ_L s16*)L->norm, ns, al)) return 0; // 2^6 = 64 entries
        RevBits b;
        if (!rev_init(b, p + 1 + hdr, hb - hdr)) return 0;
        u32 s1 = rev_read(b, al), s2 = rev_read(b, al);
        for (;;) {
            if (nw >= 254) return 0;
            weights[nw++] = (u8)(wt[s1] >> 24);
            s1 = (wt[s1] & 0xFFFFu) + rev_read(b, (wt[s1] >> 16) & 0xFFu);
            if (b.pos < 0) {
                weights[nw++] = (u8)(wt[s2] >> 24);
                break;
            }
            if (nw >= 254) return 0;
            weights[nw++] = (u8)(wt[s2] >> 24);
            s2 = (wt[s2] & 0xFFFFu) + rev_read(b, (wt[s2] >> 16) & 0xFFu);
            if (b.pos < 0) {
                weights[nw++] = (u8)(wt[s1] >> 24);
                break;
            }
        }
    }
    // the last weight completes the sum to a power of two
    u32 sum = 0;
    for (u32 i = 0; i < nw; i++) {
        if (weights[i] > 12) return 0;
        if (weights[i]) sum += 1u << (weights[i] - 1);
    }
    if (sum == 0) return 0;
    u32 const maxBits = sq_highbit32(sum) + 1;
    u32 const left = (1u << maxBits) - sum;
    if (left & (left - 1)) return 0;
    weights[nw++] = (u8)(sq_highbit32(left) + 1);
    if (!huf_build(L, nw)) return 0;
    return used;
}

// One Huffman stream p[0 .. n) -> count symbols at out (global memory); false = corrupt.  One lane.
SQ_DEV bool huf_decode_stream(const DecLds* L, const u8* p, u32 n, u8* out, u32 count) {
    const SQ_L u16* const tab = (const SQ_L u16*)L->huf;
    u32 const maxBits = L->hufBits;
    RevBits b;
    if (!rev_init(b, p, n)) return false;
    u32 i = 0;
    // four symbols per store where the output allows it
    for (; i + 4 <= count; i += 4) {
        u32 v = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            u32 const e = tab[rev_peek(b, maxBits)];
            v |= (e & 0xFFu) << (8 * k);
            b.pos -= (s32)(e >> 8);
        }
#if SQ_ON_DEVICE
        *(SQ_G sq_u32_ua*)(out + i) = v;
#else
        memcpy(out + i, &v, 4);
#endif
    }
    for (; i < count; i++) {
        u32 const e = tab[rev_peek(b, maxBits)];
        sq_gwr8(out + i, e & 0xFFu);
        b.pos -= (s32)(e >> 8);
    }
    return b.pos == 0; // every bit of the stream belongs to a symbol
}

// ---- copies.  dst: the output (global).  Stores are not waited for here: the caller keeps track of what is complete.
// n bytes from a read-only source (frame bytes, literal buffer), all lanes together
SQ_DEV void wcopy_in(u8* dst, const u8* src, u32 n) {
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u32 pos = 16 * lane;
    for (; pos + 3 * 1024 + 16 <= n; pos += 4096) {
        sq_v4u v[4];
#pragma unroll
        for (u32 k = 0; k < 4; k++) v[k] = sq_grd128(src + pos + 1024 * k);
#pragma unroll
        for (u32 k = 0; k < 4; k++) sq_gwr128(dst + pos + 1024 * k, v[k]);
    }
    for (; pos + 16 <= n; pos += 1024) sq_gwr128(dst + pos, sq_grd128(src + pos));
    // the tail (< 16 bytes beyond the last whole 16): one byte per lane
    u32 const tail0 = n & ~15u;
    if (tail0 + lane < n) sq_gwr8(dst + tail0 + lane, sq_grd8(src + tail0 + lane));
#else
    memcpy(dst, src, n);
#endif
}
SQ_DEV void wfill(u8* dst, u32 byte, u32 n) {
#if SQ_ON_DEVICE
    for (u32 k = wv_lane(); k < n; k += 64) sq_gwr8(dst + k, byte);
#else
    memset(dst, (int)byte, n);
#endif
}
// a match, all lanes together: n bytes from dst - off to dst, every byte read from the off bytes in front of dst (a
// self-overlapping match repeats them).  The source was written by this wave: the caller has waited for those stores and
// dropped the CU's L1 lines (wv_mem_sync_inv) since, so plain loads see them.
SQ_DEV void wcopy_match(u8* dst, u32 off, u32 n) {
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    const u8* const src = dst - off;
    if (off >= n && n >= 64) { // plain copy, 16 bytes per lane (unaligned)
        u32 pos = 16 * lane;
        for (; pos + 16 <= n; pos += 1024) {
            sq_gwr128(dst + pos, sq_grd128(src + pos));
        }
        u32 const tail0 = n & ~15u;
        if (tail0 + lane < n) sq_gwr8(dst + tail0 + lane, sq_grd8(src + tail0 + lane));
        return;
    }
    // byte k comes from src[k mod off]
    u32 k = lane, m = lane % off;
    u32 const stepm = 64 % off;
    for (; k < n; k += 64) {
        sq_gwr8(dst + k, sq_grd8(src + m));
        m += stepm;
        if (m >= off) m -= off;
    }
#else
    for (u32 k = 0; k < n; k++) dst[k] = dst[(s64)k - (s64)off];
#endif
}

// XXH64 (seed 0) of p[0 .. n), for the content checksum.  Serial; p is this wave's own output, complete in memory.
SQ_DEV u64 xxh64_round(u64 acc, u64 input) {
    acc += input * 0xC2B2AE3D27D4EB4FULL;
    acc = (acc << 31) | (acc >> 33);
    return acc * 0x9E3779B185EBCA87ULL;
}
SQ_DEV u64 xxh64_merge(u64 h, u64 v) {
    h ^= xxh64_round(0, v);
    return h * 0x9E3779B185EBCA87ULL + 0x85EBCA77C2B2AE63ULL;
}
SQ_DEV u64 xxh64_serial(const u8* p, u64 n) {
    u64 const P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL,
              P5 = 0x27D4EB2F165667C5ULL;
    u64 h, i = 0;
    if (n >= 32) {
        u64 v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; i + 32 <= n; i += 32) {
            v1 = xxh64_round(v1, sq_grd64(p + i));
            v2 = xxh64_round(v2, sq_grd64(p + i + 8));
            v3 = xxh64_round(v3, sq_grd64(p + i + 16));
            v4 = xxh64_round(v4, sq_grd64(p + i + 24));
        }
        h = ((v1 << 1) | (v1 >> 63)) + ((v2 << 7) | (v2 >> 57)) + ((v3 << 12) | (v3 >> 52)) + ((v4 << 18) | (v4 >> 46));
        h = xxh64_merge(h, v1);
        h = xxh64_merge(h, v2);
        h = xxh64_merge(h, v3);
        h = xxh64_merge(h, v4);
    } else
        h = P5;
    h += n;
    for (; i + 8 <= n; i += 8) {
        h ^= xxh64_round(0, sq_grd64(p + i));
        h = ((h << 27) | (h >> 37)) * P1 + P4;
    }
    if (i + 4 <= n) {
        h ^= (u64)sq_grd32(p + i) * P1;
        h = ((h << 23) | (h >> 41)) * P2 + P3;
        i += 4;
    }
    for (; i < n; i++) {
        h ^= (u64)sq_grd8(p + i) * P5;
        h = ((h << 11) | (h >> 53)) * P1;
    }
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}

// ---- one compressed block p[0 .. n) -> dst[op ..); frameStart: where the frame's output began (offsets reach no
// further back).  Returns DZ_*; *opOut = the new output position.
SQ_DEV u32 decode_block(DecLds* L, const u8* p, u32 n, u8* dst, u32 cap, u32 frameStart, u32 op, u8* litbuf, u32* opOut) {
    SQ_L u32* const ret = (SQ_L u32*)L->ret;
    // ---- literals section header (serial): ret[0] status, [1] type, [2] regenerated size, [3] compressed size, [4] streams, [5] header size
    WV_SERIAL {
        u32 st = DZ_OK, regen = 0, csize = 0, hdr = 0, streams = 1, ltype = 0;
        if (n < 1)
            st = DZ_CORRUPT;
        else {
            u32 const b0 = sq_grd8(p);
            ltype = b0 & 3;
            u32 const sf = (b0 >> 2) & 3;
            if (ltype < 2) {
                if (sf == 0 || sf == 2) {
                    regen = b0 >> 3;
                    hdr = 1;
                } else if (sf == 1) {
                    if (n < 2) st = DZ_CORRUPT; else regen = (b0 >> 4) | (sq_grd8(p + 1) << 4);
                    hdr = 2;
                } else {
                    if (n < 3) st = DZ_CORRUPT; else regen = (b0 >> 4) | (sq_grd8(p + 1) << 4) | (sq_grd8(p + 2) << 12);
                    hdr = 3;
                }
            } else {
                if (sf == 0 || sf == 1) {
                    if (n < 3) st = DZ_CORRUPT;
                    else {
                        u32 const v = b0 | (sq_grd8(p + 1) << 8) | (sq_grd8(p + 2) << 16);
                        regen = (v >> 4) & 0x3FF;
                        csize = (v >> 14) & 0x3FF;
                    }
                    hdr = 3;
                    streams = sf == 0 ? 1 : 4;
                } else if (sf == 2) {
                    if (n < 4) st = DZ_CORRUPT;
                    else {
                        u32 const v = b0 | (sq_grd8(p + 1) << 8) | (sq_grd8(p + 2) << 16) | (sq_grd8(p + 3) << 24);
                        regen = (v >> 4) & 0x3FFF;
                        csize = (v >> 18) & 0x3FFF;
                    }
                    hdr = 4;
                    streams = 4;
                } else {
                    if (n < 5) st = DZ_CORRUPT;
                    else {
                        u64 const v = (u64)b0 | ((u64)sq_grd8(p + 1) << 8) | ((u64)sq_grd8(p + 2) << 16) | ((u64)sq_grd8(p + 3) << 24) |
                                      ((u64)sq_grd8(p + 4) << 32);
                        regen = (u32)((v >> 4) & 0x3FFFF);
                        csize = (u32)((v >> 22) & 0x3FFFF);
                    }
                    hdr = 5;
                    streams = 4;
                }
            }
            if (regen > SQDZ_BLOCK_MAX) st = DZ_CORRUPT;
        }
        ret[0] = st;
        ret[1] = ltype;
        ret[2] = regen;
        ret[3] = csize;
        ret[4] = streams;
        ret[5] = hdr;
    }
    wv_fence();
    if (wv_uni(ret[0])) return wv_uni(ret[0]);
    u32 const ltype = wv_uni(ret[1]), regen = wv_uni(ret[2]), csize = wv_uni(ret[3]), streams = wv_uni(ret[4]);
    u32 at = wv_uni(ret[5]);
    const u8* litSrc = litbuf; // where the literals are read from
    u32 litSafe = SQDZ_LITBUF_BYTES; // bytes that may be READ from litSrc (the lane-parallel copies read whole 8-byte words)
    u32 rleByte = 0;
    bool litRle = false;
    if (ltype == 0) {
        if (at + regen > n) return DZ_CORRUPT;
        litSrc = p + at;
        litSafe = n - at;
        at += regen;
    } else if (ltype == 1) {
        if (at + 1 > n) return DZ_CORRUPT;
        rleByte = srd8(p + at);
        litRle = true;
        at += 1;
    } else {
        if (at + csize > n) return DZ_CORRUPT;
        // tree (serial), then the streams: one lane each
        WV_SERIAL {
            u32 st = DZ_OK, t = 0;
            if (ltype == 2) {
                t = read_huf_tree(L, p + at, csize);
                if (t == 0) st = DZ_CORRUPT;
            } else if (!L->hufValid)
                st = DZ_CORRUPT;
            ret[0] = st;
            ret[1] = t;
        }
        wv_fence();
        if (wv_uni(ret[0])) return DZ_CORRUPT;
        u32 const t = wv_uni(ret[1]);
        const u8* const q = p + at + t;
        u32 const left = csize - t;
        if (t > csize) return DZ_CORRUPT;
        if (streams == 1) {
            WV_SERIAL ret[0] = huf_decode_stream(L, q, left, litbuf, regen) ? DZ_OK : DZ_CORRUPT;
        } else {
            if (left < 6) return DZ_CORRUPT;
            u32 const s1 = srd16(q), s2 = srd16(q + 2), s3 = srd16(q + 4);
            if (6 + s1 + s2 + s3 > left) return DZ_CORRUPT;
            u32 const s4 = left - 6 - s1 - s2 - s3;
            u32 const seg = (regen + 3) / 4;
            if (3 * seg > regen) return DZ_CORRUPT;
            WV_SERIAL ret[0] = DZ_OK;
            wv_fence();
#if SQ_ON_DEVICE
            u32 const lane = wv_lane();
            if (lane < 4) {
                u32 const so = lane == 0 ? 0u : lane == 1 ? s1 : lane == 2 ? s1 + s2 : s1 + s2 + s3;
                u32 const sn = lane == 0 ? s1 : lane == 1 ? s2 : lane == 2 ? s3 : s4;
                u32 const cnt = lane < 3 ? seg : regen - 3 * seg;
                if (!huf_decode_stream(L, q + 6 + so, sn, litbuf + lane * seg, cnt)) ret[0] = DZ_CORRUPT;
            }
#else
            u32 const so[4] = {0, s1, s1 + s2, s1 + s2 + s3}, sn[4] = {s1, s2, s3, s4};
            for (u32 k = 0; k < 4; k++)
                if (!huf_decode_stream(L, q + 6 + so[k], sn[k], litbuf + k * seg, k < 3 ? seg : regen - 3 * seg)) ret[0] = DZ_CORRUPT;
#endif
        }
        wv_mem_sync(); // the literal buffer is complete before anybody copies from it
        if (wv_uni(ret[0])) return DZ_CORRUPT;
        at += csize;
    }
    // ---- sequences section header and tables (serial): ret[0] status, [1] number of sequences, [2] position of the bit stream
    WV_SERIAL {
        u32 st = DZ_OK, nseq = 0, a = at;
        if (a >= n)
            st = DZ_CORRUPT;
        else {
            nseq = sq_grd8(p + a++);
            if (nseq >= 128) {
                if (nseq == 255) {
                    if (a + 2 > n) st = DZ_CORRUPT; else nseq = sq_grd8(p + a) + (sq_grd8(p + a + 1) << 8) + 0x7F00;
                    a += 2;
                } else {
                    if (a + 1 > n) st = DZ_CORRUPT; else nseq = ((nseq - 128) << 8) + sq_grd8(p + a);
                    a += 1;
                }
            }
        }
        if (!st && nseq) {
            if (a >= n)
                st = DZ_CORRUPT;
            else {
                u32 const modes = sq_grd8(p + a++);
                if (modes & 3) st = DZ_CORRUPT;
                for (u32 k = 0; k < 3 && !st; k++) {
                    u32 const mode = (modes >> (6 - 2 * k)) & 3;
                    u32 const maxSym = k == 0 ? 35u : k == 1 ? 31u : 52u, maxLog = k == 1 ? 8u : 9u;
                    SQ_L u32* const tab = (SQ_L u32*)L->fse[k];
                    if (mode == 0) { // predefined distribution
                        u32 const dn = k == 0 ? 36u : k == 1 ? 29u : 53u, dl = k == 1 ? 5u : 6u;
                        for (u32 i = 0; i < dn; i++) L->norm[i] = k == 0 ? kLLDefault[i] : k == 1 ? kOFDefault[i] : kMLDefault[i];
                        if (!fse_build(tab, (SQ_L u16*)L->next, (const SQ_L s16*)L->norm, dn, dl)) st = DZ_CORRUPT;
                        L->fseLog[k] = dl;
                        L->have[k] = 1;
                    } else if (mode == 1) { // one symbol
                        if (a >= n || sq_grd8(p + a) > maxSym)
                            st = DZ_CORRUPT;
                        else {
                            tab[0] = sq_grd8(p + a) << 24;
                            L->fseLog[k] = 0;
                            L->have[k] = 1;
                            a += 1;
                        }
                    } else if (mode == 2) {
                        u32 ns = 0, al = 0;
                        u32 const used = a < n ? read_ncount(p + a, n - a, maxSym, maxLog, (SQ_L s16*)L->norm, &ns, &al) : 0u;
                        if (used == 0 || !fse_build(tab, (SQ_L u16*)L->next, (const SQ_L s16*)L->norm, ns, al))
                            st = DZ_CORRUPT;
                        else {
                            L->fseLog[k] = al;
                            L->have[k] = 1;
                            a += used;
                        }
                    } else if (!L->have[k])
                        st = DZ_CORRUPT;
                }
                if (!st && a >= n) st = DZ_CORRUPT;
            }
        }
        ret[0] = st;
        ret[1] = nseq;
        ret[2] = a;
    }
    wv_fence();
    if (wv_uni(ret[0])) return DZ_CORRUPT;
    u32 const nseq = wv_uni(ret[1]);
    at = wv_uni(ret[2]);
    auto putLits = [&](u32 o, u32 from, u32 len) { // len literals starting at literal number `from` -> dst + o
        if (!len) return;
        if (litRle)
            wfill(dst + o, rleByte, len);
        else
            wcopy_in(dst + o, litSrc + from, len);
    };
    if (nseq == 0) {
        if (at != n) return DZ_CORRUPT;
        if ((u64)op + regen > cap) return DZ_DSTSIZE;
        putLits(op, 0, regen);
        *opOut = op + regen;
        return DZ_OK;
    }
    // ---- sequences: decoded a tile at a time by lane 0 (its registers carry the reader and the states from tile to tile),
    // executed by the wave
    RevBits b;
    u32 sLL = 0, sOF = 0, sML = 0, rep0 = 0, rep1 = 0, rep2 = 0;
    WV_SERIAL {
        u32 st = DZ_OK;
        if (!rev_init(b, p + at, n - at))
            st = DZ_CORRUPT;
        else {
            sLL = rev_read(b, L->fseLog[0]);
            sOF = rev_read(b, L->fseLog[1]);
            sML = rev_read(b, L->fseLog[2]);
            if (b.pos < 0) st = DZ_CORRUPT;
        }
        rep0 = L->ret[8];
        rep1 = L->ret[9];
        rep2 = L->ret[10];
        ret[0] = st;
    }
    wv_fence();
    if (wv_uni(ret[0])) return DZ_CORRUPT;
    u32 litPos = 0;
    u32 synced = op; // output below this position is complete in memory (no store of this wave pending)
    for (u32 s0 = 0; s0 < nseq; s0 += SQDZ_TILE) {
        u32 const cnt = nseq - s0 < SQDZ_TILE ? nseq - s0 : SQDZ_TILE;
        WV_SERIAL {
            u32 st = DZ_OK;
            const SQ_L u32* const tLL = (const SQ_L u32*)L->fse[0];
            const SQ_L u32* const tOF = (const SQ_L u32*)L->fse[1];
            const SQ_L u32* const tML = (const SQ_L u32*)L->fse[2];
            for (u32 i = 0; i < cnt && !st; i++) {
                u32 const eLL = tLL[sLL], eOF = tOF[sOF], eML = tML[sML];
                u32 const ofCode = eOF >> 24, mlCode = eML >> 24, llCode = eLL >> 24;
                if (ofCode > 31 || mlCode > 52 || llCode > 35) {
                    st = DZ_CORRUPT;
                    break;
                }
                // (an offset code above 24 has more extra bits than one refill of the window holds: two reads)
                u32 ofExtra;
                if (ofCode > 24) {
                    u32 const hi = rev_read(b, ofCode - 16);
                    ofExtra = (hi << 16) | rev_read(b, 16);
                } else
                    ofExtra = rev_read(b, ofCode);
                u64 const ofVal = (1ull << ofCode) + ofExtra;
                u32 const ml = ((const SQ_L u32*)L->mlBase)[mlCode] + rev_read(b, ((const SQ_L u8*)L->mlBits)[mlCode]);
                u32 const ll = ((const SQ_L u32*)L->llBase)[llCode] + rev_read(b, ((const SQ_L u8*)L->llBits)[llCode]);
                if (b.pos < 0) {
                    st = DZ_CORRUPT;
                    break;
                }
                u64 offset;
                if (ofVal > 3) {
                    offset = ofVal - 3;
                    rep2 = rep1;
                    rep1 = rep0;
                    rep0 = (u32)offset;
                } else {
                    u32 const idx = (u32)ofVal - 1 + (ll == 0 ? 1u : 0u); // 0..3
                    if (idx == 0)
                        offset = rep0;
                    else {
                        offset = idx == 1 ? rep1 : idx == 2 ? rep2 : rep0 - 1;
                        if (offset == 0) {
                            st = DZ_CORRUPT;
                            break;
                        }
                        if (idx > 1) rep2 = rep1;
                        rep1 = rep0;
                        rep0 = (u32)offset;
                    }
                }
                if (offset > 0xFFFFFFFFull) {
                    st = DZ_CORRUPT;
                    break;
                }
                if (s0 + i + 1 < nseq) { // states are refreshed in the order LL, ML, OF
                    sLL = (eLL & 0xFFFFu) + rev_read(b, (eLL >> 16) & 0xFFu);
                    sML = (eML & 0xFFFFu) + rev_read(b, (eML >> 16) & 0xFFu);
                    sOF = (eOF & 0xFFFFu) + rev_read(b, (eOF >> 16) & 0xFFu);
                    if (b.pos < 0) {
                        st = DZ_CORRUPT;
                        break;
                    }
                }
                L->tile[0][i] = ll;
                L->tile[1][i] = ml;
                L->tile[2][i] = (u32)offset;
            }
            if (!st && s0 + cnt == nseq && b.pos != 0) st = DZ_CORRUPT; // every bit belongs to a sequence
            ret[0] = st;
        }
        wv_fence();
        if (wv_uni(ret[0])) return DZ_CORRUPT;
        // ---- execute the tile.  Lane i holds sequence i.
#if SQ_ON_DEVICE
        u32 const lane = wv_lane();
        bool const mine = lane < cnt;
        const SQ_L u32* const tl = (const SQ_L u32*)L->tile[0];
        u32 const ll = mine ? tl[lane] : 0u, ml = mine ? tl[SQDZ_TILE + lane] : 0u, off = mine ? tl[2 * SQDZ_TILE + lane] : 1u;
        u32 totLL = 0, totAll = 0;
        u32 const litAt = litPos + wv_excl_scan(ll, &totLL);       // first literal of this lane's sequence
        u32 const outAt = op + wv_excl_scan(ll + ml, &totAll);     // where its output starts
        if (litPos + totLL > regen) return DZ_CORRUPT;
        if ((u64)op + totAll > cap) return DZ_DSTSIZE;
        u32 const mDst = outAt + ll;                                // where its match goes
        // an offset reaches no further back than the frame's first byte
        if (wv_ballot(mine && off > mDst - frameStart)) return DZ_CORRUPT;
        // literal runs: one sequence per lane, up to 32 bytes each in flight; longer runs by the whole wave
        {
            bool const shortRun = mine && ll <= 32 && !litRle && litAt + 32 <= litSafe;
            if (wv_ballot(shortRun)) {
                u64 v[4];
#pragma unroll
                for (u32 k = 0; k < 4; k++) v[k] = (shortRun && 8 * k < ll) ? sq_grd64(litSrc + litAt + 8 * k) : 0ull; // (the literal buffer has 64 bytes of slack; raw literals are followed by the sequences section)
#pragma unroll
                for (u32 k = 0; k < 4; k++)
                    if (shortRun && 8 * k < ll) {
                        if (8 * k + 8 <= ll)
                            *(SQ_G sq_u64_ua*)(dst + outAt + 8 * k) = v[k];
                        else
                            for (u32 t = 0; 8 * k + t < ll; t++) sq_gwr8(dst + outAt + 8 * k + t, (u32)(v[k] >> (8 * t)) & 0xFFu);
                    }
            }
            u64 longRuns = wv_ballot(mine && ll && !shortRun);
            while (longRuns) {
                u32 const i = (u32)__builtin_ctzll(longRuns);
                longRuns &= longRuns - 1;
                putLits(wv_readlane(outAt, i), wv_readlane(litAt, i), wv_readlane(ll, i));
            }
        }
        // matches.  A match may read what the literal runs above or earlier matches of this tile wrote: rounds.  A lane is
        // READY when no earlier unfinished match of the tile writes into the bytes it reads.
        {
            u32 const srcLo = mDst - off, srcHi = off < ml ? mDst : srcLo + ml; // the bytes it reads: [srcLo, srcHi)
            // which earlier sequences' MATCH output intersects them (bit j of dep)
            u64 dep = 0;
            for (u32 j = 0; j + 1 < cnt; j++) {
                u32 const mj = wv_readlane(mDst, j), lj = wv_readlane(ml, j);
                if (lane > j && srcLo < mj + lj && srcHi > mj && lj) dep |= 1ull << j;
            }
            u64 todo = wv_ballot(mine && ml);
            bool first = true;
            while (todo) {
                // everything written so far (literal runs, earlier rounds) must be in memory before it is read
                if (first ? wv_ballot(mine && ml && srcHi > synced) != 0 : true) {
                    wv_mem_sync_inv(); // stores complete, and no line of the output cached before they landed is hit again
                    synced = op;       // everything in front of this tile is complete (its own literal runs too, but they are not one range)
                }
                first = false;
                bool const ready = ((todo >> lane) & 1) && (dep & todo) == 0;
                u64 const rdy = wv_ballot(ready);
                // short non-overlapping matches of the ready set: one per lane, up to 64 bytes in flight each
                bool const fastm = ready && ml <= 64 && off >= ml && (u64)srcLo + 64 <= cap;
                if (wv_ballot(fastm)) {
                    u64 v[8];
#pragma unroll
                    for (u32 k = 0; k < 8; k++)
                        v[k] = (fastm && 8 * k < ml) ? sq_grd64(dst + srcLo + 8 * k) : 0ull; // (reads at most 7 bytes beyond the match's source, inside dst[0 .. cap): fastm)
#pragma unroll
                    for (u32 k = 0; k < 8; k++)
                        if (fastm && 8 * k < ml) {
                            if (8 * k + 8 <= ml)
                                *(SQ_G sq_u64_ua*)(dst + mDst + 8 * k) = v[k];
                            else
                                for (u32 t = 0; 8 * k + t < ml; t++) sq_gwr8(dst + mDst + 8 * k + t, (u32)(v[k] >> (8 * t)) & 0xFFu);
                        }
                }
                // the others of the ready set: the whole wave, one after the other (they do not depend on each other)
                u64 slow = wv_ballot(ready && !fastm);
                while (slow) {
                    u32 const i = (u32)__builtin_ctzll(slow);
                    slow &= slow - 1;
                    wcopy_match(dst + wv_readlane(mDst, i), wv_readlane(off, i), wv_readlane(ml, i));
                }
                todo &= ~rdy;
            }
        }
        litPos += totLL;
        op += totAll;
#else
        for (u32 i = 0; i < cnt; i++) {
            u32 const ll = L->tile[0][i], ml = L->tile[1][i], off = L->tile[2][i];
            if (litPos + ll > regen) return DZ_CORRUPT;
            if ((u64)op + ll + ml > cap) return DZ_DSTSIZE;
            putLits(op, litPos, ll);
            litPos += ll;
            op += ll;
            if (off > op - frameStart) return DZ_CORRUPT;
            wcopy_match(dst + op, off, ml);
            op += ml;
        }
        (void)synced;
#endif
    }
    // the literals behind the last sequence
    if ((u64)op + (regen - litPos) > cap) return DZ_DSTSIZE;
    putLits(op, litPos, regen - litPos);
    op += regen - litPos;
    WV_SERIAL {
        L->ret[8] = rep0;
        L->ret[9] = rep1;
        L->ret[10] = rep2;
    }
    wv_fence();
    *opOut = op;
    return DZ_OK;
}

// Decodes every frame in src[0 .. n) into dst[0 .. cap); *outLen = bytes produced.  Returns DZ_*.
SQ_DEV u32 decode_frames_wave(const u8* src, u32 n, u8* dst, u32 cap, u8* litbuf, DecLds* L, u32* outLen) {
    SQ_L u32* const ret = (SQ_L u32*)L->ret;
    u32 at = 0, op = 0;
    *outLen = 0;
    if (n == 0) return DZ_SRCSIZE;
    while (at < n) {
        // ---- frame header (serial): ret[0] status (0xFFFF: a skippable frame, skipped), [1] new `at`, [2] flags (bit 0: content
        // size present, bit 1: checksum), [3..4] content size
        WV_SERIAL {
            u32 st = DZ_OK, a = at, fl = 0;
            u64 fcs = 0;
            if (n - a < 4)
                st = DZ_SRCSIZE;
            else {
                u32 const magic = sq_grd32(src + a);
                if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { // skippable frame
                    if (n - a < 8)
                        st = DZ_SRCSIZE;
                    else {
                        u32 const sz = sq_grd32(src + a + 4);
                        if ((u64)sz + 8 > n - a)
                            st = DZ_SRCSIZE;
                        else {
                            a += 8 + sz;
                            st = 0xFFFF;
                        }
                    }
                } else if (magic != 0xFD2FB528u)
                    st = DZ_UNKNOWN;
                else {
                    a += 4;
                    if (a >= n)
                        st = DZ_SRCSIZE;
                    else {
                        u32 const fhd = sq_grd8(src + a++);
                        u32 const fcsFlag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dictFlag = fhd & 3;
                        if (fhd & 0x08) st = DZ_UNSUPPORTED;
                        if (!st && !single) {
                            if (a >= n) st = DZ_SRCSIZE; else a++; // window descriptor: the output buffer is the window
                        }
                        u32 const dictBytes = dictFlag == 0 ? 0u : dictFlag == 1 ? 1u : dictFlag == 2 ? 2u : 4u;
                        if (!st) {
                            if (a + dictBytes > n)
                                st = DZ_SRCSIZE;
                            else {
                                u32 dictId = 0;
                                for (u32 i = 0; i < dictBytes; i++) dictId |= sq_grd8(src + a + i) << (8 * i);
                                a += dictBytes;
                                if (dictId) st = DZ_DICT;
                            }
                        }
                        u32 const fcsBytes = fcsFlag == 0 ? (single ? 1u : 0u) : (fcsFlag == 1 ? 2u : (fcsFlag == 2 ? 4u : 8u));
                        if (!st) {
                            if (a + fcsBytes > n)
                                st = DZ_SRCSIZE;
                            else {
                                for (u32 i = 0; i < fcsBytes; i++) fcs |= (u64)sq_grd8(src + a + i) << (8 * i);
                                if (fcsBytes == 2) fcs += 256;
                                a += fcsBytes;
                                fl = (fcsBytes ? 1u : 0u) | (checksum ? 2u : 0u);
                            }
                        }
                    }
                }
            }
            ret[0] = st;
            ret[1] = a;
            ret[2] = fl;
            ret[3] = (u32)fcs;
            ret[4] = (u32)(fcs >> 32);
            ret[8] = 1; // repeat offsets of a fresh frame
            ret[9] = 4;
            ret[10] = 8;
            L->hufValid = 0;
            L->have[0] = L->have[1] = L->have[2] = 0;
        }
        wv_fence();
        u32 const hst = wv_uni(ret[0]);
        at = wv_uni(ret[1]);
        if (hst == 0xFFFF) continue;
        if (hst) return hst;
        u32 const fl = wv_uni(ret[2]);
        u64 const fcs = ((u64)wv_uni(ret[4]) << 32) | wv_uni(ret[3]);
        u32 const frameStart = op;
        for (;;) {
            if (at + 3 > n) return DZ_SRCSIZE;
            u32 const bh = srd8(src + at) | (srd8(src + at + 1) << 8) | (srd8(src + at + 2) << 16);
            at += 3;
            u32 const last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
            if (type == 3) return DZ_CORRUPT;
            if (type == 0) {
                if (at + bsize > n) return DZ_SRCSIZE;
                if ((u64)op + bsize > cap) return DZ_DSTSIZE;
                wcopy_in(dst + op, src + at, bsize);
                op += bsize;
                at += bsize;
            } else if (type == 1) {
                if (at + 1 > n) return DZ_SRCSIZE;
                if ((u64)op + bsize > cap) return DZ_DSTSIZE;
                wfill(dst + op, srd8(src + at), bsize);
                op += bsize;
                at += 1;
            } else {
                if (bsize > SQDZ_BLOCK_MAX || at + bsize > n) return DZ_CORRUPT;
                u32 nop = op;
                u32 const st = decode_block(L, src + at, bsize, dst, cap, frameStart, op, litbuf, &nop);
                if (st) return st;
                op = nop;
                at += bsize;
            }
            if (last) break;
        }
        if ((fl & 1) && (u64)(op - frameStart) != fcs) return DZ_CORRUPT;
        if (fl & 2) {
            if (at + 4 > n) return DZ_SRCSIZE;
            wv_mem_sync(); // the frame's output is complete
            WV_SERIAL ret[0] = ((u32)xxh64_serial(dst + frameStart, op - frameStart) == sq_grd32(src + at)) ? DZ_OK : DZ_CHECKSUM;
            wv_fence();
            if (wv_uni(ret[0])) return DZ_CHECKSUM;
            at += 4;
        }
    }
    wv_mem_sync();
    *outLen = op;
    return DZ_OK;
}

#if SQ_ON_DEVICE
// (out of line: the queue loop below stays small and shares no register with the body)
__device__ __noinline__ u32 decode_one(DecLds* L, const u8* src, u32 n, u8* dst, u32 cap, u8* litbuf, u32* outLen) {
    return decode_frames_wave(src, n, dst, cap, litbuf, L, outLen);
}
// One wave per input.  status[i] = DZ_*, out_len[i] = bytes produced (0 on failure).  litbuf: SQDZ_LITBUF_BYTES per wave
// of the grid.  The grid's waves pull inputs from *queue until it is exhausted (every wave reaches q >= n).
__global__ void __launch_bounds__(64, 3)
k_decompress_batch(const u8* src, const u64* src_off, const u32* src_len, u32 n, u8* dst, const u64* dst_off, const u32* dst_cap,
                   u32* out_len, u32* status, u8* litbuf, u32* queue) {
    __shared__ DecLds lds;
    dec_lds_init(&lds);
    u8* const lb = litbuf + (u64)blockIdx.x * SQDZ_LITBUF_BYTES;
    for (;;) {
        u32 q = 0;
        if (wv_lane() == 0) q = atomicAdd(queue, 1u);
        q = wv_uni(q);
        if (q >= n) break;
        u32 len = 0;
        u32 const st = decode_one(&lds, src + src_off[q], src_len[q], dst + dst_off[q], dst_cap[q], lb, &len);
        if (wv_lane() == 0) {
            status[q] = st;
            out_len[q] = st ? 0u : len;
        }
    }
}
#endif

} // namespace sqdz
