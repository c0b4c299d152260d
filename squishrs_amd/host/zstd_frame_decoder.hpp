// zstd_frame_decoder.hpp -- Zstandard frame decoder (RFC 8878) for the `unpack` side of the C++ host
// driver.  It replaces the reference's call
//     zstd::bulk::decompress(&compressed_data, orig_size)      /root/reference/src/archive/reader.rs:302-303
// (zstd-sys 2.0.15 = C libzstd 1.5.7, not vendored under /root/reference).  Host-only: decoding is
// outside the GPU hot path (SURVEY.md 8(f) row f2 "host first").  It decodes everything a conforming
// encoder may emit without a dictionary: raw / RLE / compressed blocks, raw / RLE / Huffman (1 or 4
// streams, direct or FSE-coded weights) / treeless literals, predefined / RLE / FSE / repeat sequence
// tables, repeat offsets, several concatenated frames, skippable frames.  The optional content checksum
// is skipped, not verified.  Written from the format specification; independent of oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zfd {

struct DecodeError {
    std::string msg; // worded like libzstd's ZSTD_getErrorName() where the reference would show it
};

namespace detail {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

[[noreturn]] inline void corrupt() { throw DecodeError{"Data corruption detected"}; }
inline u32 highbit(u32 v) { return 31u - (u32)__builtin_clz(v); }

// ---- forward (little-endian) bit reader: frame/block headers, FSE table descriptions
struct FwdBits {
    const u8* p;
    size_t n, bit = 0;
    FwdBits(const u8* p_, size_t n_) : p(p_), n(n_) {}
    u32 read(u32 k) {
        u32 v = 0;
        for (u32 i = 0; i < k; i++, bit++) {
            if ((bit >> 3) >= n) corrupt();
            v |= (u32)((p[bit >> 3] >> (bit & 7)) & 1) << i;
        }
        return v;
    }
    void rewind(u32 k) { bit -= k; }
    size_t bytes_used() const { return (bit + 7) >> 3; }
};

// ---- backward bit reader: Huffman and FSE payloads are written forward and read from their end;
// the last byte carries a 1 bit that marks the end of the stream.  Reading beyond the beginning
// yields zero bits (the format allows the last symbols' reads to run over).
struct RevBits {
    const u8* p;
    size_t n;
    int64_t pos; // number of unread bits
    RevBits(const u8* p_, size_t n_) : p(p_), n(n_) {
        if (n == 0 || p[n - 1] == 0) corrupt();
        pos = (int64_t)(n - 1) * 8 + highbit(p[n - 1]);
    }
    u64 peek(u32 k) const { // the k bits below pos, k <= 56
        if (k == 0) return 0;
        int64_t lo = pos - (int64_t)k;
        u32 drop = 0;
        if (lo < 0) {
            drop = (u32)(-lo);
            if (drop >= k) return 0;
            lo = 0;
        }
        size_t const byte = (size_t)(lo >> 3);
        u64 w = 0;
        size_t const take = n - byte < 8 ? n - byte : 8;
        memcpy(&w, p + byte, take);
        w >>= (lo & 7);
        u32 const have = k - drop;
        w &= (have >= 64) ? ~0ull : ((1ull << have) - 1);
        return w << drop;
    }
    u64 read(u32 k) {
        u64 const v = peek(k);
        pos -= k;
        return v;
    }
};

// ---- FSE decoding table
struct FseTable {
    u32 log = 0;
    std::vector<u8> sym, nb;
    std::vector<u16> base;
    void build(const int16_t* norm, u32 nsym, u32 log_) {
        log = log_;
        u32 const size = 1u << log;
        sym.assign(size, 0);
        nb.assign(size, 0);
        base.assign(size, 0);
        std::vector<u16> next(nsym);
        u32 high = size;
        for (u32 s = 0; s < nsym; s++) {
            if (norm[s] == -1) {
                sym[--high] = (u8)s;
                next[s] = 1;
            } else
                next[s] = (u16)norm[s];
        }
        u32 const step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
        u32 pos = 0;
        for (u32 s = 0; s < nsym; s++) {
            for (int i = 0; i < norm[s]; i++) {
                sym[pos] = (u8)s;
                do {
                    pos = (pos + step) & mask;
                } while (pos >= high);
            }
        }
        if (pos != 0) corrupt();
        for (u32 i = 0; i < size; i++) {
            u32 const x = next[sym[i]]++;
            if (x == 0) corrupt();
            nb[i] = (u8)(log - highbit(x));
            base[i] = (u16)((x << nb[i]) - size);
        }
    }
    void build_rle(u8 s) {
        log = 0;
        sym.assign(1, s);
        nb.assign(1, 0);
        base.assign(1, 0);
    }
};

// FSE table description ("NCount"); returns the number of bytes it occupied
inline size_t read_ncount(const u8* p, size_t n, u32 max_sym, u32 max_log, int16_t* norm, u32* nsym, u32* log) {
    FwdBits b(p, n);
    u32 const al = 5 + b.read(4);
    if (al > max_log) corrupt();
    int32_t remaining = 1 << al;
    u32 s = 0;
    while (remaining > 0 && s <= max_sym) {
        u32 const bits = highbit((u32)remaining + 1) + 1;
        u32 val = b.read(bits);
        u32 const lower = (1u << (bits - 1)) - 1;
        u32 const threshold = (1u << bits) - 1 - ((u32)remaining + 1);
        if ((val & lower) < threshold) {
            b.rewind(1);
            val &= lower;
        } else if (val > lower)
            val -= threshold;
        int32_t const proba = (int32_t)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        norm[s++] = (int16_t)proba;
        if (proba == 0) {
            u32 rep = b.read(2);
            for (;;) {
                for (u32 i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
                if (rep != 3) break;
                rep = b.read(2);
            }
        }
    }
    if (remaining != 0 || s > max_sym + 1) corrupt();
    *nsym = s;
    *log = al;
    return b.bytes_used();
}

// ---- Huffman decoding table (canonical, longest codes first)
struct HufTable {
    u32 maxBits = 0;
    std::vector<u8> sym, nb;
    bool valid = false;
    void build(const u8* weights, u32 n /* including the deduced last one */) {
        u32 sum = 0;
        for (u32 i = 0; i < n; i++)
            if (weights[i] > 12) corrupt(); else if (weights[i]) sum += 1u << (weights[i] - 1);
        if (sum == 0) corrupt();
        maxBits = highbit(sum);
        if ((1u << maxBits) != sum || maxBits > 11 || maxBits == 0) corrupt();
        u32 const size = 1u << maxBits;
        sym.assign(size, 0);
        nb.assign(size, 0);
        u32 rankCount[16] = {0}, rankIdx[16] = {0};
        u8 bits[256];
        for (u32 i = 0; i < n; i++) {
            bits[i] = weights[i] ? (u8)(maxBits + 1 - weights[i]) : 0;
            rankCount[bits[i]]++;
        }
        rankIdx[maxBits] = 0;
        for (u32 i = maxBits; i >= 1; i--) {
            rankIdx[i - 1] = rankIdx[i] + rankCount[i] * (1u << (maxBits - i));
            for (u32 k = rankIdx[i]; k < rankIdx[i - 1]; k++) nb[k] = (u8)i;
        }
        if (rankIdx[0] != size) corrupt();
        for (u32 i = 0; i < n; i++) {
            if (!bits[i]) continue;
            u32 const len = 1u << (maxBits - bits[i]);
            for (u32 k = 0; k < len; k++) sym[rankIdx[bits[i]] + k] = (u8)i;
            rankIdx[bits[i]] += len;
        }
        valid = true;
    }
    void decode_stream(const u8* p, size_t n, u8* out, size_t count) const {
        RevBits b(p, n);
        for (size_t i = 0; i < count; i++) {
            u32 const idx = (u32)b.peek(maxBits);
            out[i] = sym[idx];
            b.pos -= nb[idx];
        }
        if (b.pos != 0) corrupt(); // every bit of the stream belongs to a symbol
    }
};

// Huffman tree description; returns its size in bytes
inline size_t read_huf_tree(const u8* p, size_t n, HufTable& t) {
    if (n < 1) corrupt();
    u8 weights[256] = {0};
    u32 nw = 0;
    size_t used;
    u32 const hb = p[0];
    if (hb >= 128) { // 4-bit weights, two per byte
        nw = hb - 127;
        used = 1 + (nw + 1) / 2;
        if (used > n) corrupt();
        for (u32 i = 0; i < nw; i++) weights[i] = (i & 1) ? (p[1 + i / 2] & 15) : (p[1 + i / 2] >> 4);
    } else { // FSE-coded weights: two interleaved states until the stream runs dry
        used = 1 + hb;
        if (hb == 0 || used > n) corrupt();
        int16_t norm[16];
        u32 ns, al;
        size_t const hdr = read_ncount(p + 1, hb, 12, 6, norm, &ns, &al);
        if (hdr >= hb) corrupt();
        FseTable ft;
        ft.build(norm, ns, al);
        RevBits b(p + 1 + hdr, hb - hdr);
        u32 s1 = (u32)b.read(al), s2 = (u32)b.read(al);
        for (;;) {
            if (nw >= 254) corrupt();
            weights[nw++] = ft.sym[s1];
            s1 = ft.base[s1] + (u32)b.read(ft.nb[s1]);
            if (b.pos < 0) {
                weights[nw++] = ft.sym[s2];
                break;
            }
            if (nw >= 254) corrupt();
            weights[nw++] = ft.sym[s2];
            s2 = ft.base[s2] + (u32)b.read(ft.nb[s2]);
            if (b.pos < 0) {
                weights[nw++] = ft.sym[s1];
                break;
            }
        }
    }
    // the last weight completes the sum to a power of two
    u32 sum = 0;
    for (u32 i = 0; i < nw; i++)
        if (weights[i] > 12) corrupt(); else if (weights[i]) sum += 1u << (weights[i] - 1);
    if (sum == 0) corrupt();
    u32 const maxBits = highbit(sum) + 1;
    u32 const left = (1u << maxBits) - sum;
    if (left & (left - 1)) corrupt();
    weights[nw++] = (u8)(highbit(left) + 1);
    t.build(weights, nw);
    return used;
}

// ---- sequence code tables
static const u32 kLLBase[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const u8 kLLBits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const u32 kMLBase[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
                                31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195,
                                16387, 32771, 65539};
static const u8 kMLBits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                               0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const int16_t kLLDefault[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const int16_t kMLDefault[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const int16_t kOFDefault[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

// state that survives from block to block within a frame
struct FrameState {
    HufTable huf;
    FseTable ll, of, ml;
    bool haveLL = false, haveOF = false, haveML = false;
    u32 rep[3] = {1, 4, 8};
};

inline size_t read_seq_table(u32 mode, const u8* p, size_t n, FseTable& t, bool& have, const int16_t* def, u32 defN,
                             u32 defLog, u32 maxSym, u32 maxLog) {
    switch (mode) {
    case 0:
        t.build(def, defN, defLog);
        have = true;
        return 0;
    case 1:
        if (n < 1 || p[0] > maxSym) corrupt();
        t.build_rle(p[0]);
        have = true;
        return 1;
    case 2: {
        int16_t norm[64];
        u32 ns, al;
        size_t const used = read_ncount(p, n, maxSym, maxLog, norm, &ns, &al);
        t.build(norm, ns, al);
        have = true;
        return used;
    }
    default:
        if (!have) corrupt();
        return 0;
    }
}

inline void decode_block(FrameState& fs, const u8* p, size_t n, std::vector<u8>& out, size_t frameStart, std::vector<u8>& lit) {
    // ---- literals section
    if (n < 1) corrupt();
    u32 const ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
    size_t regen, csize = 0, hdr;
    u32 streams = 1;
    if (ltype < 2) { // raw / RLE
        if (sf == 0 || sf == 2) {
            regen = p[0] >> 3;
            hdr = 1;
        } else if (sf == 1) {
            if (n < 2) corrupt();
            regen = (p[0] >> 4) | ((size_t)p[1] << 4);
            hdr = 2;
        } else {
            if (n < 3) corrupt();
            regen = (p[0] >> 4) | ((size_t)p[1] << 4) | ((size_t)p[2] << 12);
            hdr = 3;
        }
    } else {
        if (sf == 0 || sf == 1) {
            if (n < 3) corrupt();
            u32 const v = p[0] | (p[1] << 8) | (p[2] << 16);
            regen = (v >> 4) & 0x3FF;
            csize = (v >> 14) & 0x3FF;
            hdr = 3;
            streams = sf == 0 ? 1 : 4;
        } else if (sf == 2) {
            if (n < 4) corrupt();
            u32 const v = p[0] | (p[1] << 8) | (p[2] << 16) | ((u32)p[3] << 24);
            regen = (v >> 4) & 0x3FFF;
            csize = (v >> 18) & 0x3FFF;
            hdr = 4;
            streams = 4;
        } else {
            if (n < 5) corrupt();
            u64 const v = p[0] | (p[1] << 8) | (p[2] << 16) | ((u64)p[3] << 24) | ((u64)p[4] << 32);
            regen = (size_t)((v >> 4) & 0x3FFFF);
            csize = (size_t)((v >> 22) & 0x3FFFF);
            hdr = 5;
            streams = 4;
        }
    }
    if (regen > (128u << 10)) corrupt();
    lit.resize(regen);
    size_t at = hdr;
    if (ltype == 0) {
        if (at + regen > n) corrupt();
        memcpy(lit.data(), p + at, regen);
        at += regen;
    } else if (ltype == 1) {
        if (at + 1 > n) corrupt();
        memset(lit.data(), p[at], regen);
        at += 1;
    } else {
        if (at + csize > n) corrupt();
        const u8* q = p + at;
        size_t left = csize;
        if (ltype == 2) {
            size_t const t = read_huf_tree(q, left, fs.huf);
            q += t;
            left -= t;
        } else if (!fs.huf.valid)
            corrupt();
        if (streams == 1)
            fs.huf.decode_stream(q, left, lit.data(), regen);
        else {
            if (left < 6) corrupt();
            size_t const s1 = q[0] | (q[1] << 8), s2 = q[2] | (q[3] << 8), s3 = q[4] | (q[5] << 8);
            if (6 + s1 + s2 + s3 > left) corrupt();
            size_t const s4 = left - 6 - s1 - s2 - s3;
            size_t const seg = (regen + 3) / 4;
            if (3 * seg > regen) corrupt();
            const u8* b = q + 6;
            fs.huf.decode_stream(b, s1, lit.data(), seg);
            fs.huf.decode_stream(b + s1, s2, lit.data() + seg, seg);
            fs.huf.decode_stream(b + s1 + s2, s3, lit.data() + 2 * seg, seg);
            fs.huf.decode_stream(b + s1 + s2 + s3, s4, lit.data() + 3 * seg, regen - 3 * seg);
        }
        at += csize;
    }
    // ---- sequences section
    if (at >= n) corrupt();
    u32 nseq = p[at++];
    if (nseq >= 128) {
        if (nseq == 255) {
            if (at + 2 > n) corrupt();
            nseq = p[at] + (p[at + 1] << 8) + 0x7F00;
            at += 2;
        } else {
            if (at + 1 > n) corrupt();
            nseq = ((nseq - 128) << 8) + p[at++];
        }
    }
    size_t litPos = 0;
    if (nseq == 0) {
        if (at != n) corrupt();
        out.insert(out.end(), lit.begin(), lit.end());
        return;
    }
    if (at >= n) corrupt();
    u32 const modes = p[at++];
    if (modes & 3) corrupt();
    at += read_seq_table(modes >> 6, p + at, n - at, fs.ll, fs.haveLL, kLLDefault, 36, 6, 35, 9);
    at += read_seq_table((modes >> 4) & 3, p + at, n - at, fs.of, fs.haveOF, kOFDefault, 29, 5, 31, 8);
    at += read_seq_table((modes >> 2) & 3, p + at, n - at, fs.ml, fs.haveML, kMLDefault, 53, 6, 52, 9);
    if (at >= n) corrupt();
    RevBits b(p + at, n - at);
    u32 sLL = (u32)b.read(fs.ll.log), sOF = (u32)b.read(fs.of.log), sML = (u32)b.read(fs.ml.log);
    if (b.pos < 0) corrupt();
    for (u32 i = 0; i < nseq; i++) {
        u32 const ofCode = fs.of.sym[sOF], mlCode = fs.ml.sym[sML], llCode = fs.ll.sym[sLL];
        if (ofCode > 31 || mlCode > 52 || llCode > 35) corrupt();
        u64 const ofVal = (1ull << ofCode) + b.read(ofCode);
        u32 const ml = kMLBase[mlCode] + (u32)b.read(kMLBits[mlCode]);
        u32 const ll = kLLBase[llCode] + (u32)b.read(kLLBits[llCode]);
        if (b.pos < 0) corrupt();
        u64 offset;
        if (ofVal > 3) {
            offset = ofVal - 3;
            fs.rep[2] = fs.rep[1];
            fs.rep[1] = fs.rep[0];
            fs.rep[0] = (u32)offset;
        } else {
            u32 idx = (u32)ofVal - 1 + (ll == 0 ? 1 : 0); // 0..3
            if (idx == 0)
                offset = fs.rep[0];
            else {
                offset = idx < 3 ? fs.rep[idx] : fs.rep[0] - 1;
                if (offset == 0) corrupt();
                if (idx > 1) fs.rep[2] = fs.rep[1];
                fs.rep[1] = fs.rep[0];
                fs.rep[0] = (u32)offset;
            }
        }
        if (i + 1 < nseq) { // states are refreshed in the order LL, ML, OF
            sLL = fs.ll.base[sLL] + (u32)b.read(fs.ll.nb[sLL]);
            sML = fs.ml.base[sML] + (u32)b.read(fs.ml.nb[sML]);
            sOF = fs.of.base[sOF] + (u32)b.read(fs.of.nb[sOF]);
            if (b.pos < 0) corrupt();
        }
        // execute
        if (litPos + ll > lit.size()) corrupt();
        out.insert(out.end(), lit.begin() + (long)litPos, lit.begin() + (long)(litPos + ll));
        litPos += ll;
        size_t const have = out.size() - frameStart;
        if (offset > have) corrupt();
        size_t const from = out.size() - (size_t)offset;
        out.resize(out.size() + ml);
        u8* const d = out.data();
        if (offset >= ml)
            memcpy(d + from + (size_t)offset, d + from, ml);
        else
            for (u32 k = 0; k < ml; k++) d[from + (size_t)offset + k] = d[from + k];
    }
    if (b.pos != 0) corrupt();
    out.insert(out.end(), lit.begin() + (long)litPos, lit.end());
}

} // namespace detail

// Decodes every frame in [src, src + n) and returns the concatenation.  `capacity`: like the destination
// capacity handed to zstd::bulk::decompress -- more output than that is an error.
inline std::vector<uint8_t> decompress(const uint8_t* src, size_t n, size_t capacity) {
    using namespace detail;
    std::vector<u8> out, lit;
    size_t at = 0;
    if (n == 0) throw DecodeError{"Src size is incorrect"};
    while (at < n) {
        if (n - at < 4) throw DecodeError{"Src size is incorrect"};
        u32 const magic = src[at] | (src[at + 1] << 8) | (src[at + 2] << 16) | ((u32)src[at + 3] << 24);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) { // skippable frame
            if (n - at < 8) throw DecodeError{"Src size is incorrect"};
            u32 const sz = src[at + 4] | (src[at + 5] << 8) | (src[at + 6] << 16) | ((u32)src[at + 7] << 24);
            if ((size_t)sz + 8 > n - at) throw DecodeError{"Src size is incorrect"};
            at += 8 + sz;
            continue;
        }
        if (magic != 0xFD2FB528u) throw DecodeError{"Unknown frame descriptor"};
        at += 4;
        if (at >= n) throw DecodeError{"Src size is incorrect"};
        u32 const fhd = src[at++];
        u32 const fcsFlag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dictFlag = fhd & 3;
        if (fhd & 0x08) throw DecodeError{"Unsupported frame parameter"};
        u64 windowSize = 0;
        if (!single) {
            if (at >= n) throw DecodeError{"Src size is incorrect"};
            u32 const wd = src[at++];
            u64 const wbase = 1ull << (10 + (wd >> 3));
            windowSize = wbase + (wbase >> 3) * (wd & 7);
        }
        static const u32 dictBytes[4] = {0, 1, 2, 4};
        u32 dictId = 0;
        if (at + dictBytes[dictFlag] > n) throw DecodeError{"Src size is incorrect"};
        for (u32 i = 0; i < dictBytes[dictFlag]; i++) dictId |= (u32)src[at + i] << (8 * i);
        at += dictBytes[dictFlag];
        if (dictId) throw DecodeError{"Dictionary mismatch"};
        u32 const fcsBytes = fcsFlag == 0 ? (single ? 1u : 0u) : (fcsFlag == 1 ? 2u : (fcsFlag == 2 ? 4u : 8u));
        if (at + fcsBytes > n) throw DecodeError{"Src size is incorrect"};
        u64 fcs = 0;
        for (u32 i = 0; i < fcsBytes; i++) fcs |= (u64)src[at + i] << (8 * i);
        if (fcsBytes == 2) fcs += 256;
        at += fcsBytes;
        (void)windowSize;
        size_t const frameStart = out.size();
        FrameState fs;
        for (;;) {
            if (at + 3 > n) throw DecodeError{"Src size is incorrect"};
            u32 const bh = src[at] | (src[at + 1] << 8) | (src[at + 2] << 16);
            at += 3;
            u32 const last = bh & 1, type = (bh >> 1) & 3, bsize = bh >> 3;
            if (type == 3) corrupt();
            if (type == 0) {
                if (at + bsize > n) throw DecodeError{"Src size is incorrect"};
                out.insert(out.end(), src + at, src + at + bsize);
                at += bsize;
            } else if (type == 1) {
                if (at + 1 > n) throw DecodeError{"Src size is incorrect"};
                out.insert(out.end(), bsize, src[at]);
                at += 1;
            } else {
                if (bsize > (128u << 10) || at + bsize > n) corrupt();
                decode_block(fs, src + at, bsize, out, frameStart, lit);
                at += bsize;
            }
            if (out.size() > capacity) throw DecodeError{"Destination buffer is too small"};
            if (last) break;
        }
        if (fcsBytes && out.size() - frameStart != fcs) corrupt();
        if (checksum) {
            if (at + 4 > n) throw DecodeError{"Src size is incorrect"};
            at += 4; // XXH64 low 32 bits of the content: not verified here
        }
    }
    return out;
}

} // namespace zfd
