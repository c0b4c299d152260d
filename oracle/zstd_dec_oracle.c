/*
 * zstd_dec_oracle.c -- CPU ORACLE (test infrastructure, see sq_oracle.h).
 * Minimal single-frame zstd decoder written from RFC 8878, used to round-trip frames emitted by
 * the oracle and by the HIP path (the counterpart of zstd::bulk::decompress at
 * /root/reference/src/archive/reader.rs:302-303 and of the zstd::stream::Decoder check in
 * /root/reference/src/util/tests.rs:146-166).  Independent of the encoder files on purpose.
 */
#include "sq_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;

#define BAD ((size_t)-1)
static u32 hb32(u32 v) { return 31 - (u32)__builtin_clz(v); }

/* ---------------- forward bit reader (FSE table descriptions) */
typedef struct {
    const u8* p;
    size_t n;
    size_t bitpos;
} fbr;
static u32 fbr_peek(const fbr* b, u32 nb) {
    u64 v = 0;
    size_t byte = b->bitpos >> 3;
    for (int i = 0; i < 8; i++)
        if (byte + i < b->n) v |= (u64)b->p[byte + i] << (8 * i);
    v >>= (b->bitpos & 7);
    return (u32)(v & ((1ULL << nb) - 1));
}

/* ---------------- backward bit reader */
typedef struct {
    const u8* p;
    long bitpos; /* number of unread bits below the cursor; may go negative (zero fill) */
    int bad;
} bbr;
static int bbr_init(bbr* b, const u8* p, size_t n) {
    if (n == 0 || p[n - 1] == 0) return -1;
    b->p = p;
    b->bitpos = (long)(n - 1) * 8 + (long)hb32(p[n - 1]);
    b->bad = 0;
    return 0;
}
static u64 bbr_bits_at(const bbr* b, long pos, u32 nb) { /* bits [pos, pos+nb) */
    u64 v = 0;
    for (u32 i = 0; i < nb; i++) {
        long q = pos + (long)i;
        if (q >= 0) v |= (u64)((b->p[q >> 3] >> (q & 7)) & 1) << i;
    }
    return v;
}
static u64 bbr_read(bbr* b, u32 nb) {
    if (nb == 0) return 0;
    b->bitpos -= (long)nb;
    return bbr_bits_at(b, b->bitpos, nb);
}
static u64 bbr_peek(const bbr* b, u32 nb) { return bbr_bits_at(b, b->bitpos - (long)nb, nb); }

/* ---------------- FSE decoding tables */
typedef struct {
    u8 symbol;
    u8 nbBits;
    u16 newState;
} fse_de;
typedef struct {
    fse_de t[512];
    u32 log;
} fse_dt;

static int fse_build_dt(fse_dt* dt, const s16* norm, u32 maxSym, u32 log) {
    u32 const size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    u16 next[64];
    u32 high = size - 1, pos = 0;
    if (log > 9) return -1;
    dt->log = log;
    for (u32 s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) {
            dt->t[high--].symbol = (u8)s;
            next[s] = 1;
        } else
            next[s] = (u16)norm[s];
    }
    for (u32 s = 0; s <= maxSym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            dt->t[pos].symbol = (u8)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    if (pos != 0) return -1;
    for (u32 u = 0; u < size; u++) {
        u8 s = dt->t[u].symbol;
        u32 ns = next[s]++;
        dt->t[u].nbBits = (u8)(log - hb32(ns));
        dt->t[u].newState = (u16)((ns << dt->t[u].nbBits) - size);
    }
    return 0;
}
static void fse_build_rle(fse_dt* dt, u8 sym) {
    dt->log = 0;
    dt->t[0].symbol = sym;
    dt->t[0].nbBits = 0;
    dt->t[0].newState = 0;
}
/* returns bytes consumed or BAD */
static size_t fse_read_ncount(s16* norm, u32* maxSymPtr, u32* logPtr, const u8* src, size_t n,
                              u32 maxLog) {
    fbr b = {src, n, 0};
    u32 log, sym = 0, maxSym = *maxSymPtr;
    int remaining, threshold, nbBits, prev0 = 0;
    if (n < 1) return BAD;
    log = fbr_peek(&b, 4) + 5;
    b.bitpos += 4;
    if (log > maxLog) return BAD;
    remaining = (1 << log) + 1;
    threshold = 1 << log;
    nbBits = (int)log + 1;
    memset(norm, 0, (maxSym + 1) * sizeof(s16));
    while (remaining > 1 && sym <= maxSym) {
        if (prev0) {
            for (;;) {
                u32 r = fbr_peek(&b, 2);
                b.bitpos += 2;
                sym += r;
                if (r != 3) break;
            }
            if (sym > maxSym) return BAD;
        }
        {
            int const max = (2 * threshold - 1) - remaining;
            int count;
            u32 v = fbr_peek(&b, (u32)nbBits);
            if ((int)(v & (u32)(threshold - 1)) < max) {
                count = (int)(v & (u32)(threshold - 1));
                b.bitpos += (size_t)(nbBits - 1);
            } else {
                count = (int)(v & (u32)(2 * threshold - 1));
                if (count >= threshold) count -= max;
                b.bitpos += (size_t)nbBits;
            }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[sym++] = (s16)count;
            prev0 = !count;
            while (remaining < threshold) {
                nbBits--;
                threshold >>= 1;
            }
        }
    }
    if (remaining != 1) return BAD;
    *maxSymPtr = sym - 1;
    *logPtr = log;
    {
        size_t used = (b.bitpos + 7) >> 3;
        if (used > n) return BAD;
        return used;
    }
}

/* ---------------- Huffman */
typedef struct {
    u8 sym[1 << 11];
    u8 nb[1 << 11];
    u32 log;
} huf_dt;

static int huf_build(huf_dt* dt, const u8* weights, u32 nsym /* including last */) {
    u32 rankCount[14] = {0}, rankStart[14];
    u32 total = 0, log;
    for (u32 i = 0; i + 1 < nsym; i++) {
        if (weights[i] > 11) return -1;
        if (weights[i]) total += 1u << (weights[i] - 1);
    }
    if (total == 0) return -1;
    log = hb32(total) + 1;
    if (log > 11) return -1;
    {
        u32 rest = (1u << log) - total;
        u32 lastW;
        u8* w = (u8*)weights;
        if (rest == 0 || (rest & (rest - 1))) return -1;
        lastW = hb32(rest) + 1;
        w[nsym - 1] = (u8)lastW;
    }
    for (u32 i = 0; i < nsym; i++) rankCount[weights[i]]++;
    {
        u32 next = 0;
        for (u32 w = 1; w <= log; w++) {
            rankStart[w] = next;
            next += rankCount[w] << (w - 1);
        }
    }
    dt->log = log;
    for (u32 s = 0; s < nsym; s++) {
        u32 w = weights[s];
        if (!w) continue;
        u32 len = (1u << w) >> 1;
        for (u32 k = 0; k < len; k++) {
            dt->sym[rankStart[w] + k] = (u8)s;
            dt->nb[rankStart[w] + k] = (u8)(log + 1 - w);
        }
        rankStart[w] += len;
    }
    return 0;
}
/* reads tree description; returns bytes consumed */
static size_t huf_read_tree(huf_dt* dt, const u8* src, size_t n) {
    u8 weights[257];
    u32 nw;
    size_t used;
    if (n < 1) return BAD;
    memset(weights, 0, sizeof(weights));
    if (src[0] >= 128) {
        nw = src[0] - 127;
        used = 1 + (nw + 1) / 2;
        if (used > n) return BAD;
        for (u32 i = 0; i < nw; i++)
            weights[i] = (i & 1) ? (src[1 + i / 2] & 15) : (src[1 + i / 2] >> 4);
    } else {
        size_t const csize = src[0];
        s16 norm[16];
        u32 maxSym = 12, log;
        fse_dt fdt;
        bbr b;
        size_t h;
        u32 s1, s2;
        used = 1 + csize;
        if (used > n || csize < 2) return BAD;
        h = fse_read_ncount(norm, &maxSym, &log, src + 1, csize, 6);
        if (h == BAD) return BAD;
        if (fse_build_dt(&fdt, norm, maxSym, log)) return BAD;
        if (bbr_init(&b, src + 1 + h, csize - h)) return BAD;
        s1 = (u32)bbr_read(&b, log);
        s2 = (u32)bbr_read(&b, log);
        nw = 0;
        for (;;) {
            if (nw >= 255) return BAD;
            weights[nw++] = fdt.t[s1].symbol;
            if (b.bitpos < (long)fdt.t[s1].nbBits) { /* not enough bits to update: flush other */
                weights[nw++] = fdt.t[s2].symbol;
                break;
            }
            s1 = fdt.t[s1].newState + (u32)bbr_read(&b, fdt.t[s1].nbBits);
            if (nw >= 255) return BAD;
            weights[nw++] = fdt.t[s2].symbol;
            if (b.bitpos < (long)fdt.t[s2].nbBits) {
                weights[nw++] = fdt.t[s1].symbol;
                break;
            }
            s2 = fdt.t[s2].newState + (u32)bbr_read(&b, fdt.t[s2].nbBits);
        }
    }
    if (huf_build(dt, weights, nw + 1)) return BAD;
    return used;
}
static int huf_decode_stream(const huf_dt* dt, const u8* src, size_t n, u8* dst, size_t count) {
    bbr b;
    if (bbr_init(&b, src, n)) return -1;
    for (size_t i = 0; i < count; i++) {
        u32 idx = (u32)bbr_peek(&b, dt->log);
        dst[i] = dt->sym[idx];
        b.bitpos -= dt->nb[idx];
    }
    return b.bitpos == 0 ? 0 : -1;
}

/* ---------------- sequences */
static const u32 LL_base[36] = {0,  1,  2,   3,   4,   5,    6,    7,    8,    9,     10,    11,
                                12, 13, 14,  15,  16,  18,   20,   22,   24,   28,    32,    40,
                                48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
static const u8 LL_bitsD[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  1,  1,
                                1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const u32 ML_base[53] = {3,  4,  5,  6,  7,  8,  9,  10,  11,  12,  13,   14,   15,   16,
                                17, 18, 19, 20, 21, 22, 23, 24,  25,  26,  27,   28,   29,   30,
                                31, 32, 33, 34, 35, 37, 39, 41,  43,  47,  51,   59,   67,   83,
                                99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
static const u8 ML_bitsD[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,
                                0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,  1,  1,  1,
                                2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const s16 LL_defNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                   2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const s16 ML_defNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const s16 OF_defNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

typedef struct {
    huf_dt huf;
    int huf_valid;
    fse_dt ll, of, ml;
    int ll_valid, of_valid, ml_valid;
    u32 rep[3];
} dctx;

static size_t read_table(fse_dt* dt, int* valid, u32 mode, const u8* p, size_t n, u32 maxSym,
                         u32 maxLog, const s16* defNorm, u32 defLog, u32 defMax) {
    switch (mode) {
    case 0:
        if (fse_build_dt(dt, defNorm, defMax, defLog)) return BAD;
        *valid = 1;
        return 0;
    case 1:
        if (n < 1 || p[0] > maxSym) return BAD;
        fse_build_rle(dt, p[0]);
        *valid = 1;
        return 1;
    case 2: {
        s16 norm[64];
        u32 ms = maxSym, log;
        size_t h = fse_read_ncount(norm, &ms, &log, p, n, maxLog);
        if (h == BAD) return BAD;
        if (fse_build_dt(dt, norm, ms, log)) return BAD;
        *valid = 1;
        return h;
    }
    default: return *valid ? 0 : BAD;
    }
}

static size_t decode_block(dctx* d, const u8* src, size_t n, u8* dst0, u8* dst, u8* dend) {
    u8* lits = NULL;
    size_t litSize = 0, pos = 0;
    u8* const ostart = dst;
    u8* litbuf = (u8*)malloc(128 * 1024 + 32);
    size_t ret = BAD;
    if (!litbuf || n < 1) goto done;
    { /* literals section */
        u32 const type = src[0] & 3, sf = (src[0] >> 2) & 3;
        if (type < 2) {
            size_t hs;
            if (sf == 0 || sf == 2) {
                litSize = src[0] >> 3;
                hs = 1;
            } else if (sf == 1) {
                if (n < 2) goto done;
                litSize = (src[0] >> 4) + ((size_t)src[1] << 4);
                hs = 2;
            } else {
                if (n < 3) goto done;
                litSize = (src[0] >> 4) + ((size_t)src[1] << 4) + ((size_t)src[2] << 12);
                hs = 3;
            }
            if (litSize > 128 * 1024) goto done;
            if (type == 0) {
                if (hs + litSize > n) goto done;
                memcpy(litbuf, src + hs, litSize);
                pos = hs + litSize;
            } else {
                if (hs + 1 > n) goto done;
                memset(litbuf, src[hs], litSize);
                pos = hs + 1;
            }
        } else {
            size_t hs, csize;
            u32 streams;
            if (sf == 0 || sf == 1) {
                u32 v;
                if (n < 3) goto done;
                v = src[0] | (src[1] << 8) | (src[2] << 16);
                litSize = (v >> 4) & 0x3FF;
                csize = (v >> 14) & 0x3FF;
                hs = 3;
                streams = sf == 0 ? 1 : 4;
            } else if (sf == 2) {
                u32 v;
                if (n < 4) goto done;
                v = src[0] | (src[1] << 8) | (src[2] << 16) | ((u32)src[3] << 24);
                litSize = (v >> 4) & 0x3FFF;
                csize = (v >> 18) & 0x3FFF;
                hs = 4;
                streams = 4;
            } else {
                u64 v;
                if (n < 5) goto done;
                v = (u64)src[0] | ((u64)src[1] << 8) | ((u64)src[2] << 16) | ((u64)src[3] << 24) |
                    ((u64)src[4] << 32);
                litSize = (v >> 4) & 0x3FFFF;
                csize = (v >> 22) & 0x3FFFF;
                hs = 5;
                streams = 4;
            }
            if (hs + csize > n || litSize > 128 * 1024) goto done;
            {
                const u8* p = src + hs;
                size_t rem = csize;
                if (type == 2) {
                    size_t t = huf_read_tree(&d->huf, p, rem);
                    if (t == BAD) goto done;
                    p += t;
                    rem -= t;
                    d->huf_valid = 1;
                } else if (!d->huf_valid)
                    goto done;
                if (streams == 1) {
                    if (huf_decode_stream(&d->huf, p, rem, litbuf, litSize)) goto done;
                } else {
                    size_t s1, s2, s3, seg = (litSize + 3) / 4;
                    if (rem < 6) goto done;
                    s1 = p[0] | (p[1] << 8);
                    s2 = p[2] | (p[3] << 8);
                    s3 = p[4] | (p[5] << 8);
                    if (6 + s1 + s2 + s3 > rem || seg * 3 > litSize) goto done;
                    if (huf_decode_stream(&d->huf, p + 6, s1, litbuf, seg)) goto done;
                    if (huf_decode_stream(&d->huf, p + 6 + s1, s2, litbuf + seg, seg)) goto done;
                    if (huf_decode_stream(&d->huf, p + 6 + s1 + s2, s3, litbuf + 2 * seg, seg)) goto done;
                    if (huf_decode_stream(&d->huf, p + 6 + s1 + s2 + s3, rem - 6 - s1 - s2 - s3,
                                          litbuf + 3 * seg, litSize - 3 * seg))
                        goto done;
                }
            }
            pos = hs + csize;
        }
        lits = litbuf;
    }
    { /* sequences section */
        size_t nbSeq, litPos = 0;
        if (pos >= n) goto done;
        if (src[pos] < 128) {
            nbSeq = src[pos];
            pos += 1;
        } else if (src[pos] < 255) {
            if (pos + 2 > n) goto done;
            nbSeq = ((size_t)(src[pos] - 128) << 8) + src[pos + 1];
            pos += 2;
        } else {
            if (pos + 3 > n) goto done;
            nbSeq = (size_t)src[pos + 1] + ((size_t)src[pos + 2] << 8) + 0x7F00;
            pos += 3;
        }
        if (nbSeq) {
            u8 modes;
            size_t t;
            bbr b;
            u32 sLL, sOF, sML;
            if (pos >= n) goto done;
            modes = src[pos++];
            if (modes & 3) goto done;
            t = read_table(&d->ll, &d->ll_valid, modes >> 6, src + pos, n - pos, 35, 9, LL_defNorm, 6, 35);
            if (t == BAD) goto done;
            pos += t;
            t = read_table(&d->of, &d->of_valid, (modes >> 4) & 3, src + pos, n - pos, 31, 8, OF_defNorm, 5, 28);
            if (t == BAD) goto done;
            pos += t;
            t = read_table(&d->ml, &d->ml_valid, (modes >> 2) & 3, src + pos, n - pos, 52, 9, ML_defNorm, 6, 52);
            if (t == BAD) goto done;
            pos += t;
            if (bbr_init(&b, src + pos, n - pos)) goto done;
            sLL = (u32)bbr_read(&b, d->ll.log);
            sOF = (u32)bbr_read(&b, d->of.log);
            sML = (u32)bbr_read(&b, d->ml.log);
            for (size_t i = 0; i < nbSeq; i++) {
                u32 const ofc = d->of.t[sOF].symbol, mlc = d->ml.t[sML].symbol, llc = d->ll.t[sLL].symbol;
                u32 ofv, ml, ll, offset;
                if (ofc > 31 || mlc > 52 || llc > 35) goto done;
                ofv = (1u << ofc) + (u32)bbr_read(&b, ofc);
                ml = ML_base[mlc] + (u32)bbr_read(&b, ML_bitsD[mlc]);
                ll = LL_base[llc] + (u32)bbr_read(&b, LL_bitsD[llc]);
                if (ofv > 3) {
                    offset = ofv - 3;
                    d->rep[2] = d->rep[1];
                    d->rep[1] = d->rep[0];
                    d->rep[0] = offset;
                } else {
                    u32 idx = ofv - 1 + (ll == 0);
                    if (idx == 0)
                        offset = d->rep[0];
                    else {
                        offset = (idx == 3) ? d->rep[0] - 1 : d->rep[idx];
                        if (idx > 1) d->rep[2] = d->rep[1];
                        d->rep[1] = d->rep[0];
                        d->rep[0] = offset;
                    }
                }
                if (i + 1 < nbSeq) {
                    sLL = d->ll.t[sLL].newState + (u32)bbr_read(&b, d->ll.t[sLL].nbBits);
                    sML = d->ml.t[sML].newState + (u32)bbr_read(&b, d->ml.t[sML].nbBits);
                    sOF = d->of.t[sOF].newState + (u32)bbr_read(&b, d->of.t[sOF].nbBits);
                }
                if (litPos + ll > litSize) goto done;
                if ((size_t)(dend - dst) < (size_t)ll + ml) goto done;
                memcpy(dst, lits + litPos, ll);
                dst += ll;
                litPos += ll;
                if (offset == 0 || offset > (size_t)(dst - dst0)) goto done;
                for (u32 k = 0; k < ml; k++) dst[k] = dst[(long)k - (long)offset];
                dst += ml;
            }
            if (b.bitpos != 0) goto done;
        } else if (pos != n)
            goto done;
        if ((size_t)(dend - dst) < litSize - litPos) goto done;
        memcpy(dst, lits + litPos, litSize - litPos);
        dst += litSize - litPos;
    }
    ret = (size_t)(dst - ostart);
done:
    free(litbuf);
    return ret;
}

size_t sqo_decompress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    size_t pos = 5;
    u8 fhd;
    u32 fcsCode, single, dictCode;
    u64 fcs = 0;
    int haveFcs = 0;
    u8* op = dst;
    dctx* d;
    size_t result = BAD;
    if (n < 6 || src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) return BAD;
    fhd = src[4];
    fcsCode = fhd >> 6;
    single = (fhd >> 5) & 1;
    dictCode = fhd & 3;
    if (fhd & 0x08) return BAD;
    if (!single) pos += 1;
    pos += (size_t[]){0, 1, 2, 4}[dictCode];
    {
        size_t const fl = fcsCode == 0 ? (single ? 1 : 0) : (fcsCode == 1 ? 2 : (fcsCode == 2 ? 4 : 8));
        if (pos + fl > n) return BAD;
        if (fl) {
            memcpy(&fcs, src + pos, fl);
            if (fl == 2) fcs += 256;
            haveFcs = 1;
        }
        pos += fl;
    }
    d = (dctx*)calloc(1, sizeof(dctx));
    if (!d) return BAD;
    d->rep[0] = 1;
    d->rep[1] = 4;
    d->rep[2] = 8;
    for (;;) {
        u32 h, last, type, size;
        if (pos + 3 > n) goto out;
        h = src[pos] | (src[pos + 1] << 8) | ((u32)src[pos + 2] << 16);
        pos += 3;
        last = h & 1;
        type = (h >> 1) & 3;
        size = h >> 3;
        if (type == 0) {
            if (pos + size > n || (size_t)(dst + cap - op) < size) goto out;
            memcpy(op, src + pos, size);
            op += size;
            pos += size;
        } else if (type == 1) {
            if (pos + 1 > n || (size_t)(dst + cap - op) < size) goto out;
            memset(op, src[pos], size);
            op += size;
            pos += 1;
        } else if (type == 2) {
            size_t r;
            if (pos + size > n) goto out;
            r = decode_block(d, src + pos, size, dst, op, dst + cap);
            if (r == BAD) goto out;
            op += r;
            pos += size;
        } else
            goto out;
        if (last) break;
    }
    if (fhd & 4) pos += 4;
    if (pos != n) goto out;
    if (haveFcs && fcs != (u64)(op - dst)) goto out;
    result = (size_t)(op - dst);
out:
    free(d);
    return result;
}
