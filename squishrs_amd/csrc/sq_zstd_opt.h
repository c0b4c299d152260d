// sq_zstd_opt.h -- btopt parser (zstd strategy 7, optLevel 0) used by level 12 for chunks of
// <= 16 KiB (minMatch 3, searchLog 4, targetLength 24); bit-exact with libzstd 1.5.7
// ZSTD_compressBlock_btopt, reached from /root/reference/src/util/chunk.rs:89-90.
//
// Such chunks are rare on the pack path (file tails) and tiny, and the optimal parser is a
// dynamic program with a serial price-update chain, so this whole block compressor runs as ONE
// serial region on lane 0 (plain scalar code, tables in the wave's HBM workspace); the entropy
// stage that follows is the shared wave-level one.  Functions here must only be called from
// inside WV_SERIAL.
#pragma once
#include "sq_zstd_match.h"

namespace sqz {
namespace opt {

#define BLOCKSIZE_MAX (128u << 10)
#define MINMATCH 3
#define REP_NUM 3
#define WINDOW_START_INDEX 2u
#define REPCODE_TO_OFFBASE(r) (r)

struct mstate { // serial view of the match state
    const u8* base;
    u32 nextToUpdate;
    u32* hashTable;
    u32* chainTable;
    u32* hashTable3;
    u32 hashLog3;
    CParams cp;
    void* opt;
};
struct seqstore {
    SeqDef* seqs;
    u8* lits;
    u32 nseq, nlits, longLengthType, longLengthPos;
};

SQ_DEV u32 highbit32(u32 v) { return sq_highbit32(v); }
SQ_DEV u32 rd32(const u8* p) { return sq_rd32(p); }
// (codes below 16 / 32 are the value itself and carry no extra bits -- nearly every length the parser prices: two table loads
// per price less, on a lane whose every load is a trip of its own)
SQ_DEV u32 zs_LLcode(u32 v) { return v < 16 ? v : LLcode(v); }
SQ_DEV u32 zs_MLcode(u32 v) { return v < 32 ? v : MLcode(v); }
SQ_DEV u32 zs_LL_bits_of(u32 code) { return code < 16 ? 0u : LL_bits[code]; }
SQ_DEV u32 zs_ML_bits_of(u32 code) { return code < 32 ? 0u : ML_bits[code]; }
SQ_DEV size_t count_match(const u8* pIn, const u8* pMatch, const u8* const pInLimit) {
    const u8* const pStart = pIn;
    while (pIn + 8 <= pInLimit) {
        u64 x = sq_rd64(pIn) ^ sq_rd64(pMatch);
        if (x) return (size_t)(pIn - pStart) + (size_t)(__builtin_ctzll(x) >> 3);
        pIn += 8;
        pMatch += 8;
    }
    while (pIn < pInLimit && *pIn == *pMatch) {
        pIn++;
        pMatch++;
    }
    return (size_t)(pIn - pStart);
}
SQ_DEV size_t hashPtr(const u8* p, u32 hBits, u32 mls) {
    (void)mls; /* ZSTD_hashPtr has no 3-byte case: minMatch 3 hashes 4 bytes */
    return (size_t)hash4(sq_rd32(p), hBits);
}
SQ_DEV size_t hash3Ptr(const u8* p, u32 hBits) { return (size_t)hash3(sq_rd32(p), hBits); }
SQ_DEV void storeSeq(seqstore* ss, size_t litLength, const u8* literals, u32 offBase, size_t matchLength) {
    for (size_t i = 0; i < litLength; i++) ss->lits[ss->nlits + i] = literals[i];
    ss->nlits += (u32)litLength;
    if (litLength > 0xFFFF) {
        ss->longLengthType = 1;
        ss->longLengthPos = ss->nseq;
    }
    ss->seqs[ss->nseq].litLength = (u16)litLength;
    ss->seqs[ss->nseq].offBase = offBase;
    {
        size_t const mlBase = matchLength - MINMATCH;
        if (mlBase > 0xFFFF) {
            ss->longLengthType = 2;
            ss->longLengthPos = ss->nseq;
        }
        ss->seqs[ss->nseq].mlBase = (u16)mlBase;
    }
    ss->nseq++;
}

#define OPT_NUM (1 << 12)
#define OPT_SIZE (OPT_NUM + 3)
#define OPT_MAX_PRICE (1 << 30)
#define OPT_PREDEF_THRESHOLD 8
#define BITCOST_ACCURACY 8
#define BITCOST_MULTIPLIER (1 << BITCOST_ACCURACY)
#define OPT_LITFREQ_ADD 2
#define O_MaxLit 255
#define O_MaxLL 35
#define O_MaxML 52
#define O_MaxOff 31

typedef struct {
    int price;
    u32 off, mlen, litlen;
    u32 rep[REP_NUM];
} opt_t;
typedef struct {
    u32 off, len;
} match_t;

typedef struct {
    // the statistics every price comes from (1.5 KB): while a block is parsed they live in LDS (OptLds below)
    u32 litFreq[O_MaxLit + 1], litLengthFreq[O_MaxLL + 1], matchLengthFreq[O_MaxML + 1],
        offCodeFreq[O_MaxOff + 1];
    u32 litSum, litLengthSum, matchLengthSum, offCodeSum;
    u32 litSumBasePrice, litLengthSumBasePrice, matchLengthSumBasePrice, offCodeSumBasePrice;
    int predef; /* priceType == zop_predef */
    // (in the wave's workspace only)
    match_t matchTable[OPT_SIZE];
    opt_t priceTable[OPT_SIZE];
} optstate;
// The parse is one serial region on lane 0, and 74 of its loads per input byte went to these tables in the wave's workspace
// (SQ counters of 12 KiB text chunks, round 4: 65 % of the wave's cycles waiting for them).  The statistics, the matches of
// the position at hand (at most 1 << searchLog + 4 = 20 at this size class) and the first OPT_LDS_N entries of the price table
// -- the series of positions the parser is working on rarely grows beyond them -- are kept in LDS for the duration of
// compressBlock_btopt (the 8 KB the entropy stage and the other finders use otherwise); entries beyond stay in the workspace.
#define OPT_HOT_WORDS ((O_MaxLit + 1) + (O_MaxLL + 1) + (O_MaxML + 1) + (O_MaxOff + 1) + 4 + 4 + 1)
#define OPT_LDS_N 224u
#define OPT_LDS_MATCHES 32u
typedef struct {
    u32 hot[OPT_HOT_WORDS];
    match_t matches[OPT_LDS_MATCHES];
    opt_t win[OPT_LDS_N];
} OptLds;
static_assert(sizeof(OptLds) <= 8192, "fits the union of LdsState");
static_assert(sizeof(opt_t) == 28 && offsetof(optstate, matchTable) == 4 * OPT_HOT_WORDS, "layout of the statistics");


/* optLevel 0 => WEIGHT == ZSTD_bitWeight */
SQ_DEV u32 opt_weight(u32 stat) { return highbit32(stat + 1) * BITCOST_MULTIPLIER; }

SQ_DEVFN void opt_setBasePrices(optstate* o) {
    o->litSumBasePrice = opt_weight(o->litSum);
    o->litLengthSumBasePrice = opt_weight(o->litLengthSum);
    o->matchLengthSumBasePrice = opt_weight(o->matchLengthSum);
    o->offCodeSumBasePrice = opt_weight(o->offCodeSum);
}
SQ_DEVFN u32 opt_downscale(u32* table, u32 lastEltIndex, u32 shift, int base1) {
    u32 s, sum = 0;
    for (s = 0; s < lastEltIndex + 1; s++) {
        u32 const base = base1 ? 1 : (table[s] > 0);
        u32 const newStat = base + (table[s] >> shift);
        sum += newStat;
        table[s] = newStat;
    }
    return sum;
}
SQ_DEVFN u32 opt_scaleStats(u32* table, u32 lastEltIndex, u32 logTarget) {
    u32 prevsum = 0, factor;
    for (u32 s = 0; s <= lastEltIndex; s++) prevsum += table[s];
    factor = prevsum >> logTarget;
    if (factor <= 1) return prevsum;
    return opt_downscale(table, lastEltIndex, highbit32(factor), 1);
}
SQ_DEVFN void opt_rescaleFreqs(optstate* o, const u8* src, size_t srcSize) { /* ZSTD_rescaleFreqs */
    o->predef = 0;
    if (o->litLengthSum == 0) {
        if (srcSize <= OPT_PREDEF_THRESHOLD) o->predef = 1;
        {
            for (u32 i_ = 0; i_ <= O_MaxLit; i_++) o->litFreq[i_] = 0;
            for (size_t i = 0; i < srcSize; i++) o->litFreq[src[i]]++;
            o->litSum = opt_downscale(o->litFreq, O_MaxLit, 8, 0);
        }
        {
            const u32 baseLLfreqs[O_MaxLL + 1] = {4, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                         1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
            for (u32 i_ = 0; i_ <= O_MaxLL; i_++) o->litLengthFreq[i_] = baseLLfreqs[i_];
            o->litLengthSum = 0;
            for (u32 s = 0; s <= O_MaxLL; s++) o->litLengthSum += baseLLfreqs[s];
        }
        for (u32 ml = 0; ml <= O_MaxML; ml++) o->matchLengthFreq[ml] = 1;
        o->matchLengthSum = O_MaxML + 1;
        {
            const u32 baseOFCfreqs[O_MaxOff + 1] = {6, 2, 1, 1, 2, 3, 4, 4, 4, 3, 2,
                                                           1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                                           1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
            for (u32 i_ = 0; i_ <= O_MaxOff; i_++) o->offCodeFreq[i_] = baseOFCfreqs[i_];
            o->offCodeSum = 0;
            for (u32 s = 0; s <= O_MaxOff; s++) o->offCodeSum += baseOFCfreqs[s];
        }
    } else {
        o->litSum = opt_scaleStats(o->litFreq, O_MaxLit, 12);
        o->litLengthSum = opt_scaleStats(o->litLengthFreq, O_MaxLL, 11);
        o->matchLengthSum = opt_scaleStats(o->matchLengthFreq, O_MaxML, 11);
        o->offCodeSum = opt_scaleStats(o->offCodeFreq, O_MaxOff, 11);
    }
    opt_setBasePrices(o);
}

SQ_DEV u32 opt_rawLiteralsCost(const u8* literals, u32 litLength, const optstate* o) {
    if (litLength == 0) return 0;
    if (o->predef) return (litLength * 6) * BITCOST_MULTIPLIER;
    {
        u32 price = o->litSumBasePrice * litLength;
        u32 const litPriceMax = o->litSumBasePrice - BITCOST_MULTIPLIER;
        for (u32 u = 0; u < litLength; u++) {
            u32 litPrice = opt_weight(o->litFreq[literals[u]]);
            if (litPrice > litPriceMax) litPrice = litPriceMax;
            price -= litPrice;
        }
        return price;
    }
}
SQ_DEV u32 opt_litLengthPrice(u32 litLength, const optstate* o) {
    if (o->predef) return opt_weight(litLength);
    u32 extra = 0;
    if (litLength == BLOCKSIZE_MAX) { // (ZSTD_litLengthPrice prices 128 KiB as one bit more than 128 KiB - 1)
        extra = BITCOST_MULTIPLIER;
        litLength = BLOCKSIZE_MAX - 1;
    }
    {
        u32 const llCode = zs_LLcode(litLength);
        return extra + (zs_LL_bits_of(llCode) * BITCOST_MULTIPLIER) + o->litLengthSumBasePrice -
               opt_weight(o->litLengthFreq[llCode]);
    }
}
SQ_DEV u32 opt_getMatchPrice(u32 offBase, u32 matchLength, const optstate* o) {
    u32 price;
    u32 const offCode = highbit32(offBase);
    u32 const mlBase = matchLength - MINMATCH;
    if (o->predef) return opt_weight(mlBase) + ((16 + offCode) * BITCOST_MULTIPLIER);
    price = (offCode * BITCOST_MULTIPLIER) +
            (o->offCodeSumBasePrice - opt_weight(o->offCodeFreq[offCode]));
    if (offCode >= 20) price += (offCode - 19) * 2 * BITCOST_MULTIPLIER; /* optLevel < 2 */
    {
        u32 const mlCode = zs_MLcode(mlBase);
        price += (zs_ML_bits_of(mlCode) * BITCOST_MULTIPLIER) +
                 (o->matchLengthSumBasePrice - opt_weight(o->matchLengthFreq[mlCode]));
    }
    price += BITCOST_MULTIPLIER / 5;
    return price;
}
SQ_DEVFN void opt_updateStats(optstate* o, u32 litLength, const u8* literals, u32 offBase,
                            u32 matchLength) {
    for (u32 u = 0; u < litLength; u++) o->litFreq[literals[u]] += OPT_LITFREQ_ADD;
    o->litSum += litLength * OPT_LITFREQ_ADD;
    o->litLengthFreq[zs_LLcode(litLength)]++;
    o->litLengthSum++;
    o->offCodeFreq[highbit32(offBase)]++;
    o->offCodeSum++;
    o->matchLengthFreq[zs_MLcode(matchLength - MINMATCH)]++;
    o->matchLengthSum++;
}

SQ_DEVFN u32 readMINMATCH(const u8* p, u32 length) {
    return (length == 3) ? rd32(p) << 8 : rd32(p);
}

SQ_DEVFN u32 insertAndFindFirstIndexHash3(const mstate* ms, u32* nextToUpdate3, const u8* ip) {
    u32* const hashTable3 = ms->hashTable3;
    u32 const hashLog3 = ms->hashLog3;
    const u8* const base = ms->base;
    u32 idx = *nextToUpdate3;
    u32 const target = (u32)(ip - base);
    size_t const hash3 = hash3Ptr(ip, hashLog3);
    while (idx < target) {
        hashTable3[hash3Ptr(base + idx, hashLog3)] = idx;
        idx++;
    }
    *nextToUpdate3 = target;
    return hashTable3[hash3];
}

/* ZSTD_insertBt1 */
SQ_DEVFN u32 insertBt1(const mstate* ms, const u8* ip, const u8* iend, u32 target, u32 mls) {
    u32* const hashTable = ms->hashTable;
    u32 const hashLog = ms->cp.hashLog;
    size_t const h = hashPtr(ip, hashLog, mls);
    u32* const bt = ms->chainTable;
    u32 const btLog = ms->cp.chainLog - 1;
    u32 const btMask = (1u << btLog) - 1;
    u32 matchIndex = hashTable[h];
    size_t commonLengthSmaller = 0, commonLengthLarger = 0;
    const u8* const base = ms->base;
    const u8* match;
    const u32 curr = (u32)(ip - base);
    const u32 btLow = btMask >= curr ? 0 : curr - btMask;
    u32* smallerPtr = bt + 2 * (curr & btMask);
    u32* largerPtr = smallerPtr + 1;
    u32 dummy32;
    u32 const maxDistance = 1U << ms->cp.windowLog;
    u32 const windowLow =
        (target - WINDOW_START_INDEX > maxDistance) ? target - maxDistance : WINDOW_START_INDEX;
    u32 matchEndIdx = curr + 8 + 1;
    size_t bestLength = 8;
    u32 nbCompares = 1U << ms->cp.searchLog;

    hashTable[h] = curr;
    for (; nbCompares && (matchIndex >= windowLow); --nbCompares) {
        u32* const nextPtr = bt + 2 * (matchIndex & btMask);
        size_t matchLength =
            commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        match = base + matchIndex;
        matchLength += count_match(ip + matchLength, match + matchLength, iend);
        if (matchLength > bestLength) {
            bestLength = matchLength;
            if (matchLength > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)matchLength;
        }
        if (ip + matchLength == iend) break;
        if (match[matchLength] < ip[matchLength]) {
            *smallerPtr = matchIndex;
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerPtr = &dummy32;
                break;
            }
            smallerPtr = nextPtr + 1;
            matchIndex = nextPtr[1];
        } else {
            *largerPtr = matchIndex;
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerPtr = &dummy32;
                break;
            }
            largerPtr = nextPtr;
            matchIndex = nextPtr[0];
        }
    }
    *smallerPtr = *largerPtr = 0;
    {
        u32 positions = 0;
        u32 const adv = matchEndIdx - (curr + 8);
        if (bestLength > 384) positions = (192 < (u32)(bestLength - 384)) ? 192 : (u32)(bestLength - 384);
        return positions > adv ? positions : adv;
    }
}
SQ_DEVFN void updateTree(mstate* ms, const u8* ip, const u8* iend, u32 mls) {
    const u8* const base = ms->base;
    u32 const target = (u32)(ip - base);
    u32 idx = ms->nextToUpdate;
    while (idx < target) idx += insertBt1(ms, base + idx, iend, target, mls);
    ms->nextToUpdate = target;
}

/* ZSTD_insertBtAndGetAllMatches (noDict) */
SQ_DEVFN u32 insertBtAndGetAllMatches(match_t* matches, mstate* ms, u32* nextToUpdate3, const u8* ip,
                                    const u8* iLimit, const u32 rep[REP_NUM], u32 ll0,
                                    u32 lengthToBeat, u32 mls) {
    u32 const sufficient_len = ms->cp.targetLength < OPT_NUM - 1 ? ms->cp.targetLength : OPT_NUM - 1;
    const u8* const base = ms->base;
    u32 const curr = (u32)(ip - base);
    u32 const hashLog = ms->cp.hashLog;
    u32 const minMatch = (mls == 3) ? 3 : 4;
    u32* const hashTable = ms->hashTable;
    size_t const h = hashPtr(ip, hashLog, mls);
    u32 matchIndex = hashTable[h];
    u32* const bt = ms->chainTable;
    u32 const btLog = ms->cp.chainLog - 1;
    u32 const btMask = (1U << btLog) - 1;
    size_t commonLengthSmaller = 0, commonLengthLarger = 0;
    u32 const dictLimit = WINDOW_START_INDEX;
    u32 const btLow = (btMask >= curr) ? 0 : curr - btMask;
    u32 const maxDistance = 1U << ms->cp.windowLog;
    u32 const windowLow =
        (curr - WINDOW_START_INDEX > maxDistance) ? curr - maxDistance : WINDOW_START_INDEX;
    u32 const matchLow = windowLow ? windowLow : 1;
    u32* smallerPtr = bt + 2 * (curr & btMask);
    u32* largerPtr = bt + 2 * (curr & btMask) + 1;
    u32 matchEndIdx = curr + 8 + 1;
    u32 dummy32;
    u32 mnum = 0;
    u32 nbCompares = 1U << ms->cp.searchLog;
    size_t bestLength = lengthToBeat - 1;

    {
        u32 const lastR = REP_NUM + ll0;
        for (u32 repCode = ll0; repCode < lastR; repCode++) {
            u32 const repOffset = (repCode == REP_NUM) ? (rep[0] - 1) : rep[repCode];
            u32 const repIndex = curr - repOffset;
            u32 repLen = 0;
            if (repOffset - 1 < curr - dictLimit) {
                if ((repIndex >= windowLow) &
                    (readMINMATCH(ip, minMatch) == readMINMATCH(ip - repOffset, minMatch))) {
                    repLen = (u32)count_match(ip + minMatch, ip + minMatch - repOffset, iLimit) + minMatch;
                }
            }
            if (repLen > bestLength) {
                bestLength = repLen;
                matches[mnum].off = REPCODE_TO_OFFBASE(repCode - ll0 + 1);
                matches[mnum].len = repLen;
                mnum++;
                if ((repLen > sufficient_len) | (ip + repLen == iLimit)) return mnum;
            }
        }
    }
    if ((mls == 3) && (bestLength < mls)) {
        u32 const matchIndex3 = insertAndFindFirstIndexHash3(ms, nextToUpdate3, ip);
        if ((matchIndex3 >= matchLow) & (curr - matchIndex3 < (1 << 18))) {
            size_t const mlen = count_match(ip, base + matchIndex3, iLimit);
            if (mlen >= mls) {
                bestLength = mlen;
                matches[0].off = (curr - matchIndex3 + REP_NUM);
                matches[0].len = (u32)mlen;
                mnum = 1;
                if ((mlen > sufficient_len) | (ip + mlen == iLimit)) {
                    ms->nextToUpdate = curr + 1;
                    return 1;
                }
            }
        }
    }
    hashTable[h] = curr;
    for (; nbCompares && (matchIndex >= matchLow); --nbCompares) {
        u32* const nextPtr = bt + 2 * (matchIndex & btMask);
        const u8* match = base + matchIndex;
        size_t matchLength =
            commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        matchLength += count_match(ip + matchLength, match + matchLength, iLimit);
        if (matchLength > bestLength) {
            if (matchLength > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)matchLength;
            bestLength = matchLength;
            matches[mnum].off = (curr - matchIndex + REP_NUM);
            matches[mnum].len = (u32)matchLength;
            mnum++;
            if ((matchLength > OPT_NUM) | (ip + matchLength == iLimit)) break;
        }
        if (match[matchLength] < ip[matchLength]) {
            *smallerPtr = matchIndex;
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerPtr = &dummy32;
                break;
            }
            smallerPtr = nextPtr + 1;
            matchIndex = nextPtr[1];
        } else {
            *largerPtr = matchIndex;
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerPtr = &dummy32;
                break;
            }
            largerPtr = nextPtr;
            matchIndex = nextPtr[0];
        }
    }
    *smallerPtr = *largerPtr = 0;
    ms->nextToUpdate = matchEndIdx - 8;
    return mnum;
}

SQ_DEVFN u32 btGetAllMatches(match_t* matches, mstate* ms, u32* nextToUpdate3, const u8* ip,
                           const u8* iHighLimit, const u32 rep[REP_NUM], u32 ll0, u32 lengthToBeat,
                           u32 mls) {
    if (ip < ms->base + ms->nextToUpdate) return 0;
    updateTree(ms, ip, iHighLimit, mls);
    return insertBtAndGetAllMatches(matches, ms, nextToUpdate3, ip, iHighLimit, rep, ll0,
                                    lengthToBeat, mls);
}

SQ_DEVFN void opt_newRep(u32 out[REP_NUM], const u32 rep[REP_NUM], u32 offBase, u32 ll0) {
    /* ZSTD_newRep / ZSTD_updateRep */
    u32 r[REP_NUM] = {rep[0], rep[1], rep[2]};
    if (((offBase) > REP_NUM)) {
        r[2] = r[1];
        r[1] = r[0];
        r[0] = ((offBase) - REP_NUM);
    } else {
        u32 const repCode = offBase - 1 + ll0;
        if (repCode > 0) {
            u32 const currentOffset = (repCode == REP_NUM) ? (r[0] - 1) : r[repCode];
            r[2] = (repCode >= 2) ? r[1] : r[2];
            r[1] = r[0];
            r[0] = currentOffset;
        }
    }
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
}

#ifndef SQZ_OPT_LASTSTRETCH_ELSE
#define SQZ_OPT_LASTSTRETCH_ELSE 0 /* libzstd 1.5.7 ships `} {` (no else) in the reverse traversal */
#endif

/* ZSTD_compressBlock_opt_generic, optLevel 0, noDict */
SQ_DEVFN size_t compressBlock_btopt(mstate* ms, seqstore* ss, u32 rep[REP_NUM], const u8* src, size_t srcSize, void* ldsArea) {
    optstate* const og = (optstate*)ms->opt; // (workspace)
    OptLds* const lds = (OptLds*)ldsArea;
    for (u32 i_ = 0; i_ < OPT_HOT_WORDS; i_++) lds->hot[i_] = ((const u32*)og)[i_];
    optstate* const o = (optstate*)lds->hot; // (only the statistics: matchTable / priceTable are reached through og)
    const u8* const istart = src;
    const u8* ip = istart;
    const u8* anchor = istart;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = iend - 8;
    const u8* const base = ms->base;
    const u8* const prefixStart = base + WINDOW_START_INDEX;
    u32 const sufficient_len = ms->cp.targetLength < OPT_NUM - 1 ? ms->cp.targetLength : OPT_NUM - 1;
    u32 const mls = ms->cp.minMatch < 3 ? 3 : (ms->cp.minMatch > 6 ? 6 : ms->cp.minMatch);
    u32 const minMatch = (ms->cp.minMatch == 3) ? 3 : 4;
    u32 nextToUpdate3 = ms->nextToUpdate;
    opt_t* const gopt = og->priceTable;
    opt_t* const lopt = lds->win;
#define OPTE(i_) (*(((u32)(i_)) < OPT_LDS_N ? &lopt[(i_)] : &gopt[(i_)]))
    match_t* const matches = lds->matches;
    opt_t lastStretch;
    lastStretch.price = 0; lastStretch.off = lastStretch.mlen = lastStretch.litlen = 0; lastStretch.rep[0] = lastStretch.rep[1] = lastStretch.rep[2] = 0;

#define LIT_PRICE(_p) (int)opt_rawLiteralsCost(_p, 1, o)
#define LL_PRICE(_l) (int)opt_litLengthPrice(_l, o)
#define LL_INCPRICE(_l) (LL_PRICE(_l) - LL_PRICE(_l - 1))

    opt_rescaleFreqs(o, src, srcSize);
    ip += (ip == prefixStart);

    while (ip < ilimit) {
        u32 cur, last_pos = 0;
        {
            u32 const litlen = (u32)(ip - anchor);
            u32 const ll0 = !litlen;
            u32 nbMatches = btGetAllMatches(matches, ms, &nextToUpdate3, ip, iend, rep, ll0, minMatch, mls);
            if (!nbMatches) {
                ip++;
                continue;
            }
            OPTE(0).mlen = 0;
            OPTE(0).litlen = litlen;
            OPTE(0).price = LL_PRICE(litlen);
            OPTE(0).rep[0] = rep[0]; OPTE(0).rep[1] = rep[1]; OPTE(0).rep[2] = rep[2];
            {
                u32 const maxML = matches[nbMatches - 1].len;
                u32 const maxOffBase = matches[nbMatches - 1].off;
                if (maxML > sufficient_len) {
                    lastStretch.litlen = 0;
                    lastStretch.mlen = maxML;
                    lastStretch.off = maxOffBase;
                    cur = 0;
                    last_pos = maxML;
                    goto _shortestPath;
                }
            }
            {
                u32 pos, matchNb;
                for (pos = 1; pos < minMatch; pos++) {
                    OPTE(pos).price = OPT_MAX_PRICE;
                    OPTE(pos).mlen = 0;
                    OPTE(pos).litlen = litlen + pos;
                }
                for (matchNb = 0; matchNb < nbMatches; matchNb++) {
                    u32 const offBase = matches[matchNb].off;
                    u32 const end = matches[matchNb].len;
                    for (; pos <= end; pos++) {
                        int const matchPrice = (int)opt_getMatchPrice(offBase, pos, o);
                        int const sequencePrice = OPTE(0).price + matchPrice;
                        OPTE(pos).mlen = pos;
                        OPTE(pos).off = offBase;
                        OPTE(pos).litlen = 0;
                        OPTE(pos).price = sequencePrice + LL_PRICE(0);
                    }
                }
                last_pos = pos - 1;
                OPTE(pos).price = OPT_MAX_PRICE;
            }
        }
        for (cur = 1; cur <= last_pos; cur++) {
            const u8* const inr = ip + cur;
            {
                u32 const litlen = OPTE(cur - 1).litlen + 1;
                int const price = OPTE(cur - 1).price + LIT_PRICE(ip + cur - 1) + LL_INCPRICE(litlen);
                if (price <= OPTE(cur).price) {
                    OPTE(cur) = OPTE(cur - 1);
                    OPTE(cur).litlen = litlen;
                    OPTE(cur).price = price;
                }
            }
            if (OPTE(cur).litlen == 0) {
                u32 const prev = cur - OPTE(cur).mlen;
                opt_newRep(OPTE(cur).rep, OPTE(prev).rep, OPTE(cur).off, OPTE(prev).litlen == 0);
            }
            if (inr > ilimit) continue;
            if (cur == last_pos) break;
            if (OPTE(cur + 1).price <= OPTE(cur).price + (BITCOST_MULTIPLIER / 2)) continue;
            {
                u32 const ll0 = (OPTE(cur).litlen == 0);
                int const previousPrice = OPTE(cur).price;
                int const basePrice = previousPrice + LL_PRICE(0);
                u32 nbMatches = btGetAllMatches(matches, ms, &nextToUpdate3, inr, iend, OPTE(cur).rep,
                                                ll0, minMatch, mls);
                u32 matchNb;
                if (!nbMatches) continue;
                {
                    u32 const longestML = matches[nbMatches - 1].len;
                    if ((longestML > sufficient_len) || (cur + longestML >= OPT_NUM) ||
                        (ip + cur + longestML >= iend)) {
                        lastStretch.mlen = longestML;
                        lastStretch.off = matches[nbMatches - 1].off;
                        lastStretch.litlen = 0;
                        last_pos = cur + longestML;
                        goto _shortestPath;
                    }
                }
                for (matchNb = 0; matchNb < nbMatches; matchNb++) {
                    u32 const offset = matches[matchNb].off;
                    u32 const lastML = matches[matchNb].len;
                    u32 const startML = (matchNb > 0) ? matches[matchNb - 1].len + 1 : minMatch;
                    u32 mlen;
                    for (mlen = lastML; mlen >= startML; mlen--) {
                        u32 const pos = cur + mlen;
                        int const price = basePrice + (int)opt_getMatchPrice(offset, mlen, o);
                        if ((pos > last_pos) || (price < OPTE(pos).price)) {
                            while (last_pos < pos) {
                                last_pos++;
                                OPTE(last_pos).price = OPT_MAX_PRICE;
                                OPTE(last_pos).litlen = !0;
                            }
                            OPTE(pos).mlen = mlen;
                            OPTE(pos).off = offset;
                            OPTE(pos).litlen = 0;
                            OPTE(pos).price = price;
                        } else {
                            break; /* optLevel == 0 */
                        }
                    }
                }
            }
            OPTE(last_pos + 1).price = OPT_MAX_PRICE;
        }
        lastStretch = OPTE(last_pos);
        cur = last_pos - lastStretch.mlen;

    _shortestPath:
        if (lastStretch.mlen == 0) {
            ip += last_pos;
            continue;
        }
        if (lastStretch.litlen == 0) {
            u32 reps[REP_NUM];
            opt_newRep(reps, OPTE(cur).rep, lastStretch.off, OPTE(cur).litlen == 0);
            rep[0] = reps[0]; rep[1] = reps[1]; rep[2] = reps[2];
        } else {
            rep[0] = lastStretch.rep[0]; rep[1] = lastStretch.rep[1]; rep[2] = lastStretch.rep[2];
            cur -= lastStretch.litlen;
        }
        {
            u32 const storeEnd = cur + 2;
            u32 storeStart = storeEnd;
            u32 stretchPos = cur;
            if (lastStretch.litlen > 0) {
                OPTE(storeEnd).litlen = lastStretch.litlen;
                OPTE(storeEnd).mlen = 0;
                storeStart = storeEnd - 1;
                OPTE(storeStart) = lastStretch;
            }
#if SQZ_OPT_LASTSTRETCH_ELSE
            else
#endif
            {
                OPTE(storeEnd) = lastStretch;
                storeStart = storeEnd;
            }
            while (1) {
                opt_t nextStretch = OPTE(stretchPos);
                OPTE(storeStart).litlen = nextStretch.litlen;
                if (nextStretch.mlen == 0) break;
                storeStart--;
                OPTE(storeStart) = nextStretch;
                stretchPos -= nextStretch.litlen + nextStretch.mlen;
            }
            for (u32 storePos = storeStart; storePos <= storeEnd; storePos++) {
                u32 const llen = OPTE(storePos).litlen;
                u32 const mlen = OPTE(storePos).mlen;
                u32 const offBase = OPTE(storePos).off;
                u32 const advance = llen + mlen;
                if (mlen == 0) {
                    ip = anchor + llen;
                    continue;
                }
                opt_updateStats(o, llen, anchor, offBase, mlen);
                storeSeq(ss, llen, anchor, offBase, mlen);
                anchor += advance;
                ip = anchor;
            }
            opt_setBasePrices(o);
        }
    }
#undef LIT_PRICE
#undef LL_PRICE
#undef LL_INCPRICE
#undef OPTE
    for (u32 i_ = 0; i_ < OPT_HOT_WORDS; i_++) ((u32*)og)[i_] = lds->hot[i_]; // (a later block of the chunk goes on from them)
    return (size_t)(iend - anchor);
}

#undef BLOCKSIZE_MAX
#undef MINMATCH
#undef REP_NUM
#undef WINDOW_START_INDEX
#undef REPCODE_TO_OFFBASE
} // namespace opt
} // namespace sqz
