/*
 * zstd12_oracle.c -- CPU ORACLE (test infrastructure, see sq_oracle.h).
 *
 * Restates, as plain scalar C, what /root/reference/src/util/chunk.rs:89-90
 *     let compressed = compress(chunk, COMPRESSION_LEVEL)      // zstd::bulk::compress(chunk, 12)
 * computes: one self-contained zstd frame per chunk, fresh context, level 12.
 * The arithmetic lives in zstd-sys 2.0.15+zstd.1.5.7 (Cargo.lock:1637-1640), i.e. C libzstd
 * 1.5.7, which is not vendored in /root/reference; this file restates that library's published
 * algorithm (RFC 8878 format + the 1.5.7 encoder's decisions).  libzstd routine names are quoted
 * in comments.  Pinned against libzstd 1.5.7 outputs: tools/gen_golden.py, tests/golden/.
 *
 * Level-12 strategy by chunk length n (measured with ZSTD_getCParams):
 *     n <= 16 KiB           btopt   (minMatch 3)
 *     16 KiB < n <= 256 KiB btlazy2 (minMatch 4, DUBT binary tree)
 *     n > 256 KiB           lazy2   (minMatch 5, row-hash match finder, 64-entry rows)
 */
#include "sq_oracle.h"
#include "zstd12_entropy.h"
#include <stdlib.h>
#include <string.h>

#define KB *(1 << 10)
#define BLOCKSIZE_MAX (128 KB)
#define MINMATCH 3
#define REP_NUM 3
#define WINDOW_START_INDEX 2
#define OFFSET_TO_OFFBASE(o) ((o) + REP_NUM)
#define OFFBASE_TO_OFFSET(o) ((o)-REP_NUM)
#define OFFBASE_IS_OFFSET(o) ((o) > REP_NUM)
#define REPCODE1_TO_OFFBASE 1
#define REPCODE_TO_OFFBASE(r) (r)

enum { STRAT_LAZY2 = 5, STRAT_BTLAZY2 = 6, STRAT_BTOPT = 7 };

static u32 rd32(const u8* p) {
    u32 v;
    memcpy(&v, p, 4);
    return v;
}
static u64 rd64(const u8* p) {
    u64 v;
    memcpy(&v, p, 8);
    return v;
}
static u16 rd16(const u8* p) {
    u16 v;
    memcpy(&v, p, 2);
    return v;
}

/* ------------------------------------------------------------------ parameters */
sqo_cparams sqo_get_cparams(size_t n) {
    /* ZSTD_getCParams(12, n, 0) after ZSTD_adjustCParams_internal; table measured against
     * libzstd 1.5.7.  Boundaries are 2^k + 1. */
    sqo_cparams p;
    if (n == 0) n = 1; /* unknown size is never used on this path */
    if (n <= 16 KB) {
        u32 w = 0;
        while (((size_t)1 << w) < n) w++; /* ceil(log2 n) */
        p.windowLog = w < 10 ? 10 : w;
        p.chainLog = (w + 1 < 7) ? 7 : w + 1;
        p.hashLog = p.chainLog > 14 ? 14 : p.chainLog;
        p.searchLog = 4;
        p.minMatch = 3;
        p.targetLength = 24;
        p.strategy = STRAT_BTOPT;
    } else if (n <= 256 KB) {
        u32 w = 15;
        while (((size_t)1 << w) < n) w++;
        p.windowLog = w;
        p.chainLog = w + 1;
        p.hashLog = (w == 17) ? 17 : w + 1;
        p.searchLog = 7;
        p.minMatch = 4;
        p.targetLength = 12;
        p.strategy = STRAT_BTLAZY2;
    } else {
        u32 w = 19;
        while (((size_t)1 << w) < n && w < 21) w++;
        p.windowLog = w;
        p.chainLog = (w == 19) ? 19 : w;
        p.hashLog = w + 1;
        p.searchLog = 6;
        p.minMatch = 5;
        p.targetLength = 32;
        p.strategy = STRAT_LAZY2;
    }
    return p;
}

size_t sqo_compress_bound(size_t n) {
    return n + (n >> 8) + ((n < (128 KB)) ? (((128 KB) - n) >> 11) : 0);
}

/* ------------------------------------------------------------------ match state */
typedef struct {
    const u8* base; /* base + index = byte; first source byte has index WINDOW_START_INDEX */
    u32 nextToUpdate;
    u32* hashTable;
    u32* chainTable;
    u8* tagTable;
    u32* hashTable3;
    u32 hashLog3;
    u32 rowHashLog;
    int lazySkipping;
    sqo_cparams cp;
    void* opt; /* btopt state, see zstd12_opt.inc */
} mstate;

static size_t count_match(const u8* pIn, const u8* pMatch, const u8* const pInLimit) {
    /* ZSTD_count */
    const u8* const pStart = pIn;
    while (pIn < pInLimit && *pIn == *pMatch) {
        pIn++;
        pMatch++;
    }
    return (size_t)(pIn - pStart);
}

static const u32 prime3bytes = 506832829U;
static const u32 prime4bytes = 2654435761U;
static const u64 prime5bytes = 889523592379ULL;
static const u64 prime6bytes = 227718039650203ULL;
static size_t hashPtr(const u8* p, u32 hBits, u32 mls) { /* ZSTD_hashPtr, salt 0 */
    switch (mls) {
    default:
    case 4: return (size_t)((rd32(p) * prime4bytes) >> (32 - hBits));
    case 5: return (size_t)(((rd64(p) << (64 - 40)) * prime5bytes) >> (64 - hBits));
    case 6: return (size_t)(((rd64(p) << (64 - 48)) * prime6bytes) >> (64 - hBits));
    }
}
static size_t hash3Ptr(const u8* p, u32 hBits) { /* ZSTD_hash3Ptr */
    return (size_t)(((rd32(p) << 8) * prime3bytes) >> (32 - hBits));
}

static void storeSeq(seqstore* ss, size_t litLength, const u8* literals, u32 offBase,
                     size_t matchLength) { /* ZSTD_storeSeq */
    memcpy(ss->lits + ss->nlits, literals, litLength);
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

/* ------------------------------------------------------------------ DUBT (btlazy2) */
#define DUBT_UNSORTED_MARK 1

static void updateDUBT(mstate* ms, const u8* ip, u32 mls) { /* ZSTD_updateDUBT */
    u32* const hashTable = ms->hashTable;
    u32 const hashLog = ms->cp.hashLog;
    u32* const bt = ms->chainTable;
    u32 const btLog = ms->cp.chainLog - 1;
    u32 const btMask = (1u << btLog) - 1;
    const u8* const base = ms->base;
    u32 const target = (u32)(ip - base);
    u32 idx = ms->nextToUpdate;
    for (; idx < target; idx++) {
        size_t const h = hashPtr(base + idx, hashLog, mls);
        u32 const matchIndex = hashTable[h];
        u32* const nextCandidatePtr = bt + 2 * (idx & btMask);
        u32* const sortMarkPtr = nextCandidatePtr + 1;
        hashTable[h] = idx;
        *nextCandidatePtr = matchIndex;
        *sortMarkPtr = DUBT_UNSORTED_MARK;
    }
    ms->nextToUpdate = target;
}

static void insertDUBT1(const mstate* ms, u32 curr, const u8* inputEnd, u32 nbCompares, u32 btLow) {
    /* ZSTD_insertDUBT1 (noDict) */
    u32* const bt = ms->chainTable;
    u32 const btLog = ms->cp.chainLog - 1;
    u32 const btMask = (1u << btLog) - 1;
    size_t commonLengthSmaller = 0, commonLengthLarger = 0;
    const u8* const base = ms->base;
    const u8* const ip = base + curr;
    const u8* const iend = inputEnd;
    const u8* match;
    u32* smallerPtr = bt + 2 * (curr & btMask);
    u32* largerPtr = smallerPtr + 1;
    u32 matchIndex = *smallerPtr;
    u32 dummy32;
    u32 const windowValid = WINDOW_START_INDEX;
    u32 const maxDistance = 1U << ms->cp.windowLog;
    u32 const windowLow = (curr - windowValid > maxDistance) ? curr - maxDistance : windowValid;

    for (; nbCompares && (matchIndex > windowLow); --nbCompares) {
        u32* const nextPtr = bt + 2 * (matchIndex & btMask);
        size_t matchLength =
            commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        match = base + matchIndex;
        matchLength += count_match(ip + matchLength, match + matchLength, iend);
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
}

static size_t DUBT_findBestMatch(mstate* ms, const u8* const ip, const u8* const iend,
                                 size_t* offBasePtr, u32 const mls) {
    /* ZSTD_DUBT_findBestMatch (noDict) */
    u32* const hashTable = ms->hashTable;
    u32 const hashLog = ms->cp.hashLog;
    size_t const h = hashPtr(ip, hashLog, mls);
    u32 matchIndex = hashTable[h];
    const u8* const base = ms->base;
    u32 const curr = (u32)(ip - base);
    u32 const maxDistance = 1U << ms->cp.windowLog;
    u32 const lowestValid = WINDOW_START_INDEX;
    u32 const windowLow = (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
    u32* const bt = ms->chainTable;
    u32 const btLog = ms->cp.chainLog - 1;
    u32 const btMask = (1u << btLog) - 1;
    u32 const btLow = (btMask >= curr) ? 0 : curr - btMask;
    u32 const unsortLimit = btLow > windowLow ? btLow : windowLow;
    u32* nextCandidate = bt + 2 * (matchIndex & btMask);
    u32* unsortedMark = bt + 2 * (matchIndex & btMask) + 1;
    u32 nbCompares = 1U << ms->cp.searchLog;
    u32 nbCandidates = nbCompares;
    u32 previousCandidate = 0;

    while ((matchIndex > unsortLimit) && (*unsortedMark == DUBT_UNSORTED_MARK) &&
           (nbCandidates > 1)) {
        *unsortedMark = previousCandidate;
        previousCandidate = matchIndex;
        matchIndex = *nextCandidate;
        nextCandidate = bt + 2 * (matchIndex & btMask);
        unsortedMark = bt + 2 * (matchIndex & btMask) + 1;
        nbCandidates--;
    }
    if ((matchIndex > unsortLimit) && (*unsortedMark == DUBT_UNSORTED_MARK)) {
        *nextCandidate = *unsortedMark = 0;
    }
    matchIndex = previousCandidate;
    while (matchIndex) {
        u32* const nextCandidateIdxPtr = bt + 2 * (matchIndex & btMask) + 1;
        u32 const nextCandidateIdx = *nextCandidateIdxPtr;
        insertDUBT1(ms, matchIndex, iend, nbCandidates, unsortLimit);
        matchIndex = nextCandidateIdx;
        nbCandidates++;
    }
    {
        size_t commonLengthSmaller = 0, commonLengthLarger = 0;
        u32* smallerPtr = bt + 2 * (curr & btMask);
        u32* largerPtr = bt + 2 * (curr & btMask) + 1;
        u32 matchEndIdx = curr + 8 + 1;
        u32 dummy32;
        size_t bestLength = 0;

        matchIndex = hashTable[h];
        hashTable[h] = curr;

        for (; nbCompares && (matchIndex > windowLow); --nbCompares) {
            u32* const nextPtr = bt + 2 * (matchIndex & btMask);
            size_t matchLength =
                commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
            const u8* match = base + matchIndex;
            matchLength += count_match(ip + matchLength, match + matchLength, iend);
            if (matchLength > bestLength) {
                if (matchLength > matchEndIdx - matchIndex) matchEndIdx = matchIndex + (u32)matchLength;
                if ((4 * (int)(matchLength - bestLength)) >
                    (int)(highbit32(curr - matchIndex + 1) - highbit32((u32)*offBasePtr))) {
                    bestLength = matchLength;
                    *offBasePtr = OFFSET_TO_OFFBASE(curr - matchIndex);
                }
                if (ip + matchLength == iend) break;
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
        return bestLength;
    }
}

static size_t BtFindBestMatch(mstate* ms, const u8* ip, const u8* iLimit, size_t* offBasePtr,
                              u32 mls) { /* ZSTD_BtFindBestMatch */
    if (ip < ms->base + ms->nextToUpdate) return 0;
    updateDUBT(ms, ip, mls);
    return DUBT_findBestMatch(ms, ip, iLimit, offBasePtr, mls);
}

/* ------------------------------------------------------------------ row hash (lazy2, n>256K) */
#define ROW_TAG_BITS 8
#define ROW_TAG_MASK 0xFFu
#define ROW_CACHE_SIZE 8

static u32 row_nextIndex(u8* tagRow, u32 rowMask) { /* ZSTD_row_nextIndex */
    u32 next = (*tagRow - 1) & rowMask;
    next += (next == 0) ? rowMask : 0;
    *tagRow = (u8)next;
    return next;
}
static void row_insert(mstate* ms, u32 idx, u32 mls, u32 rowLog, u32 rowMask) {
    u32 const hash = (u32)hashPtr(ms->base + idx, ms->rowHashLog + ROW_TAG_BITS, mls);
    u32 const relRow = (hash >> ROW_TAG_BITS) << rowLog;
    u32* const row = ms->hashTable + relRow;
    u8* tagRow = ms->tagTable + relRow;
    u32 const pos = row_nextIndex(tagRow, rowMask);
    tagRow[pos] = (u8)(hash & ROW_TAG_MASK);
    row[pos] = idx;
}
static void row_update(mstate* ms, const u8* ip, u32 mls, u32 rowLog, u32 rowMask) {
    /* ZSTD_row_update_internal (useCache=1).  The 8-entry hash cache of libzstd is a pure
     * prefetch optimisation (it always yields hash(base+idx)), so it is not modelled. */
    u32 idx = ms->nextToUpdate;
    const u32 target = (u32)(ip - ms->base);
    const u32 kSkipThreshold = 384;
    const u32 kMaxMatchStartPositionsToUpdate = 96;
    const u32 kMaxMatchEndPositionsToUpdate = 32;
    if (target - idx > kSkipThreshold) {
        u32 const bound = idx + kMaxMatchStartPositionsToUpdate;
        for (; idx < bound; ++idx) row_insert(ms, idx, mls, rowLog, rowMask);
        idx = target - kMaxMatchEndPositionsToUpdate;
    }
    for (; idx < target; ++idx) row_insert(ms, idx, mls, rowLog, rowMask);
    ms->nextToUpdate = target;
}

static size_t RowFindBestMatch(mstate* ms, const u8* const ip, const u8* const iLimit,
                               size_t* offsetPtr, const u32 mls, const u32 rowLog) {
    /* ZSTD_RowFindBestMatch (noDict) */
    u32* const hashTable = ms->hashTable;
    u8* const tagTable = ms->tagTable;
    const u32 hashLog = ms->rowHashLog;
    const u8* const base = ms->base;
    const u32 curr = (u32)(ip - base);
    const u32 maxDistance = 1U << ms->cp.windowLog;
    const u32 lowestValid = WINDOW_START_INDEX;
    const u32 lowLimit = (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
    const u32 rowEntries = (1U << rowLog);
    const u32 rowMask = rowEntries - 1;
    const u32 cappedSearchLog = ms->cp.searchLog < rowLog ? ms->cp.searchLog : rowLog;
    u32 nbAttempts = 1U << cappedSearchLog;
    size_t ml = 4 - 1;
    u32 hash;

    if (!ms->lazySkipping) {
        row_update(ms, ip, mls, rowLog, rowMask);
    } else {
        ms->nextToUpdate = curr;
    }
    hash = (u32)hashPtr(ip, hashLog + ROW_TAG_BITS, mls);
    {
        u32 const relRow = (hash >> ROW_TAG_BITS) << rowLog;
        u32 const tag = hash & ROW_TAG_MASK;
        u32* const row = hashTable + relRow;
        u8* tagRow = tagTable + relRow;
        u32 const head = *tagRow & rowMask;
        u32 matchBuffer[64];
        size_t numMatches = 0, currMatch = 0;
        /* ZSTD_row_getMatchMask + rotate: visit positions head, head+1, ... (newest first) */
        for (u32 j = 0; j < rowEntries && nbAttempts > 0; j++) {
            u32 const matchPos = (head + j) & rowMask;
            u32 matchIndex;
            if (tagRow[matchPos] != (u8)tag) continue;
            matchIndex = row[matchPos];
            if (matchPos == 0) continue;
            if (matchIndex < lowLimit) break;
            matchBuffer[numMatches++] = matchIndex;
            --nbAttempts;
        }
        {
            u32 const pos = row_nextIndex(tagRow, rowMask);
            tagRow[pos] = (u8)tag;
            row[pos] = ms->nextToUpdate++;
        }
        for (; currMatch < numMatches; ++currMatch) {
            u32 const matchIndex = matchBuffer[currMatch];
            size_t currentMl = 0;
            const u8* const match = base + matchIndex;
            if (rd32(match + ml - 3) == rd32(ip + ml - 3)) currentMl = count_match(ip, match, iLimit);
            if (currentMl > ml) {
                ml = currentMl;
                *offsetPtr = OFFSET_TO_OFFBASE(curr - matchIndex);
                if (ip + currentMl == iLimit) break;
            }
        }
    }
    return ml;
}

/* ------------------------------------------------------------------ lazy parser */
enum { SEARCH_BT = 0, SEARCH_ROW = 1 };

static size_t searchMax(mstate* ms, const u8* ip, const u8* iend, size_t* offBase, u32 mls,
                        u32 rowLog, int method) {
    if (method == SEARCH_ROW) return RowFindBestMatch(ms, ip, iend, offBase, mls, rowLog);
    return BtFindBestMatch(ms, ip, iend, offBase, mls);
}

/* ZSTD_compressBlock_lazy_generic (noDict, depth 2) */
static size_t compressBlock_lazy(mstate* ms, seqstore* ss, u32 rep[REP_NUM], const u8* src,
                                 size_t srcSize, int method) {
    const u32 depth = 2;
    const u8* const istart = src;
    const u8* ip = istart;
    const u8* anchor = istart;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = (method == SEARCH_ROW) ? iend - 8 - ROW_CACHE_SIZE : iend - 8;
    const u8* const base = ms->base;
    const u32 prefixLowestIndex = WINDOW_START_INDEX;
    const u8* const prefixLowest = base + prefixLowestIndex;
    u32 mls = ms->cp.minMatch;
    u32 rowLog = ms->cp.searchLog;
    u32 offset_1 = rep[0], offset_2 = rep[1];
    u32 offsetSaved1 = 0, offsetSaved2 = 0;
    if (mls < 4) mls = 4;
    if (mls > 6) mls = 6;
    if (rowLog < 4) rowLog = 4;
    if (rowLog > 6) rowLog = 6;

    ip += (ip == prefixLowest);
    {
        u32 const curr = (u32)(ip - base);
        u32 const maxDistance = 1U << ms->cp.windowLog;
        u32 const windowLow =
            (curr - prefixLowestIndex > maxDistance) ? curr - maxDistance : prefixLowestIndex;
        u32 const maxRep = curr - windowLow;
        if (offset_2 > maxRep) offsetSaved2 = offset_2, offset_2 = 0;
        if (offset_1 > maxRep) offsetSaved1 = offset_1, offset_1 = 0;
    }
    ms->lazySkipping = 0;

    while (ip < ilimit) {
        size_t matchLength = 0;
        size_t offBase = REPCODE1_TO_OFFBASE;
        const u8* start = ip + 1;

        if ((offset_1 > 0) & (rd32(ip + 1 - offset_1) == rd32(ip + 1))) {
            matchLength = count_match(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
        }
        {
            size_t offbaseFound = 999999999;
            size_t const ml2 = searchMax(ms, ip, iend, &offbaseFound, mls, rowLog, method);
            if (ml2 > matchLength) matchLength = ml2, start = ip, offBase = offbaseFound;
        }
        if (matchLength < 4) {
            size_t const step = ((size_t)(ip - anchor) >> 8) + 1; /* kSearchStrength */
            ip += step;
            ms->lazySkipping = step > 8; /* kLazySkippingStep */
            continue;
        }
        if (depth >= 1)
            while (ip < ilimit) {
                ip++;
                if ((offBase) && ((offset_1 > 0) & (rd32(ip) == rd32(ip - offset_1)))) {
                    size_t const mlRep = count_match(ip + 4, ip + 4 - offset_1, iend) + 4;
                    int const gain2 = (int)(mlRep * 3);
                    int const gain1 = (int)(matchLength * 3 - highbit32((u32)offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1))
                        matchLength = mlRep, offBase = REPCODE1_TO_OFFBASE, start = ip;
                }
                {
                    size_t ofbCandidate = 999999999;
                    size_t const ml2 = searchMax(ms, ip, iend, &ofbCandidate, mls, rowLog, method);
                    int const gain2 = (int)(ml2 * 4 - highbit32((u32)ofbCandidate));
                    int const gain1 = (int)(matchLength * 4 - highbit32((u32)offBase) + 4);
                    if ((ml2 >= 4) && (gain2 > gain1)) {
                        matchLength = ml2, offBase = ofbCandidate, start = ip;
                        continue;
                    }
                }
                if ((depth == 2) && (ip < ilimit)) {
                    ip++;
                    if ((offBase) && ((offset_1 > 0) & (rd32(ip) == rd32(ip - offset_1)))) {
                        size_t const mlRep = count_match(ip + 4, ip + 4 - offset_1, iend) + 4;
                        int const gain2 = (int)(mlRep * 4);
                        int const gain1 = (int)(matchLength * 4 - highbit32((u32)offBase) + 1);
                        if ((mlRep >= 4) && (gain2 > gain1))
                            matchLength = mlRep, offBase = REPCODE1_TO_OFFBASE, start = ip;
                    }
                    {
                        size_t ofbCandidate = 999999999;
                        size_t const ml2 =
                            searchMax(ms, ip, iend, &ofbCandidate, mls, rowLog, method);
                        int const gain2 = (int)(ml2 * 4 - highbit32((u32)ofbCandidate));
                        int const gain1 = (int)(matchLength * 4 - highbit32((u32)offBase) + 7);
                        if ((ml2 >= 4) && (gain2 > gain1)) {
                            matchLength = ml2, offBase = ofbCandidate, start = ip;
                            continue;
                        }
                    }
                }
                break;
            }

        if (OFFBASE_IS_OFFSET(offBase)) {
            while (((start > anchor) & (start - OFFBASE_TO_OFFSET(offBase) > prefixLowest)) &&
                   (start[-1] == (start - OFFBASE_TO_OFFSET(offBase))[-1])) {
                start--;
                matchLength++;
            }
            offset_2 = offset_1;
            offset_1 = (u32)OFFBASE_TO_OFFSET(offBase);
        }
        {
            size_t const litLength = (size_t)(start - anchor);
            storeSeq(ss, litLength, anchor, (u32)offBase, matchLength);
            anchor = ip = start + matchLength;
        }
        if (ms->lazySkipping) ms->lazySkipping = 0;

        while (((ip <= ilimit) & (offset_2 > 0)) && (rd32(ip) == rd32(ip - offset_2))) {
            matchLength = count_match(ip + 4, ip + 4 - offset_2, iend) + 4;
            offBase = offset_2;
            offset_2 = offset_1;
            offset_1 = (u32)offBase;
            storeSeq(ss, 0, anchor, REPCODE1_TO_OFFBASE, matchLength);
            ip += matchLength;
            anchor = ip;
        }
    }
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (size_t)(iend - anchor);
}

#include "zstd12_opt.inc" /* btopt (n <= 16 KiB): compressBlock_btopt() */

/* ------------------------------------------------------------------ pre-block splitter */
#define PS_HASHTABLESIZE 1024
typedef struct {
    u32 events[PS_HASHTABLESIZE];
    size_t nbEvents;
} Fingerprint;

static u32 ps_hash2(const u8* p, u32 hashLog) {
    if (hashLog == 8) return (u32)p[0];
    return (u32)(rd16(p)) * 0x9e3779b9U >> (32 - hashLog);
}
static void ps_record(Fingerprint* fp, const u8* src, size_t srcSize, size_t samplingRate,
                      u32 hashLog) { /* recordFingerprint_generic */
    size_t const limit = srcSize - 2 + 1;
    memset(fp->events, 0, sizeof(u32) * ((size_t)1 << hashLog));
    fp->nbEvents = 0;
    for (size_t n = 0; n < limit; n += samplingRate) fp->events[ps_hash2(src + n, hashLog)]++;
    fp->nbEvents += limit / samplingRate;
}
static u64 ps_abs64(int64_t v) { return (u64)((v < 0) ? -v : v); }
static u64 ps_fpDistance(const Fingerprint* a, const Fingerprint* b, u32 hashLog) {
    u64 distance = 0;
    for (size_t n = 0; n < ((size_t)1 << hashLog); n++)
        distance += ps_abs64((int64_t)a->events[n] * (int64_t)b->nbEvents -
                             (int64_t)b->events[n] * (int64_t)a->nbEvents);
    return distance;
}
static int ps_compare(const Fingerprint* ref, const Fingerprint* newfp, int penalty, u32 hashLog) {
    u64 p50 = (u64)ref->nbEvents * (u64)newfp->nbEvents;
    u64 deviation = ps_fpDistance(ref, newfp, hashLog);
    u64 threshold = p50 * (u64)(16 - 2 + penalty) / 16;
    return deviation >= threshold;
}
static void ps_merge(Fingerprint* acc, const Fingerprint* newfp) {
    for (size_t n = 0; n < PS_HASHTABLESIZE; n++) acc->events[n] += newfp->events[n];
    acc->nbEvents += newfp->nbEvents;
}
/* ZSTD_splitBlock_byChunks(level) : level 0..3 */
static size_t splitBlock_byChunks(const u8* p, size_t blockSize, int level) {
    static const size_t rates[] = {43, 11, 5, 1};
    static const u32 hashParams[] = {8, 9, 10, 10};
    static __thread Fingerprint past, newEv; /* per thread: the oracle is called from thread pools (bench_threads.c, tools/gpu_fuzz.py) */
    int penalty = 3;
    size_t pos;
    const size_t CHUNKSIZE = 8 << 10;
    memset(&past, 0, sizeof(past));
    memset(&newEv, 0, sizeof(newEv));
    ps_record(&past, p, CHUNKSIZE, rates[level], hashParams[level]);
    for (pos = CHUNKSIZE; pos <= blockSize - CHUNKSIZE; pos += CHUNKSIZE) {
        ps_record(&newEv, p + pos, CHUNKSIZE, rates[level], hashParams[level]);
        if (ps_compare(&past, &newEv, penalty, hashParams[level])) {
            return pos;
        } else {
            ps_merge(&past, &newEv);
            if (penalty > 0) penalty--;
        }
    }
    return blockSize;
}
/* ZSTD_optimalBlockSize */
static size_t optimalBlockSize(const u8* src, size_t srcSize, size_t blockSizeMax, u32 strat,
                               int64_t savings) {
    static const int splitLevels[] = {0, 0, 1, 2, 2, 3, 3, 4, 4, 4};
    int splitLevel;
    if (srcSize < 128 KB || blockSizeMax < 128 KB) return srcSize < blockSizeMax ? srcSize : blockSizeMax;
    if (savings < 3) return 128 KB;
    splitLevel = splitLevels[strat];
    if (splitLevel == 0) return 128 KB; /* fromBorders variant: not reachable at level 12 */
    return splitBlock_byChunks(src, blockSizeMax, splitLevel - 1);
}

/* ------------------------------------------------------------------ frame / block loop */
typedef struct {
    entropy_t ent;
    u32 rep[REP_NUM];
} blockstate;

typedef struct {
    mstate ms;
    seqstore ss;
    blockstate bs[2];
    int prevIdx;
    int isFirstBlock;
    /* sequence dump (ZSTD_generateSequences analogue) */
    sqo_sequence* dump;
    size_t dumpCap, dumpN;
} cctx;

static int isRLE(const u8* src, size_t length) {
    for (size_t i = 1; i < length; i++)
        if (src[i] != src[0]) return 0;
    return 1;
}

static void dump_block(cctx* c, const u32 repStart[REP_NUM], size_t lastLits) {
    /* mirrors ZSTD_copyBlockSequences closely enough for diffing (offset resolved from repcodes) */
    u32 rep[REP_NUM] = {repStart[0], repStart[1], repStart[2]};
    for (u32 i = 0; i < c->ss.nseq && c->dumpN < c->dumpCap; i++) {
        sqo_sequence* o = &c->dump[c->dumpN++];
        u32 ll = c->ss.seqs[i].litLength, ml = c->ss.seqs[i].mlBase + MINMATCH;
        u32 ob = c->ss.seqs[i].offBase;
        if (c->ss.longLengthType == 1 && c->ss.longLengthPos == i) ll += 0x10000;
        if (c->ss.longLengthType == 2 && c->ss.longLengthPos == i) ml += 0x10000;
        o->litLength = ll;
        o->matchLength = ml;
        o->rep = 0;
        if (OFFBASE_IS_OFFSET(ob)) {
            o->offset = OFFBASE_TO_OFFSET(ob);
            rep[2] = rep[1];
            rep[1] = rep[0];
            rep[0] = o->offset;
        } else {
            u32 const repCode = ob - 1 + (ll == 0);
            o->rep = ob;
            if (repCode == 3)
                o->offset = rep[0] - 1;
            else
                o->offset = rep[repCode];
            if (repCode > 0) {
                u32 cur = o->offset;
                if (repCode >= 2) rep[2] = rep[1];
                rep[1] = rep[0];
                rep[0] = cur;
            }
        }
    }
    if (c->dumpN < c->dumpCap) {
        sqo_sequence* o = &c->dump[c->dumpN++];
        o->offset = 0;
        o->litLength = (u32)lastLits;
        o->matchLength = 0;
        o->rep = 0;
    }
}

/* ZSTD_buildSeqStore + ZSTD_compressBlock_internal.  Returns body size, 0 => raw block,
 * 1 => RLE block (dst[0] holds the byte). */
static size_t compressBlock_internal(cctx* c, u8* dst, size_t dstCap, const u8* src,
                                     size_t srcSize) {
    blockstate* const prev = &c->bs[c->prevIdx];
    blockstate* const next = &c->bs[c->prevIdx ^ 1];
    mstate* const ms = &c->ms;
    size_t cSize;
    if (srcSize < 1 + 1 + 3 + 1 + 1) { /* MIN_CBLOCK_SIZE + ZSTD_blockHeaderSize + 1 + 1 */
        cSize = 0; /* ZSTDbss_noCompress */
        goto out;
    }
    c->ss.nseq = 0;
    c->ss.nlits = 0;
    c->ss.longLengthType = 0;
    {
        u32 const curr = (u32)(src - ms->base);
        if (curr > ms->nextToUpdate + 384) {
            u32 const d = curr - ms->nextToUpdate - 384;
            ms->nextToUpdate = curr - (192 < d ? 192 : d);
        }
    }
    {
        size_t lastLLSize;
        for (int i = 0; i < REP_NUM; ++i) next->rep[i] = prev->rep[i];
        switch (ms->cp.strategy) {
        case STRAT_LAZY2: lastLLSize = compressBlock_lazy(ms, &c->ss, next->rep, src, srcSize, SEARCH_ROW); break;
        case STRAT_BTLAZY2: lastLLSize = compressBlock_lazy(ms, &c->ss, next->rep, src, srcSize, SEARCH_BT); break;
        default: lastLLSize = compressBlock_btopt(ms, &c->ss, next->rep, &prev->ent, src, srcSize); break;
        }
        memcpy(c->ss.lits + c->ss.nlits, src + srcSize - lastLLSize, lastLLSize);
        c->ss.nlits += (u32)lastLLSize;
        if (c->dump) dump_block(c, prev->rep, lastLLSize);
    }
    cSize = entropy_compress_seqstore(&c->ss, &prev->ent, &next->ent, ms->cp.strategy, dst, dstCap,
                                      srcSize);
    if (is_err(cSize)) return cSize;
    if (!c->isFirstBlock && cSize < 25 /* rleMaxLength */ && isRLE(src, srcSize)) {
        cSize = 1;
        dst[0] = src[0];
    }
out:
    if (cSize > 1) c->prevIdx ^= 1; /* ZSTD_blockState_confirmRepcodesAndEntropyTables */
    if (c->bs[c->prevIdx].ent.of_repeat == REP_VALID) c->bs[c->prevIdx].ent.of_repeat = REP_CHECK;
    return cSize;
}

static void free_ctx(cctx* c) {
    free(c->ms.hashTable);
    free(c->ms.chainTable);
    free(c->ms.tagTable);
    free(c->ms.hashTable3);
    opt_free(&c->ms);
    free(c->ss.seqs);
    free(c->ss.lits);
    free(c->ss.llCode);
    free(c->ss.ofCode);
    free(c->ss.mlCode);
}

static size_t compress_frame(cctx* c, const u8* src, size_t n, u8* dst, size_t cap) {
    u8* op = dst;
    u8* const oend = dst + cap;
    const u8* ip = src;
    size_t remaining = n;
    int64_t savings = 0;
    sqo_cparams const cp = sqo_get_cparams(n);

    /* ---- ZSTD_writeFrameHeader: magic, FHD (single segment, content size), FCS ---- */
    if (cap < 18) return 0;
    {
        u32 const fcsCode = (n >= 256) + (n >= 65536 + 256) + (n >= 0xFFFFFFFFU);
        op[0] = 0x28;
        op[1] = 0xB5;
        op[2] = 0x2F;
        op[3] = 0xFD;
        op[4] = (u8)((fcsCode << 6) + (1 << 5));
        op += 5;
        switch (fcsCode) {
        case 0: *op++ = (u8)n; break;
        case 1: {
            u16 v = (u16)(n - 256);
            op[0] = (u8)v;
            op[1] = (u8)(v >> 8);
            op += 2;
            break;
        }
        case 2: {
            u32 v = (u32)n;
            memcpy(op, &v, 4);
            op += 4;
            break;
        }
        default: {
            u64 v = (u64)n;
            memcpy(op, &v, 8);
            op += 8;
            break;
        }
        }
    }
    /* ---- fresh context (ZSTD_resetCCtx_internal / ZSTD_reset_matchState) ---- */
    memset(&c->ms, 0, sizeof(c->ms));
    c->ms.cp = cp;
    c->ms.base = src - WINDOW_START_INDEX;
    c->ms.nextToUpdate = WINDOW_START_INDEX;
    if (cp.strategy == STRAT_LAZY2) {
        size_t const hSize = (size_t)1 << cp.hashLog;
        u32 rowLog = cp.searchLog < 4 ? 4 : (cp.searchLog > 6 ? 6 : cp.searchLog);
        c->ms.hashTable = (u32*)calloc(hSize, sizeof(u32));
        c->ms.tagTable = (u8*)calloc(hSize, 1);
        c->ms.rowHashLog = cp.hashLog - rowLog;
    } else {
        c->ms.hashTable = (u32*)calloc((size_t)1 << cp.hashLog, sizeof(u32));
        c->ms.chainTable = (u32*)calloc((size_t)1 << cp.chainLog, sizeof(u32));
        if (cp.minMatch == 3) {
            c->ms.hashLog3 = cp.windowLog < 17 ? cp.windowLog : 17;
            c->ms.hashTable3 = (u32*)calloc((size_t)1 << c->ms.hashLog3, sizeof(u32));
        }
    }
    if (cp.strategy == STRAT_BTOPT) opt_alloc(&c->ms);
    {
        size_t const maxNbSeq = BLOCKSIZE_MAX / (cp.minMatch == 3 ? 3 : 4);
        c->ss.seqs = (seqdef*)malloc((maxNbSeq + 1) * sizeof(seqdef));
        c->ss.lits = (u8*)malloc(BLOCKSIZE_MAX + 32);
        c->ss.llCode = (u8*)malloc(maxNbSeq + 1);
        c->ss.ofCode = (u8*)malloc(maxNbSeq + 1);
        c->ss.mlCode = (u8*)malloc(maxNbSeq + 1);
    }
    for (int k = 0; k < 2; k++) {
        entropy_reset(&c->bs[k].ent);
        c->bs[k].rep[0] = 1;
        c->bs[k].rep[1] = 4;
        c->bs[k].rep[2] = 8;
    }
    c->prevIdx = 0;
    c->isFirstBlock = 1;

    /* ---- ZSTD_compress_frameChunk ---- */
    if (n == 0) { /* ZSTD_writeLastEmptyBlock */
        op[0] = 1;
        op[1] = 0;
        op[2] = 0;
        op += 3;
    }
    while (remaining) {
        size_t const blockSize = optimalBlockSize(ip, remaining, BLOCKSIZE_MAX, cp.strategy, savings);
        u32 const lastBlock = (blockSize == remaining);
        size_t cSize;
        if ((size_t)(oend - op) < 3 + 2 + 1) goto fail;
        if (c->ms.nextToUpdate < WINDOW_START_INDEX) c->ms.nextToUpdate = WINDOW_START_INDEX;
        cSize = compressBlock_internal(c, op + 3, (size_t)(oend - op) - 3, ip, blockSize);
        if (is_err(cSize)) goto fail;
        if (cSize == 0) { /* ZSTD_noCompressBlock */
            u32 const h = lastBlock + (0u << 1) + (u32)(blockSize << 3);
            if (blockSize + 3 > (size_t)(oend - op)) goto fail;
            op[0] = (u8)h;
            op[1] = (u8)(h >> 8);
            op[2] = (u8)(h >> 16);
            memcpy(op + 3, ip, blockSize);
            cSize = 3 + blockSize;
        } else {
            u32 const h = (cSize == 1) ? lastBlock + (1u << 1) + (u32)(blockSize << 3)
                                       : lastBlock + (2u << 1) + (u32)(cSize << 3);
            op[0] = (u8)h;
            op[1] = (u8)(h >> 8);
            op[2] = (u8)(h >> 16);
            cSize += 3;
        }
        savings += (int64_t)blockSize - (int64_t)cSize;
        ip += blockSize;
        remaining -= blockSize;
        op += cSize;
        c->isFirstBlock = 0;
    }
    free_ctx(c);
    return (size_t)(op - dst);
fail:
    free_ctx(c);
    return 0;
}

size_t sqo_compress_chunk(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    cctx* c = (cctx*)calloc(1, sizeof(cctx));
    size_t r;
    if (!c) return 0;
    r = compress_frame(c, src, n, dst, cap);
    free(c);
    return r;
}

size_t sqo_generate_sequences(const uint8_t* src, size_t n, sqo_sequence* out, size_t cap) {
    cctx* c = (cctx*)calloc(1, sizeof(cctx));
    size_t const bound = sqo_compress_bound(n) + 64;
    u8* tmp = (u8*)malloc(bound);
    size_t r;
    c->dump = out;
    c->dumpCap = cap;
    compress_frame(c, src, n, tmp, bound);
    r = c->dumpN;
    free(tmp);
    free(c);
    return r;
}
