/*
 * zstd12_entropy.c -- CPU ORACLE (test infrastructure, see sq_oracle.h).
 * Restates the entropy stage of libzstd 1.5.7 (zstd-sys 2.0.15+zstd.1.5.7, Cargo.lock:1637-1640)
 * as reached from /root/reference/src/util/chunk.rs:89-90  `compress(chunk, COMPRESSION_LEVEL)`.
 * Comments name the libzstd routine each function restates.
 */
#include "zstd12_entropy.h"
#include <string.h>

/* ------------------------------------------------------------------ bit writer (BIT_CStream_t) */
typedef struct {
    u64 acc;
    u32 nbits;
    u8* p;
    u8* start;
    u8* end;
    int overflow;
} bitw;

static void bw_init(bitw* b, u8* dst, size_t cap) {
    b->acc = 0;
    b->nbits = 0;
    b->p = b->start = dst;
    b->end = dst + cap;
    b->overflow = 0;
}
static void bw_flush(bitw* b) {
    while (b->nbits >= 8) {
        if (b->p < b->end)
            *b->p++ = (u8)b->acc;
        else
            b->overflow = 1;
        b->acc >>= 8;
        b->nbits -= 8;
    }
}
static void bw_add(bitw* b, u64 v, u32 n) { /* BIT_addBits: low n bits of v, LSB first */
    if (n == 0) return;
    v &= (n >= 64) ? ~0ULL : ((1ULL << n) - 1);
    b->acc |= v << b->nbits;
    b->nbits += n;
    bw_flush(b); /* n <= 31 always here, acc never overflows */
}
static size_t bw_close(bitw* b) { /* BIT_closeCStream: end mark bit then byte-align */
    bw_add(b, 1, 1);
    if (b->nbits) {
        if (b->p < b->end)
            *b->p++ = (u8)b->acc;
        else
            b->overflow = 1;
        b->nbits = 0;
    }
    if (b->overflow) return 0;
    return (size_t)(b->p - b->start);
}

/* ------------------------------------------------------------------ histogram (HIST_count) */
static u32 hist_count(u32* count, u32* maxSymPtr, const u8* src, size_t n) {
    u32 maxSym = *maxSymPtr, largest = 0;
    memset(count, 0, (maxSym + 1) * sizeof(u32));
    if (n == 0) {
        *maxSymPtr = 0;
        return 0;
    }
    for (size_t i = 0; i < n; i++) count[src[i]]++;
    while (!count[maxSym]) maxSym--;
    *maxSymPtr = maxSym;
    for (u32 s = 0; s <= maxSym; s++)
        if (count[s] > largest) largest = count[s];
    return largest;
}

/* ------------------------------------------------------------------ FSE */
static u32 fse_minTableLog(size_t srcSize, u32 maxSym) { /* FSE_minTableLog */
    u32 minBitsSrc = highbit32((u32)srcSize) + 1;
    u32 minBitsSymbols = highbit32(maxSym) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}
static u32 fse_optimalTableLog_internal(u32 maxTableLog, size_t srcSize, u32 maxSym, u32 minus) {
    u32 maxBitsSrc = highbit32((u32)(srcSize - 1)) - minus;
    u32 tableLog = maxTableLog;
    u32 minBits = fse_minTableLog(srcSize, maxSym);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}
static u32 fse_optimalTableLog(u32 maxTableLog, size_t srcSize, u32 maxSym) {
    return fse_optimalTableLog_internal(maxTableLog, srcSize, maxSym, 2);
}

static size_t fse_normalizeM2(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSym,
                              s16 lowProbCount) { /* FSE_normalizeM2 */
    const s16 NOT_YET = -2;
    u32 s, distributed = 0, ToDistribute;
    u32 const lowThreshold = (u32)(total >> tableLog);
    u32 lowOne = (u32)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) {
            norm[s] = 0;
            continue;
        }
        if (count[s] <= lowThreshold) {
            norm[s] = lowProbCount;
            distributed++;
            total -= count[s];
            continue;
        }
        if (count[s] <= lowOne) {
            norm[s] = 1;
            distributed++;
            total -= count[s];
            continue;
        }
        norm[s] = NOT_YET;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (u32)((total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSym; s++) {
            if ((norm[s] == NOT_YET) && (count[s] <= lowOne)) {
                norm[s] = 1;
                distributed++;
                total -= count[s];
            }
        }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++)
            if (count[s] > maxC) {
                maxV = s;
                maxC = count[s];
            }
        norm[maxV] += (s16)ToDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSym + 1))
            if (norm[s] > 0) {
                ToDistribute--;
                norm[s]++;
            }
        return 0;
    }
    {
        u64 const vStepLog = 62 - tableLog;
        u64 const mid = (1ULL << (vStepLog - 1)) - 1;
        u64 const rStep = ((((u64)1 << vStepLog) * ToDistribute) + mid) / (u32)total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == NOT_YET) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog);
                u32 const sEnd = (u32)(end >> vStepLog);
                u32 const weight = sEnd - sStart;
                if (weight < 1) return ERR;
                norm[s] = (s16)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

static size_t fse_normalizeCount(s16* norm, u32 tableLog, const u32* count, size_t total,
                                 u32 maxSym, u32 useLowProbCount) { /* FSE_normalizeCount */
    static const u32 rtbTable[] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    if (tableLog == 0) tableLog = 11;
    if (tableLog < 5) return ERR;
    if (tableLog > 12) return ERR;
    if (tableLog < fse_minTableLog(total, maxSym)) return ERR;
    {
        s16 const lowProbCount = useLowProbCount ? -1 : 1;
        u64 const scale = 62 - tableLog;
        u64 const step = ((u64)1 << 62) / (u32)total;
        u64 const vStep = 1ULL << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        u32 s, largest = 0;
        s16 largestP = 0;
        u32 lowThreshold = (u32)(total >> tableLog);
        for (s = 0; s <= maxSym; s++) {
            if (count[s] == total) return 0; /* rle special case */
            if (count[s] == 0) {
                norm[s] = 0;
                continue;
            }
            if (count[s] <= lowThreshold) {
                norm[s] = lowProbCount;
                stillToDistribute--;
            } else {
                s16 proba = (s16)((count[s] * step) >> scale);
                if (proba < 8) {
                    u64 restToBeat = vStep * rtbTable[proba];
                    proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
                }
                if (proba > largestP) {
                    largestP = proba;
                    largest = s;
                }
                norm[s] = proba;
                stillToDistribute -= proba;
            }
        }
        if (-stillToDistribute >= (norm[largest] >> 1)) {
            size_t e = fse_normalizeM2(norm, tableLog, count, total, maxSym, lowProbCount);
            if (is_err(e)) return e;
        } else
            norm[largest] += (s16)stillToDistribute;
    }
    return tableLog;
}

static size_t fse_writeNCount(u8* out0, size_t cap, const s16* norm, u32 maxSym, u32 tableLog) {
    /* FSE_writeNCount_generic */
    u8* out = out0;
    u8* const oend = out0 + cap;
    int nbBits;
    const int tableSize = 1 << tableLog;
    int remaining, threshold;
    u32 bitStream = 0;
    int bitCount = 0;
    u32 symbol = 0;
    u32 const alphabetSize = maxSym + 1;
    int previousIs0 = 0;

    bitStream += (tableLog - 5) << bitCount;
    bitCount += 4;
    remaining = tableSize + 1;
    threshold = tableSize;
    nbBits = (int)tableLog + 1;

    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            u32 start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFU << bitCount;
                if (out > oend - 2) return ERR;
                out[0] = (u8)bitStream;
                out[1] = (u8)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
            }
            while (symbol >= start + 3) {
                start += 3;
                bitStream += 3U << bitCount;
                bitCount += 2;
            }
            bitStream += (symbol - start) << bitCount;
            bitCount += 2;
            if (bitCount > 16) {
                if (out > oend - 2) return ERR;
                out[0] = (u8)bitStream;
                out[1] = (u8)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
                bitCount -= 16;
            }
        }
        {
            int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return ERR;
            while (remaining < threshold) {
                nbBits--;
                threshold >>= 1;
            }
        }
        if (bitCount > 16) {
            if (out > oend - 2) return ERR;
            out[0] = (u8)bitStream;
            out[1] = (u8)(bitStream >> 8);
            out += 2;
            bitStream >>= 16;
            bitCount -= 16;
        }
    }
    if (remaining != 1) return ERR;
    if (out > oend - 2) return ERR;
    out[0] = (u8)bitStream;
    out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - out0);
}

static size_t fse_buildCTable(fse_ct* ct, const s16* norm, u32 maxSym, u32 tableLog) {
    /* FSE_buildCTable_wksp */
    u32 const tableSize = 1u << tableLog;
    u32 const tableMask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 const maxSV1 = maxSym + 1;
    u16 cumul[FSE_MAXSYM + 3];
    u8 tableSymbol[1 << FSE_MAXLOG];
    u32 highThreshold = tableSize - 1;
    if (tableLog > FSE_MAXLOG || maxSym > FSE_MAXSYM) return ERR;
    ct->tableLog = (u16)tableLog;
    ct->maxSym = (u16)maxSym;
    {
        u32 u;
        cumul[0] = 0;
        for (u = 1; u <= maxSV1; u++) {
            if (norm[u - 1] == -1) {
                cumul[u] = cumul[u - 1] + 1;
                tableSymbol[highThreshold--] = (u8)(u - 1);
            } else {
                cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
            }
        }
        cumul[maxSV1] = (u16)(tableSize + 1);
    }
    {
        u32 position = 0, symbol;
        for (symbol = 0; symbol < maxSV1; symbol++) {
            int n, freq = norm[symbol];
            for (n = 0; n < freq; n++) {
                tableSymbol[position] = (u8)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    {
        u32 u;
        for (u = 0; u < tableSize; u++) {
            u8 s = tableSymbol[u];
            ct->stateTable[cumul[s]++] = (u16)(tableSize + u);
        }
    }
    {
        u32 total = 0, s;
        for (s = 0; s <= maxSym; s++) {
            switch (norm[s]) {
            case 0:
                ct->tt[s].deltaNbBits = ((tableLog + 1) << 16) - (1u << tableLog);
                ct->tt[s].deltaFindState = 0;
                break;
            case -1:
            case 1:
                ct->tt[s].deltaNbBits = (tableLog << 16) - (1u << tableLog);
                ct->tt[s].deltaFindState = (int)(total - 1);
                total++;
                break;
            default: {
                u32 const maxBitsOut = tableLog - highbit32((u32)norm[s] - 1);
                u32 const minStatePlus = (u32)norm[s] << maxBitsOut;
                ct->tt[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                ct->tt[s].deltaFindState = (int)(total - (u32)norm[s]);
                total += (u32)norm[s];
            }
            }
        }
    }
    return 0;
}

static void fse_buildCTable_rle(fse_ct* ct, u8 sym) { /* FSE_buildCTable_rle */
    ct->tableLog = 0;
    ct->maxSym = sym;
    ct->stateTable[0] = 0;
    ct->stateTable[1] = 0;
    ct->tt[sym].deltaNbBits = 0;
    ct->tt[sym].deltaFindState = 0;
}

typedef struct {
    u32 value;
    const fse_ct* ct;
} fse_cstate;

static void fse_initCState2(fse_cstate* st, const fse_ct* ct, u32 symbol) {
    fse_tt const tt = ct->tt[symbol];
    u32 nbBitsOut = (u32)((tt.deltaNbBits + (1 << 15)) >> 16);
    st->ct = ct;
    st->value = (nbBitsOut << 16) - tt.deltaNbBits;
    st->value = ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static void fse_encodeSymbol(bitw* b, fse_cstate* st, u32 symbol) {
    fse_tt const tt = st->ct->tt[symbol];
    u32 const nbBitsOut = (u32)((st->value + tt.deltaNbBits) >> 16);
    bw_add(b, st->value, nbBitsOut);
    st->value = st->ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static void fse_flushCState(bitw* b, const fse_cstate* st) {
    bw_add(b, st->value, st->ct->tableLog);
}

/* FSE_compress_usingCTable: 2 interleaved states; even index -> state1, odd index -> state2 */
static size_t fse_compress_usingCTable(u8* dst, size_t cap, const u8* src, size_t n,
                                       const fse_ct* ct) {
    bitw b;
    fse_cstate s1, s2;
    size_t i;
    if (n <= 2) return 0;
    if (cap < 9) return 0; /* BIT_initCStream needs sizeof(container)+1 */
    bw_init(&b, dst, cap);
    i = n;
    if (n & 1) {
        fse_initCState2(&s1, ct, src[--i]);
        fse_initCState2(&s2, ct, src[--i]);
        fse_encodeSymbol(&b, &s1, src[--i]);
    } else {
        fse_initCState2(&s2, ct, src[--i]);
        fse_initCState2(&s1, ct, src[--i]);
    }
    while (i > 0) {
        fse_encodeSymbol(&b, &s2, src[--i]);
        fse_encodeSymbol(&b, &s1, src[--i]);
    }
    fse_flushCState(&b, &s2);
    fse_flushCState(&b, &s1);
    return bw_close(&b);
}

/* ------------------------------------------------------------------ Huffman */
typedef struct {
    u32 count;
    u16 parent;
    u8 byte;
    u8 nbBits;
} nodeElt;

#define HUF_TABLELOG_MAX 12
#define RANK_POSITION_TABLE_SIZE 192
#define RANK_POSITION_LOG_BUCKETS_BEGIN ((RANK_POSITION_TABLE_SIZE - 1) - 32 - 1) /* 158 */
#define RANK_POSITION_DISTINCT_COUNT_CUTOFF \
    (RANK_POSITION_LOG_BUCKETS_BEGIN + 7 /* highbit32(158) */)

static u32 huf_getIndex(u32 count) {
    return (count < RANK_POSITION_DISTINCT_COUNT_CUTOFF)
               ? count
               : highbit32(count) + RANK_POSITION_LOG_BUCKETS_BEGIN;
}
static void huf_swap(nodeElt* a, nodeElt* b) {
    nodeElt t = *a;
    *a = *b;
    *b = t;
}
static void huf_insertionSort(nodeElt* arr, int low, int high) {
    int i, size = high - low + 1;
    arr += low;
    for (i = 1; i < size; ++i) {
        nodeElt key = arr[i];
        int j = i - 1;
        while (j >= 0 && arr[j].count < key.count) {
            arr[j + 1] = arr[j];
            j--;
        }
        arr[j + 1] = key;
    }
}
static int huf_partition(nodeElt* arr, int low, int high) {
    u32 pivot = arr[high].count;
    int i = low - 1, j = low;
    for (; j < high; j++) {
        if (arr[j].count > pivot) {
            i++;
            huf_swap(&arr[i], &arr[j]);
        }
    }
    huf_swap(&arr[i + 1], &arr[high]);
    return i + 1;
}
static void huf_quickSort(nodeElt* arr, int low, int high) {
    if (high - low < 8) {
        huf_insertionSort(arr, low, high);
        return;
    }
    while (low < high) {
        int idx = huf_partition(arr, low, high);
        if (idx - low < high - idx) {
            huf_quickSort(arr, low, idx - 1);
            low = idx + 1;
        } else {
            huf_quickSort(arr, idx + 1, high);
            high = idx - 1;
        }
    }
}
static void huf_sort(nodeElt* huffNode, const u32* count, u32 maxSym) { /* HUF_sort */
    struct {
        u16 base, curr;
    } rankPosition[RANK_POSITION_TABLE_SIZE];
    u32 n, maxSV1 = maxSym + 1;
    memset(rankPosition, 0, sizeof(rankPosition));
    for (n = 0; n < maxSV1; ++n) rankPosition[huf_getIndex(count[n])].base++;
    for (n = RANK_POSITION_TABLE_SIZE - 1; n > 0; --n) {
        rankPosition[n - 1].base += rankPosition[n].base;
        rankPosition[n - 1].curr = rankPosition[n - 1].base;
    }
    for (n = 0; n < maxSV1; ++n) {
        u32 c = count[n];
        u32 r = huf_getIndex(c) + 1;
        u32 pos = rankPosition[r].curr++;
        huffNode[pos].count = c;
        huffNode[pos].byte = (u8)n;
    }
    for (n = RANK_POSITION_DISTINCT_COUNT_CUTOFF; n < RANK_POSITION_TABLE_SIZE - 1; ++n) {
        int bucketSize = rankPosition[n].curr - rankPosition[n].base;
        u32 bucketStartIdx = rankPosition[n].base;
        if (bucketSize > 1) huf_quickSort(huffNode + bucketStartIdx, 0, bucketSize - 1);
    }
}

#define STARTNODE 256
static int huf_buildTree(nodeElt* huffNode, u32 maxSym) { /* HUF_buildTree */
    nodeElt* const huffNode0 = huffNode - 1;
    int nonNullRank, lowS, lowN, nodeNb = STARTNODE, n, nodeRoot;
    nonNullRank = (int)maxSym;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank;
    nodeRoot = nodeNb + lowS - 1;
    lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (u16)nodeNb;
    nodeNb++;
    lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = (u32)(1U << 30);
    huffNode0[0].count = (u32)(1U << 31);
    while (nodeNb <= nodeRoot) {
        int const n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        int const n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= STARTNODE; n--)
        huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++)
        huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    return nonNullRank;
}

static u32 huf_setMaxHeight(nodeElt* huffNode, u32 lastNonNull, u32 targetNbBits) {
    /* HUF_setMaxHeight */
    u32 const largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    {
        int totalCost = 0;
        const u32 baseCost = 1 << (largestBits - targetNbBits);
        int n = (int)lastNonNull;
        while (huffNode[n].nbBits > targetNbBits) {
            totalCost += baseCost - (1 << (largestBits - huffNode[n].nbBits));
            huffNode[n].nbBits = (u8)targetNbBits;
            n--;
        }
        while (huffNode[n].nbBits == targetNbBits) --n;
        totalCost >>= (largestBits - targetNbBits);
        {
            u32 const noSymbol = 0xF0F0F0F0;
            u32 rankLast[HUF_TABLELOG_MAX + 2];
            memset(rankLast, 0xF0, sizeof(rankLast));
            {
                u32 currentNbBits = targetNbBits;
                int pos;
                for (pos = n; pos >= 0; pos--) {
                    if (huffNode[pos].nbBits >= currentNbBits) continue;
                    currentNbBits = huffNode[pos].nbBits;
                    rankLast[targetNbBits - currentNbBits] = (u32)pos;
                }
            }
            while (totalCost > 0) {
                u32 nBitsToDecrease = highbit32((u32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    u32 const highPos = rankLast[nBitsToDecrease];
                    u32 const lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    {
                        u32 const highTotal = huffNode[highPos].count;
                        u32 const lowTotal = 2 * huffNode[lowPos].count;
                        if (highTotal <= lowTotal) break;
                    }
                }
                while ((nBitsToDecrease <= HUF_TABLELOG_MAX) &&
                       (rankLast[nBitsToDecrease] == noSymbol))
                    nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                huffNode[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease - 1] == noSymbol)
                    rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                if (rankLast[nBitsToDecrease] == 0)
                    rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (huffNode[rankLast[nBitsToDecrease]].nbBits != targetNbBits - nBitsToDecrease)
                        rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (huffNode[n].nbBits == targetNbBits) n--;
                    huffNode[n + 1].nbBits--;
                    rankLast[1] = (u32)(n + 1);
                    totalCost++;
                    continue;
                }
                huffNode[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return targetNbBits;
}

/* HUF_buildCTable_wksp : returns maxNbBits */
static size_t huf_buildCTable(huf_ct* ct, const u32* count, u32 maxSym, u32 maxNbBits) {
    nodeElt table[2 * 256 + 2];
    nodeElt* const huffNode = table + 1;
    int nonNullRank, n;
    memset(table, 0, sizeof(table));
    huf_sort(huffNode, count, maxSym);
    nonNullRank = huf_buildTree(huffNode, maxSym);
    maxNbBits = huf_setMaxHeight(huffNode, (u32)nonNullRank, maxNbBits);
    if (maxNbBits > HUF_TABLELOG_MAX) return ERR;
    { /* HUF_buildCTableFromTree */
        u16 nbPerRank[HUF_TABLELOG_MAX + 1] = {0};
        u16 valPerRank[HUF_TABLELOG_MAX + 1] = {0};
        int const alphabetSize = (int)(maxSym + 1);
        for (n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
        {
            u16 min = 0;
            for (n = (int)maxNbBits; n > 0; n--) {
                valPerRank[n] = min;
                min += nbPerRank[n];
                min >>= 1;
            }
        }
        memset(ct->nbBits, 0, sizeof(ct->nbBits));
        memset(ct->val, 0, sizeof(ct->val));
        for (n = 0; n < alphabetSize; n++) ct->nbBits[huffNode[n].byte] = huffNode[n].nbBits;
        for (n = 0; n < alphabetSize; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
        ct->maxSym = maxSym;
        ct->tableLog = maxNbBits;
    }
    return maxNbBits;
}

/* HUF_compressWeights */
static size_t huf_compressWeights(u8* dst, size_t cap, const u8* weights, size_t wtSize) {
    u8* op = dst;
    u8* const oend = dst + cap;
    u32 maxSym = HUF_TABLELOG_MAX;
    u32 tableLog = 6;
    u32 count[HUF_TABLELOG_MAX + 1];
    s16 norm[HUF_TABLELOG_MAX + 1];
    fse_ct ct;
    if (wtSize <= 1) return 0;
    {
        u32 const maxCount = hist_count(count, &maxSym, weights, wtSize);
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = fse_optimalTableLog(tableLog, wtSize, maxSym);
    {
        size_t e = fse_normalizeCount(norm, tableLog, count, wtSize, maxSym, 0);
        if (is_err(e)) return e;
    }
    {
        size_t hSize = fse_writeNCount(op, (size_t)(oend - op), norm, maxSym, tableLog);
        if (is_err(hSize)) return hSize;
        op += hSize;
    }
    if (is_err(fse_buildCTable(&ct, norm, maxSym, tableLog))) return ERR;
    {
        size_t cSize = fse_compress_usingCTable(op, (size_t)(oend - op), weights, wtSize, &ct);
        if (is_err(cSize)) return cSize;
        if (cSize == 0) return 0;
        op += cSize;
    }
    return (size_t)(op - dst);
}

/* HUF_writeCTable_wksp */
static size_t huf_writeCTable(u8* dst, size_t cap, const huf_ct* ct, u32 maxSym, u32 huffLog) {
    u8 bitsToWeight[HUF_TABLELOG_MAX + 1];
    u8 huffWeight[256];
    u8* op = dst;
    u32 n;
    bitsToWeight[0] = 0;
    for (n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (u8)(huffLog + 1 - n);
    for (n = 0; n < maxSym; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    if (cap < 1) return ERR;
    {
        size_t hSize = huf_compressWeights(op + 1, cap - 1, huffWeight, maxSym);
        if (is_err(hSize)) return hSize;
        if ((hSize > 1) & (hSize < maxSym / 2)) {
            op[0] = (u8)hSize;
            return hSize + 1;
        }
    }
    if (maxSym > (256 - 128)) return ERR;
    if (((maxSym + 1) / 2) + 1 > cap) return ERR;
    op[0] = (u8)(128 + (maxSym - 1));
    huffWeight[maxSym] = 0;
    for (n = 0; n < maxSym; n += 2) op[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}

/* HUF_compress1X_usingCTable_internal: symbols last->first, LSB-first, closing 1 bit */
static size_t huf_compress1X(u8* dst, size_t cap, const u8* src, size_t n, const huf_ct* ct) {
    bitw b;
    if (cap < 8) return 0;
    bw_init(&b, dst, cap);
    for (size_t i = n; i > 0; i--) {
        u8 s = src[i - 1];
        bw_add(&b, ct->val[s], ct->nbBits[s]);
    }
    return bw_close(&b);
}
/* HUF_compress4X_usingCTable_internal */
static size_t huf_compress4X(u8* dst, size_t cap, const u8* src, size_t n, const huf_ct* ct) {
    size_t const segmentSize = (n + 3) / 4;
    const u8* ip = src;
    const u8* const iend = src + n;
    u8* op = dst;
    u8* const oend = dst + cap;
    if (cap < 6 + 1 + 1 + 1 + 8) return 0;
    if (n < 12) return 0;
    op += 6;
    for (int k = 0; k < 3; k++) {
        size_t cSize = huf_compress1X(op, (size_t)(oend - op), ip, segmentSize, ct);
        if (cSize == 0 || cSize > 65535) return 0;
        dst[2 * k] = (u8)cSize;
        dst[2 * k + 1] = (u8)(cSize >> 8);
        op += cSize;
        ip += segmentSize;
    }
    {
        size_t cSize = huf_compress1X(op, (size_t)(oend - op), ip, (size_t)(iend - ip), ct);
        if (cSize == 0 || cSize > 65535) return 0;
        op += cSize;
    }
    return (size_t)(op - dst);
}
/* HUF_compressCTable_internal */
static size_t huf_compressCTable(u8* ostart, u8* op, u8* oend, const u8* src, size_t n,
                                 int singleStream, const huf_ct* ct) {
    size_t const cSize = singleStream ? huf_compress1X(op, (size_t)(oend - op), src, n, ct)
                                      : huf_compress4X(op, (size_t)(oend - op), src, n, ct);
    if (cSize == 0) return 0;
    op += cSize;
    if ((size_t)(op - ostart) >= n - 1) return 0;
    return (size_t)(op - ostart);
}
static int huf_validateCTable(const huf_ct* ct, const u32* count, u32 maxSym) {
    int bad = 0;
    if (ct->maxSym < maxSym) return 0;
    for (u32 s = 0; s <= maxSym; ++s) bad |= (count[s] != 0) & (ct->nbBits[s] == 0);
    return !bad;
}
static size_t huf_estimateCompressedSize(const huf_ct* ct, const u32* count, u32 maxSym) {
    size_t nbBits = 0;
    for (u32 s = 0; s <= maxSym; ++s) nbBits += (size_t)ct->nbBits[s] * count[s];
    return nbBits >> 3;
}

/* HUF_compress_internal (flags: no preferRepeat (strategy>=lazy), no optimalDepth (<btultra)) */
static size_t huf_compress(u8* dst, size_t dstSize, const u8* src, size_t srcSize, int singleStream,
                           huf_ct* oldHufTable, u32* repeat, int suspectUncompressible) {
    u8* const ostart = dst;
    u8* const oend = dst + dstSize;
    u8* op = ostart;
    u32 count[256];
    u32 maxSym = 255;
    u32 huffLog = 11;
    huf_ct newTable;
    if (!srcSize) return 0;
    if (!dstSize) return 0;
    if (srcSize > 128 * 1024) return ERR;

    if (suspectUncompressible && srcSize >= 4096 * 10) {
        size_t largestTotal = 0;
        {
            u32 m = 255;
            largestTotal += hist_count(count, &m, src, 4096);
        }
        {
            u32 m = 255;
            largestTotal += hist_count(count, &m, src + srcSize - 4096, 4096);
        }
        if (largestTotal <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {
        u32 largest = hist_count(count, &maxSym, src, srcSize);
        if (largest == srcSize) {
            *ostart = src[0];
            return 1;
        }
        if (largest <= (srcSize >> 7) + 4) return 0;
    }
    if (*repeat == REP_CHECK && !huf_validateCTable(oldHufTable, count, maxSym)) *repeat = REP_NONE;

    huffLog = fse_optimalTableLog_internal(huffLog, srcSize, maxSym, 1); /* HUF_optimalTableLog */
    {
        size_t maxBits = huf_buildCTable(&newTable, count, maxSym, huffLog);
        if (is_err(maxBits)) return maxBits;
        huffLog = (u32)maxBits;
    }
    {
        size_t hSize = huf_writeCTable(op, dstSize, &newTable, maxSym, huffLog);
        if (is_err(hSize)) return hSize;
        if (*repeat != REP_NONE) {
            size_t const oldSize = huf_estimateCompressedSize(oldHufTable, count, maxSym);
            size_t const newSize = huf_estimateCompressedSize(&newTable, count, maxSym);
            if (oldSize <= hSize + newSize || hSize + 12 >= srcSize)
                return huf_compressCTable(ostart, op, oend, src, srcSize, singleStream, oldHufTable);
        }
        if (hSize + 12ul >= srcSize) return 0;
        op += hSize;
        *repeat = REP_NONE;
        *oldHufTable = newTable;
    }
    return huf_compressCTable(ostart, op, oend, src, srcSize, singleStream, &newTable);
}

/* ------------------------------------------------------------------ literals section */
static size_t min_gain(size_t srcSize, u32 strategy) { /* ZSTD_minGain */
    u32 const minlog = (strategy >= 8) ? strategy - 1 : 6;
    return (srcSize >> minlog) + 2;
}
static size_t noCompressLiterals(u8* dst, size_t cap, const u8* src, size_t n) {
    u32 const flSize = 1 + (n > 31) + (n > 4095);
    if (n + flSize > cap) return ERR;
    switch (flSize) {
    case 1: dst[0] = (u8)(0 + (n << 3)); break;
    case 2: {
        u16 v = (u16)(0 + (1 << 2) + (n << 4));
        dst[0] = (u8)v;
        dst[1] = (u8)(v >> 8);
        break;
    }
    default: {
        u32 v = (u32)(0 + (3 << 2) + (n << 4));
        dst[0] = (u8)v;
        dst[1] = (u8)(v >> 8);
        dst[2] = (u8)(v >> 16);
        /* MEM_writeLE32 writes a 4th byte (zero) that the memcpy below overwrites */
        break;
    }
    }
    memcpy(dst + flSize, src, n);
    return n + flSize;
}
static size_t rleLiterals(u8* dst, size_t cap, const u8* src, size_t n) {
    u32 const flSize = 1 + (n > 31) + (n > 4095);
    (void)cap;
    switch (flSize) {
    case 1: dst[0] = (u8)(1 + (n << 3)); break;
    case 2: {
        u16 v = (u16)(1 + (1 << 2) + (n << 4));
        dst[0] = (u8)v;
        dst[1] = (u8)(v >> 8);
        break;
    }
    default: {
        u32 v = (u32)(1 + (3 << 2) + (n << 4));
        dst[0] = (u8)v;
        dst[1] = (u8)(v >> 8);
        dst[2] = (u8)(v >> 16);
        break;
    }
    }
    dst[flSize] = src[0];
    return flSize + 1;
}
static int allBytesIdentical(const u8* src, size_t n) {
    for (size_t i = 1; i < n; i++)
        if (src[i] != src[0]) return 0;
    return 1;
}

/* ZSTD_compressLiterals */
static size_t compressLiterals(u8* dst, size_t dstCap, const u8* src, size_t srcSize,
                               const entropy_t* prev, entropy_t* next, u32 strategy,
                               int suspectUncompressible) {
    size_t const lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    u8* const ostart = dst;
    u32 singleStream = srcSize < 256;
    u32 hType = SET_COMPRESSED;
    size_t cLitSize;

    next->huf = prev->huf;
    next->huf_repeat = prev->huf_repeat;
    {
        int const shift = (9 - (int)strategy) < 3 ? (9 - (int)strategy) : 3;
        size_t const mintc = (prev->huf_repeat == REP_VALID) ? 6 : (size_t)8 << shift;
        if (srcSize < mintc) return noCompressLiterals(dst, dstCap, src, srcSize);
    }
    if (dstCap < lhSize + 1) return ERR;
    {
        u32 repeat = prev->huf_repeat;
        if (repeat == REP_VALID && lhSize == 3) singleStream = 1;
        cLitSize = huf_compress(ostart + lhSize, dstCap - lhSize, src, srcSize, (int)singleStream,
                                &next->huf, &repeat, suspectUncompressible);
        if (repeat != REP_NONE) hType = SET_REPEAT;
    }
    {
        size_t const minGain = min_gain(srcSize, strategy);
        if ((cLitSize == 0) || (cLitSize >= srcSize - minGain) || is_err(cLitSize)) {
            next->huf = prev->huf;
            next->huf_repeat = prev->huf_repeat;
            return noCompressLiterals(dst, dstCap, src, srcSize);
        }
    }
    if (cLitSize == 1) {
        if ((srcSize >= 8) || allBytesIdentical(src, srcSize)) {
            next->huf = prev->huf;
            next->huf_repeat = prev->huf_repeat;
            return rleLiterals(dst, dstCap, src, srcSize);
        }
    }
    if (hType == SET_COMPRESSED) next->huf_repeat = REP_CHECK;
    switch (lhSize) {
    case 3: {
        u32 const lhc =
            hType + ((u32)(!singleStream) << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 14);
        ostart[0] = (u8)lhc;
        ostart[1] = (u8)(lhc >> 8);
        ostart[2] = (u8)(lhc >> 16);
        break;
    }
    case 4: {
        u32 const lhc = hType + (2 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 18);
        ostart[0] = (u8)lhc;
        ostart[1] = (u8)(lhc >> 8);
        ostart[2] = (u8)(lhc >> 16);
        ostart[3] = (u8)(lhc >> 24);
        break;
    }
    default: {
        u32 const lhc = hType + (3 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 22);
        ostart[0] = (u8)lhc;
        ostart[1] = (u8)(lhc >> 8);
        ostart[2] = (u8)(lhc >> 16);
        ostart[3] = (u8)(lhc >> 24);
        ostart[4] = (u8)(cLitSize >> 10);
        break;
    }
    }
    return lhSize + cLitSize;
}

/* ------------------------------------------------------------------ sequences section */
const u8 zs_LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  1,  1,
                               1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
const u8 zs_ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,
                               0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,  1,  1,  1,
                               2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const s16 LL_defaultNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                       2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const s16 ML_defaultNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static const s16 OF_defaultNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                       1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define DefaultMaxOff 28
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8

u32 zs_LLcode(u32 litLength) { /* ZSTD_LLcode */
    static const u8 LL_Code[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                                   16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                                   22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23,
                                   24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
    return (litLength > 63) ? highbit32(litLength) + 19 : LL_Code[litLength];
}
u32 zs_MLcode(u32 mlBase) { /* ZSTD_MLcode */
    static const u8 ML_Code[128] = {
        0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
        22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36,
        37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40,
        40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41,
        41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42,
        42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
    return (mlBase > 127) ? highbit32(mlBase) + 36 : ML_Code[mlBase];
}

static u32 kInvProbLog256[256];
__attribute__((constructor)) static void init_invprob(void) {  /* filled at load time: callers run on thread pools */
    /* kInverseProbabilityLog256[x] = floor(-log2(x/256) * 256), [0] = 0.  Computed with exact
     * integer arithmetic: find largest v with 2^(v/256) * x <= 256, i.e. compare
     * x^256 * 2^v <= 256^256 via repeated squaring in fixed point.  To stay simple and exact we
     * use long double log2l and verify against the anchor values below at first use. */
    static int done = 0;
    if (done) return;
    kInvProbLog256[0] = 0;
    for (int x = 1; x < 256; x++) {
        long double v = -__builtin_log2l((long double)x / 256.0L) * 256.0L;
        kInvProbLog256[x] = (u32)v; /* floor */
    }
    done = 1;
}

static size_t entropyCost(const u32* count, u32 max, size_t total) { /* ZSTD_entropyCost */
    u32 cost = 0;
    for (u32 s = 0; s <= max; ++s) {
        u32 norm = (u32)((256 * count[s]) / total);
        if (count[s] != 0 && norm == 0) norm = 1;
        cost += count[s] * kInvProbLog256[norm];
    }
    return cost >> 8;
}
static size_t crossEntropyCost(const s16* norm, u32 accuracyLog, const u32* count, u32 max) {
    u32 const shift = 8 - accuracyLog;
    size_t cost = 0;
    for (u32 s = 0; s <= max; ++s) {
        u32 const normAcc = (norm[s] != -1) ? (u32)norm[s] : 1;
        u32 const norm256 = normAcc << shift;
        cost += count[s] * kInvProbLog256[norm256];
    }
    return cost >> 8;
}
static u32 fse_bitCost(const fse_ct* ct, u32 tableLog, u32 sym, u32 accuracyLog) { /* FSE_bitCost */
    u32 const minNbBits = ct->tt[sym].deltaNbBits >> 16;
    u32 const threshold = (minNbBits + 1) << 16;
    u32 const tableSize = 1u << tableLog;
    u32 const deltaFromThreshold = threshold - (ct->tt[sym].deltaNbBits + tableSize);
    u32 const normalizedDeltaFromThreshold = (deltaFromThreshold << accuracyLog) >> tableLog;
    u32 const bitMultiplier = 1u << accuracyLog;
    return (minNbBits + 1) * bitMultiplier - normalizedDeltaFromThreshold;
}
static size_t fseBitCost(const fse_ct* ct, const u32* count, u32 max) { /* ZSTD_fseBitCost */
    u32 const kAccuracyLog = 8;
    size_t cost = 0;
    if (ct->maxSym < max) return ERR;
    for (u32 s = 0; s <= max; ++s) {
        u32 const tableLog = ct->tableLog;
        u32 const badCost = (tableLog + 1) << kAccuracyLog;
        u32 bitCost;
        if (count[s] == 0) continue;
        bitCost = fse_bitCost(ct, tableLog, s, kAccuracyLog);
        if (bitCost >= badCost) return ERR;
        cost += (size_t)count[s] * bitCost;
    }
    return cost >> kAccuracyLog;
}
static size_t NCountCost(const u32* count, u32 max, size_t nbSeq, u32 FSELog) {
    u8 wksp[512];
    s16 norm[MaxML + 1];
    const u32 tableLog = fse_optimalTableLog(FSELog, nbSeq, max);
    size_t e = fse_normalizeCount(norm, tableLog, count, nbSeq, max, nbSeq >= 2048);
    if (is_err(e)) return e;
    return fse_writeNCount(wksp, sizeof(wksp), norm, max, tableLog);
}

/* ZSTD_selectEncodingType, strategy >= ZSTD_lazy branch */
static u32 selectEncodingType(u32* repeatMode, const u32* count, u32 max, size_t mostFrequent,
                              size_t nbSeq, u32 FSELog, const fse_ct* prevCTable,
                              const s16* defaultNorm, u32 defaultNormLog, int isDefaultAllowed) {
    if (mostFrequent == nbSeq) {
        *repeatMode = REP_NONE;
        if (isDefaultAllowed && nbSeq <= 2) return SET_BASIC;
        return SET_RLE;
    }
    {
        size_t const basicCost =
            isDefaultAllowed ? crossEntropyCost(defaultNorm, defaultNormLog, count, max) : ERR;
        size_t const repeatCost = *repeatMode != REP_NONE ? fseBitCost(prevCTable, count, max) : ERR;
        size_t const ncc = NCountCost(count, max, nbSeq, FSELog);
        size_t const compressedCost = (ncc << 3) + entropyCost(count, max, nbSeq);
        if (basicCost <= repeatCost && basicCost <= compressedCost) {
            *repeatMode = REP_NONE;
            return SET_BASIC;
        }
        if (repeatCost <= compressedCost) return SET_REPEAT;
    }
    *repeatMode = REP_CHECK;
    return SET_COMPRESSED;
}

/* ZSTD_buildCTable */
static size_t buildCTable(u8* dst, size_t dstCap, fse_ct* nextCTable, u32 FSELog, u32 type,
                          u32* count, u32 max, const u8* codeTable, size_t nbSeq,
                          const s16* defaultNorm, u32 defaultNormLog, u32 defaultMax,
                          const fse_ct* prevCTable) {
    switch (type) {
    case SET_RLE:
        fse_buildCTable_rle(nextCTable, (u8)max);
        if (dstCap == 0) return ERR;
        *dst = codeTable[0];
        return 1;
    case SET_REPEAT: *nextCTable = *prevCTable; return 0;
    case SET_BASIC:
        if (is_err(fse_buildCTable(nextCTable, defaultNorm, defaultMax, defaultNormLog))) return ERR;
        return 0;
    case SET_COMPRESSED: {
        s16 norm[MaxML + 1];
        size_t nbSeq_1 = nbSeq;
        const u32 tableLog = fse_optimalTableLog(FSELog, nbSeq, max);
        if (count[codeTable[nbSeq - 1]] > 1) {
            count[codeTable[nbSeq - 1]]--;
            nbSeq_1--;
        }
        {
            size_t e = fse_normalizeCount(norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048);
            if (is_err(e)) return e;
        }
        {
            size_t const NCountSize = fse_writeNCount(dst, dstCap, norm, max, tableLog);
            if (is_err(NCountSize)) return NCountSize;
            if (is_err(fse_buildCTable(nextCTable, norm, max, tableLog))) return ERR;
            return NCountSize;
        }
    }
    default: return ERR;
    }
}

/* ZSTD_encodeSequences (64-bit path) */
static size_t encodeSequences(u8* dst, size_t dstCap, const fse_ct* ctML, const u8* mlCodeTable,
                              const fse_ct* ctOF, const u8* ofCodeTable, const fse_ct* ctLL,
                              const u8* llCodeTable, const seqdef* sequences, size_t nbSeq) {
    bitw b;
    fse_cstate stML, stOF, stLL;
    if (dstCap < 9) return ERR;
    bw_init(&b, dst, dstCap);
    fse_initCState2(&stML, ctML, mlCodeTable[nbSeq - 1]);
    fse_initCState2(&stOF, ctOF, ofCodeTable[nbSeq - 1]);
    fse_initCState2(&stLL, ctLL, llCodeTable[nbSeq - 1]);
    bw_add(&b, sequences[nbSeq - 1].litLength, zs_LL_bits[llCodeTable[nbSeq - 1]]);
    bw_add(&b, sequences[nbSeq - 1].mlBase, zs_ML_bits[mlCodeTable[nbSeq - 1]]);
    bw_add(&b, sequences[nbSeq - 1].offBase, ofCodeTable[nbSeq - 1]);
    {
        size_t n;
        for (n = nbSeq - 2; n < nbSeq; n--) {
            u8 const llCode = llCodeTable[n];
            u8 const ofCode = ofCodeTable[n];
            u8 const mlCode = mlCodeTable[n];
            fse_encodeSymbol(&b, &stOF, ofCode);
            fse_encodeSymbol(&b, &stML, mlCode);
            fse_encodeSymbol(&b, &stLL, llCode);
            bw_add(&b, sequences[n].litLength, zs_LL_bits[llCode]);
            bw_add(&b, sequences[n].mlBase, zs_ML_bits[mlCode]);
            bw_add(&b, sequences[n].offBase, ofCode);
        }
    }
    fse_flushCState(&b, &stML);
    fse_flushCState(&b, &stOF);
    fse_flushCState(&b, &stLL);
    {
        size_t const streamSize = bw_close(&b);
        if (streamSize == 0) return ERR;
        return streamSize;
    }
}

static u32 hist_codes(u32* count, u32* maxPtr, const u8* codes, size_t n) {
    return hist_count(count, maxPtr, codes, n);
}

void entropy_reset(entropy_t* e) {
    memset(e, 0, sizeof(*e));
    e->huf_repeat = REP_NONE;
    e->ll_repeat = e->of_repeat = e->ml_repeat = REP_NONE;
}

/* ZSTD_entropyCompressSeqStore_internal + the compressibility wrapper */
size_t entropy_compress_seqstore(const seqstore* ss, const entropy_t* prev, entropy_t* next,
                                 u32 strategy, u8* dst, size_t dstCap, size_t blockSize) {
    u8* const ostart = dst;
    u8* const oend = dst + dstCap;
    u8* op = ostart;
    size_t const nbSeq = ss->nseq;
    size_t lastCountSize = 0;
    u32 count[MaxML + 1];
    init_invprob();

    { /* literals */
        size_t const litSize = ss->nlits;
        int const suspectUncompressible = (nbSeq == 0) || (litSize / nbSeq >= 20);
        size_t const cSize = compressLiterals(op, dstCap, ss->lits, litSize, prev, next, strategy,
                                              suspectUncompressible);
        if (is_err(cSize)) return (blockSize <= dstCap) ? 0 : ERR;
        op += cSize;
    }
    if ((oend - op) < 3 + 1) return 0;
    if (nbSeq < 128) {
        *op++ = (u8)nbSeq;
    } else if (nbSeq < 0x7F00) {
        op[0] = (u8)((nbSeq >> 8) + 0x80);
        op[1] = (u8)nbSeq;
        op += 2;
    } else {
        op[0] = 0xFF;
        op[1] = (u8)(nbSeq - 0x7F00);
        op[2] = (u8)((nbSeq - 0x7F00) >> 8);
        op += 3;
    }
    if (nbSeq == 0) {
        next->ll = prev->ll;
        next->of = prev->of;
        next->ml = prev->ml;
        next->ll_repeat = prev->ll_repeat;
        next->of_repeat = prev->of_repeat;
        next->ml_repeat = prev->ml_repeat;
        goto check;
    }
    {
        u8* const seqHead = op++;
        u32 LLtype, Offtype, MLtype;
        /* ZSTD_seqToCodes */
        for (size_t u = 0; u < nbSeq; u++) {
            ss->llCode[u] = (u8)zs_LLcode(ss->seqs[u].litLength);
            ss->ofCode[u] = (u8)highbit32(ss->seqs[u].offBase);
            ss->mlCode[u] = (u8)zs_MLcode(ss->seqs[u].mlBase);
        }
        if (ss->longLengthType == 1) ss->llCode[ss->longLengthPos] = MaxLL;
        if (ss->longLengthType == 2) ss->mlCode[ss->longLengthPos] = MaxML;

        { /* LL */
            u32 max = MaxLL;
            size_t const mostFrequent = hist_codes(count, &max, ss->llCode, nbSeq);
            next->ll_repeat = prev->ll_repeat;
            LLtype = selectEncodingType(&next->ll_repeat, count, max, mostFrequent, nbSeq, LLFSELog,
                                        &prev->ll, LL_defaultNorm, 6, 1);
            {
                size_t const countSize =
                    buildCTable(op, (size_t)(oend - op), &next->ll, LLFSELog, LLtype, count, max,
                                ss->llCode, nbSeq, LL_defaultNorm, 6, MaxLL, &prev->ll);
                if (is_err(countSize)) return (blockSize <= dstCap) ? 0 : ERR;
                if (LLtype == SET_COMPRESSED) lastCountSize = countSize;
                op += countSize;
            }
        }
        { /* OF */
            u32 max = MaxOff;
            size_t const mostFrequent = hist_codes(count, &max, ss->ofCode, nbSeq);
            int const defaultAllowed = (max <= DefaultMaxOff);
            next->of_repeat = prev->of_repeat;
            Offtype = selectEncodingType(&next->of_repeat, count, max, mostFrequent, nbSeq,
                                         OffFSELog, &prev->of, OF_defaultNorm, 5, defaultAllowed);
            {
                size_t const countSize =
                    buildCTable(op, (size_t)(oend - op), &next->of, OffFSELog, Offtype, count, max,
                                ss->ofCode, nbSeq, OF_defaultNorm, 5, DefaultMaxOff, &prev->of);
                if (is_err(countSize)) return (blockSize <= dstCap) ? 0 : ERR;
                if (Offtype == SET_COMPRESSED) lastCountSize = countSize;
                op += countSize;
            }
        }
        { /* ML */
            u32 max = MaxML;
            size_t const mostFrequent = hist_codes(count, &max, ss->mlCode, nbSeq);
            next->ml_repeat = prev->ml_repeat;
            MLtype = selectEncodingType(&next->ml_repeat, count, max, mostFrequent, nbSeq, MLFSELog,
                                        &prev->ml, ML_defaultNorm, 6, 1);
            {
                size_t const countSize =
                    buildCTable(op, (size_t)(oend - op), &next->ml, MLFSELog, MLtype, count, max,
                                ss->mlCode, nbSeq, ML_defaultNorm, 6, MaxML, &prev->ml);
                if (is_err(countSize)) return (blockSize <= dstCap) ? 0 : ERR;
                if (MLtype == SET_COMPRESSED) lastCountSize = countSize;
                op += countSize;
            }
        }
        *seqHead = (u8)((LLtype << 6) + (Offtype << 4) + (MLtype << 2));
        {
            size_t const bitstreamSize =
                encodeSequences(op, (size_t)(oend - op), &next->ml, ss->mlCode, &next->of,
                                ss->ofCode, &next->ll, ss->llCode, ss->seqs, nbSeq);
            if (is_err(bitstreamSize)) return (blockSize <= dstCap) ? 0 : ERR;
            op += bitstreamSize;
            if (lastCountSize && (lastCountSize + bitstreamSize) < 4) return 0;
        }
    }
check: {
    size_t const cSize = (size_t)(op - ostart);
    size_t const maxCSize = blockSize - min_gain(blockSize, strategy);
    if (cSize >= maxCSize) return 0;
    return cSize;
}
}
