// sq_zstd_entropy.h -- entropy stage of the level-12 zstd block compressor on gfx950:
// literals section (Huffman) and sequences section (FSE), bit-exact with libzstd 1.5.7
// (zstd-sys 2.0.15+zstd.1.5.7, Cargo.lock:1637-1640), the library that
// /root/reference/src/util/chunk.rs:89-90 reaches through zstd::bulk::compress(chunk, 12).
//
// Where the work runs (one wave = one chunk):
//   * byte / code histograms: 64 lanes, LDS atomics (wv_hist_bytes)
//   * table construction (Huffman tree, FSE normalisation, NCount, encoding-type cost model):
//     a few hundred scalar steps on <= 256 symbols -> serial region on lane 0, tables in LDS
//   * bit-stream emission: wave-parallel -- code lengths -> prefix scan over the lanes -> every lane ORs its bits
//     into an LDS window (wvb_tile); only the three FSE state recurrences are serial (lanes 0..2)
// All tables live in LDS (g_lds); the previous block's tables stay resident for the
// repeat / treeless modes.  Comments name the libzstd routine each function corresponds to.
#pragma once
#include "sq_wave.h"

namespace sqz {

#define SQZ_ERR 0xFFFFFFFFu

enum { REP_NONE = 0, REP_CHECK = 1, REP_VALID = 2 };
enum { SET_BASIC = 0, SET_RLE = 1, SET_COMPRESSED = 2, SET_REPEAT = 3 };
enum { MaxLL = 35, MaxML = 52, MaxOff = 31, DefaultMaxOff = 28, LLFSELog = 9, MLFSELog = 9, OffFSELog = 8 };

struct FseCT { // FSE_CTable
    u16 tableLog, maxSym;
    u16 stateTable[512];
    s16 dfs[MaxML + 1]; // deltaFindState
    u32 dnb[MaxML + 1]; // deltaNbBits
};
struct HufCT { // HUF_CElt[] + header
    u8 nbBits[256];
    u16 val[256];
    u32 maxSym;
};
struct Entropy { // ZSTD_entropyCTables_t
    HufCT huf;
    u32 huf_repeat;
    FseCT ll, of, ml;
    u32 ll_repeat, of_repeat, ml_repeat;
};
struct NodeElt {
    u32 count;
    u16 parent;
    u8 byte, nbBits;
};
struct SeqDef {
    u32 offBase;
    u16 litLength, mlBase;
};

// Everything a wave keeps in LDS (14.4 KB since round 4: the two sets of entropy tables -- previous and next block's, 9.8 KB --
// moved to the wave's workspace in HBM, where the encoders read them through the L1 / L2; what they borrowed out to the match
// finder as scratch is `scr`): 11 waves per CU fit in the 160 KB of a CU.
struct EntropyScratch { // live only inside the entropy stage of a block
    u32 count[256];
    NodeElt huffNodeTbl[2 * 256 + 2];
    struct {
        u16 base, curr;
    } rankPosition[192];
    HufCT newHuf;
    u8 huffWeight[256];
    s16 norm[64];
    u16 cumul[64];
    u8 tableSymbol[512];
};
#define SQZ_BLOOM_WORDS 2048 // 65,536-bit Bloom filter of "live" hash buckets (match-finding phase)
// candidate list of a row-hash search batch (sq_zstd_rowbatch.h): (owner lane, slot) descriptors, then the
// candidates' positions and match lengths, balanced over the lanes
#define SQZ_RB_PAIRS 256u
struct RbLds {
    u16 desc[SQZ_RB_PAIRS];
    u32 cand[SQZ_RB_PAIRS];
    u32 rowId[64], pbase[64]; // per owner lane: its row, the start of its pairs
    u32 stop[64], best[64], flag[64]; // per owner lane: first pair that ends the scan; (len << 8 | 255 - order) maximum; "long candidate" flag
    u32 prev[128];                    // per owner lane: the 8 bytes in front of its position
    u8 be[SQZ_RB_PAIRS];              // per pair: equal bytes in front of position and candidate (0..8)
};
struct LdsState {
    Entropy* entp; // the two sets of entropy tables (in the wave's workspace: WsLayout::ent)
    u32 scr[1024]; // match finder: survivors of an insertion super-batch (3 x 256 words) / the certificate's overflow list
    union {
        EntropyScratch es;
        u32 bloom[SQZ_BLOOM_WORDS];
        RbLds rb; // row-hash search batches (match-finding phase of the lazy2 strategy)
    } u;
    alignas(16) u8 grp[2048]; // wv_group scratch; the 64 x 32 source bytes of a row-hash search batch
    u32 ret[8];   // results handed from a serial region to the whole wave
    u32* trace;   // 56 debug words in the wave's workspace (read by the host watchdog)
#ifdef SQ_PROF
    unsigned long long prof[32]; // per-phase 100 MHz tick sums (profiling build only)
    unsigned long long prof0[32], profLong[32]; // prof[] when the current chunk began; phase sums of the wave's longest chunk
#endif
};
SQ_LDS LdsState g_lds;

// phase timers (profiling build only): s_memrealtime ticks (100 MHz) summed per phase
#if defined(SQ_PROF) && SQ_ON_DEVICE
#define PROF_T0(v) unsigned long long v = __builtin_amdgcn_s_memrealtime()
#define PROF_ADD(slot, v)                                                              \
    do {                                                                               \
        unsigned long long e_ = __builtin_amdgcn_s_memrealtime();                      \
        if (wv_lane() == 0) g_lds.prof[slot] += e_ - (v);                              \
    } while (0)
#define PROF_CNT(slot, n)                                                              \
    do {                                                                               \
        if (wv_lane() == 0) g_lds.prof[slot] += (n);                                   \
    } while (0)
#else
#define PROF_T0(v) do { } while (0)
#define PROF_ADD(slot, v) do { } while (0)
#define PROF_CNT(slot, n) do { } while (0)
#endif

// progress markers for bring-up: slot 0 = stage, others free-form
#if SQ_ON_DEVICE
#define SQ_TRACE(slot, val)                                        \
    do {                                                           \
        if (wv_lane() == 0) g_lds.trace[slot] = (u32)(val);        \
    } while (0)
#else
#define SQ_TRACE(slot, val) do { g_lds.trace[slot] = (u32)(val); } while (0)
#endif

// ---------------------------------------------------------------- constant tables
#if SQ_ON_DEVICE
#define SQZ_CONST __device__ const
#else
#define SQZ_CONST static const
#endif
SQZ_CONST u8 LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  1,  1,
                            1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
SQZ_CONST u8 ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,
                            0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,  1,  1,  1,
                            2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
SQZ_CONST s16 LL_defaultNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                    2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
SQZ_CONST s16 ML_defaultNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
SQZ_CONST s16 OF_defaultNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
SQZ_CONST u8 LL_Code[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,
                            16, 16, 17, 17, 18, 18, 19, 19, 20, 20, 20, 20, 21, 21, 21, 21,
                            22, 22, 22, 22, 22, 22, 22, 22, 23, 23, 23, 23, 23, 23, 23, 23,
                            24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24};
SQZ_CONST u8 ML_Code[128] = {
    0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
    22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 36,
    37, 37, 37, 37, 38, 38, 38, 38, 38, 38, 38, 38, 39, 39, 39, 39, 39, 39, 39, 39, 40, 40,
    40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 41, 41, 41, 41, 41, 41, 41, 41,
    41, 41, 41, 41, 41, 41, 41, 41, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42,
    42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42, 42};
// kInverseProbabilityLog256[x] = floor(-log2(x/256)*256)
SQZ_CONST u16 kInvProbLog256[256] = {
    0,    2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
    1024, 1001, 980,  960,  941,  923,  906,  889,  874,  859,  844,  830,  817,  804,  791,  779,
    768,  756,  745,  734,  724,  714,  704,  694,  685,  676,  667,  658,  650,  642,  633,  626,
    618,  610,  603,  595,  588,  581,  574,  567,  561,  554,  548,  542,  535,  529,  523,  517,
    512,  506,  500,  495,  489,  484,  478,  473,  468,  463,  458,  453,  448,  443,  438,  434,
    429,  424,  420,  415,  411,  407,  402,  398,  394,  390,  386,  382,  377,  373,  370,  366,
    362,  358,  354,  350,  347,  343,  339,  336,  332,  329,  325,  322,  318,  315,  311,  308,
    305,  302,  298,  295,  292,  289,  286,  282,  279,  276,  273,  270,  267,  264,  261,  258,
    256,  253,  250,  247,  244,  241,  239,  236,  233,  230,  228,  225,  222,  220,  217,  215,
    212,  209,  207,  204,  202,  199,  197,  194,  192,  190,  187,  185,  182,  180,  178,  175,
    173,  171,  168,  166,  164,  162,  159,  157,  155,  153,  151,  149,  146,  144,  142,  140,
    138,  136,  134,  132,  130,  128,  126,  123,  121,  119,  117,  115,  114,  112,  110,  108,
    106,  104,  102,  100,  98,   96,   94,   93,   91,   89,   87,   85,   83,   82,   80,   78,
    76,   74,   73,   71,   69,   67,   66,   64,   62,   61,   59,   57,   55,   54,   52,   50,
    49,   47,   46,   44,   42,   41,   39,   37,   36,   34,   33,   31,   30,   28,   26,   25,
    23,   22,   20,   19,   17,   16,   14,   13,   11,   10,   8,    7,    5,    4,    2,    1};

SQ_DEV u32 LLcode(u32 litLength) { return (litLength > 63) ? sq_highbit32(litLength) + 19 : LL_Code[litLength]; }
SQ_DEV u32 MLcode(u32 mlBase) { return (mlBase > 127) ? sq_highbit32(mlBase) + 36 : ML_Code[mlBase]; }
SQ_DEV bool is_err(u32 c) { return c == SQZ_ERR; }

// ---------------------------------------------------------------- serial bit writer (lane 0)
struct BitW {
    u64 acc;
    u32 nbits;
    u8* p;
    u8* start;
    u8* end;
    u32 overflow;
};
SQ_DEV void bw_init(BitW& b, u8* dst, u32 cap) {
    b.acc = 0;
    b.nbits = 0;
    b.p = b.start = dst;
    b.end = dst + cap;
    b.overflow = 0;
}
SQ_DEV void bw_add(BitW& b, u64 v, u32 n) {
    if (n == 0) return;
    v &= ((1ULL << n) - 1);
    b.acc |= v << b.nbits;
    b.nbits += n;
    while (b.nbits >= 8) {
        if (b.p < b.end)
            *b.p++ = (u8)b.acc;
        else
            b.overflow = 1;
        b.acc >>= 8;
        b.nbits -= 8;
    }
}
SQ_DEV u32 bw_close(BitW& b) {
    bw_add(b, 1, 1);
    if (b.nbits) {
        if (b.p < b.end)
            *b.p++ = (u8)b.acc;
        else
            b.overflow = 1;
        b.nbits = 0;
    }
    if (b.overflow) return 0;
    return (u32)(b.p - b.start);
}

// ================================================================ FSE (serial)
SQ_DEV u32 fse_minTableLog(u32 srcSize, u32 maxSym) {
    u32 a = sq_highbit32(srcSize) + 1, b = sq_highbit32(maxSym) + 2;
    return a < b ? a : b;
}
SQ_DEV u32 fse_optimalTableLog_internal(u32 maxTableLog, u32 srcSize, u32 maxSym, u32 minus) {
    u32 maxBitsSrc = sq_highbit32(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    u32 minBits = fse_minTableLog(srcSize, maxSym);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}
SQ_DEV u32 fse_optimalTableLog(u32 maxTableLog, u32 srcSize, u32 maxSym) {
    return fse_optimalTableLog_internal(maxTableLog, srcSize, maxSym, 2);
}

SQ_DEVFN u32 fse_normalizeM2(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSym, s16 lowProbCount) {
    const s16 NOT_YET = -2;
    u32 s, distributed = 0, ToDistribute;
    u32 const lowThreshold = total >> tableLog;
    u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
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
        lowOne = (u32)(((u64)total * 3) / (ToDistribute * 2));
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
        u64 const rStep = ((((u64)1 << vStepLog) * ToDistribute) + mid) / total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == NOT_YET) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog);
                u32 const sEnd = (u32)(end >> vStepLog);
                u32 const weight = sEnd - sStart;
                if (weight < 1) return SQZ_ERR;
                norm[s] = (s16)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

SQ_DEVFN u32 fse_normalizeCount(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSym,
                                u32 useLowProbCount) {
    const u32 rtbTable[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    if (tableLog == 0) tableLog = 11;
    if (tableLog < 5 || tableLog > 12) return SQZ_ERR;
    if (tableLog < fse_minTableLog(total, maxSym)) return SQZ_ERR;
    {
        s16 const lowProbCount = useLowProbCount ? -1 : 1;
        u64 const scale = 62 - tableLog;
        u64 const step = ((u64)1 << 62) / total;
        u64 const vStep = 1ULL << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        u32 s, largest = 0;
        s16 largestP = 0;
        u32 lowThreshold = total >> tableLog;
        for (s = 0; s <= maxSym; s++) {
            if (count[s] == total) return 0;
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
            u32 e = fse_normalizeM2(norm, tableLog, count, total, maxSym, lowProbCount);
            if (is_err(e)) return e;
        } else
            norm[largest] += (s16)stillToDistribute;
    }
    return tableLog;
}

SQ_DEVFN u32 fse_writeNCount(u8* out0, u32 cap, const s16* norm, u32 maxSym, u32 tableLog) {
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
    if (cap < 4) return SQZ_ERR;
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
                if (out > oend - 2) return SQZ_ERR;
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
                if (out > oend - 2) return SQZ_ERR;
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
            if (remaining < 1) return SQZ_ERR;
            while (remaining < threshold) {
                nbBits--;
                threshold >>= 1;
            }
        }
        if (bitCount > 16) {
            if (out > oend - 2) return SQZ_ERR;
            out[0] = (u8)bitStream;
            out[1] = (u8)(bitStream >> 8);
            out += 2;
            bitStream >>= 16;
            bitCount -= 16;
        }
    }
    if (remaining != 1) return SQZ_ERR;
    if (out > oend - 2) return SQZ_ERR;
    out[0] = (u8)bitStream;
    out[1] = (u8)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (u32)(out - out0);
}

// FSE_buildCTable_wksp; uses g_lds.u.es.cumul / g_lds.u.es.tableSymbol
SQ_DEVFN u32 fse_buildCTable(FseCT* ct, const s16* norm, u32 maxSym, u32 tableLog) {
    u32 const tableSize = 1u << tableLog;
    u32 const tableMask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u32 const maxSV1 = maxSym + 1;
    u16* const cumul = g_lds.u.es.cumul;
    u8* const tableSymbol = g_lds.u.es.tableSymbol;
    u32 highThreshold = tableSize - 1;
    if (tableLog > 9 || maxSym > MaxML) return SQZ_ERR;
    ct->tableLog = (u16)tableLog;
    ct->maxSym = (u16)maxSym;
    cumul[0] = 0;
    for (u32 u = 1; u <= maxSV1; u++) {
        if (norm[u - 1] == -1) {
            cumul[u] = cumul[u - 1] + 1;
            tableSymbol[highThreshold--] = (u8)(u - 1);
        } else {
            cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
        }
    }
    cumul[maxSV1] = (u16)(tableSize + 1);
    {
        u32 position = 0;
        for (u32 symbol = 0; symbol < maxSV1; symbol++) {
            int const freq = norm[symbol];
            for (int n = 0; n < freq; n++) {
                tableSymbol[position] = (u8)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u32 u = 0; u < tableSize; u++) {
        u8 s = tableSymbol[u];
        ct->stateTable[cumul[s]++] = (u16)(tableSize + u);
    }
    {
        u32 total = 0;
        for (u32 s = 0; s <= maxSym; s++) {
            int const n = norm[s];
            if (n == 0) {
                ct->dnb[s] = ((tableLog + 1) << 16) - (1u << tableLog);
                ct->dfs[s] = 0;
            } else if (n == -1 || n == 1) {
                ct->dnb[s] = (tableLog << 16) - (1u << tableLog);
                ct->dfs[s] = (s16)((int)total - 1);
                total++;
            } else {
                u32 const maxBitsOut = tableLog - sq_highbit32((u32)n - 1);
                u32 const minStatePlus = (u32)n << maxBitsOut;
                ct->dnb[s] = (maxBitsOut << 16) - minStatePlus;
                ct->dfs[s] = (s16)((int)total - n);
                total += (u32)n;
            }
        }
    }
    return 0;
}
SQ_DEV void fse_buildCTable_rle(FseCT* ct, u32 sym) {
    ct->tableLog = 0;
    ct->maxSym = (u16)sym;
    ct->stateTable[0] = 0;
    ct->stateTable[1] = 0;
    ct->dnb[sym] = 0;
    ct->dfs[sym] = 0;
}

struct FseState {
    u32 value;
    const FseCT* ct;
};
SQ_DEV void fse_initCState2(FseState& st, const FseCT* ct, u32 symbol) {
    u32 const dnb = ct->dnb[symbol];
    u32 const nbBitsOut = (dnb + (1u << 15)) >> 16;
    st.ct = ct;
    st.value = (nbBitsOut << 16) - dnb;
    st.value = ct->stateTable[(st.value >> nbBitsOut) + ct->dfs[symbol]];
}
SQ_DEV void fse_encodeSymbol(BitW& b, FseState& st, u32 symbol) {
    u32 const nbBitsOut = (st.value + st.ct->dnb[symbol]) >> 16;
    bw_add(b, st.value, nbBitsOut);
    st.value = st.ct->stateTable[(st.value >> nbBitsOut) + st.ct->dfs[symbol]];
}
SQ_DEV void fse_flushCState(BitW& b, const FseState& st) { bw_add(b, st.value, st.ct->tableLog); }

// FSE_compress_usingCTable (Huffman weights): even index -> state1, odd index -> state2
SQ_DEVFN u32 fse_compress_usingCTable(u8* dst, u32 cap, const u8* src, u32 n, const FseCT* ct) {
    BitW b;
    FseState s1, s2;
    u32 i = n;
    if (n <= 2) return 0;
    if (cap < 9) return 0;
    bw_init(b, dst, cap);
    if (n & 1) {
        fse_initCState2(s1, ct, src[--i]);
        fse_initCState2(s2, ct, src[--i]);
        fse_encodeSymbol(b, s1, src[--i]);
    } else {
        fse_initCState2(s2, ct, src[--i]);
        fse_initCState2(s1, ct, src[--i]);
    }
    while (i > 0) {
        fse_encodeSymbol(b, s2, src[--i]);
        fse_encodeSymbol(b, s1, src[--i]);
    }
    fse_flushCState(b, s2);
    fse_flushCState(b, s1);
    return bw_close(b);
}

// ================================================================ Huffman (serial, tables in LDS)
#define SQZ_HUF_TABLELOG_MAX 12
#define SQZ_RANK_LOG_BUCKETS_BEGIN 158
#define SQZ_RANK_DISTINCT_CUTOFF 165

SQ_DEV u32 huf_getIndex(u32 count) {
    return (count < SQZ_RANK_DISTINCT_CUTOFF) ? count : sq_highbit32(count) + SQZ_RANK_LOG_BUCKETS_BEGIN;
}
SQ_DEV void huf_swap(NodeElt* a, NodeElt* b) {
    NodeElt t = *a;
    *a = *b;
    *b = t;
}
SQ_DEVFN void huf_insertionSort(NodeElt* arr, int low, int high) {
    int const size = high - low + 1;
    arr += low;
    for (int i = 1; i < size; ++i) {
        NodeElt key = arr[i];
        int j = i - 1;
        while (j >= 0 && arr[j].count < key.count) {
            arr[j + 1] = arr[j];
            j--;
        }
        arr[j + 1] = key;
    }
}
SQ_DEVFN int huf_partition(NodeElt* arr, int low, int high) {
    u32 const pivot = arr[high].count;
    int i = low - 1;
    for (int j = low; j < high; j++) {
        if (arr[j].count > pivot) {
            i++;
            huf_swap(&arr[i], &arr[j]);
        }
    }
    huf_swap(&arr[i + 1], &arr[high]);
    return i + 1;
}
// HUF_simpleQuickSort without recursion.  An explicit stack holds the ranges the original
// passes to its recursive calls (smaller side; entry check: < 8 apart -> insertion sort); the
// larger side is iterated without that check, exactly like the original's while loop.  The two
// sides are disjoint, so processing order does not change the result.
SQ_DEVFN void huf_quickSort(NodeElt* arr, int low0, int high0) {
    int stackLo[40], stackHi[40];
    int sp = 0;
    if (high0 - low0 < 8) {
        huf_insertionSort(arr, low0, high0);
        return;
    }
    stackLo[0] = low0;
    stackHi[0] = high0;
    sp = 1;
    while (sp > 0) {
        sp--;
        int low = stackLo[sp], high = stackHi[sp];
        while (low < high) {
            int const idx = huf_partition(arr, low, high);
            int cl, ch; // the "recursive call" range
            if (idx - low < high - idx) {
                cl = low;
                ch = idx - 1;
                low = idx + 1;
            } else {
                cl = idx + 1;
                ch = high;
                high = idx - 1;
            }
            if (ch - cl < 8) {
                huf_insertionSort(arr, cl, ch);
            } else {
                stackLo[sp] = cl;
                stackHi[sp] = ch;
                sp++;
            }
        }
    }
}
SQ_DEVFN void huf_sort(NodeElt* huffNode, const u32* count, u32 maxSym) {
    u32 const maxSV1 = maxSym + 1;
    for (u32 n = 0; n < 192; n++) {
        g_lds.u.es.rankPosition[n].base = 0;
        g_lds.u.es.rankPosition[n].curr = 0;
    }
    for (u32 n = 0; n < maxSV1; ++n) g_lds.u.es.rankPosition[huf_getIndex(count[n])].base++;
    for (u32 n = 191; n > 0; --n) {
        g_lds.u.es.rankPosition[n - 1].base += g_lds.u.es.rankPosition[n].base;
        g_lds.u.es.rankPosition[n - 1].curr = g_lds.u.es.rankPosition[n - 1].base;
    }
    for (u32 n = 0; n < maxSV1; ++n) {
        u32 const c = count[n];
        u32 const r = huf_getIndex(c) + 1;
        u32 const pos = g_lds.u.es.rankPosition[r].curr++;
        huffNode[pos].count = c;
        huffNode[pos].byte = (u8)n;
    }
    for (u32 n = SQZ_RANK_DISTINCT_CUTOFF; n < 191; ++n) {
        int const bucketSize = g_lds.u.es.rankPosition[n].curr - g_lds.u.es.rankPosition[n].base;
        u32 const bucketStartIdx = g_lds.u.es.rankPosition[n].base;
        if (bucketSize > 1) huf_quickSort(huffNode + bucketStartIdx, 0, bucketSize - 1);
    }
}
SQ_DEVFN int huf_buildTree(NodeElt* huffNode, u32 maxSym) {
    NodeElt* const huffNode0 = huffNode - 1;
    int nonNullRank, lowS, lowN, nodeNb = 256, n, nodeRoot;
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
    for (n = nodeRoot - 1; n >= 256; n--) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    return nonNullRank;
}
SQ_DEVFN u32 huf_setMaxHeight(NodeElt* huffNode, u32 lastNonNull, u32 targetNbBits) {
    u32 const largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    {
        int totalCost = 0;
        const u32 baseCost = 1u << (largestBits - targetNbBits);
        int n = (int)lastNonNull;
        while (huffNode[n].nbBits > targetNbBits) {
            totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
            huffNode[n].nbBits = (u8)targetNbBits;
            n--;
        }
        while (huffNode[n].nbBits == targetNbBits) --n;
        totalCost >>= (largestBits - targetNbBits);
        {
            u32 const noSymbol = 0xF0F0F0F0;
            u32 rankLast[SQZ_HUF_TABLELOG_MAX + 2];
            for (u32 i = 0; i < SQZ_HUF_TABLELOG_MAX + 2; i++) rankLast[i] = noSymbol;
            {
                u32 currentNbBits = targetNbBits;
                for (int pos = n; pos >= 0; pos--) {
                    if (huffNode[pos].nbBits >= currentNbBits) continue;
                    currentNbBits = huffNode[pos].nbBits;
                    rankLast[targetNbBits - currentNbBits] = (u32)pos;
                }
            }
            while (totalCost > 0) {
                u32 nBitsToDecrease = sq_highbit32((u32)totalCost) + 1;
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
                while ((nBitsToDecrease <= SQZ_HUF_TABLELOG_MAX) && (rankLast[nBitsToDecrease] == noSymbol))
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
// HUF_buildCTable_wksp into g_lds.u.es.newHuf; returns maxNbBits
SQ_DEVFN u32 huf_buildCTable(const u32* count, u32 maxSym, u32 maxNbBits) {
    NodeElt* const table = g_lds.u.es.huffNodeTbl;
    NodeElt* const huffNode = table + 1;
    HufCT* const ct = &g_lds.u.es.newHuf;
    for (u32 i = 0; i < 2 * 256 + 2; i++) {
        table[i].count = 0;
        table[i].parent = 0;
        table[i].byte = 0;
        table[i].nbBits = 0;
    }
    huf_sort(huffNode, count, maxSym);
    int const nonNullRank = huf_buildTree(huffNode, maxSym);
    maxNbBits = huf_setMaxHeight(huffNode, (u32)nonNullRank, maxNbBits);
    if (maxNbBits > SQZ_HUF_TABLELOG_MAX) return SQZ_ERR;
    {
        u16 nbPerRank[SQZ_HUF_TABLELOG_MAX + 1], valPerRank[SQZ_HUF_TABLELOG_MAX + 1];
        for (u32 i = 0; i <= SQZ_HUF_TABLELOG_MAX; i++) nbPerRank[i] = valPerRank[i] = 0;
        int const alphabetSize = (int)(maxSym + 1);
        for (int n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
        {
            u16 min = 0;
            for (int n = (int)maxNbBits; n > 0; n--) {
                valPerRank[n] = min;
                min += nbPerRank[n];
                min >>= 1;
            }
        }
        for (u32 i = 0; i < 256; i++) {
            ct->nbBits[i] = 0;
            ct->val[i] = 0;
        }
        for (int n = 0; n < alphabetSize; n++) ct->nbBits[huffNode[n].byte] = huffNode[n].nbBits;
        for (int n = 0; n < alphabetSize; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
        ct->maxSym = maxSym;
    }
    return maxNbBits;
}
// HUF_compressWeights; FSE table built in the scratch CTable `wct`
SQ_DEVFN u32 huf_compressWeights(u8* dst, u32 cap, const u8* weights, u32 wtSize, FseCT* wct) {
    u8* op = dst;
    u8* const oend = dst + cap;
    u32 maxSym = SQZ_HUF_TABLELOG_MAX;
    u32 tableLog = 6;
    u32 count[SQZ_HUF_TABLELOG_MAX + 1];
    s16 norm[SQZ_HUF_TABLELOG_MAX + 1];
    if (wtSize <= 1) return 0;
    {
        for (u32 i = 0; i <= SQZ_HUF_TABLELOG_MAX; i++) count[i] = 0;
        for (u32 i = 0; i < wtSize; i++) count[weights[i]]++;
        while (!count[maxSym]) maxSym--;
        u32 maxCount = 0;
        for (u32 s = 0; s <= maxSym; s++)
            if (count[s] > maxCount) maxCount = count[s];
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = fse_optimalTableLog(tableLog, wtSize, maxSym);
    {
        u32 e = fse_normalizeCount(norm, tableLog, count, wtSize, maxSym, 0);
        if (is_err(e)) return e;
    }
    {
        u32 hSize = fse_writeNCount(op, (u32)(oend - op), norm, maxSym, tableLog);
        if (is_err(hSize)) return hSize;
        op += hSize;
    }
    if (is_err(fse_buildCTable(wct, norm, maxSym, tableLog))) return SQZ_ERR;
    {
        u32 cSize = fse_compress_usingCTable(op, (u32)(oend - op), weights, wtSize, wct);
        if (cSize == 0) return 0;
        op += cSize;
    }
    return (u32)(op - dst);
}
// HUF_writeCTable_wksp for g_lds.u.es.newHuf
SQ_DEVFN u32 huf_writeCTable(u8* dst, u32 cap, u32 maxSym, u32 huffLog, FseCT* wct) {
    const HufCT* ct = &g_lds.u.es.newHuf;
    u8 bitsToWeight[SQZ_HUF_TABLELOG_MAX + 1];
    u8* const huffWeight = g_lds.u.es.huffWeight;
    u8* op = dst;
    bitsToWeight[0] = 0;
    for (u32 n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (u8)(huffLog + 1 - n);
    for (u32 n = 0; n < maxSym; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    if (cap < 1) return SQZ_ERR;
    {
        u32 hSize = huf_compressWeights(op + 1, cap - 1, huffWeight, maxSym, wct);
        if (is_err(hSize)) return hSize;
        if ((hSize > 1) & (hSize < maxSym / 2)) {
            op[0] = (u8)hSize;
            return hSize + 1;
        }
    }
    if (maxSym > (256 - 128)) return SQZ_ERR;
    if (((maxSym + 1) / 2) + 1 > cap) return SQZ_ERR;
    op[0] = (u8)(128 + (maxSym - 1));
    huffWeight[maxSym] = 0;
    for (u32 n = 0; n < maxSym; n += 2) op[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}
// ---------------------------------------------------------------- wave-parallel bit-stream emission
// A zstd bit stream is written LSB first, front to back, by symbols taken from the END of their array towards its
// start.  So the bit position of a symbol's code is the sum of the code lengths of everything behind it: one
// exclusive prefix scan over the lanes of a tile (wv_excl_scan), after which every lane ORs its own bits into an
// LDS window (ds_or) and the wave stores the window's complete 32-bit words with coalesced stores.  Only what is a
// true recurrence stays serial (the three FSE states, see wv_encode_sequences).
// The window lives in g_lds.grp (2 KiB, free during the entropy stage): word 0 holds the < 32 carried bits.
#if SQ_ON_DEVICE
#define SQZ_WIN_WORDS 192u
struct WvBitOut {
    u8* dst;
    u32 cap, outBytes, carry, overflow;
};
SQ_DEV void wvb_init(WvBitOut& o, u8* dst, u32 cap) {
    o.dst = dst;
    o.cap = cap;
    o.outBytes = 0;
    o.carry = 0;
    o.overflow = 0;
    SQ_L u32* const win = (SQ_L u32*)g_lds.grp;
    for (u32 i = wv_lane(); i < SQZ_WIN_WORDS; i += 64) win[i] = 0;
    wv_fence();
}
// OR `len` bits (len <= 64; bits above len are zero) of this lane at window bit position `pos`
SQ_DEV void wvb_or(u32 pos, u64 bits, u32 len) {
    if (len == 0) return;
    SQ_L u32* const win = (SQ_L u32*)g_lds.grp;
    u32 const w = pos >> 5, sh = pos & 31;
    u64 const lo = bits << sh;
    u32 const hi = sh ? (u32)(bits >> (64 - sh)) : 0u;
    if ((u32)lo) __hip_atomic_fetch_or(&win[w], (u32)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((u32)(lo >> 32)) __hip_atomic_fetch_or(&win[w + 1], (u32)(lo >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (hi) __hip_atomic_fetch_or(&win[w + 2], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One tile: lane l contributes part A (lenA bits) directly followed by part B (lenB bits); lanes in lane order.
// total bits of a tile <= (SQZ_WIN_WORDS - 4) * 32 - 32.
SQ_DEV void wvb_tile(WvBitOut& o, u64 a, u32 lenA, u64 b, u32 lenB) {
    u32 total;
    u32 const excl = wv_excl_scan(lenA + lenB, &total);
    u32 const pos = o.carry + excl;
    wvb_or(pos, a, lenA);
    wvb_or(pos + lenA, b, lenB);
    wv_fence();
    SQ_L u32* const win = (SQ_L u32*)g_lds.grp;
    u32 const bitsNow = o.carry + total;
    u32 const D = bitsNow >> 5; // complete words
    if (o.outBytes + 4 * D > o.cap) o.overflow = 1;
    if (!o.overflow)
        for (u32 i = wv_lane(); i < D; i += 64) *(SQ_G sq_u32_ua*)(o.dst + o.outBytes + 4 * i) = win[i];
    u32 const rem = win[D];
    wv_fence();
    for (u32 i = wv_lane(); i <= D + 3 && i < SQZ_WIN_WORDS; i += 64) win[i] = 0;
    wv_fence();
    if (wv_lane() == 0) win[0] = rem;
    wv_fence();
    o.outBytes += 4 * D;
    o.carry = bitsNow & 31;
}
// the closing '1' bit and the last partial bytes; returns the stream size, 0 if it did not fit
SQ_DEV u32 wvb_close(WvBitOut& o) {
    SQ_L u32* const win = (SQ_L u32*)g_lds.grp;
    u32 const v = win[0] | (1u << o.carry);
    u32 const nb = (o.carry + 1 + 7) >> 3;
    if (o.outBytes + nb > o.cap) o.overflow = 1;
    if (!o.overflow && wv_lane() == 0)
        for (u32 k = 0; k < nb; k++) sq_gwr8(o.dst + o.outBytes + k, (v >> (8 * k)) & 0xFF);
    wv_fence();
    return o.overflow ? 0u : o.outBytes + nb;
}
#endif

// HUF_compress1X_usingCTable: symbols src[n-1] ... src[0], then the end mark.  Wave-level.  ct: table in LDS.
SQ_DEVFN u32 huf_compress1X(u8* dst, u32 cap, const u8* src, u32 n, const HufCT* ct) {
    if (cap < 8) return 0;
#if SQ_ON_DEVICE
    WvBitOut o;
    wvb_init(o, dst, cap);
    const u8* const nbT = (const u8*)ct->nbBits; // (the table is in LDS -- a block's new one -- or in the workspace -- the previous block's)
    const u16* const valT = (const u16*)ct->val;
    u32 const lane = wv_lane();
    // tiles of 256 symbols from the end; lane l owns the 4 symbols at hi-4l-4 .. hi-4l-1 (written last-first)
    s32 a = (s32)n - 4 * (s32)(lane + 1);
    u32 w = 0;
    auto load4 = [&](s32 at) -> u32 {
        if (at >= 0) return sq_grd32(src + at);
        u32 r = 0;
        for (int k = 0; k < 4; k++)
            if (at + k >= 0) r |= sq_grd8(src + at + k) << (8 * k);
        return r;
    };
    if (a > -4) w = load4(a);
    for (u32 hi = n; hi > 0; hi = hi > 256 ? hi - 256 : 0) {
        // prefetch the next tile's word before working on this one
        s32 const an = a - 256;
        u32 wn = 0;
        if (hi > 256 && an > -4) wn = load4(an);
        u64 acc = 0;
        u32 len = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (a + k >= 0) {
                u32 const sym = (w >> (8 * k)) & 0xFF;
                u32 const nb = nbT[sym];
                acc |= (u64)(valT[sym] & ((1u << nb) - 1)) << len;
                len += nb;
            }
        }
        wvb_tile(o, acc, len, 0, 0);
        a = an;
        w = wn;
    }
    return wvb_close(o);
#else
    BitW b;
    bw_init(b, dst, cap);
    for (u32 i = n; i > 0; i--) {
        u8 s = src[i - 1];
        bw_add(b, ct->val[s], ct->nbBits[s]);
    }
    return bw_close(b);
#endif
}
// HUF_compress4X_usingCTable.  Wave-level.
SQ_DEVFN u32 huf_compress4X(u8* dst, u32 cap, const u8* src, u32 n, const HufCT* ct) {
    u32 const segmentSize = (n + 3) / 4;
    const u8* ip = src;
    const u8* const iend = src + n;
    u8* op = dst;
    u8* const oend = dst + cap;
    if (cap < 6 + 1 + 1 + 1 + 8) return 0;
    if (n < 12) return 0;
    op += 6;
    for (int k = 0; k < 3; k++) {
        u32 cSize = huf_compress1X(op, (u32)(oend - op), ip, segmentSize, ct);
        if (cSize == 0 || cSize > 65535) return 0;
        gst8(dst + 2 * k, cSize & 0xFF);
        gst8(dst + 2 * k + 1, cSize >> 8);
        op += cSize;
        ip += segmentSize;
    }
    {
        u32 cSize = huf_compress1X(op, (u32)(oend - op), ip, (u32)(iend - ip), ct);
        if (cSize == 0 || cSize > 65535) return 0;
        op += cSize;
    }
    return (u32)(op - dst);
}
SQ_DEV u32 huf_compressCTable(u8* ostart, u8* op, u8* oend, const u8* src, u32 n, u32 singleStream, const HufCT* ct) {
    u32 const cSize = singleStream ? huf_compress1X(op, (u32)(oend - op), src, n, ct)
                                   : huf_compress4X(op, (u32)(oend - op), src, n, ct);
    if (cSize == 0) return 0;
    op += cSize;
    if ((u32)(op - ostart) >= n - 1) return 0;
    return (u32)(op - ostart);
}

SQ_DEV u32 min_gain(u32 srcSize, u32 strategy) {
    u32 const minlog = (strategy >= 8) ? strategy - 1 : 6;
    return (srcSize >> minlog) + 2;
}

// ---------------------------------------------------------------- literals section
// raw literals header (ZSTD_noCompressLiterals); the payload is copied by the whole wave
SQ_DEV u32 lit_header_basic(u8* dst, u32 n, u32 type /*0 raw, 1 rle*/) {
    u32 const flSize = 1 + (n > 31) + (n > 4095);
    WV_SERIAL {
        if (flSize == 1)
            dst[0] = (u8)(type + (n << 3));
        else if (flSize == 2) {
            u32 v = type + (1 << 2) + (n << 4);
            dst[0] = (u8)v;
            dst[1] = (u8)(v >> 8);
        } else {
            u32 v = type + (3 << 2) + (n << 4);
            dst[0] = (u8)v;
            dst[1] = (u8)(v >> 8);
            dst[2] = (u8)(v >> 16);
        }
    }
    return flSize;
}
// `discard`: the caller already knows that a raw literals section makes the block a raw block (a
// block without sequences: section + 1 > block size), so only the size matters, not the bytes.
SQ_DEVFN u32 noCompressLiterals(u8* dst, u32 cap, const u8* src, u32 n, u32 discard) {
    u32 const flSize = 1 + (n > 31) + (n > 4095);
    if (n + flSize > cap) return SQZ_ERR;
    if (discard) return n + flSize;
    lit_header_basic(dst, n, 0);
    wv_memcpy(dst + flSize, src, n);
    return n + flSize;
}

// ZSTD_compressLiterals + HUF_compress_internal.  Wave-level function.
// prev/next: entropy tables in LDS.  Returns section size or SQZ_ERR.
SQ_DEVFN u32 compressLiterals(u8* dst, u32 dstCap, const u8* src, u32 srcSize, const Entropy* prev,
                              Entropy* next, u32 strategy, u32 suspectUncompressible, u32 discardRaw) {
    u32 const lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    u32 const singleStream = srcSize < 256;
    // nextHuf = prevHuf (table + repeat mode)
    WV_SERIAL {
        next->huf = prev->huf;
        next->huf_repeat = prev->huf_repeat;
        g_lds.ret[6] = 0; // becomes 1 when anything is written to dst before the literals' mode is decided
    }
    wv_mem_sync();
    {
        int const shift = (9 - (int)strategy) < 3 ? (9 - (int)strategy) : 3;
        u32 const mintc = (prev->huf_repeat == REP_VALID) ? 6u : (8u << shift);
        if (srcSize < mintc) return noCompressLiterals(dst, dstCap, src, srcSize, discardRaw);
    }
    if (dstCap < lhSize + 1) return SQZ_ERR;

    // --- HUF_compress_internal: statistics by the whole wave
    u32 early = 0xFFFFFFFEu; // "continue"
    if (suspectUncompressible && srcSize >= 4096 * 10) {
        u32 l1, l2, dummy;
        wv_zero_lds(g_lds.u.es.count, 256);
        wv_hist_bytes(g_lds.u.es.count, src, 4096);
        wv_max_and_last(g_lds.u.es.count, 256, &l1, &dummy);
        wv_zero_lds(g_lds.u.es.count, 256);
        wv_hist_bytes(g_lds.u.es.count, src + srcSize - 4096, 4096);
        wv_max_and_last(g_lds.u.es.count, 256, &l2, &dummy);
        if (l1 + l2 <= ((2 * 4096) >> 7) + 4) early = 0;
    }
    u32 largest = 0, maxSym = 255;
    if (early != 0) {
        wv_zero_lds(g_lds.u.es.count, 256);
        wv_hist_bytes(g_lds.u.es.count, src, srcSize);
        wv_max_and_last(g_lds.u.es.count, 256, &largest, &maxSym);
    }
    // --- table construction and header: serial on lane 0.  Leaves in g_lds.ret: [0] what to do next (0 = cLitSize
    // decided, 1 = emit the streams with the previous table, 2 = with the new table), [1] cLitSize when decided,
    // [2] bytes of table description in front of the streams, [3] the repeat mode so far.
    u8* const ostart = dst + lhSize;
    u8* const oend = ostart + (dstCap - lhSize);
    WV_SERIAL {
        u32 const hcap = dstCap - lhSize;
        u8* op = ostart;
        u32 cLitSize = 0, todo = 0, hdrBytes = 0;
        u32 repeat = prev->huf_repeat;
        const u32* count = g_lds.u.es.count;
        do {
            if (early == 0) {
                cLitSize = 0;
                break;
            }
            if (largest == srcSize) {
                g_lds.ret[6] = 1;
                *ostart = src[0];
                cLitSize = 1;
                break;
            }
            if (largest <= (srcSize >> 7) + 4) {
                cLitSize = 0;
                break;
            }
            if (repeat == REP_CHECK) { // HUF_validateCTable
                int bad = 0;
                if (prev->huf.maxSym < maxSym)
                    bad = 1;
                else
                    for (u32 s = 0; s <= maxSym; ++s) bad |= (count[s] != 0) & (prev->huf.nbBits[s] == 0);
                if (bad) repeat = REP_NONE;
            }
            u32 huffLog = fse_optimalTableLog_internal(11, srcSize, maxSym, 1);
            {
                u32 maxBits = huf_buildCTable(count, maxSym, huffLog);
                if (is_err(maxBits)) {
                    cLitSize = SQZ_ERR;
                    break;
                }
                huffLog = maxBits;
            }
            {
                // scratch FSE table for the weights: next->ll is rebuilt later in this block
                g_lds.ret[6] = 1;
                u32 hSize = huf_writeCTable(op, hcap, maxSym, huffLog, &next->ll);
                if (is_err(hSize)) {
                    cLitSize = SQZ_ERR;
                    break;
                }
                if (repeat != REP_NONE) {
                    u32 oldBits = 0, newBits = 0;
                    for (u32 s = 0; s <= maxSym; ++s) {
                        oldBits += (u32)prev->huf.nbBits[s] * count[s];
                        newBits += (u32)g_lds.u.es.newHuf.nbBits[s] * count[s];
                    }
                    u32 const oldSize = oldBits >> 3, newSize = newBits >> 3;
                    if (oldSize <= hSize + newSize || hSize + 12 >= srcSize) {
                        todo = 1; // streams with the previous table, no description
                        break;
                    }
                }
                if (hSize + 12u >= srcSize) {
                    cLitSize = 0;
                    break;
                }
                hdrBytes = hSize;
                repeat = REP_NONE;
                next->huf = g_lds.u.es.newHuf;
                todo = 2;
            }
        } while (0);
        g_lds.ret[0] = todo;
        g_lds.ret[1] = cLitSize;
        g_lds.ret[2] = hdrBytes;
        g_lds.ret[3] = repeat;
    }
    wv_mem_sync();
    // --- stream emission by the whole wave (huf_compress1X / 4X)
    u32 cLit = g_lds.ret[1];
    {
        u32 const todo = g_lds.ret[0], hdrBytes = g_lds.ret[2];
        wv_fence();
        if (todo) {
            const HufCT* const table = (todo == 1) ? &prev->huf : &g_lds.u.es.newHuf;
            cLit = huf_compressCTable(ostart, ostart + hdrBytes, oend, src, srcSize, singleStream, table);
            wv_mem_sync();
        }
    }
    WV_SERIAL {
        u32 const cLitSize = cLit;
        u32 const repeat = g_lds.ret[3];
        u32 hType = SET_COMPRESSED;
        if (repeat != REP_NONE) hType = SET_REPEAT;
        // ZSTD_compressLiterals post-processing
        u32 mode; // 0 raw, 1 rle, 2 compressed
        {
            u32 const minGain = min_gain(srcSize, strategy);
            if ((cLitSize == 0) || is_err(cLitSize) || (cLitSize >= srcSize - minGain)) {
                next->huf = prev->huf;
                next->huf_repeat = prev->huf_repeat;
                mode = 0;
            } else if (cLitSize == 1 && srcSize >= 8) { /* (srcSize >= 8) || allBytesIdentical; srcSize >= mintc >= 32 here */
                next->huf = prev->huf;
                next->huf_repeat = prev->huf_repeat;
                mode = 1;
            } else {
                if (hType == SET_COMPRESSED) next->huf_repeat = REP_CHECK;
                mode = 2;
                if (lhSize == 3) {
                    u32 const lhc = hType + ((u32)(!singleStream) << 2) + (srcSize << 4) + (cLitSize << 14);
                    dst[0] = (u8)lhc;
                    dst[1] = (u8)(lhc >> 8);
                    dst[2] = (u8)(lhc >> 16);
                } else if (lhSize == 4) {
                    u32 const lhc = hType + (2 << 2) + (srcSize << 4) + (cLitSize << 18);
                    dst[0] = (u8)lhc;
                    dst[1] = (u8)(lhc >> 8);
                    dst[2] = (u8)(lhc >> 16);
                    dst[3] = (u8)(lhc >> 24);
                } else {
                    u32 const lhc = hType + (3 << 2) + (srcSize << 4) + (cLitSize << 22);
                    dst[0] = (u8)lhc;
                    dst[1] = (u8)(lhc >> 8);
                    dst[2] = (u8)(lhc >> 16);
                    dst[3] = (u8)(lhc >> 24);
                    dst[4] = (u8)(cLitSize >> 10);
                }
            }
        }
        g_lds.ret[0] = mode;
        g_lds.ret[1] = cLitSize;
    }
    wv_mem_sync();
    u32 const mode = g_lds.ret[0], cLitSize = g_lds.ret[1];
    if (mode == 0) return noCompressLiterals(dst, dstCap, src, srcSize, discardRaw);
    if (mode == 1) { // ZSTD_compressRleLiteralsBlock
        u32 const fl = lit_header_basic(dst, srcSize, 1);
        WV_SERIAL { dst[fl] = src[0]; }
        return fl + 1;
    }
    return lhSize + cLitSize;
}

// ---------------------------------------------------------------- sequences section (serial parts)
SQ_DEV u32 entropyCost(const u32* count, u32 max, u32 total) {
    u32 cost = 0;
    for (u32 s = 0; s <= max; ++s) {
        u32 norm = (256 * count[s]) / total;
        if (count[s] != 0 && norm == 0) norm = 1;
        cost += count[s] * kInvProbLog256[norm];
    }
    return cost >> 8;
}
SQ_DEV u32 crossEntropyCost(const s16* norm, u32 accuracyLog, const u32* count, u32 max) {
    u32 const shift = 8 - accuracyLog;
    u32 cost = 0;
    for (u32 s = 0; s <= max; ++s) {
        u32 const normAcc = (norm[s] != -1) ? (u32)norm[s] : 1;
        u32 const norm256 = normAcc << shift;
        cost += count[s] * kInvProbLog256[norm256];
    }
    return cost >> 8;
}
SQ_DEV u32 fse_bitCost(const FseCT* ct, u32 tableLog, u32 sym, u32 accuracyLog) {
    u32 const minNbBits = ct->dnb[sym] >> 16;
    u32 const threshold = (minNbBits + 1) << 16;
    u32 const tableSize = 1u << tableLog;
    u32 const deltaFromThreshold = threshold - (ct->dnb[sym] + tableSize);
    u32 const normalizedDeltaFromThreshold = (deltaFromThreshold << accuracyLog) >> tableLog;
    u32 const bitMultiplier = 1u << accuracyLog;
    return (minNbBits + 1) * bitMultiplier - normalizedDeltaFromThreshold;
}
SQ_DEV u32 fseBitCost(const FseCT* ct, const u32* count, u32 max) {
    u32 const kAccuracyLog = 8;
    u32 cost = 0;
    if (ct->maxSym < max) return SQZ_ERR;
    for (u32 s = 0; s <= max; ++s) {
        u32 const tableLog = ct->tableLog;
        u32 const badCost = (tableLog + 1) << kAccuracyLog;
        if (count[s] == 0) continue;
        u32 const bitCost = fse_bitCost(ct, tableLog, s, kAccuracyLog);
        if (bitCost >= badCost) return SQZ_ERR;
        cost += count[s] * bitCost;
    }
    return cost >> kAccuracyLog;
}
SQ_DEVFN u32 NCountCost(const u32* count, u32 max, u32 nbSeq, u32 FSELog) {
    u8 wksp[512];
    const u32 tableLog = fse_optimalTableLog(FSELog, nbSeq, max);
    u32 e = fse_normalizeCount(g_lds.u.es.norm, tableLog, count, nbSeq, max, nbSeq >= 2048);
    if (is_err(e)) return e;
    return fse_writeNCount(wksp, sizeof(wksp), g_lds.u.es.norm, max, tableLog);
}
// SQZ_ERR compares as "infinite" exactly like libzstd's size_t error codes
SQ_DEVFN u32 selectEncodingType(u32* repeatMode, const u32* count, u32 max, u32 mostFrequent, u32 nbSeq,
                                u32 FSELog, const FseCT* prevCTable, const s16* defaultNorm,
                                u32 defaultNormLog, u32 isDefaultAllowed) {
    if (mostFrequent == nbSeq) {
        *repeatMode = REP_NONE;
        if (isDefaultAllowed && nbSeq <= 2) return SET_BASIC;
        return SET_RLE;
    }
    {
        u32 const basicCost = isDefaultAllowed ? crossEntropyCost(defaultNorm, defaultNormLog, count, max) : SQZ_ERR;
        u32 const repeatCost = *repeatMode != REP_NONE ? fseBitCost(prevCTable, count, max) : SQZ_ERR;
        u32 const ncc = NCountCost(count, max, nbSeq, FSELog);
        u64 const compressedCost = ((u64)ncc << 3) + entropyCost(count, max, nbSeq);
        if (basicCost <= repeatCost && (u64)basicCost <= compressedCost) {
            *repeatMode = REP_NONE;
            return SET_BASIC;
        }
        if ((u64)repeatCost <= compressedCost) return SET_REPEAT;
    }
    *repeatMode = REP_CHECK;
    return SET_COMPRESSED;
}
SQ_DEVFN u32 buildCTable(u8* dst, u32 dstCap, FseCT* nextCTable, u32 FSELog, u32 type, u32* count, u32 max,
                         const u8* codeTable, u32 nbSeq, const s16* defaultNorm, u32 defaultNormLog,
                         u32 defaultMax, const FseCT* prevCTable) {
    switch (type) {
    case SET_RLE:
        fse_buildCTable_rle(nextCTable, max);
        if (dstCap == 0) return SQZ_ERR;
        *dst = codeTable[0];
        return 1;
    case SET_REPEAT: *nextCTable = *prevCTable; return 0;
    case SET_BASIC:
        for (u32 s = 0; s <= defaultMax; s++) g_lds.u.es.norm[s] = defaultNorm[s];
        if (is_err(fse_buildCTable(nextCTable, g_lds.u.es.norm, defaultMax, defaultNormLog))) return SQZ_ERR;
        return 0;
    case SET_COMPRESSED: {
        u32 nbSeq_1 = nbSeq;
        const u32 tableLog = fse_optimalTableLog(FSELog, nbSeq, max);
        if (count[codeTable[nbSeq - 1]] > 1) {
            count[codeTable[nbSeq - 1]]--;
            nbSeq_1--;
        }
        {
            u32 e = fse_normalizeCount(g_lds.u.es.norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048);
            if (is_err(e)) return e;
        }
        {
            u32 const NCountSize = fse_writeNCount(dst, dstCap, g_lds.u.es.norm, max, tableLog);
            if (is_err(NCountSize)) return NCountSize;
            if (is_err(fse_buildCTable(nextCTable, g_lds.u.es.norm, max, tableLog))) return SQZ_ERR;
            return NCountSize;
        }
    }
    default: return SQZ_ERR;
    }
}
// ZSTD_encodeSequences.  Wave-level.  The three FSE states are true recurrences: lanes 0..2 run them (OF, ML, LL)
// side by side over all sequences, last to first, and leave (nbBits << 12 | bits) per sequence and stream in
// stBits (4 x u16 per sequence, workspace); then tiles of 64 sequences are packed in parallel: every lane builds
// its sequence's bits (state bits of OF, ML, LL; extra bits of LL, ML, OF), a prefix scan places them (wvb_tile).
SQ_DEVFN u32 encodeSequences(u8* dst, u32 dstCap, const FseCT* ctML, const u8* mlCodeTable, const FseCT* ctOF,
                             const u8* ofCodeTable, const FseCT* ctLL, const u8* llCodeTable,
                             const SeqDef* sequences, u32 nbSeq, u16* stBits) {
    if (dstCap < 9) return SQZ_ERR;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u32 finalValue = 0, tableLog = 0;
    if (lane < 3) {
        const FseCT* const ctg = lane == 0 ? ctOF : lane == 1 ? ctML : ctLL;
        const FseCT* const ct = (const FseCT*)ctg;
        const u8* const codes = lane == 0 ? ofCodeTable : lane == 1 ? mlCodeTable : llCodeTable;
        tableLog = ct->tableLog;
        u32 value;
        { // FSE_initCState2 with the last sequence's code
            u32 const sym = sq_grd8(codes + nbSeq - 1);
            u32 const dnb = ct->dnb[sym];
            u32 const nbBitsOut = (dnb + (1u << 15)) >> 16;
            value = (nbBitsOut << 16) - dnb;
            value = ct->stateTable[(value >> nbBitsOut) + ct->dfs[sym]];
        }
        if (nbSeq >= 2) {
            // sequences nbSeq-2 .. 0 in groups of 16 code bytes (one aligned 16-byte load per group, the next
            // group's load in flight while this one is worked on)
            s32 g = (s32)((nbSeq - 2) >> 4);
            sq_v4u cur = *(const SQ_G sq_v4u*)(codes + 16 * g);
            for (; g >= 0; g--) {
                sq_v4u nxt = cur;
                if (g > 0) nxt = *(const SQ_G sq_v4u*)(codes + 16 * (g - 1));
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    u32 const n = 16u * (u32)g + (u32)k;
                    if (n <= nbSeq - 2) {
                        u32 const sym = (cur[k >> 2] >> (8 * (k & 3))) & 0xFF;
                        u32 const nb = (value + ct->dnb[sym]) >> 16;
                        *(SQ_G u16*)(stBits + 4 * n + lane) = (u16)((nb << 12) | (value & ((1u << nb) - 1)));
                        value = ct->stateTable[(value >> nb) + ct->dfs[sym]];
                    }
                }
                cur = nxt;
            }
        }
        finalValue = value;
    }
    wv_mem_sync(); // the state bits have landed
    WvBitOut o;
    wvb_init(o, dst, dstCap);
    for (s32 hi = (s32)nbSeq; hi > 0; hi -= 64) {
        s32 const n = hi - 1 - (s32)lane;
        u64 A = 0, B = 0;
        u32 lenA = 0, lenB = 0;
        if (n >= 0) {
            sq_v2u const sd = *(const SQ_G sq_v2u*)(sequences + n); // offBase | litLength, mlBase
            u32 const ll = sq_grd8(llCodeTable + n), of = sq_grd8(ofCodeTable + n), ml = sq_grd8(mlCodeTable + n);
            if ((u32)n != nbSeq - 1) {
                sq_v2u const sb = *(const SQ_G sq_v2u*)(stBits + 4 * n);
                u32 const fOF = sb.x & 0xFFFF, fML = sb.x >> 16, fLL = sb.y & 0xFFFF;
                A = fOF & 0xFFF;
                lenA = fOF >> 12;
                A |= (u64)(fML & 0xFFF) << lenA;
                lenA += fML >> 12;
                A |= (u64)(fLL & 0xFFF) << lenA;
                lenA += fLL >> 12;
            }
            u32 const llb = LL_bits[ll], mlb = ML_bits[ml];
            A |= (u64)((sd.y & 0xFFFF) & ((1u << llb) - 1)) << lenA;
            lenA += llb;
            A |= (u64)((sd.y >> 16) & ((1u << mlb) - 1)) << lenA;
            lenA += mlb;
            B = sd.x & (u32)((1ull << of) - 1);
            lenB = of;
        }
        wvb_tile(o, A, lenA, B, lenB);
    }
    { // FSE_flushCState x 3: ML, OF, LL
        u32 const vOF = wv_readlane(finalValue, 0), vML = wv_readlane(finalValue, 1), vLL = wv_readlane(finalValue, 2);
        u32 const tOF = wv_readlane(tableLog, 0), tML = wv_readlane(tableLog, 1), tLL = wv_readlane(tableLog, 2);
        u64 A = vML & ((1u << tML) - 1);
        u32 len = tML;
        A |= (u64)(vOF & ((1u << tOF) - 1)) << len;
        len += tOF;
        A |= (u64)(vLL & ((1u << tLL) - 1)) << len;
        len += tLL;
        wvb_tile(o, lane == 0 ? A : 0, lane == 0 ? len : 0, 0, 0);
    }
    u32 const streamSize = wvb_close(o);
    wv_mem_sync();
    return streamSize ? streamSize : SQZ_ERR;
#else
    (void)stBits;
    BitW b;
    FseState stML, stOF, stLL;
    bw_init(b, dst, dstCap);
    fse_initCState2(stML, ctML, mlCodeTable[nbSeq - 1]);
    fse_initCState2(stOF, ctOF, ofCodeTable[nbSeq - 1]);
    fse_initCState2(stLL, ctLL, llCodeTable[nbSeq - 1]);
    bw_add(b, sequences[nbSeq - 1].litLength, LL_bits[llCodeTable[nbSeq - 1]]);
    bw_add(b, sequences[nbSeq - 1].mlBase, ML_bits[mlCodeTable[nbSeq - 1]]);
    bw_add(b, sequences[nbSeq - 1].offBase, ofCodeTable[nbSeq - 1]);
    for (u32 n = nbSeq - 2; n < nbSeq; n--) {
        u32 const llCode = llCodeTable[n], ofCode = ofCodeTable[n], mlCode = mlCodeTable[n];
        fse_encodeSymbol(b, stOF, ofCode);
        fse_encodeSymbol(b, stML, mlCode);
        fse_encodeSymbol(b, stLL, llCode);
        bw_add(b, sequences[n].litLength, LL_bits[llCode]);
        bw_add(b, sequences[n].mlBase, ML_bits[mlCode]);
        bw_add(b, sequences[n].offBase, ofCode);
    }
    fse_flushCState(b, stML);
    fse_flushCState(b, stOF);
    fse_flushCState(b, stLL);
    {
        u32 const streamSize = bw_close(b);
        if (streamSize == 0) return SQZ_ERR;
        return streamSize;
    }
#endif
}

// Sequence store of one block (arrays in the wave's global workspace)
struct SeqStore {
    SeqDef* seqs;
    u8* lits;
    u8 *llCode, *ofCode, *mlCode;
    u16* stBits; // 4 x u16 per sequence: FSE state bits of the sequences bit stream (encodeSequences)
    u32 nseq, nlits;
    u32 longLengthType, longLengthPos;
};

SQ_DEV void entropy_reset(Entropy* e) { // serial
    e->huf_repeat = REP_NONE;
    e->ll_repeat = e->of_repeat = e->ml_repeat = REP_NONE;
    e->huf.maxSym = 0;
    e->ll.tableLog = e->of.tableLog = e->ml.tableLog = 0;
    e->ll.maxSym = e->of.maxSym = e->ml.maxSym = 0;
}

// ZSTD_seqToCodes + the three code histograms, by the whole wave.  hist: 3 x 64 u32 in LDS
// (g_lds.u.es.count[0..191]); most-frequent / max computed later in the serial part.
SQ_DEV void seqToCodes_and_hist(const SeqStore& ss) {
    wv_zero_lds(g_lds.u.es.count, 192);
#if SQ_ON_DEVICE
    for (u32 u = wv_lane(); u < ss.nseq; u += 64) {
        SeqDef s = ss.seqs[u];
        u32 ll = LLcode(s.litLength), of = sq_highbit32(s.offBase), ml = MLcode(s.mlBase);
        if (ss.longLengthType == 1 && ss.longLengthPos == u) ll = MaxLL;
        if (ss.longLengthType == 2 && ss.longLengthPos == u) ml = MaxML;
        ss.llCode[u] = (u8)ll;
        ss.ofCode[u] = (u8)of;
        ss.mlCode[u] = (u8)ml;
        atomicAdd(&g_lds.u.es.count[ll], 1u);
        atomicAdd(&g_lds.u.es.count[64 + of], 1u);
        atomicAdd(&g_lds.u.es.count[128 + ml], 1u);
    }
    wv_mem_sync();
#else
    for (u32 u = 0; u < ss.nseq; u++) {
        u32 ll = LLcode(ss.seqs[u].litLength), of = sq_highbit32(ss.seqs[u].offBase), ml = MLcode(ss.seqs[u].mlBase);
        if (ss.longLengthType == 1 && ss.longLengthPos == u) ll = MaxLL;
        if (ss.longLengthType == 2 && ss.longLengthPos == u) ml = MaxML;
        ss.llCode[u] = (u8)ll;
        ss.ofCode[u] = (u8)of;
        ss.mlCode[u] = (u8)ml;
        g_lds.u.es.count[ll]++;
        g_lds.u.es.count[64 + of]++;
        g_lds.u.es.count[128 + ml]++;
    }
#endif
}

SQ_DEV void hist_stats(const u32* count, u32 maxIn, u32* maxOut, u32* mostFrequent) { // serial
    u32 mx = maxIn;
    while (mx > 0 && count[mx] == 0) mx--;
    u32 mf = 0;
    for (u32 s = 0; s <= mx; s++)
        if (count[s] > mf) mf = count[s];
    *maxOut = mx;
    *mostFrequent = mf;
}

// ZSTD_entropyCompressSeqStore (+ _internal).  Wave-level.  Returns compressed body size,
// 0 = emit a raw block, SQZ_ERR = internal error.
SQ_DEVFN u32 entropy_compress_seqstore(const SeqStore& ss, const Entropy* prev, Entropy* next, u32 strategy,
                                       u8* dst, u32 dstCap, u32 blockSize) {
    PROF_T0(t_ent);
    u32 const nbSeq = ss.nseq;
    u32 litSectionSize;
    WV_SERIAL { g_lds.ret[7] = 1; } // "dst may have been written to" (0 only on the early way out below)
    {
        u32 const litSize = ss.nlits;
        u32 const suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);
        // A raw literals section of litSize + 3 bytes, the sequence count (1 byte) and -- when there are sequences -- at
        // least the modes byte and one byte of bit stream: if that much is not below blockSize - minGain the block is
        // emitted raw whatever the rest costs (ZSTD_entropyCompressSeqStore: cSize >= maxCSize -> 0).  Then the raw
        // literals are not copied and nothing else is written either (discardRaw): incompressible chunks with a few
        // short matches -- what stage 1 hands over on random data -- stop here, and stage 1's raw payload, if it is in
        // place in dst, stays intact (ret[7] = 0 tells the caller).
        u32 const certainRaw = (litSize > 4095) && (litSize + 3 + (nbSeq ? 3u : 1u) >= blockSize - min_gain(blockSize, strategy));
        litSectionSize = compressLiterals(dst, dstCap, ss.lits, litSize, prev, next, strategy, suspect, certainRaw);
        if (is_err(litSectionSize)) return (blockSize <= dstCap) ? 0 : SQZ_ERR;
        if (certainRaw && litSectionSize == litSize + 3) { // the literals' mode came out raw
            wv_mem_sync();
            u32 const touched = g_lds.ret[6];
            wv_fence();
            WV_SERIAL { g_lds.ret[7] = touched; }
            wv_mem_sync();
            PROF_ADD(6, t_ent);
            return 0;
        }
    }
    SQ_TRACE(0, 81);
    SQ_TRACE(6, litSectionSize);
    if (nbSeq) seqToCodes_and_hist(ss);
    SQ_TRACE(0, 82);
    WV_SERIAL {
        u8* const ostart = dst;
        u8* const oend = dst + dstCap;
        u8* op = dst + litSectionSize;
        u32 result = 0; // 0 => raw
        u32 lastCountSize = 0;
        g_lds.ret[3] = 0;
        do {
            if ((oend - op) < 3 + 1) {
                result = 0;
                break;
            }
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
                result = (u32)(op - ostart);
            } else {
                u8* const seqHead = op++;
                u32 LLtype, Offtype, MLtype;
                u32 bad = 0;
                { // LL
                    u32 max, mf;
                    u32* count = g_lds.u.es.count;
                    hist_stats(count, MaxLL, &max, &mf);
                    next->ll_repeat = prev->ll_repeat;
                    LLtype = selectEncodingType(&next->ll_repeat, count, max, mf, nbSeq, LLFSELog, &prev->ll,
                                                LL_defaultNorm, 6, 1);
                    u32 const countSize = buildCTable(op, (u32)(oend - op), &next->ll, LLFSELog, LLtype, count, max,
                                                      ss.llCode, nbSeq, LL_defaultNorm, 6, MaxLL, &prev->ll);
                    if (is_err(countSize))
                        bad = 1;
                    else {
                        if (LLtype == SET_COMPRESSED) lastCountSize = countSize;
                        op += countSize;
                    }
                }
                if (!bad) { // OF
                    u32 max, mf;
                    u32* count = g_lds.u.es.count + 64;
                    hist_stats(count, MaxOff, &max, &mf);
                    u32 const defaultAllowed = (max <= DefaultMaxOff);
                    next->of_repeat = prev->of_repeat;
                    Offtype = selectEncodingType(&next->of_repeat, count, max, mf, nbSeq, OffFSELog, &prev->of,
                                                 OF_defaultNorm, 5, defaultAllowed);
                    u32 const countSize = buildCTable(op, (u32)(oend - op), &next->of, OffFSELog, Offtype, count, max,
                                                      ss.ofCode, nbSeq, OF_defaultNorm, 5, DefaultMaxOff, &prev->of);
                    if (is_err(countSize))
                        bad = 1;
                    else {
                        if (Offtype == SET_COMPRESSED) lastCountSize = countSize;
                        op += countSize;
                    }
                }
                if (!bad) { // ML
                    u32 max, mf;
                    u32* count = g_lds.u.es.count + 128;
                    hist_stats(count, MaxML, &max, &mf);
                    next->ml_repeat = prev->ml_repeat;
                    MLtype = selectEncodingType(&next->ml_repeat, count, max, mf, nbSeq, MLFSELog, &prev->ml,
                                                ML_defaultNorm, 6, 1);
                    u32 const countSize = buildCTable(op, (u32)(oend - op), &next->ml, MLFSELog, MLtype, count, max,
                                                      ss.mlCode, nbSeq, ML_defaultNorm, 6, MaxML, &prev->ml);
                    if (is_err(countSize))
                        bad = 1;
                    else {
                        if (MLtype == SET_COMPRESSED) lastCountSize = countSize;
                        op += countSize;
                    }
                }
                if (bad) {
                    result = (blockSize <= dstCap) ? 0 : SQZ_ERR;
                    break;
                }
                *seqHead = (u8)((LLtype << 6) + (Offtype << 4) + (MLtype << 2));
                // the bit stream itself is written by the whole wave, below
                g_lds.ret[3] = 1;
                g_lds.ret[4] = (u32)(op - ostart);
                g_lds.ret[5] = lastCountSize;
                break;
            }
            {
                u32 const maxCSize = blockSize - min_gain(blockSize, strategy);
                if (result >= maxCSize) result = 0;
            }
        } while (0);
        g_lds.ret[2] = result;
    }
    wv_mem_sync();
    if (nbSeq && g_lds.ret[3] == 1) {
        u32 const opOff = g_lds.ret[4], lastCountSize = g_lds.ret[5];
        wv_fence();
        u32 result;
        u32 const bitstreamSize = encodeSequences(dst + opOff, dstCap - opOff, &next->ml, ss.mlCode, &next->of, ss.ofCode,
                                                  &next->ll, ss.llCode, ss.seqs, nbSeq, ss.stBits);
        if (is_err(bitstreamSize)) {
            result = (blockSize <= dstCap) ? 0 : SQZ_ERR;
        } else if (lastCountSize && (lastCountSize + bitstreamSize) < 4) {
            result = 0;
        } else {
            result = opOff + bitstreamSize;
            u32 const maxCSize = blockSize - min_gain(blockSize, strategy);
            if (result >= maxCSize) result = 0;
        }
        PROF_ADD(6, t_ent);
        return result;
    }
    wv_mem_sync();
    PROF_ADD(6, t_ent);
    return g_lds.ret[2];
}

} // namespace sqz
