// sq_zstd_match.h -- LZ match finders and the lazy parser of zstd level 12 on gfx950, one wave
// per chunk, bit-exact with libzstd 1.5.7 (reached from /root/reference/src/util/chunk.rs:89-90).
//
//   chunk > 256 KiB        : lazy2 + row-hash match finder (ZSTD_RowFindBestMatch, 64-entry rows)
//   16 KiB < chunk <= 256K : btlazy2, DUBT binary tree (ZSTD_DUBT_findBestMatch)
//
// Wave mapping: the parser state machine is wave-uniform; the 64 lanes are spent on
//   * hash insertion in batches of 64 positions (hash per lane, in-wave ranking of equal
//     rows/hashes, coalesced bt writes)                       -> dubt_update / row_update
//   * the 64-entry tag row: one lane per slot, ballot -> candidate mask   -> row_find
//   * one candidate per lane match-length evaluation          -> row_find
//   * 8 bytes per lane match extension                        -> wv_count
// Tables (hash / chain / tag) are in the wave's HBM workspace; see sq_zstd_kernel.h.
#pragma once
#include "sq_zstd_entropy.h"

namespace sqz {

#define SQZ_WINDOW_START 2u
#define SQZ_BLOCKSIZE_MAX (128u << 10)
#define SQZ_REP_NUM 3
#define SQZ_MINMATCH 3
#define OFFSET_TO_OFFBASE(o) ((o) + SQZ_REP_NUM)
#define OFFBASE_TO_OFFSET(o) ((o)-SQZ_REP_NUM)
#define OFFBASE_IS_OFFSET(o) ((o) > SQZ_REP_NUM)

enum { STRAT_LAZY2 = 5, STRAT_BTLAZY2 = 6, STRAT_BTOPT = 7 };

struct CParams {
    u32 windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
};

// ZSTD_getCParams(12, n) after ZSTD_adjustCParams (table measured on libzstd 1.5.7)
SQ_DEV CParams get_cparams(u32 n) {
    CParams p;
    if (n == 0) n = 1;
    if (n <= (16u << 10)) {
        u32 w = 0;
        while ((1u << w) < n) w++;
        p.windowLog = w < 10 ? 10 : w;
        p.chainLog = (w + 1 < 7) ? 7 : w + 1;
        p.hashLog = p.chainLog > 14 ? 14 : p.chainLog;
        p.searchLog = 4;
        p.minMatch = 3;
        p.targetLength = 24;
        p.strategy = STRAT_BTOPT;
    } else if (n <= (256u << 10)) {
        u32 w = 15;
        while ((1u << w) < n) w++;
        p.windowLog = w;
        p.chainLog = w + 1;
        p.hashLog = (w == 17) ? 17 : w + 1;
        p.searchLog = 7;
        p.minMatch = 4;
        p.targetLength = 12;
        p.strategy = STRAT_BTLAZY2;
    } else {
        u32 w = 19;
        while ((1u << w) < n && w < 21) w++;
        p.windowLog = w;
        p.chainLog = w;
        p.hashLog = w + 1;
        p.searchLog = 6;
        p.minMatch = 5;
        p.targetLength = 32;
        p.strategy = STRAT_LAZY2;
    }
    return p;
}

struct MatchState {
    const u8* base; // base + index; first source byte has index SQZ_WINDOW_START
    u32* hashTable;
    u32* chainTable;
    u8* tagTable;
    u32* hashTable3;
    u32* dummy; // 2 words of scratch in the workspace (the "dummy32" of libzstd)
    u32 hashLog3;
    u32 rowHashLog;
    u32 nextToUpdate;
    u32 lazySkipping;
    u32 fault; // set when a table yields an index that cannot exist (>= current position): the
               // frame is then reported as failed instead of emitting an undecodable match
    // ---- btlazy2 insert filter (see "dead-insert elision" below)
    u32 eager;       // 1 = every position is inserted (libzstd's literal behaviour)
    u32 nRepairs;    // schedule re-predictions so far in this chunk
    u32 anchorPos;   // parser anchor (index) at the current search
    u32 blockEndPos; // index one past the current block
    u32 chunkEndPos; // index one past the chunk
    u32* sched;      // predicted search positions (workspace)
    u32* liveList;   // bucket ids made live so far (workspace); rebuilds the Bloom filter per block
    u32* skipList;   // [from,to) position ranges libzstd itself never inserts (workspace)
    u32* blockTab;   // (start, end) index pairs of the blocks compressed so far (workspace)
    u32 nLive, nSkip, nBlocks;
    u32 specOff;     // 1 = speculative search epochs disabled
    u32 nxEnt;       // index of the entropy-table slot that is scratch during match finding
    u32 nEpoch, nSpecFail, nReplay; // statistics (trace words 34..36)
    // ---- whole-chunk no-match certificate (see cert_build / cert_scan)
    const u32* schedTab; // constant no-match schedules of a 128 KiB block (sched_tab_fill)
    u32 certCnt0;    // scheduled searches in block 0 of this chunk
    u32 nCert;       // scheduled searches in the whole chunk (blocks 0 and 1)
    u32 nCertDone;   // leading ones that were skipped as certainly failing
    u32 certEnd;     // every scheduled search of block 0 at a position < certEnd certainly fails; 0 = none
    u32 certEnd1;    // the same for block 1 (given that block 1 starts with repcode offset 1)
    u32* tailTab;        // table of the tail certificate (workspace; see tail_certify)
    u32* tailTabClean;   // -> the wave's flag: that table has been cleared in this launch
    u32 tailTried;       // tail certificates attempted in this chunk (at most SQZ_TAIL_TRIES)
    u32 tailAt, tailSeqs; // where the last one was attempted, and how many sequences the block had then
    u32 tailBlkTried;    // one has been attempted in the current block
    u32 sbMax;       // current super-batch length of the filtered insertion stream (256..2048)
    u32 genBits;     // btlazy2 hash-table generation tag of this chunk, in place (bits 19..29)
    u32 certPairs;   // lazy2 (chunks above 256 KiB): bit k = stage 1 certified that no search of blocks 2k, 2k + 1 finds a match
                     // (while the repcode offset is 1 and the blocks are the full 128 KiB ones) -- row_replay_nomatch
    CParams cp;
};

// hash-table word of the btlazy2 finder: head index + two flag bits
#define SQZ_HT_LIVE 0x80000000u
#define SQZ_HT_PEND 0x40000000u
#define SQZ_HT_IDX 0x3FFFFFFFu
// Generation tag: a btlazy2 chunk is at most 256 KiB, so an index needs 19 bits; bits 19..29 hold
// the number of the chunk (within this wave's launch) that wrote the word.  A word with another
// tag reads as 0 = "empty bucket", so the 2 MiB table is cleared once per launch and wave instead
// of once per chunk (SQZ_HT_GEN_MAX chunks later it is cleared again).
#define SQZ_HT_TAG 0x3FF80000u
#define SQZ_HT_GEN_SHIFT 19
#define SQZ_HT_GEN_MAX 2046u // 2047 is the tag of the all-ones SEARCHED mark, never used
#define SQZ_SCHED_CAP 4096u
#define SQZ_LIVE_CAP (96u << 10)
#define SQZ_SKIP_CAP 256u
#define SQZ_MAX_REPAIRS 24u
#define SQZ_BLOCKTAB_CAP 64u
#define SQZ_TAIL_TRIES 8u  // tail certificates per chunk (each later one first clears its 2 MiB table)
#ifndef SQZ_EPOCH_AFTER_FAILS
#define SQZ_EPOCH_AFTER_FAILS 16u // plain no-match searches in a row before speculative epochs start
#endif
#ifndef SQZ_EPOCH_MIN_K
#define SQZ_EPOCH_MIN_K 8u // an epoch needs at least this many scheduled searches left in the block
#endif
#define SQZ_DUBT_SEARCHED 0xFFFFFFFFu // sort mark of a deferred (not yet replayed) search node

SQ_DEV u32 ht_dec_g(u32 genBits, u32 w) { return ((w & SQZ_HT_TAG) == genBits) ? (w & ~SQZ_HT_TAG) : 0u; }
SQ_DEV u32 ht_enc_g(u32 genBits, u32 w) { return w | genBits; }
SQ_DEV u32 ht_dec(const MatchState& ms, u32 w) { return ht_dec_g(ms.genBits, w); }
SQ_DEV u32 ht_enc(const MatchState& ms, u32 w) { return ht_enc_g(ms.genBits, w); }

SQ_DEV u32 hash4(u32 v, u32 hBits) { return (v * 2654435761U) >> (32 - hBits); }
SQ_DEV u32 hash5(u64 v, u32 hBits) { return (u32)(((v << 24) * 889523592379ULL) >> (64 - hBits)); }
SQ_DEV u32 hash3(u32 v, u32 hBits) { return ((v << 8) * 506832829U) >> (32 - hBits); }

// ZSTD_storeSeq: literal run copied by the wave, sequence record by lane 0
SQ_DEV void storeSeq(SeqStore& ss, u32 litLength, const u8* literals, u32 offBase, u32 matchLength) {
    PROF_T0(t_cp);
    if (litLength) wv_memcpy(ss.lits + ss.nlits, literals, litLength);
    PROF_ADD(7, t_cp);
    ss.nlits += litLength;
    if (litLength > 0xFFFF) {
        ss.longLengthType = 1;
        ss.longLengthPos = ss.nseq;
    }
    u32 const mlBase = matchLength - SQZ_MINMATCH;
    if (mlBase > 0xFFFF) {
        ss.longLengthType = 2;
        ss.longLengthPos = ss.nseq;
    }
#if SQ_ON_DEVICE
    if (wv_lane() == 0) sq_gst64v((u32*)&ss.seqs[ss.nseq], offBase, (litLength & 0xFFFF) | (mlBase << 16));
#else
    ss.seqs[ss.nseq].offBase = offBase;
    ss.seqs[ss.nseq].litLength = (u16)litLength;
    ss.seqs[ss.nseq].mlBase = (u16)mlBase;
#endif
    ss.nseq++;
}

// ================================================================ DUBT (btlazy2)
#define SQZ_DUBT_UNSORTED 1u

// ---------------------------------------------------------------- dead-insert elision
// libzstd inserts EVERY position into the hash table / unsorted chain (ZSTD_updateDUBT).  The state
// of a hash bucket is observable only through a later search that lands in the same bucket, buckets
// do not interact, and all state dies with the chunk.  On incompressible data the lazy parser
// searches one position in `step` (step grows to 512) and > 99 % of the 2^19 buckets are never
// searched: their insertions are dead.  The finder therefore keeps buckets either LIVE (every
// insertion applied, exactly as libzstd) or untouched:
//   * at chunk start the parser's no-match search schedule is predicted (it depends only on the
//     anchor and block bounds); the buckets of those positions (+1, +2 for the depth-1/2 searches)
//     become LIVE; a 64 Kibit Bloom filter in LDS pre-filters positions so that the insertion
//     stream touches the 2 MiB table only for (probably) live buckets;
//   * a search that lands in a bucket that is not LIVE (a match changed the schedule) triggers a
//     repair: re-predict from the real parser state, mark the new buckets PENDING, replay the
//     chunk's earlier positions into exactly those buckets (a never-searched bucket is a pure
//     insertion chain, so late replay is equivalent), then flip them LIVE;
//   * after SQZ_MAX_REPAIRS repairs (compressible data) the finder replays everything once and
//     switches to eager mode for the rest of the chunk.
// Results are identical to inserting everything; only unobservable work is skipped.
#if SQ_ON_DEVICE
SQ_DEV void bloom_slots(u32 h, u32* w1, u32* m1, u32* w2, u32* m2) {
    u32 const g1 = h & 0xFFFFu, g2 = (h * 0x9E3779B1u) >> 16;
    *w1 = g1 >> 5;
    *m1 = 1u << (g1 & 31);
    *w2 = g2 >> 5;
    *m2 = 1u << (g2 & 31);
}
SQ_DEV bool bloom_test(u32 h) {
    u32 w1, m1, w2, m2;
    bloom_slots(h, &w1, &m1, &w2, &m2);
    const SQ_L u32* const B = (const SQ_L u32*)g_lds.u.bloom;
    return (B[w1] & m1) && (B[w2] & m2);
}
SQ_DEV void bloom_add(u32 h) {
    u32 w1, m1, w2, m2;
    bloom_slots(h, &w1, &m1, &w2, &m2);
    SQ_L u32* const B = (SQ_L u32*)g_lds.u.bloom;
    __hip_atomic_fetch_or(&B[w1], m1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(&B[w2], m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#else
SQ_DEV bool bloom_test(u32 h) {
    u32 const g1 = h & 0xFFFFu, g2 = (h * 0x9E3779B1u) >> 16;
    return (g_lds.u.bloom[g1 >> 5] & (1u << (g1 & 31))) && (g_lds.u.bloom[g2 >> 5] & (1u << (g2 & 31)));
}
SQ_DEV void bloom_add(u32 h) {
    u32 const g1 = h & 0xFFFFu, g2 = (h * 0x9E3779B1u) >> 16;
    g_lds.u.bloom[g1 >> 5] |= 1u << (g1 & 31);
    g_lds.u.bloom[g2 >> 5] |= 1u << (g2 & 31);
}
#endif
SQ_DEVFN void bloom_rebuild(MatchState& ms) {
    PROF_T0(t_br);
    wv_zero_lds(g_lds.u.bloom, SQZ_BLOOM_WORDS);
#if SQ_ON_DEVICE
    for (u32 i = wv_lane(); i < ms.nLive; i += 64) bloom_add(sq_gld32(ms.liveList + i));
    wv_mem_sync();
#else
    for (u32 i = 0; i < ms.nLive; i++) bloom_add(ms.liveList[i]);
#endif
    PROF_ADD(15, t_br);
}

#if SQ_ON_DEVICE
// Phase 1 of the filtered position stream, shared by the insertion sweep and the certificate scan:
// hash `lim` (<= 2048) consecutive positions starting at idx0, test them against the Bloom filter
// in LDS and leave the positives as (position, hash) pairs, in position order, in the 256-entry
// staging list (g_lds.grp).  Returns the number of positives (only the first 256 are stored).
// Rounds of 16 coalesced loads, then per group of 4 loads: hashes, all 8 Bloom probes issued back
// to back, ballot compaction.  FULL = exactly 2048 positions: no address clamping, no tail masks.
template <bool FULL>
SQ_DEV u32 filter_phase1(const u8* const base, u32 const idx0, u32 const lim, u32 const hashLog) {
    u32 const lane = wv_lane();
    u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    SQ_L u32* const stage = (SQ_L u32*)g_lds.grp;
    const SQ_L u32* const B = (const SQ_L u32*)g_lds.u.bloom;
    u32 const lastPos = idx0 + lim - 1;
    u32 cnt = 0;
#pragma unroll 1
    for (u32 r = 0; r < 2; r++) {
        if (!FULL && 1024 * r >= lim) break;
        u32 const rl = FULL ? 1024u : lim - 1024 * r; // positions of this round (may exceed 1024)
        u32 w[16];
#pragma unroll
        for (u32 u = 0; u < 16; u++) {
            u32 const pz = idx0 + 1024 * r + 64 * u + lane;
            w[u] = sq_grd32(base + ((FULL || pz < lastPos) ? pz : lastPos));
        }
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            if (!FULL && 256 * q >= rl) break;
            u32 p1[4], p2[4];
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                u32 const h = hash4(w[4 * q + k], hashLog);
                w[4 * q + k] = h;
                p1[k] = B[(h & 0xFFFFu) >> 5];
                p2[k] = B[(h * 0x9E3779B1u) >> 21];
            }
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                u32 const h = w[4 * q + k];
                u32 const u = 4 * q + k;
                bool const c = (FULL || (64 * u + lane < rl)) & (((p1[k] >> (h & 31)) & (p2[k] >> (((h * 0x9E3779B1u) >> 16) & 31)) & 1u) != 0);
                u64 const m = wv_ballot(c);
                u32 const slot = cnt + (u32)__popcll(m & below);
                if (c && slot < 256) {
                    stage[2 * slot] = idx0 + 1024 * r + 64 * u + lane;
                    stage[2 * slot + 1] = h;
                }
                cnt += (u32)__popcll(m);
            }
        }
        wv_fence();
    }
    return cnt;
}
#endif

// Insert positions [from, to) as unsorted candidates (ZSTD_updateDUBT) into the buckets selected
// by MODE: 0 = regular update (live buckets; all buckets in eager mode), 1 = repair replay
// (PENDING buckets), 2 = go-eager replay (every bucket that is not LIVE).
template <int MODE>
SQ_DEVFN void dubt_insert_range(MatchState& ms, u32 from, u32 to) {
    u32* const hashTable = ms.hashTable;
    u32 const hashLog = ms.cp.hashLog;
    u32* const bt = ms.chainTable;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
    const u8* const base = ms.base;
    bool const all = (MODE == 0) && ms.eager;
    // The fields of `ms` used inside the loops are read ONCE: this function is out of line, `ms` reaches it
    // as a generic pointer, and every flat_load makes hipcc drain vmcnt and lgkmcnt to 0 before its use --
    // with table stores pending, a full memory round trip per access.
    u32 const genBits = ms.genBits;
    u32 sbMax = ms.sbMax;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32 idx0 = from;
    if (MODE != 2 && !all) {
        // Filtered stream: super-batches of up to 2048 positions.  Phase 1 hashes all of them and keeps
        // the Bloom-positive ones (a few %) in an LDS staging list, in position order; phase 2
        // gathers their table words in one go and keeps the positions whose bucket qualifies;
        // phase 3 inserts those (usually < 64) with the ordered per-step procedure.  Memory round
        // trips per 2048 positions: ~4 instead of ~2 per 64.
        // staging: candidates (idx, h) in the wv_group scratch, survivors (idx, h, head) in the
        // "next" entropy-table slot, which is dead until the entropy stage of this block
        SQ_L u32* const stage = (SQ_L u32*)g_lds.grp;
        SQ_L u32* const surv = (SQ_L u32*)g_lds.scr;
        bool pendingStores = false;
        // ranges of <= 128 positions (the updates between two exact searches) go straight to the
        // per-64 loop below; everything longer is cut into super-batches of up to 2048 positions
        for (; idx0 < to && to - idx0 > 128;) {
            // super-batch length: adapts downwards when the Bloom filter has become dense enough
            // (after schedule repairs) to overflow the 256-entry candidate list
            u32 const lim = (to - idx0 < sbMax) ? to - idx0 : sbMax;
            PROF_T0(t_p1);
            u32 const cnt = (lim == 2048) ? filter_phase1<true>(base, idx0, lim, hashLog) : filter_phase1<false>(base, idx0, lim, hashLog);
            PROF_ADD(12, t_p1);
            PROF_CNT(18, 1);
            PROF_CNT(19, cnt);
            if (cnt > 256) {
                PROF_CNT(17, 1);
                if (sbMax > 256) { // retry this range with half the length
                    sbMax >>= 1;
                    continue;
                }
                break; // saturated filter: finish with the per-64 loop below
            }
            if (cnt <= 96 && lim == sbMax && sbMax < 2048) sbMax <<= 1; // sparse again: grow back
            wv_fence();
            PROF_T0(t_p23);
            if (pendingStores) { // the previous super-batch's table stores must land before we gather
                wv_mem_sync();
                pendingStores = false;
            }
            // phase 2: one round trip for all candidates' table words; survivors -> surv[]
            u32 cidx[4], ch[4], chead[4];
#pragma unroll
            for (u32 g = 0; g < 4; g++) {
                u32 const j = 64 * g + lane;
                cidx[g] = ch[g] = chead[g] = 0;
                if (j < cnt) {
                    cidx[g] = stage[2 * j];
                    ch[g] = stage[2 * j + 1];
                    chead[g] = ht_dec_g(genBits, sq_tld32(hashTable + ch[g]));
                }
            }
            u32 ntake = 0;
#pragma unroll
            for (u32 g = 0; g < 4; g++) {
                bool take = (64 * g + lane) < cnt;
                if (MODE == 0)
                    take = take && (chead[g] & SQZ_HT_LIVE);
                else
                    take = take && (chead[g] & SQZ_HT_PEND) && !(chead[g] & SQZ_HT_LIVE);
                u64 const m = wv_ballot(take);
                if (take) {
                    u32 const slot = ntake + (u32)__popcll(m & below);
                    surv[3 * slot] = cidx[g];
                    surv[3 * slot + 1] = ch[g];
                    surv[3 * slot + 2] = chead[g];
                }
                ntake += (u32)__popcll(m);
            }
            wv_fence();
            PROF_CNT(20, ntake);
            // phase 3: ordered insertion of the survivors, 64 per step (usually a single step that
            // reuses the heads gathered in phase 2)
            for (u32 j0 = 0; j0 < ntake; j0 += 64) {
                u32 const j = j0 + lane;
                bool const act = j < ntake;
                u32 idx = 0, h = 0, head = 0;
                if (act) {
                    idx = surv[3 * j];
                    h = surv[3 * j + 1];
                    head = surv[3 * j + 2];
                }
                if (j0 > 0) { // an earlier step of this super-batch may have rewritten these heads
                    wv_mem_sync();
                    if (act) head = ht_dec_g(genBits, sq_tld32(hashTable + h));
                }
                WvGroup const g = wv_group(h, act, g_lds.grp);
                u32 const prevPos = (u32)__shfl((int)idx, (int)(g.prevLane & 63), 64);
                if (act) {
                    u32 const prev = (g.prevLane < 64) ? prevPos : (head & SQZ_HT_IDX);
                    sq_gst64v(bt + 2 * (idx & btMask), prev, SQZ_DUBT_UNSORTED);
                    if (g.rank + 1 == g.gsize) sq_gst32(hashTable + h, ht_enc_g(genBits, idx | (head & ~SQZ_HT_IDX)));
                }
                pendingStores = true;
            }
            PROF_ADD(13, t_p23);
            idx0 += lim;
        }
        if (pendingStores) wv_mem_sync();
        ms.sbMax = sbMax;
    }
    PROF_T0(t_tail);
    // 64 positions per step.  A gather may hit a head written by the previous step and a load is
    // not ordered behind an older store of the same wave, so a step that stored drains first.
    for (; idx0 < to; idx0 += 64) {
        PROF_CNT(16, 1);
        u32 const idx = idx0 + lane;
        bool const active = idx < to;
        u32 h = 0;
        if (active) h = hash4(sq_grd32(base + idx), hashLog);
        bool cand = active;
        if (MODE != 2 && !all) cand = active && bloom_test(h);
        u32 head = 0;
        if (cand) head = ht_dec_g(genBits, sq_tld32(hashTable + h));
        bool take;
        if (MODE == 0)
            take = cand && (all || (head & SQZ_HT_LIVE));
        else if (MODE == 1)
            take = cand && (head & SQZ_HT_PEND) && !(head & SQZ_HT_LIVE);
        else
            take = cand && !(head & SQZ_HT_LIVE);
        if (wv_ballot(take) == 0) continue;
        WvGroup const g = wv_group(h, take, g_lds.grp);
        if (take) {
            u32 const prev = (g.prevLane < 64) ? idx0 + g.prevLane : (head & SQZ_HT_IDX);
            sq_gst64v(bt + 2 * (idx & btMask), prev, SQZ_DUBT_UNSORTED);
            if (g.rank + 1 == g.gsize) sq_gst32(hashTable + h, ht_enc_g(genBits, idx | (head & ~SQZ_HT_IDX)));
        }
        wv_mem_sync();
    }
    PROF_ADD(14, t_tail);
#else
    for (u32 idx = from; idx < to; idx++) {
        u32 const h = hash4(sq_rd32(base + idx), hashLog);
        if (MODE != 2 && !all && !bloom_test(h)) continue;
        u32 const head = ht_dec_g(genBits, hashTable[h]);
        bool take;
        if (MODE == 0)
            take = all || (head & SQZ_HT_LIVE);
        else if (MODE == 1)
            take = (head & SQZ_HT_PEND) && !(head & SQZ_HT_LIVE);
        else
            take = !(head & SQZ_HT_LIVE);
        if (!take) continue;
        hashTable[h] = ht_enc_g(genBits, idx | (head & ~SQZ_HT_IDX));
        bt[2 * (idx & btMask)] = head & SQZ_HT_IDX;
        bt[2 * (idx & btMask) + 1] = SQZ_DUBT_UNSORTED;
    }
#endif
}

// replay [2, to) minus the ranges libzstd never inserted
template <int MODE>
SQ_DEVFN void dubt_replay(MatchState& ms, u32 to) {
    u32 pos = SQZ_WINDOW_START;
    for (u32 k = 0; k < ms.nSkip && pos < to; k++) {
        u32 const a = gld32(ms.skipList + 2 * k), b = gld32(ms.skipList + 2 * k + 1);
        if (a > pos) dubt_insert_range<MODE>(ms, pos, a < to ? a : to);
        if (b > pos) pos = b;
    }
    if (pos < to) dubt_insert_range<MODE>(ms, pos, to);
}

// ---------------------------------------------------------------- no-match schedule tables
// While the lazy parser finds nothing, the positions it searches in a block depend only on where
// the block starts relative to the anchor: step = ((ip - anchor) >> 8) + 1.  Two constant tables
// (relative positions, ascending) describe any block of a chunk: "first" (the search starts one
// byte into the frame's first block) and "later"; a shorter block uses a prefix.  Layout (u32):
// [0] entries of "first", [1] entries of "later", then the two tables, SQZ_SCHEDTAB_N slots each.
#define SQZ_SCHEDTAB_N 2048u
// ... followed by one word per 256-byte level k (512 levels) of each table: the smallest scheduled
// relative position >= 256 k (17 bits, 0x1FFFF = none) | ceil(65536 / (k + 1)) << 17, which lets a
// lane decide with two multiplications whether one of its 16 positions is a scheduled one.
#define SQZ_SCHEDLVL_N 512u
#define SQZ_SCHEDLVL_OFF (2u + 2u * SQZ_SCHEDTAB_N)
// ... followed by the two tables once more as bitmaps, 16 positions (one u16) per 16-byte window of the block:
// for chunks that start on a 16-byte boundary the certificate scan reads a lane's scheduled positions
// with one 2-byte load instead of computing them.
#define SQZ_SCHEDBITS_N 8192u // 16-position windows of a 128 KiB block
#define SQZ_SCHEDBITS_OFF (SQZ_SCHEDLVL_OFF + 2u * SQZ_SCHEDLVL_N)
#define SQZ_SCHEDTAB_WORDS (SQZ_SCHEDBITS_OFF + 2u * (SQZ_SCHEDBITS_N / 2u))

// number of table entries < limit (entries ascending); whole wave, ~28 coalesced loads per lane
SQ_DEVFN u32 sched_count(const u32* const entries, u32 const nEntries, u32 const limit) {
#if SQ_ON_DEVICE
    u32 c = 0;
    for (u32 i = wv_lane(); i < nEntries; i += 64) c += (sq_gld32(entries + i) < limit) ? 1u : 0u;
    for (int sft = 32; sft >= 1; sft >>= 1) c += (u32)__shfl_xor((int)c, sft, 64);
    return wv_uni(c);
#else
    u32 c = 0;
    while (c < nEntries && entries[c] < limit) c++;
    return c;
#endif
}
// position (index) of the i-th scheduled search of the chunk, i < ms.nCert (a chunk of this
// strategy has at most two blocks)
SQ_DEV u32 cert_pos_t(const u32* schedTab, u32 certCnt0, u32 i) { // one load, no branch
    bool const first = i < certCnt0;
    u32 const slot = first ? i : SQZ_SCHEDTAB_N + (i - certCnt0);
    return sq_gld32(schedTab + 2 + slot) + (first ? SQZ_WINDOW_START : SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX);
}
SQ_DEV u32 cert_pos(const MatchState& ms, u32 i) { return cert_pos_t(ms.schedTab, ms.certCnt0, i); }

// Searches skipped under the no-match certificate were never executed, so the buckets they would
// have re-sorted hold them (after a replay) as plain unsorted nodes.  Give them the mark of a
// deferred search, exactly what a speculative epoch leaves behind; dubt_materialize replays them if
// the bucket is ever searched.  MODE 1: buckets being made live by the current repair (PENDING);
// MODE 2: every bucket that was not live when the finder went eager.
template <int MODE>
SQ_DEVFN void cert_mark_deferred(MatchState& ms) {
    if (ms.nCertDone == 0) return;
    u32* const bt = ms.chainTable;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
#if SQ_ON_DEVICE
    for (u32 i = wv_lane(); i < ms.nCertDone; i += 64) {
        u32 const s = cert_pos(ms, i);
        if (s >= ms.nextToUpdate) continue; // not inserted yet (cannot happen: the search inserts it)
        u32 const h = hash4(sq_grd32(ms.base + s), ms.cp.hashLog);
        u32 const v = ht_dec(ms, sq_tld32(ms.hashTable + h));
        bool const mine = (MODE == 1) ? ((v & SQZ_HT_PEND) && !(v & SQZ_HT_LIVE)) : !(v & SQZ_HT_LIVE);
        if (mine) sq_gst32(bt + 2 * (s & btMask) + 1, SQZ_DUBT_SEARCHED);
    }
    wv_mem_sync();
#else
    for (u32 i = 0; i < ms.nCertDone; i++) {
        u32 const s = cert_pos(ms, i);
        if (s >= ms.nextToUpdate) continue;
        u32 const h = hash4(sq_rd32(ms.base + s), ms.cp.hashLog);
        u32 const v = ht_dec(ms, ms.hashTable[h]);
        bool const mine = (MODE == 1) ? ((v & SQZ_HT_PEND) && !(v & SQZ_HT_LIVE)) : !(v & SQZ_HT_LIVE);
        if (mine) bt[2 * (s & btMask) + 1] = SQZ_DUBT_SEARCHED;
    }
#endif
}

SQ_DEVFN void filter_go_eager(MatchState& ms) {
    if (ms.eager) return;
    dubt_replay<2>(ms, ms.nextToUpdate);
    cert_mark_deferred<2>(ms);
    ms.eager = 1;
}

// positions [from, to) are never inserted by libzstd (nextToUpdate jumped over them)
SQ_DEV void filter_add_skip(MatchState& ms, u32 from, u32 to) {
    if (ms.eager || ms.cp.strategy != STRAT_BTLAZY2 || to <= from) return;
    if (ms.nSkip >= SQZ_SKIP_CAP) {
        filter_go_eager(ms);
        return;
    }
    gst32(ms.skipList + 2 * ms.nSkip, from);
    gst32(ms.skipList + 2 * ms.nSkip + 1, to);
    wv_mem_sync();
    ms.nSkip++;
}

// filter_predict through the tables, for the two states the parser is in when a prediction is
// needed after a hand-over or a match: `pos` is a scheduled position of its block's own table
// (anchor = chunk start) or of the "later" table laid at the anchor (after a sequence ip == anchor).
// Returns the number of predicted positions, or 0xFFFFFFFF when (pos, anc) is not on such a lattice.
SQ_DEVFN u32 filter_predict_tab(MatchState& ms, u32 pos, u32 anc, u32* const sched) {
    bool const firstVar = anc == SQZ_WINDOW_START;
    const u32* const ent = ms.schedTab + 2 + (firstVar ? 0u : SQZ_SCHEDTAB_N);
    u32 const nEnt = gld32(ms.schedTab + (firstVar ? 0 : 1));
    u32 const rel = pos - anc;
    u32 const j0 = sched_count(ent, nEnt, rel);
    if (j0 >= nEnt || gld32(ent + j0) != rel) return 0xFFFFFFFFu;
    u32 const lim0 = ms.blockEndPos - 8 - anc; // scheduled while anc + rel < blockEnd - 8
    u32 const j1 = sched_count(ent, nEnt, lim0);
    u32 const cnt0 = j1 > j0 ? j1 - j0 : 0u;
    // at most one more block in a chunk of this strategy
    u32 cnt1 = 0;
    u32 const b1 = ms.blockEndPos;
    const u32* const later = ms.schedTab + 2 + SQZ_SCHEDTAB_N;
    if (b1 < ms.chunkEndPos) {
        u32 const len1 = (ms.chunkEndPos - b1 < SQZ_BLOCKSIZE_MAX) ? ms.chunkEndPos - b1 : SQZ_BLOCKSIZE_MAX;
        if (b1 + len1 < ms.chunkEndPos) return 0xFFFFFFFFu; // a third block: let the serial loop do it
        if (len1 >= 9) cnt1 = sched_count(later, gld32(ms.schedTab + 1), len1 - 8);
    }
    if (cnt0 + cnt1 >= SQZ_SCHED_CAP) return 0xFFFFFFFFu;
#if SQ_ON_DEVICE
    for (u32 i = wv_lane(); i < cnt0; i += 64) sq_gst32(sched + i, anc + sq_gld32(ent + j0 + i));
    for (u32 i = wv_lane(); i < cnt1; i += 64) sq_gst32(sched + cnt0 + i, b1 + sq_gld32(later + i));
    wv_mem_sync();
#else
    for (u32 i = 0; i < cnt0; i++) sched[i] = anc + ent[j0 + i];
    for (u32 i = 0; i < cnt1; i++) sched[cnt0 + i] = b1 + later[i];
#endif
    return cnt0 + cnt1;
}

// The lazy parser's no-match search schedule from (pos, anchor) to the end of the chunk.
SQ_DEVFN u32 filter_predict(MatchState& ms, u32 pos, u32 anc, u32* const sched) {
    {
        u32 const nt = filter_predict_tab(ms, pos, anc, sched);
        if (nt != 0xFFFFFFFFu) return nt;
    }
    WV_SERIAL {
        u32 n = 0, p = pos, a = anc, bEnd = ms.blockEndPos;
        for (;;) {
            u32 const ilimit = bEnd - 8;
            while (p < ilimit && n < SQZ_SCHED_CAP) {
                sched[n++] = p;
                p += ((p - a) >> 8) + 1;
            }
            if (bEnd >= ms.chunkEndPos || n >= SQZ_SCHED_CAP) break;
            u32 nb = bEnd + SQZ_BLOCKSIZE_MAX;
            if (nb > ms.chunkEndPos) nb = ms.chunkEndPos;
            if (nb - bEnd < 9) break; // too small to search
            p = a = bEnd;
            bEnd = nb;
        }
        g_lds.ret[0] = n;
    }
    wv_mem_sync();
    return g_lds.ret[0];
}

// Buckets of the n predicted positions (and +1, +2) become PENDING; returns nothing, appends to
// liveList / Bloom.  Caller guarantees capacity.  `first`: the table is still all zero, so the
// words are stored blindly as LIVE (no gather, no finalize pass).
SQ_DEVFN void filter_mark(MatchState& ms, u32 n, bool first) {
    u32 const total = n * 3;
    u32 const hashLog = ms.cp.hashLog;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    if (first) {
        u64 const belowm = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (u32 i0 = 0; i0 < total; i0 += 256) {
            u32 pp[4], ww[4];
            bool ok[4];
#pragma unroll
            for (u32 g = 0; g < 4; g++) {
                u32 const i = i0 + 64 * g + lane;
                pp[g] = 0;
                if (i < total) pp[g] = sq_gld32(ms.sched + i / 3) + (i % 3);
            }
#pragma unroll
            for (u32 g = 0; g < 4; g++) {
                u32 const i = i0 + 64 * g + lane;
                ok[g] = (i < total) && (pp[g] + 8 <= ms.chunkEndPos);
                ww[g] = 0;
                if (ok[g]) ww[g] = sq_grd32(ms.base + pp[g]);
            }
#pragma unroll
            for (u32 g = 0; g < 4; g++) {
                u32 const h = hash4(ww[g], hashLog);
                u64 const m = wv_ballot(ok[g]);
                if (ok[g]) {
                    u32 const r = (u32)__popcll(m & belowm);
                    sq_gst32(ms.hashTable + h, ht_enc(ms, SQZ_HT_LIVE));
                    sq_gst32(ms.liveList + ms.nLive + r, h);
                    bloom_add(h);
                }
                ms.nLive += (u32)__popcll(m);
            }
        }
        wv_mem_sync();
        return;
    }
    // 256 per step, every load unconditional (clamped index) so that the dependent loads (schedule ->
    // bytes -> table word) of the four groups overlap.  A bucket met twice before its PENDING store
    // has landed is simply listed twice (the list only feeds idempotent passes).
    u64 const belowm = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32 const iLast = total ? total - 1 : 0u;
    for (u32 i0 = 0; i0 < total; i0 += 256) {
        u32 pp[4], hh[4], vv[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            u32 const i = i0 + 64 * g + lane;
            u32 const ic = i < total ? i : iLast;
            pp[g] = sq_gld32(ms.sched + ic / 3) + (ic % 3);
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            u32 const pc = pp[g] + 8 <= ms.chunkEndPos ? pp[g] : ms.chunkEndPos - 8;
            hh[g] = hash4(sq_grd32(ms.base + pc), hashLog);
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) vv[g] = ht_dec(ms, sq_tld32(ms.hashTable + hh[g]));
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            bool const fresh = (i0 + 64 * g + lane < total) && (pp[g] + 8 <= ms.chunkEndPos) &&
                               !(vv[g] & (SQZ_HT_LIVE | SQZ_HT_PEND));
            u64 const m = wv_ballot(fresh);
            if (fresh) {
                sq_gst32(ms.hashTable + hh[g], ht_enc(ms, vv[g] | SQZ_HT_PEND));
                sq_gst32(ms.liveList + ms.nLive + (u32)__popcll(m & belowm), hh[g]);
                bloom_add(hh[g]);
            }
            ms.nLive += (u32)__popcll(m);
        }
        wv_mem_sync();
    }
#else
    for (u32 i = 0; i < total; i++) {
        u32 const p = ms.sched[i / 3] + (i % 3);
        if (p + 8 > ms.chunkEndPos) continue;
        u32 const h = hash4(sq_rd32(ms.base + p), hashLog);
        u32 const v = ht_dec(ms, ms.hashTable[h]);
        if (first) {
            ms.hashTable[h] = ht_enc(ms, SQZ_HT_LIVE);
            ms.liveList[ms.nLive++] = h;
            bloom_add(h);
            continue;
        }
        if (!(v & (SQZ_HT_LIVE | SQZ_HT_PEND))) {
            ms.hashTable[h] = ht_enc(ms, v | SQZ_HT_PEND);
            ms.liveList[ms.nLive++] = h;
            bloom_add(h);
        }
    }
#endif
}
// PENDING -> LIVE for liveList[start, nLive)
SQ_DEVFN void filter_finalize(MatchState& ms, u32 start) {
#if SQ_ON_DEVICE
    for (u32 i = start + wv_lane(); i < ms.nLive; i += 64) {
        u32 const h = sq_gld32(ms.liveList + i);
        u32 const v = ht_dec(ms, sq_tld32(ms.hashTable + h));
        sq_gst32(ms.hashTable + h, ht_enc(ms, (v & ~SQZ_HT_PEND) | SQZ_HT_LIVE));
    }
    wv_mem_sync();
#else
    for (u32 i = start; i < ms.nLive; i++) {
        u32 const h = ms.liveList[i];
        ms.hashTable[h] = ht_enc(ms, (ht_dec(ms, ms.hashTable[h]) & ~SQZ_HT_PEND) | SQZ_HT_LIVE);
    }
#endif
}
// (re)predict from the real parser state and make the predicted buckets live
SQ_DEVFN void filter_repair(MatchState& ms, u32 pos, u32 anc, bool first) {
    if (ms.eager) return;
    if (!first) {
        if (ms.nRepairs >= SQZ_MAX_REPAIRS) {
            filter_go_eager(ms);
            return;
        }
        ms.nRepairs++;
    }
    u32 const n = filter_predict(ms, pos, anc, ms.sched);
    if (ms.nLive + 3 * n > SQZ_LIVE_CAP) {
        filter_go_eager(ms);
        return;
    }
    u32 const start = ms.nLive;
    filter_mark(ms, n, first);
    if (!first && ms.nLive > start) {
        if (ms.nextToUpdate > SQZ_WINDOW_START) dubt_replay<1>(ms, ms.nextToUpdate);
        cert_mark_deferred<1>(ms);
        filter_finalize(ms, start);
    }
}

// ---------------------------------------------------------------- waking the finder for one search
// After a certified prefix the finder is asleep: no bucket is live, nothing has been inserted.  The
// exact code then has ONE search to do, at `curr` (plus, if it finds a match, the depth searches at
// curr + 1 and curr + 2); afterwards the tail certificate usually ends the block.  Predicting the whole
// remaining schedule and replaying every earlier position through the live-bucket filter for that
// (filter_repair) is most of such a chunk's time.  Instead only the buckets of curr, curr+1, curr+2
// become live: one streaming pass finds the earlier positions that hash into them (one or two, typically),
// they are inserted in order -- as deferred searches where they were certified searches -- and that is
// all.  Everything else stays consistent with "live buckets hold every insertion": a later search in
// another bucket triggers the ordinary repair.
#define SQZ_WAKE_CAP 48u

// is position p one of the first ms.nCertDone scheduled searches (those the certificate skipped)?
SQ_DEVFN bool cert_was_skipped(MatchState& ms, u32 p) {
    bool const second = p >= SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX && ms.nCert > ms.certCnt0;
    const u32* const ent = ms.schedTab + 2 + (second ? SQZ_SCHEDTAB_N : 0u);
    u32 const nEnt = second ? ms.nCert - ms.certCnt0 : ms.certCnt0;
    u32 const rel = p - (second ? SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX : SQZ_WINDOW_START);
    u32 const j = sched_count(ent, nEnt, rel);
    if (j >= nEnt || gld32(ent + j) != rel) return false;
    return (second ? ms.certCnt0 + j : j) < ms.nCertDone;
}

// Returns false when this shortcut does not apply (the caller falls back to filter_repair).
SQ_DEVFN bool filter_wake(MatchState& ms, u32 curr) {
    u32 const hashLog = ms.cp.hashLog;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
    u32* const bt = ms.chainTable;
    const u8* const base = ms.base;
    u32 const to = ms.nextToUpdate; // positions [2, to) minus the skipped ranges are "inserted"
    u32 hb[3];
    u32 nb = 0;
    for (u32 k = 0; k < 3; k++)
        if (curr + k + 8 <= ms.chunkEndPos) hb[nb++] = hash4(srd32(base + curr + k), hashLog);
    if (nb == 0 || ms.nLive + 3 > SQZ_LIVE_CAP) return false;
    for (u32 k = nb; k < 3; k++) hb[k] = hb[0];
    // ---- the earlier positions that hash into one of the buckets, in LDS: (position, bucket hash)
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    SQ_L u32* const found = (SQ_L u32*)g_lds.grp;
    u32 nFound = 0;
    {
        const u8* const src0 = base + SQZ_WINDOW_START;
        const u8* const a0 = src0 - ((size_t)src0 & 15);
        s32 const p0 = (s32)SQZ_WINDOW_START - (s32)(src0 - a0);
        const u8* const lastByte = base + ms.chunkEndPos - 1;
        const u8* const aLast = lastByte - ((size_t)lastByte & 15);
        auto loadRound = [&](s32 rp) -> sq_v4u {
            const u8* ad = a0 + (size_t)(u32)(rp - p0) + 16 * lane;
            return sq_grd128_s(ad < aLast ? ad : aLast);
        };
        for (s32 pos = p0; pos < (s32)to; pos += 4096) { // four rounds of 1 KiB per step, loads first
            sq_v4u w[5];
#pragma unroll
            for (u32 r = 0; r < 5; r++) w[r] = loadRound(pos + 1024 * (s32)r);
#pragma unroll
            for (u32 r = 0; r < 4; r++) {
                s32 const rp = pos + 1024 * (s32)r;
                if (rp >= (s32)to) break;
                u32 const nx = (u32)__builtin_amdgcn_update_dpp(0, (int)w[r].x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
                u32 const n63 = wv_uni(w[r + 1].x);
                u32 const d[5] = {w[r].x, w[r].y, w[r].z, w[r].w, (lane == 63) ? n63 : nx};
                s32 const lo = rp + (s32)(16 * lane);
                u32 cm = 0;
#pragma unroll
                for (u32 q = 0; q < 4; q++) {
                    u32 v[4];
                    v[0] = d[q];
                    v[1] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
                    v[2] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
                    v[3] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
#pragma unroll
                    for (u32 k = 0; k < 4; k++) {
                        u32 const h = hash4(v[k], hashLog);
                        cm |= ((h == hb[0] || h == hb[1] || h == hb[2]) ? 1u : 0u) << (4 * q + k);
                    }
                }
                { // positions in [2, to)
                    s32 f = (s32)SQZ_WINDOW_START - lo, l = (s32)to - lo;
                    f = f < 0 ? 0 : (f > 16 ? 16 : f);
                    l = l < 0 ? 0 : (l > 16 ? 16 : l);
                    cm &= ((1u << l) - 1u) & ~((1u << f) - 1u);
                }
                while (wv_ballot(cm != 0)) {
                    u64 const m = wv_ballot(cm != 0);
                    if (cm) {
                        u32 const j = (u32)__builtin_ctz(cm);
                        cm &= cm - 1;
                        u32 const slot = nFound + (u32)__popcll(m & below);
                        if (slot < SQZ_WAKE_CAP) found[slot] = (u32)(lo + (s32)j);
                    }
                    nFound += (u32)__popcll(m);
                }
            }
        }
        wv_fence();
    }
    if (nFound > SQZ_WAKE_CAP) return false; // crowded buckets: the ordinary repair
#else
    u32 found[SQZ_WAKE_CAP];
    u32 nFound = 0;
    for (u32 p = SQZ_WINDOW_START; p < to; p++) {
        u32 const h = hash4(sq_rd32(base + p), hashLog);
        if (h == hb[0] || h == hb[1] || h == hb[2]) {
            if (nFound < SQZ_WAKE_CAP) found[nFound] = p;
            nFound++;
        }
    }
    if (nFound > SQZ_WAKE_CAP) return false;
#endif
    // ---- the three buckets become live (they are empty: nothing has been inserted anywhere yet)
    u32 const start = ms.nLive;
    for (u32 k = 0; k < nb; k++) {
        bool fresh = true;
        for (u32 j = 0; j < k; j++) fresh = fresh && hb[j] != hb[k];
        if (!fresh) continue;
        gst32(ms.hashTable + hb[k], ht_enc(ms, SQZ_HT_PEND));
        gst32(ms.liveList + ms.nLive, hb[k]);
        ms.nLive++;
    }
    wv_mem_sync();
    // ---- insert what was found, in ascending position order (the list is short: pick the minimum each time)
    u32 last = 0;
    for (u32 done = 0; done < nFound; done++) {
        u32 p = 0xFFFFFFFFu;
#if SQ_ON_DEVICE
        for (u32 i = lane; i < nFound; i += 64) {
            u32 const c = found[i];
            if (c > last && c < p) p = c;
        }
        for (int sft = 32; sft >= 1; sft >>= 1) {
            u32 const y = (u32)__shfl_xor((int)p, sft, 64);
            p = y < p ? y : p;
        }
        p = wv_uni(p);
#else
        for (u32 i = 0; i < nFound; i++)
            if (found[i] > last && found[i] < p) p = found[i];
#endif
        last = p;
        bool skipped = false; // positions libzstd never inserted (nextToUpdate jumped over them)
        for (u32 k = 0; k < ms.nSkip; k++) {
            u32 const a = gld32(ms.skipList + 2 * k), b = gld32(ms.skipList + 2 * k + 1);
            skipped = skipped || (p >= a && p < b);
        }
        if (skipped) continue;
        u32 const h = hash4(srd32(base + p), hashLog);
        u32 const head = ht_dec(ms, gld32(ms.hashTable + h));
        // a certified (skipped) search sits in its bucket as a deferred search, like after a speculative epoch
        u32 const mark = cert_was_skipped(ms, p) ? SQZ_DUBT_SEARCHED : SQZ_DUBT_UNSORTED;
        gst32(bt + 2 * (p & btMask), head & SQZ_HT_IDX);
        gst32(bt + 2 * (p & btMask) + 1, mark);
        gst32(ms.hashTable + h, ht_enc(ms, p | (head & ~SQZ_HT_IDX)));
        wv_mem_sync();
    }
    filter_finalize(ms, start);
#if SQ_ON_DEVICE
    for (u32 k = 0; k < nb; k++) bloom_add(hb[k]); // idempotent
    wv_fence();
#else
    for (u32 k = 0; k < nb; k++) bloom_add(hb[k]);
#endif
    return true;
}

// ---------------------------------------------------------------- no-match certificate
// On incompressible data every scheduled search fails and the parser emits no sequence at all: the
// whole tree machinery produces nothing observable.  A search of the lazy parser at position s can
// only return a match (length >= 4) if some EARLIER position p holds the same 4 bytes as s, or if
// the repcode test at s+1 fires.  Both are properties of the bytes, not of the tree.  So before
// any table is built, one streaming pass decides, for the predicted no-match schedule S of the
// whole chunk, the first scheduled position s* for which such a p (or a repcode hit) exists:
//   build : table[hash(s)] = (s, its 4 bytes) for s in S (8-byte entries, position word generation-
//           tagged, in the chain-table area), Bloom filter over the 4-byte values in LDS; the few s
//           that share an entry with another s go to a small overflow list;
//   scan  : every position p < certEnd is hashed and Bloom-tested (64 per instruction); the ~3 %
//           positives gather their table entry and compare bytes; a hit at (p, s), p < s, lowers
//           certEnd to s.
// Searches at positions < certEnd certainly fail: the parser skips them (they are remembered as
// deferred searches, like those of a speculative epoch) and resumes the exact code at certEnd.
// Blocks that are certified entirely cost no search, no insertion and no table traffic at all.
// The certificate is conservative (p is not required to be reachable within nbCompares), so a
// "maybe" only means the exact code decides; results are identical to libzstd's either way.
// Bloom filter of the certificate scan: keyed by the 4 BYTES themselves (probe 1 from the low
// half, probe 2 from the high half), so a scanned position costs no multiplication; only the few
// positives are hashed.
SQ_DEV void vbloom_add(SQ_L u32* const B, u32 v) {
    u32 const w1 = (v & 0xFFFFu) >> 5, m1 = 1u << (v & 31), w2 = v >> 21, m2 = 1u << ((v >> 16) & 31);
#if SQ_ON_DEVICE
    __hip_atomic_fetch_or(&B[w1], m1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(&B[w2], m2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    B[w1] |= m1;
    B[w2] |= m2;
#endif
}

// Everything cert_compute_core needs; free of the compressor's MatchState / LDS image so that the
// incompressible-chunk kernel (sq_zstd_fast.h) can run it at a higher occupancy.
struct CertCtx {
    const u8* base;      // base + index; the chunk's first byte has index SQZ_WINDOW_START
    u32* E;              // certificate table: 2^eLog entries of 8 bytes (global memory)
    u32 eLog;
    u32 genBits;         // generation tag of this chunk's entries, in place (bits 19..29)
    const u32* schedTab; // constant no-match schedules
    u32 certCnt0, nCert; // scheduled searches in block 0 / in the chunk
    u32 base0Pos;        // stage 1 (cert_lds_core) only: where the first certified block starts (SQZ_WINDOW_START for a chunk of
                         // at most two blocks; a later block of a large chunk, whose earlier blocks are then only scanned)
    u32 tab0;            //   ... which schedule table that block uses: 0 "first" (the frame's first block), 1 "later" (3: see below)
    u32 tail;            //   ... searches stop this many bytes before a block's end: 8, row-hash finder 16
    u32 hashBits;        //   ... 0: any earlier position with a search's 4 bytes breaks the certificate (btlazy2); else the
                         //   row-hash finder's rule: only one whose 5-byte hash (this many bits: row + tag) is the search's too
    u32 base1Pos;        // position the "later" table is laid at (block 1's start; an anchor for a tail certificate)
    u32 tabOff1;         // first entry of the "later" table that is used
    u32 repOff;          // repcode offset of the searches' repcode test (1 at the start of a frame)
    u32 chunkEndPos;
    SQ_L u32* bloom;     // SQZ_BLOOM_WORDS words  (LDS)
    SQ_L u32* cand;      // 256 words              (LDS)
    SQ_L u32* ovf;       // 2 * SQZ_CERT_OVF_CAP   (LDS)
    u32 certEnd, certEnd1, nOvf; // results
    SQ_L u32* lvl;       // 2 * SQZ_SCHEDLVL_N level words in LDS, or nullptr: when given, the scan leaves
                         // the scheduled positions themselves out (their mutual repeats are found
                         // while the table is built)
    u32 rawEnd;          // stage 1 only: position up to which the scan has written the raw blocks' payload
    u64* hashPool;       // stage 1 only, or null: the fused XXH3 -- where the scan puts the sums of every full 1 KiB block
    u8* hashedFlag;      //   ... the byte that says they are all there
    u32 hashDone;        // result: the sums are complete
    u32 dbg;             // timing experiments (SQ_FAST_DBG): 1 = no build, 2 = no scan, 8 = build without the exchange
    // stage 1's SECOND pass only (fastpath_blocks: the tail certificate of a block that one short match keeps from being
    // certified; zero otherwise):
    // there tab0 = 3: one block, its schedule (the "later" table) laid at base0Pos = the anchor behind the match, which may
    // sit at any byte offset; the last two of the nCert entries are not from the table: the lazy parser's two depth searches
    // at wakeLim + 1 and wakeLim + 2.  (Few fields, and none that the first pass could not do without: they are live
    // through the scan, which is at its limit of registers.)
    u32 wakeLim;         // != 0: the search the match was found at; report in wakeBad whether a position p < wakeLim other than wakeSkip shares its first byte
    u32 wakeSkip;        //    with the search at wakeLim (wakeByte) AND hashes into that search's bucket (wakeHash, wakeHashLog
    u32 wakeHash, wakeHashLog, wakeByte; // bits): such a node could leave bestLength > 0 before the walk reaches the match
    u32 wakeBad;         // result
};

#define SQZ_CERT_DUP SQZ_HT_PEND // table-word flag: another scheduled position shares this bucket
#define SQZ_CERT_OVF_CAP 64u

SQ_DEV u32 cert_pos_c(const CertCtx& cx, u32 i) { // position of the i-th scheduled search; one load, no branch
    bool const first = i < cx.certCnt0;
    u32 const slot = first ? i : SQZ_SCHEDTAB_N + cx.tabOff1 + (i - cx.certCnt0);
    return sq_gld32(cx.schedTab + 2 + slot) + (first ? SQZ_WINDOW_START : cx.base1Pos);
}
// the same for stage 1's generalised form (base0Pos / tab0)
SQ_DEV u32 cert_pos_g(const CertCtx& cx, u32 i) {
    bool const x = (cx.tab0 & 2u) && i + 2 >= cx.nCert; // (second pass: the two searches that are not on the schedule)
    u32 const j = x ? 0u : i;
    bool const first = j < cx.certCnt0;
    u32 const slot = first ? (cx.tab0 ? SQZ_SCHEDTAB_N : 0u) + j : SQZ_SCHEDTAB_N + cx.tabOff1 + (j - cx.certCnt0);
    u32 const sp = sq_gld32(cx.schedTab + 2 + slot) + (first ? cx.base0Pos : cx.base1Pos);
    return x ? cx.wakeLim + 3 - (cx.nCert - i) : sp;
}

SQ_DEVFN void cert_compute_core(CertCtx& cx) {
    // certificate table: 2^eLog entries of (position | generation tag [| DUP], the 4 bytes there),
    // 8 bytes each, in the chain-table area (which the finder only reads through valid heads)
    u32* const E = cx.E;
    u32 const eLog = cx.eLog;
    const u8* const base = cx.base;
    u32 const nCert = cx.nCert;
    // one certificate per block: a match in block 0 changes that block's schedule, not block 1's
    u32 const blk1 = (nCert > cx.certCnt0) ? cx.base1Pos : 0xFFFFFFFFu;
    u32 certEnd = cx.chunkEndPos, certEnd1 = cx.chunkEndPos;
    u32 nOvf = 0;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 i = lane; i < SQZ_BLOOM_WORDS; i += 64) cx.bloom[i] = 0;
    wv_fence();
    u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    SQ_L u32* const ovf = cx.ovf;
    u32 vmin = 0xFFFFFFFFu, vmin1 = 0xFFFFFFFFu; // lane-local minima of the violating scheduled positions
    auto note = [&](u32 sviol, u32& v0, u32& v1) { // a search at sviol may succeed
        if (sviol < blk1)
            v0 = sviol < v0 ? sviol : v0;
        else
            v1 = sviol < v1 ? sviol : v1;
    };
    // ---- build, one pass over S: every s swaps its entry (position | tag, bytes) into the table with
    // one 8-byte atomic exchange; what comes back is either stale (another generation: the slot was
    // free) or the entry of another s of this chunk that it just displaced.  A displaced entry goes
    // to the overflow list (its slot now carries the DUP flag); if it holds the same 4 bytes, the
    // later of the two searches can succeed.  The repcode test (offset 1) rides along.
    // 512 scheduled positions per step, all loads unconditional (indices clamped) so that the
    // dependent accesses (schedule table -> bytes -> exchange) of the eight groups overlap.
    u32 const iLast = nCert - 1;
    for (u32 i0 = 0; i0 < ((cx.dbg & 1) ? 0u : nCert); i0 += 512) {
        u32 sp[8], v[8], v1[8];
        u64 old[8];
#pragma unroll
        for (u32 g = 0; g < 8; g++) {
            u32 const i = i0 + 64 * g + lane;
            sp[g] = cert_pos_c(cx, i < nCert ? i : iLast);
        }
        // repcode test of the search at s: read32(s + 1 - repOff) == read32(s + 1); for repOff == 1
        // the left side is the 4 bytes at s themselves
        u32 vr[8];
#pragma unroll
        for (u32 g = 0; g < 8; g++) {
            v[g] = sq_grd32(base + sp[g]);
            v1[g] = sq_grd32(base + sp[g] + 1);
            vr[g] = (cx.repOff > 1) ? sq_grd32(base + sp[g] + 1 - cx.repOff) : v[g];
        }
#pragma unroll
        for (u32 g = 0; g < 8; g++) {
            old[g] = 0;
            if (i0 + 64 * g + lane < nCert) {
                u64 const mine = ((u64)v[g] << 32) | ht_enc_g(cx.genBits, sp[g]);
                if (cx.dbg & 16) // timing experiment: plain store instead of the exchange
                    sq_gst64v(E + 2 * hash4(v[g], eLog), (u32)mine, (u32)(mine >> 32));
                else if (!(cx.dbg & 8))
                    old[g] = __hip_atomic_exchange((SQ_G u64*)(E + 2 * hash4(v[g], eLog)), mine, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                vbloom_add(cx.bloom, v[g]);
                if (cx.repOff && v1[g] == vr[g]) note(sp[g], vmin, vmin1); // repcode at s+1
            }
        }
#pragma unroll
        for (u32 g = 0; g < 8; g++) {
            u32 const od = ht_dec_g(cx.genBits, (u32)old[g]); // 0 unless an entry of this chunk
            bool const displaced = od != 0;
            u32 const opos = od & SQZ_HT_IDX, oval = (u32)(old[g] >> 32);
            if (displaced) {
                __hip_atomic_fetch_or((SQ_G u32*)(E + 2 * hash4(v[g], eLog)), SQZ_CERT_DUP, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
                if (oval == v[g]) note(opos > sp[g] ? opos : sp[g], vmin, vmin1);
            }
            u64 const m = wv_ballot(displaced);
            if (m) {
                if (displaced) {
                    u32 const slot = nOvf + (u32)__popcll(m & below);
                    if (slot < SQZ_CERT_OVF_CAP) {
                        ovf[2 * slot] = opos;
                        ovf[2 * slot + 1] = oval;
                    } else
                        note(opos, vmin, vmin1); // no room to remember it: the certificate ends there
                }
                nOvf += (u32)__popcll(m);
            }
        }
    }
    if (nOvf > SQZ_CERT_OVF_CAP) nOvf = SQZ_CERT_OVF_CAP;
    wv_mem_sync();
    // scheduled positions with equal bytes that were not neighbours in a displacement chain: every
    // displaced entry against the entry that now sits in its slot and against the other displaced ones
    if (nOvf) {
        for (u32 k0 = 0; k0 < nOvf; k0 += 64) {
            u32 const k = k0 + lane;
            if (k < nOvf) {
                u32 const so = ovf[2 * k], vo = ovf[2 * k + 1];
                u64 const raw = __hip_atomic_load((const SQ_G u64*)(E + 2 * hash4(vo, eLog)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u32 const t = ht_dec_g(cx.genBits, (u32)raw);
                if (t && (u32)(raw >> 32) == vo) note((t & SQZ_HT_IDX) > so ? (t & SQZ_HT_IDX) : so, vmin, vmin1);
                for (u32 j = 0; j < nOvf; j++) {
                    u32 const sj = ovf[2 * j], vj = ovf[2 * j + 1];
                    if (j != k && vj == vo) note(sj > so ? sj : so, vmin, vmin1);
                }
            }
        }
    }
    auto settle = [&](u32 v0, u32 v1) { // wave minima of the violations -> certEnd / certEnd1
        if (wv_ballot((v0 & v1) != 0xFFFFFFFFu)) {
            u32 x = v0, z = v1;
            for (int sft = 32; sft >= 1; sft >>= 1) {
                u32 const y = (u32)__shfl_xor((int)x, sft, 64), w = (u32)__shfl_xor((int)z, sft, 64);
                x = y < x ? y : x;
                z = w < z ? w : z;
            }
            x = wv_uni(x);
            z = wv_uni(z);
            if (x < certEnd) certEnd = x;
            if (z < certEnd1) certEnd1 = z;
        }
    };
    settle(vmin, vmin1);
    // ---- scan: positions [2, min(certEnd, last position with 4 readable bytes]), as one software
    // pipeline over rounds of 1024 bytes.  The kernel is latency-bound (2 waves per SIMD) and a
    // wave's loads complete in order, so every load of the scan is issued >= 4 rounds (about one
    // memory latency) before its first use and every round issues the same loads in the same order
    // (which lets hipcc emit exact s_waitcnt vmcnt(n) values instead of draining to 0):
    //   round k+4 : its load is issued: ONE 16-byte aligned load per lane = 1024 contiguous bytes
    //   round k   : lane l owns the 16 positions of its 16 bytes (+3 from lane l+1): Bloom tests;
    //               the positives (position, 4 bytes) are compacted into an LDS list, hashed, and
    //               their 8-byte table entries are gathered into one of 5 register slots
    //   round k+5 : that slot is judged, then refilled
    // The first 4 KiB of a block, where the schedule itself is dense (every position, every 2nd, ...),
    // are done by a simple round that judges each lane's positives in place.
    u32 const pEnd = cx.chunkEndPos - 3; // one past the last p with p + 4 <= chunkEnd
    SQ_L u32* const cand = cx.cand; // 128 x (position, bytes)
    const SQ_L u32* const B = cx.bloom;
    // the stream starts at the 16-byte boundary at or below the first source byte (same page, so
    // readable) and never reads beyond the 16-byte block that holds the chunk's last byte
    const u8* const src0 = base + SQZ_WINDOW_START;
    const u8* const a0 = src0 - ((size_t)src0 & 15);
    s32 const p0 = (s32)SQZ_WINDOW_START - (s32)(src0 - a0); // position of a0 (may be negative)
    const u8* const lastByte = base + cx.chunkEndPos - 1;
    const u8* const aLast = lastByte - ((size_t)lastByte & 15);
    s32 pos = p0; // position of the current round's first byte

    struct Slot { // candidates whose table entries are in flight
        u32 p[2], v[2];
        sq_v2u e[2];
        u32 n;
    };
    auto judge = [&](u32 p, u32 vp, sq_v2u e, bool valid, u32& v0, u32& v1) {
        u32 const t = ht_dec_g(cx.genBits, e.x);
        if (valid && t) {
            u32 const sw = t & SQZ_HT_IDX;
            if (p < sw && vp == e.y) note(sw, v0, v1);
            if (t & SQZ_CERT_DUP)
                for (u32 k = 0; k < nOvf; k++) {
                    u32 const so = ovf[2 * k], vo = ovf[2 * k + 1];
                    if (vp == vo && p < so) note(so, v0, v1);
                }
        }
    };
    // positions that can still matter: below the end of the certificate of the last block
    auto scanStop = [&]() -> u32 {
        u32 const ce = (blk1 != 0xFFFFFFFFu) ? certEnd1 : certEnd;
        return ce < pEnd ? ce : pEnd;
    };
    auto isDense = [&](s32 rp) -> bool { // does the round at rp touch the first 4 KiB of a block?
        bool d = rp < (s32)(SQZ_WINDOW_START + 4096);
        if (blk1 != 0xFFFFFFFFu) d = d || (rp + 1024 > (s32)blk1 && rp < (s32)(blk1 + 4096));
        return d;
    };
    auto loadEntry = [&](u32 vp) -> sq_v2u {
        u64 const raw = __hip_atomic_load((const SQ_G u64*)(E + 2 * hash4(vp, eLog)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (sq_v2u){(u32)raw, (u32)(raw >> 32)};
    };
    auto loadRound = [&](s32 rp) -> sq_v4u { // bytes of the round that starts at position rp
        const u8* ad = a0 + (size_t)(u32)(rp - p0) + 16 * lane;
        return sq_grd128_s(ad < aLast ? ad : aLast);
    };
    // Bloom tests of one round: returns the lane's candidate mask (bit j <-> position lo + j) and
    // leaves the lane's 20 bytes in d[]
    auto testRound = [&](sq_v4u w, sq_v4u wn, s32 rp, u32& d4) -> u32 {
        {
            u32 const nx = (u32)__builtin_amdgcn_update_dpp(0, (int)w.x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
            u32 const n63 = wv_uni(wn.x);
            d4 = (lane == 63) ? n63 : nx;
        }
        u32 const d[5] = {w.x, w.y, w.z, w.w, d4};
        u32 const stop = scanStop();
        s32 const lo = rp + (s32)(16 * lane);
        s32 f = (s32)SQZ_WINDOW_START - lo, l = (s32)stop - lo;
        f = f < 0 ? 0 : (f > 16 ? 16 : f);
        l = l < 0 ? 0 : (l > 16 ? 16 : l);
        u32 const okMask = ((1u << l) - 1u) & ~((1u << f) - 1u);
        u32 cm = 0;
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            u32 v[4], p1[4], p2[4];
            v[0] = d[q];
            v[1] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
            v[2] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
            v[3] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                p1[k] = B[(v[k] & 0xFFFFu) >> 5];
                p2[k] = B[v[k] >> 21];
            }
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                u32 const c = (p1[k] >> (v[k] & 31)) & (p2[k] >> ((v[k] >> 16) & 31)) & 1u;
                cm |= c << (4 * q + k);
            }
        }
        return cm & okMask;
    };
    // the 4 bytes at byte offset j of the lane's 20 (w, then d4); selects on registers, no array
    auto valAt = [&](sq_v4u w, u32 d4, u32 j) -> u32 {
        u32 const a = j >> 2;
        u32 const lo = a == 0 ? w.x : a == 1 ? w.y : a == 2 ? w.z : w.w;
        u32 const hi = a == 0 ? w.y : a == 1 ? w.z : a == 2 ? w.w : d4;
        return __builtin_amdgcn_alignbyte(hi, lo, j & 3);
    };
    auto judgeSlot = [&](Slot& sl) {
        u32 v0 = 0xFFFFFFFFu, v1 = 0xFFFFFFFFu;
#pragma unroll
        for (u32 g = 0; g < 2; g++) judge(sl.p[g], sl.v[g], sl.e[g], 64 * g + lane < sl.n, v0, v1);
        settle(v0, v1);
        sl.n = 0;
    };
    // The (at most one, outside the dense zones) scheduled position among a lane's 16, as a bit mask.
    auto schedMask = [&](s32 lo) -> u32 {
        bool const second = (blk1 != 0xFFFFFFFFu) && lo >= (s32)blk1;
        u32 const r = (u32)(lo - (s32)(second ? blk1 : SQZ_WINDOW_START)); // relative to the block start
        u32 const k = r >> 8;
        u32 const wd = cx.lvl[(second ? SQZ_SCHEDLVL_N : 0u) + (k < SQZ_SCHEDLVL_N ? k : SQZ_SCHEDLVL_N - 1)];
        u32 const first = wd & 0x1FFFFu, R = wd >> 17, step = k + 1;
        // first scheduled position >= r: first + ceil((r - first) / step) * step
        u32 const x = r > first ? r - first : 0u;
        u32 const q = (x * R) >> 16; // floor(x / step), exact for x < 256 and step >= 17
        u32 const m = x - q * step;
        u32 const nxt = (r > first ? r : first) + (m ? step - m : 0u);
        bool const ok = lo >= (s32)SQZ_WINDOW_START && k >= 16 && (r & 255) <= 240 && nxt < r + 16 && nxt < ((k + 1) << 8) &&
                        first != 0x1FFFFu;
        return ok ? 1u << (nxt - r) : 0u;
    };
    // a round of the pipelined part; returns false when the pipelined segment is over
    auto round = [&](sq_v4u w, sq_v4u wn, Slot& sl) -> bool {
        u32 d4;
        u32 cm = testRound(w, wn, pos, d4);
        s32 const lo = pos + (s32)(16 * lane);
        if (cx.lvl) cm &= ~schedMask(lo);
        u32 cnt = 0;
        for (;;) { // compaction: one candidate per lane and turn
            u64 const m = wv_ballot(cm != 0);
            if (!m) break;
            if (cm) {
                u32 const j = (u32)__builtin_ctz(cm);
                cm &= cm - 1;
                u32 const slot = cnt + (u32)__popcll(m & below);
                if (slot < 128) {
                    cand[2 * slot] = (u32)(lo + (s32)j);
                    cand[2 * slot + 1] = valAt(w, d4, j);
                }
            }
            cnt += (u32)__popcll(m);
        }
        wv_fence();
        judgeSlot(sl); // filled 5 rounds ago
        if (cnt > 128) { // cannot happen outside the dense zones with a working filter: stay conservative
            u32 const cut = (u32)(pos < (s32)SQZ_WINDOW_START ? (s32)SQZ_WINDOW_START : pos) + 1;
            if (cut < certEnd) certEnd = cut;
            if (cut < certEnd1) certEnd1 = cut;
            cnt = 128;
        }
        sl.n = cnt;
#pragma unroll
        for (u32 g = 0; g < 2; g++) { // always two LDS reads + two gathers: idle lanes re-read entry 0
            u32 const j = (64 * g + lane < cnt) ? 64 * g + lane : 0u;
            sl.p[g] = cand[2 * j];
            sl.v[g] = cand[2 * j + 1];
            sl.e[g] = loadEntry(sl.v[g]);
        }
        pos += 1024;
        return pos < (s32)scanStop() && !isDense(pos);
    };
    // a self-contained round: every lane judges its own positives, four at a time
    auto roundSimple = [&]() {
        u32 d4;
        sq_v4u const w = loadRound(pos);
        u32 cm = testRound(w, loadRound(pos + 1024), pos, d4);
        s32 const lo = pos + (s32)(16 * lane);
        u32 v0 = 0xFFFFFFFFu, v1 = 0xFFFFFFFFu;
        while (wv_ballot(cm != 0)) {
            u32 pj[4], vj[4];
            sq_v2u ej[4];
            bool ok[4];
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
                ok[k] = cm != 0;
                u32 const j = ok[k] ? (u32)__builtin_ctz(cm) : 0u;
                cm &= cm - 1; // 0 stays 0
                pj[k] = (u32)(lo + (s32)j);
                vj[k] = valAt(w, d4, j);
                ej[k] = loadEntry(vj[k]);
            }
#pragma unroll
            for (u32 k = 0; k < 4; k++) judge(pj[k], vj[k], ej[k], ok[k], v0, v1);
        }
        settle(v0, v1);
        pos += 1024;
    };

    cand[0] = SQZ_WINDOW_START; // entry 0 is what idle lanes re-read
    cand[1] = 0;
    wv_fence();
    while (pos < (s32)scanStop() && !(cx.dbg & 2)) {
        if (isDense(pos)) {
            roundSimple();
            continue;
        }
        Slot sA = {{0, 0}, {0, 0}, {{0, 0}, {0, 0}}, 0}, sB = sA, sC = sA, sD = sA, sE = sA;
        sq_v4u wA = loadRound(pos), wB = loadRound(pos + 1024), wC = loadRound(pos + 2048), wD = loadRound(pos + 3072), wE;
        // 5 KiB of the stream and 5 slots of gathers in flight per wave.  (hipcc still drains vmcnt to 0
        // at the head of this loop -- its wait-count analysis gives up across the back edge -- so one
        // memory latency is paid per 5 rounds; a 10-round body measured no better.)
        for (;;) {
            wE = loadRound(pos + 4096);
            if (!round(wA, wB, sA)) break;
            wA = loadRound(pos + 4096);
            if (!round(wB, wC, sB)) break;
            wB = loadRound(pos + 4096);
            if (!round(wC, wD, sC)) break;
            wC = loadRound(pos + 4096);
            if (!round(wD, wE, sD)) break;
            wD = loadRound(pos + 4096);
            if (!round(wE, wA, sE)) break;
        }
        judgeSlot(sA);
        judgeSlot(sB);
        judgeSlot(sC);
        judgeSlot(sD);
        judgeSlot(sE);
    }
    wv_fence();
#else
    // scalar twin: same table, same rule
    for (u32 i = 0; i < SQZ_BLOOM_WORDS; i++) cx.bloom[i] = 0;
    u32 ovfS[SQZ_CERT_OVF_CAP], ovfV[SQZ_CERT_OVF_CAP];
    auto note = [&](u32 sviol) { // a search at sviol may succeed
        if (sviol < blk1)
            certEnd = sviol < certEnd ? sviol : certEnd;
        else
            certEnd1 = sviol < certEnd1 ? sviol : certEnd1;
    };
    for (u32 i = 0; i < nCert; i++) {
        u32 const sp = cert_pos_c(cx, i);
        u32 const v = sq_rd32(base + sp);
        u32 const h = hash4(v, eLog);
        u32 const od = ht_dec_g(cx.genBits, E[2 * h]);
        u32 const oval = E[2 * h + 1];
        E[2 * h] = ht_enc_g(cx.genBits, sp) | (od ? SQZ_CERT_DUP : 0u); // exchange (+ DUP if it displaced one)
        E[2 * h + 1] = v;
        vbloom_add(cx.bloom, v);
        if (cx.repOff && sq_rd32(base + sp + 1) == sq_rd32(base + sp + 1 - cx.repOff)) note(sp);
        if (od) {
            u32 const opos = od & SQZ_HT_IDX;
            if (oval == v) note(opos > sp ? opos : sp);
            if (nOvf < SQZ_CERT_OVF_CAP) {
                ovfS[nOvf] = opos;
                ovfV[nOvf] = oval;
                nOvf++;
            } else
                note(opos);
        }
    }
    // (closure over the displaced entries, as on the device)
    for (u32 k = 0; k < nOvf; k++) {
        u32 const h = hash4(ovfV[k], eLog);
        u32 const t = ht_dec_g(cx.genBits, E[2 * h]);
        if (t && E[2 * h + 1] == ovfV[k]) note((t & SQZ_HT_IDX) > ovfS[k] ? (t & SQZ_HT_IDX) : ovfS[k]);
        for (u32 j = 0; j < nOvf; j++)
            if (j != k && ovfV[j] == ovfV[k]) note(ovfS[j] > ovfS[k] ? ovfS[j] : ovfS[k]);
    }
    u32 const pEnd = cx.chunkEndPos - 3;
    for (u32 p = SQZ_WINDOW_START; p < ((blk1 != 0xFFFFFFFFu) ? certEnd1 : certEnd) && p < pEnd; p++) {
        if (cx.lvl) { // leave the scheduled positions themselves out (outside the dense block starts)
            bool const second = (blk1 != 0xFFFFFFFFu) && p >= blk1;
            u32 const r = p - (second ? blk1 : SQZ_WINDOW_START), k = r >> 8;
            if (k >= 16 && k < SQZ_SCHEDLVL_N) {
                u32 const first = cx.lvl[(second ? SQZ_SCHEDLVL_N : 0u) + k] & 0x1FFFFu;
                if (first != 0x1FFFFu && r >= first && (r - first) % (k + 1) == 0) continue;
            }
        }
        u32 const vp = sq_rd32(base + p);
        if (!((cx.bloom[(vp & 0xFFFFu) >> 5] >> (vp & 31)) & (cx.bloom[vp >> 21] >> ((vp >> 16) & 31)) & 1u)) continue;
        u32 const h = hash4(vp, eLog);
        u32 const t = ht_dec_g(cx.genBits, E[2 * h]);
        if (!t) continue;
        u32 const sw = t & SQZ_HT_IDX;
        if (p < sw && vp == E[2 * h + 1]) note(sw);
        if (t & SQZ_CERT_DUP)
            for (u32 k = 0; k < nOvf; k++)
                if (vp == ovfV[k] && p < ovfS[k]) note(ovfS[k]);
    }
#endif
    cx.certEnd = certEnd;
    cx.certEnd1 = (blk1 != 0xFFFFFFFFu) ? certEnd1 : 0u;
    cx.nOvf = nOvf;
}

// the compressor's own use: table in the chain-table area, LDS pieces of the match-finding phase
SQ_DEVFN void cert_compute(MatchState& ms) {
    CertCtx cx;
    cx.base = ms.base;
    cx.E = ms.chainTable;
    cx.eLog = ms.cp.chainLog - 1;
    cx.genBits = ms.genBits;
    cx.schedTab = ms.schedTab;
    cx.certCnt0 = ms.certCnt0;
    cx.nCert = ms.nCert;
    cx.base1Pos = SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX;
    cx.tabOff1 = 0;
    cx.repOff = 1;
    cx.chunkEndPos = ms.chunkEndPos;
    cx.bloom = (SQ_L u32*)g_lds.u.bloom;
    cx.cand = (SQ_L u32*)g_lds.grp;
    cx.ovf = (SQ_L u32*)g_lds.scr;
    cx.lvl = nullptr; // no LDS to spare here
    cx.dbg = 0;
    cert_compute_core(cx);
    ms.certEnd = cx.certEnd;
    ms.certEnd1 = cx.certEnd1;
    SQ_TRACE(40, cx.certEnd);
    SQ_TRACE(41, cx.nOvf);
    SQ_TRACE(42, cx.certEnd1);
}


// Tail certificate.  After a hand-over (a 4-byte coincidence at a searched position, a real match,
// ...) the exact code has dealt with the match and the parser is back to finding nothing, at `ipPos`,
// with anchor `anc` and repcode offset `off1`.  Its remaining searches of this block are the "later"
// schedule laid at the anchor, from the entry that equals ipPos - anc on.  The same certificate
// (table in the workspace's tail-certificate area, Bloom filter in LDS) decides whether all of them
// certainly fail.  The caller uses a positive answer only where nothing later can observe the finder's
// tables, because skipping those searches leaves the tables behind.  The live-bucket Bloom filter
// shares its LDS with the certificate's: the caller rebuilds it after a negative answer.
SQ_DEVFN bool tail_certify(MatchState& ms, u32 ipPos, u32 anc, u32 off1, u32 ilimPos) {
    const u32* const later = ms.schedTab + 2 + SQZ_SCHEDTAB_N;
    u32 const nLater = gld32(ms.schedTab + 1);
    u32 const rel = ipPos - anc;
    u32 const j0 = sched_count(later, nLater, rel);
    if (j0 >= nLater || gld32(later + j0) != rel) return false; // not on the anchor's lattice
    u32 const j1 = sched_count(later, nLater, ilimPos - anc);
    if (j1 <= j0) return true; // no search left in this block
    CertCtx cx;
    cx.base = ms.base;
    cx.E = ms.tailTab;
    cx.eLog = ms.cp.chainLog - 1;
    cx.genBits = ms.genBits;
    cx.schedTab = ms.schedTab;
    cx.certCnt0 = 0;
    cx.nCert = j1 - j0;
    cx.base1Pos = anc;
    cx.tabOff1 = j0;
    cx.repOff = off1;
    cx.chunkEndPos = ms.chunkEndPos;
    cx.bloom = (SQ_L u32*)g_lds.u.bloom;
    cx.cand = (SQ_L u32*)g_lds.grp;
    cx.ovf = (SQ_L u32*)g_lds.scr;
    cx.lvl = nullptr;
    cx.dbg = 0;
    // The table must hold no entry with this chunk's tag: first use in this launch (or after a change of
    // size class), or an earlier tail certificate of this same chunk
    if (!*ms.tailTabClean || ms.tailTried > 1) {
        wv_fill32(cx.E, 0, (size_t)2 << cx.eLog);
        *ms.tailTabClean = 1;
    }
    cert_compute_core(cx);
    return cx.certEnd1 >= ilimPos;
}

// Block 1 was certified under the assumption that it starts with repcode offset 1.  When block 0
// ended with another offset (it had a match), redo the repcode test of block 1's scheduled
// searches with the real one: returns the first scheduled position below `limit` where
// read32(s + 1 - off1) == read32(s + 1), or `limit`.
SQ_DEVFN u32 cert_rep_recheck(MatchState& ms, u32 off1, u32 limit) {
    const u8* const base = ms.base;
    u32 const first = ms.certCnt0, n = ms.nCert;
    u32 best = limit;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u32 mine = limit;
    for (u32 i0 = first; i0 < n; i0 += 256) {
        u32 sp[4], a[4], b[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            u32 const i = i0 + 64 * g + lane;
            sp[g] = cert_pos(ms, i < n ? i : n - 1);
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            a[g] = sq_grd32(base + sp[g] + 1 - off1);
            b[g] = sq_grd32(base + sp[g] + 1);
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++)
            if (i0 + 64 * g + lane < n && a[g] == b[g] && sp[g] < mine) mine = sp[g];
        if (wv_uni(sp[0]) >= limit) break; // the schedule is ascending: nothing below `limit` is left
    }
    for (int sft = 32; sft >= 1; sft >>= 1) {
        u32 const y = (u32)__shfl_xor((int)mine, sft, 64);
        mine = y < mine ? y : mine;
    }
    best = wv_uni(mine);
#else
    for (u32 i = first; i < n; i++) {
        u32 const sp = cert_pos(ms, i);
        if (sp >= limit) break;
        if (sq_rd32(base + sp + 1 - off1) == sq_rd32(base + sp + 1)) {
            best = sp;
            break;
        }
    }
#endif
    return best < limit ? best : limit;
}

// ZSTD_updateDUBT
SQ_DEVFN void dubt_update(MatchState& ms, u32 target) {
    SQ_TRIP_PHASE(6);
    dubt_insert_range<0>(ms, ms.nextToUpdate, target);
    ms.nextToUpdate = target;
}

// ZSTD_insertDUBT1 (noDict).  Per tree step ONE memory round trip: the node's two children are
// fetched with one 8-byte load issued together with the match-extension loads.
// (Everything the walk reads from the finder's state is copied into wave-uniform locals first: the state lives in the
// caller's frame, and what hipcc loads from a frame it treats as different in every lane -- vector registers, exec-masked
// regions and 64-bit vector address arithmetic for a walk that is the same in all 64 lanes.  Node slots are indices into the
// tree table, not pointers; SQZ_BT_DUMMY stands for ms.dummy.  SQ counters of 256 KiB text chunks before this: 65 VALU + 69
// SALU instructions per tree step, 30 % of a wave's time spent issuing, profiles/round4f_text256k_sq_counters.txt.)
#define SQZ_BT_DUMMY 0xFFFFFFFFu
// have: what the slot is known to hold (the child word the walk loaded when it passed the slot's node; SQZ_BT_UNKNOWN for the
// new root's own two slots).  A walk that keeps to one side writes every child word back unchanged -- half of a walk's stores,
// each a dirty 64-byte line of a 2 MiB table that the L2 then has to write out (1.15 writes to HBM per tree step measured,
// profiles/round4f_text256k_mem_counters.txt): those are skipped.
#define SQZ_BT_UNKNOWN 0xFFFFFFFEu
SQ_DEV void bt_store(u32* bt, u32* dummy, u32 slot, u32 v, u32 have = SQZ_BT_UNKNOWN) {
    if (have != v) gst32(slot == SQZ_BT_DUMMY ? dummy : bt + slot, v);
}
// The unsorted chain of a bucket as the search's first pass over it (dubt_materialize) saw it: (index, next, mark) per node,
// oldest last.  The pass that threads the candidates (dubt_search_core) and the insertions that follow read the same words
// again -- one trip to memory each, 3k + 1 trips for k candidates on a path whose whole cost is trips -- so they
// are kept in LDS instead (the last quarter of g_lds.scr; lane 0 writes, every lane reads the same word).
#define SQZ_SCAN_CACHE 64u
SQ_DEV void scan_cache_put(u32 i, u32 idx, u32 nx, u32 mk) {
    if (i >= SQZ_SCAN_CACHE) return;
    SQ_L u32* const c = (SQ_L u32*)g_lds.scr + 768 + 3 * i;
#if SQ_ON_DEVICE
    if (wv_lane() == 0)
#endif
    {
        c[0] = idx;
        c[1] = nx;
        c[2] = mk;
    }
}
SQ_DEV void scan_cache_get(u32 i, u32* idx, u32* nx, u32* mk) {
    SQ_L u32* const c = (SQ_L u32*)g_lds.scr + 768 + 3 * i;
    wv_fence();
    *idx = wv_uni(c[0]);
    *nx = wv_uni(c[1]);
    *mk = wv_uni(c[2]);
}

// (first_, link_: what curr's two slots hold -- its bucket's older head, the thread to the next candidate)
SQ_DEVFN void dubt_insert1(MatchState& ms, u32 curr_, const u8* inputEnd, u32 nbCompares_, u32 btLow_, u32 first_, u32 link_) {
    u32* const bt = (u32*)wv_uni64((u64)(size_t)ms.chainTable);
    u32* const dummy = (u32*)wv_uni64((u64)(size_t)ms.dummy);
    u32 const btMask = (1u << (wv_uni(ms.cp.chainLog) - 1)) - 1;
    u32 const curr = wv_uni(curr_), btLow = wv_uni(btLow_);
    u32 nbCompares = wv_uni(nbCompares_);
    u32 commonLengthSmaller = 0, commonLengthLarger = 0;
    const u8* const base = (const u8*)wv_uni64((u64)(size_t)ms.base);
    const u8* const ip = base + curr;
    u32 const ipRoom = wv_uni((u32)(inputEnd - ip));
    u32 smallerSlot = 2 * (curr & btMask), largerSlot = smallerSlot + 1;
    SQ_TRIP_PHASE(4);
    u32 matchIndex = wv_uni(first_);
    u32 smallerHas = matchIndex, largerHas = wv_uni(link_); // what those slots are known to hold (bt_store)
    u32 const windowLow = SQZ_WINDOW_START; // window always covers the whole chunk (see get_cparams)
    for (; nbCompares && (matchIndex > windowLow); --nbCompares) {
        if (matchIndex >= curr) {
            ms.fault = 1;
            SQ_TRACE(16, 1);
            SQ_TRACE(17, matchIndex);
            SQ_TRACE(18, curr);
            SQ_TRACE(19, nbCompares);
            SQ_TRACE(20, smallerSlot);
            SQ_TRACE(21, largerSlot);
            break;
        }
        u32 const nextSlot = 2 * (matchIndex & btMask);
        u32 child0, child1, mLess = 0;
        u32 matchLength = commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        const u8* const match = base + matchIndex;
        u64 wa8, wb8;
        wv_tree_step_loads(bt + nextSlot, ip + matchLength, match + matchLength, ipRoom - matchLength >= 8, &child0, &child1, &wa8, &wb8);
        matchLength += wv_count_cmp(ip + matchLength, match + matchLength, ipRoom - matchLength, &mLess, SQ_ON_DEVICE != 0, wa8, wb8);
        matchLength = wv_uni(matchLength);
        mLess = wv_uni(mLess);
        if (matchLength == ipRoom) break;
        if (mLess) {
            bt_store(bt, dummy, smallerSlot, matchIndex, smallerHas);
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerSlot = SQZ_BT_DUMMY;
                smallerHas = SQZ_BT_UNKNOWN;
                break;
            }
            smallerSlot = nextSlot + 1;
            smallerHas = child1;
            matchIndex = child1;
        } else {
            bt_store(bt, dummy, largerSlot, matchIndex, largerHas);
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerSlot = SQZ_BT_DUMMY;
                largerHas = SQZ_BT_UNKNOWN;
                break;
            }
            largerSlot = nextSlot;
            largerHas = child0;
            matchIndex = child0;
        }
    }
    bt_store(bt, dummy, smallerSlot, 0, smallerHas);
    bt_store(bt, dummy, largerSlot, 0, largerHas);
    wv_mem_sync();
}

// Body of ZSTD_DUBT_findBestMatch (noDict, mls 4) for the position `curr`, given the bucket head
// as it was when `curr` was reached: sorts the unsorted candidates, walks the tree, makes `curr`
// the root.  Does not touch the hash table.
// nCached: the bucket's first nodes as dubt_materialize's last pass left them in the scan cache (0: none)
SQ_DEVFN u32 dubt_search_core(MatchState& ms, u32 curr_, const u8* const iend, u32 const headIndex_, u32* offBasePtr,
                              u32* matchEndIdxOut, u32 nCached_ = 0, bool drain = true) {
    SQ_TRIP_PHASE(5);
    // (wave-uniform copies of what the walk reads from the finder's state: see dubt_insert1)
    const u8* const base = (const u8*)wv_uni64((u64)(size_t)ms.base);
    u32 const curr = wv_uni(curr_), headIndex = wv_uni(headIndex_);
    const u8* const ip = base + curr;
    u32 matchIndex = headIndex;
    u32 const windowLow = SQZ_WINDOW_START;
    u32* const bt = (u32*)wv_uni64((u64)(size_t)ms.chainTable);
    u32* const dummy = (u32*)wv_uni64((u64)(size_t)ms.dummy);
    u32 const btMask = (1u << (wv_uni(ms.cp.chainLog) - 1)) - 1;
    u32 const btLow = (btMask >= curr) ? 0 : curr - btMask;
    u32 const unsortLimit = btLow > windowLow ? btLow : windowLow;
    u32 candSlot = 2 * (matchIndex & btMask); // (next candidate, sort mark) of the visited candidate
    u32 nbCompares = 1U << wv_uni(ms.cp.searchLog);
    u32 nbCandidates = nbCompares;
    u32 previousCandidate = 0;
    u32 const ipRoom = wv_uni((u32)(iend - ip));
    {
        // one 8-byte load per visited candidate: (next candidate, sort mark) -- from the scan cache where the pass before
        // this one has read it already, and into the cache otherwise: the insertions below start from it
        SQ_TRIP_PHASE(3);
        u32 const nCached = wv_uni(nCached_);
        u32 nextC = 0, mark = 0, nVisited = 0, nCand = 0;
        auto visit = [&]() {
            u32 at = 0;
            if (nVisited < nCached)
                scan_cache_get(nVisited, &at, &nextC, &mark);
            else {
                gld_pair(bt + candSlot, &nextC, &mark);
                scan_cache_put(nVisited, matchIndex, nextC, mark);
                at = matchIndex;
            }
            if (at != matchIndex) ms.fault = 1; // (cannot happen: the cache is this bucket's chain, in order)
            nVisited++;
        };
        if (matchIndex > unsortLimit) visit();
        while ((matchIndex > unsortLimit) && (mark == SQZ_DUBT_UNSORTED) && (nbCandidates > 1)) {
            gst32(bt + candSlot + 1, previousCandidate);
            previousCandidate = matchIndex;
            matchIndex = nextC;
            candSlot = 2 * (matchIndex & btMask);
            nbCandidates--;
            nCand++;
            if (matchIndex > unsortLimit) visit();
        }
        bool reread = nCand > SQZ_SCAN_CACHE; // (the threads of the candidates beyond the cache are read back below)
        if ((matchIndex > unsortLimit) && (mark == SQZ_DUBT_UNSORTED)) {
            gst32(bt + candSlot, 0);
            gst32(bt + candSlot + 1, 0);
            reread = true; // (the oldest candidate's walk starts at this node)
        }
        if (reread) wv_mem_sync(); // (otherwise nothing written above is loaded again: the waits are the walks' own)
        // the candidates, oldest first: node nCand - 1 of the scan down to node 0; each one's slots hold (the node after it
        // in the scan, the node before it: the thread written above)
        matchIndex = previousCandidate;
        for (u32 i = nCand; i-- > 0;) {
            u32 first, link;
            if (i < SQZ_SCAN_CACHE) {
                u32 at, mk;
                scan_cache_get(i, &at, &first, &mk);
                if (at != matchIndex) ms.fault = 1;
                link = 0;
                if (i > 0) {
                    u32 f2;
                    scan_cache_get(i - 1, &link, &f2, &mk);
                }
            } else
                gld_pair(bt + 2 * (matchIndex & btMask), &first, &link);
            dubt_insert1(ms, matchIndex, iend, nbCandidates, unsortLimit, first, link);
            matchIndex = link;
            nbCandidates++;
        }
    }
    u32 commonLengthSmaller = 0, commonLengthLarger = 0;
    u32 smallerSlot = 2 * (curr & btMask), largerSlot = smallerSlot + 1;
    u32 smallerHas = SQZ_BT_UNKNOWN, largerHas = SQZ_BT_UNKNOWN; // what those slots are known to hold (bt_store)
    u32 matchEndIdx = curr + 8 + 1;
    u32 bestLength = 0;
    u32 offBase = wv_uni(*offBasePtr);
    matchIndex = headIndex;
    for (; nbCompares && (matchIndex > windowLow); --nbCompares) {
        if (matchIndex >= curr) {
            ms.fault = 1;
            SQ_TRACE(16, 2);
            SQ_TRACE(17, matchIndex);
            SQ_TRACE(18, curr);
            SQ_TRACE(19, nbCompares);
            SQ_TRACE(20, headIndex);
            break;
        }
        u32 const nextSlot = 2 * (matchIndex & btMask);
        u32 child0, child1, mLess = 0;
        u32 matchLength = commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        const u8* const match = base + matchIndex;
        u64 wa8, wb8;
        wv_tree_step_loads(bt + nextSlot, ip + matchLength, match + matchLength, ipRoom - matchLength >= 8, &child0, &child1, &wa8, &wb8);
        matchLength += wv_count_cmp(ip + matchLength, match + matchLength, ipRoom - matchLength, &mLess, SQ_ON_DEVICE != 0, wa8, wb8);
        matchLength = wv_uni(matchLength);
        mLess = wv_uni(mLess);
        if (matchLength > bestLength) {
            if (matchLength > matchEndIdx - matchIndex) matchEndIdx = matchIndex + matchLength;
            if ((4 * (int)(matchLength - bestLength)) >
                (int)(sq_highbit32(curr - matchIndex + 1) - sq_highbit32(offBase))) {
                bestLength = matchLength;
                offBase = OFFSET_TO_OFFBASE(curr - matchIndex);
            }
            if (matchLength == ipRoom) break;
        }
        if (mLess) {
            bt_store(bt, dummy, smallerSlot, matchIndex, smallerHas);
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerSlot = SQZ_BT_DUMMY;
                smallerHas = SQZ_BT_UNKNOWN;
                break;
            }
            smallerSlot = nextSlot + 1;
            smallerHas = child1;
            matchIndex = child1;
        } else {
            bt_store(bt, dummy, largerSlot, matchIndex, largerHas);
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerSlot = SQZ_BT_DUMMY;
                largerHas = SQZ_BT_UNKNOWN;
                break;
            }
            largerSlot = nextSlot;
            largerHas = child0;
            matchIndex = child0;
        }
    }
    bt_store(bt, dummy, smallerSlot, 0, smallerHas);
    bt_store(bt, dummy, largerSlot, 0, largerHas);
    if (drain) wv_mem_sync(); // (dubt_find has one more word to write and waits for all of them once)
    *offBasePtr = offBase;
    *matchEndIdxOut = matchEndIdx;
    return bestLength;
}

// index one past the block that contains position `pos`
SQ_DEV u32 block_end_of(const MatchState& ms, u32 pos) {
    for (u32 k = 0; k < ms.nBlocks; k++) {
        u32 const e = gld32(ms.blockTab + 2 * k + 1);
        if (pos < e) return e;
    }
    return ms.blockEndPos;
}

// Deferred searches (speculative epochs, below) sit in a bucket's chain as nodes whose sort mark is
// SQZ_DUBT_SEARCHED and whose `next` is the bucket head at their time.  Before an exact search may
// use the bucket they are replayed, oldest first, with the very same search body: replaying the
// search of node s only touches nodes older than s, so the result is the state libzstd would hold.
// Returns how many of the bucket's first nodes its last pass -- the one that found nothing left to replay -- has put into
// the scan cache.
SQ_DEVFN u32 dubt_materialize(MatchState& ms, u32 headIndex) {
    SQ_TRIP_PHASE(2);
    u32* const bt = ms.chainTable;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
    for (;;) {
        u32 found = 0, foundNext = 0, n = 0;
        u32 idx = headIndex;
        while (idx > SQZ_WINDOW_START) {
            u32 nx, mk;
            gld_pair(bt + 2 * (idx & btMask), &nx, &mk);
            scan_cache_put(n++, idx, nx, mk);
            if (mk == SQZ_DUBT_SEARCHED) {
                found = idx;
                foundNext = nx;
            } else if (mk != SQZ_DUBT_UNSORTED)
                break; // sorted node: nothing deferred below it
            if (nx >= idx) { // cannot happen in a well-formed chain
                ms.fault = 1;
                return 0;
            }
            idx = nx;
        }
        if (!found) return n < SQZ_SCAN_CACHE ? n : SQZ_SCAN_CACHE;
        u32 off = 999999999, mend = 0;
        ms.nReplay++;
        (void)dubt_search_core(ms, found, ms.base + block_end_of(ms, found), foundNext, &off, &mend);
        if (ms.fault) return 0;
    }
}

// ZSTD_DUBT_findBestMatch (noDict, mls 4)
SQ_DEVFN u32 dubt_find(MatchState& ms, const u8* const ip, const u8* const iend, u32* offBasePtr) {
    SQ_TRIP_PHASE(1);
    u32* const hashTable = ms.hashTable;
    u32 const h = hash4(srd32(ip), ms.cp.hashLog);
    u32 const curr = (u32)(ip - ms.base);
    u32 headWord = ht_dec(ms, gld32(hashTable + h));
    if (!ms.eager && !(headWord & SQZ_HT_LIVE)) { // the schedule changed: make this bucket exact
        // first search after a certified prefix: wake only the buckets it needs (filter_wake)
        if (!(ms.nLive == 0 && ms.nCertDone > 0 && filter_wake(ms, curr))) filter_repair(ms, curr, ms.anchorPos, false);
        headWord = ht_dec(ms, gld32(hashTable + h));
    }
    u32 const headIndex = headWord & SQZ_HT_IDX;
    u32 const nCached = dubt_materialize(ms, headIndex);
    u32 matchEndIdx = curr + 9;
    u32 const bestLength = dubt_search_core(ms, curr, iend, headIndex, offBasePtr, &matchEndIdx, nCached, false);
    gst32(hashTable + h, ht_enc(ms, curr | (headWord & ~SQZ_HT_IDX)));
    wv_mem_sync();
    if (matchEndIdx - 8 > curr + 1) filter_add_skip(ms, curr + 1, matchEndIdx - 8);
    ms.nextToUpdate = matchEndIdx - 8;
    return bestLength;
}

// ---------------------------------------------------------------- speculative search epochs
// While the lazy parser finds nothing, its next searches are at predictable positions
// (step = ((ip - anchor) >> 8) + 1) and each one is a chain of dependent memory round trips that
// almost always ends in "no match".  An epoch evaluates up to 64 of those searches at once, one
// per lane: (1) the positions up to the last one are inserted in one filtered sweep; (2) every
// lane walks its own bucket and asks a conservative question -- is there ANY older position in
// the bucket with the same 4 bytes, does the repcode test fire, is the bucket anything but a plain
// chain?  (3) searches before the first "maybe" certainly fail; their only effect, the re-sorting
// of their bucket, is deferred by leaving the searched position in the chain as a SEARCHED node
// (replayed by dubt_materialize if the bucket is ever searched for real); (4) insertions at or
// beyond the first "maybe" are rolled back and the parser continues there with the exact code.
SQ_DEVFN void dubt_rollback(MatchState& ms, u32 from, u32 to) {
    u32* const hashTable = ms.hashTable;
    u32* const bt = ms.chainTable;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
    u32 const hashLog = ms.cp.hashLog;
    bool const all = ms.eager != 0;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 idx0 = from; idx0 < to; idx0 += 64) {
        u32 const idx = idx0 + lane;
        if (idx < to) {
            u32 const h = hash4(sq_grd32(ms.base + idx), hashLog);
            if (all || bloom_test(h)) {
                u32 const hw = ht_dec(ms, sq_tld32(hashTable + h));
                if ((all || (hw & SQZ_HT_LIVE)) && (hw & SQZ_HT_IDX) >= from) {
                    u32 cur = hw & SQZ_HT_IDX;
                    while (cur >= from) cur = sq_tld32(bt + 2 * (cur & btMask));
                    sq_gst32(hashTable + h, ht_enc(ms, cur | (hw & ~SQZ_HT_IDX)));
                }
            }
        }
    }
    wv_mem_sync();
#else
    for (u32 idx = from; idx < to; idx++) {
        u32 const h = hash4(sq_rd32(ms.base + idx), hashLog);
        if (!all && !bloom_test(h)) continue;
        u32 const hw = ht_dec(ms, hashTable[h]);
        if (!(all || (hw & SQZ_HT_LIVE))) continue;
        u32 cur = hw & SQZ_HT_IDX;
        while (cur >= from) cur = bt[2 * (cur & btMask)];
        hashTable[h] = ht_enc(ms, cur | (hw & ~SQZ_HT_IDX));
    }
#endif
}

// Returns how many scheduled searches, starting at pos0, certainly fail; *nextPos = where the
// parser continues (the first unproven position, or the position after the last proven one);
// *stopped = 1 when a "maybe" ended the epoch, i.e. *nextPos must be searched by the exact code.
SQ_DEVFN u32 spec_epoch(MatchState& ms, u32 pos0, u32 anc, u32 off1, u32 ilimitPos, u32* nextPos, u32* stopped) {
    u32* const hashTable = ms.hashTable;
    u32* const bt = ms.chainTable;
    u32 const btMask = (1u << (ms.cp.chainLog - 1)) - 1;
    u32 const hashLog = ms.cp.hashLog;
    const u8* const base = ms.base;
    bool const all = ms.eager != 0;
    u32 const n2u0 = ms.nextToUpdate;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    u32 myPos = 0, K = 0, lastPos = pos0, p = pos0;
    for (u32 j = 0; j < 64 && p < ilimitPos; j++) {
        if (lane == j) myPos = p;
        K++;
        lastPos = p;
        p += ((p - anc) >> 8) + 1;
    }
    u32 const nextAfter = p;
    if (K < SQZ_EPOCH_MIN_K) return 0;
    PROF_T0(t_bulk);
    if (lastPos + 1 > ms.nextToUpdate) {
        dubt_insert_range<0>(ms, ms.nextToUpdate, lastPos + 1);
        ms.nextToUpdate = lastPos + 1;
    }
    PROF_ADD(9, t_bulk);
    PROF_T0(t_pred);
    bool const have = lane < K;
    bool hit = false;
    if (have) {
        u32 const w = sq_grd32(base + myPos);
        if (off1 > 0 && sq_grd32(base + myPos + 1 - off1) == sq_grd32(base + myPos + 1)) hit = true;
        if (!hit && myPos >= n2u0) { // below nextToUpdate: "skipped area", the search is not even run
            u32 const h = hash4(w, hashLog);
            u32 const hw = ht_dec(ms, sq_tld32(hashTable + h));
            if (!all && !(hw & SQZ_HT_LIVE)) hit = true;
            u32 idx = hw & SQZ_HT_IDX, steps = 0;
            while (!hit && idx > SQZ_WINDOW_START) {
                u64 const pr = __hip_atomic_load((const SQ_G u64*)(bt + 2 * (idx & btMask)), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                u32 const nx = (u32)pr, mk = (u32)(pr >> 32);
                if (idx < myPos) {
                    if (sq_grd32(base + idx) == w || (mk != SQZ_DUBT_UNSORTED && mk != SQZ_DUBT_SEARCHED)) hit = true;
                }
                if (nx >= idx || ++steps > 64) hit = true; // malformed / too long: let the exact code decide
                idx = nx;
            }
        }
    }
    u64 const hm = wv_ballot(hit);
    u32 const first = hm ? (u32)__builtin_ctzll(hm) : K;
    if (have && lane < first && myPos >= n2u0) sq_gst32(bt + 2 * (myPos & btMask) + 1, SQZ_DUBT_SEARCHED);
    wv_mem_sync();
    PROF_ADD(10, t_pred);
    if (first < K) {
        u32 const firstPos = wv_readlane(myPos, first);
        u32 const keep = firstPos > n2u0 ? firstPos : n2u0;
        PROF_T0(t_rb);
        if (lastPos + 1 > keep) dubt_rollback(ms, keep, lastPos + 1);
        PROF_ADD(11, t_rb);
        ms.nextToUpdate = keep;
        *nextPos = firstPos;
        *stopped = 1;
    } else {
        *nextPos = nextAfter;
        *stopped = 0;
    }
    return first;
#else
    u32 pos[64];
    u32 K = 0, p = pos0;
    for (; K < 64 && p < ilimitPos; K++) {
        pos[K] = p;
        p += ((p - anc) >> 8) + 1;
    }
    u32 const nextAfter = p;
    if (K < SQZ_EPOCH_MIN_K) return 0;
    u32 const lastPos = pos[K - 1];
    if (lastPos + 1 > ms.nextToUpdate) {
        dubt_insert_range<0>(ms, ms.nextToUpdate, lastPos + 1);
        ms.nextToUpdate = lastPos + 1;
    }
    u32 first = K;
    for (u32 j = 0; j < K; j++) {
        u32 const myPos = pos[j];
        bool hit = false;
        u32 const w = sq_rd32(base + myPos);
        if (off1 > 0 && sq_rd32(base + myPos + 1 - off1) == sq_rd32(base + myPos + 1)) hit = true;
        if (!hit && myPos >= n2u0) {
            u32 const h = hash4(w, hashLog);
            u32 const hw = ht_dec(ms, hashTable[h]);
            if (!all && !(hw & SQZ_HT_LIVE)) hit = true;
            u32 idx = hw & SQZ_HT_IDX, steps = 0;
            while (!hit && idx > SQZ_WINDOW_START) {
                u32 const nx = bt[2 * (idx & btMask)], mk = bt[2 * (idx & btMask) + 1];
                if (idx < myPos) {
                    if (sq_rd32(base + idx) == w || (mk != SQZ_DUBT_UNSORTED && mk != SQZ_DUBT_SEARCHED)) hit = true;
                }
                if (nx >= idx || ++steps > 64) hit = true;
                idx = nx;
            }
        }
        if (hit) {
            first = j;
            break;
        }
    }
    for (u32 j = 0; j < first; j++)
        if (pos[j] >= n2u0) bt[2 * (pos[j] & btMask) + 1] = SQZ_DUBT_SEARCHED;
    if (first < K) {
        u32 const firstPos = pos[first];
        u32 const keep = firstPos > n2u0 ? firstPos : n2u0;
        if (lastPos + 1 > keep) dubt_rollback(ms, keep, lastPos + 1);
        ms.nextToUpdate = keep;
        *nextPos = firstPos;
        *stopped = 1;
    } else {
        *nextPos = nextAfter;
        *stopped = 0;
    }
    return first;
#endif
}

SQ_DEV u32 bt_find_best(MatchState& ms, const u8* ip, const u8* iLimit, u32* offBasePtr) {
    if (ip < ms.base + ms.nextToUpdate) return 0; // skipped area
    PROF_T0(t_u);
    dubt_update(ms, (u32)(ip - ms.base));
    PROF_ADD(2, t_u);
    PROF_T0(t_f);
    PROF_CNT(23, 1);
    u32 const r = dubt_find(ms, ip, iLimit, offBasePtr);
    PROF_ADD(3, t_f);
    return r;
}

// ================================================================ row hash (lazy2)
// position sequence of ZSTD_row_nextIndex starting from head byte hb: k-th insert (k >= 1)
SQ_DEV u32 row_pos_after(u32 hb, u32 k) {
    u32 const c0 = hb & 63;
    u32 const p1 = (c0 <= 1) ? 63 : c0 - 1;
    // positions cycle 63,62,...,1,63,...  p_k = ((p1 - 1 - (k-1)) mod 63) + 1
    u32 const m = (k - 1) % 63;
    u32 const q = p1 - 1; // 0..62
    return ((q + 63 - m) % 63) + 1;
}

// insert positions [from, to) into the row tables (ZSTD_row_update_internalImpl)
SQ_DEV void row_insert_range(MatchState& ms, u32 from, u32 to) {
    const u8* const base = ms.base;
    u32 const hBits = ms.rowHashLog + 8;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 idx0 = from; idx0 < to; idx0 += 64) {
        u32 const idx = idx0 + lane;
        bool const active = idx < to;
        u32 hash = 0;
        if (active) hash = hash5(sq_grd64(base + idx), hBits);
        u32 const relRow = (hash >> 8) << 6;
        WvGroup const g = wv_group(relRow, active, g_lds.grp);
        u32 hb = 0;
        // Table stores of the previous step / of the last search must have landed before these reads (a wave's
        // load is not ordered behind its own older store).  Draining HERE, not right after the stores, lets the
        // stores' latency overlap the chunk loads and the grouping above.
        wv_mem_sync();
        if (active) {
            u8* const tagRow = ms.tagTable + relRow;
            u32* const row = ms.hashTable + relRow;
            hb = sq_tld8(tagRow);
            u32 const pos = row_pos_after(hb, g.rank + 1);
            if (g.rank + 63 >= g.gsize) { // not overwritten by a later lane of this batch
                sq_gwr8(tagRow + pos, hash);
                sq_gst32(row + pos, idx);
            }
        }
        wv_fence();
        // heads are written after every lane has read its (old) head byte
        if (active && g.rank + 1 == g.gsize) sq_gwr8(ms.tagTable + relRow, row_pos_after(hb, g.gsize));
    }
#else
    for (u32 idx = from; idx < to; idx++) {
        u32 const hash = hash5(sq_rd64(base + idx), hBits);
        u32 const relRow = (hash >> 8) << 6;
        u8* const tagRow = ms.tagTable + relRow;
        u32 const pos = row_pos_after(tagRow[0], 1);
        tagRow[0] = (u8)pos;
        tagRow[pos] = (u8)hash;
        ms.hashTable[relRow + pos] = idx;
    }
#endif
}

// ZSTD_row_update_internal (the 8-entry hash cache is a prefetch device, not state)
SQ_DEV void row_update(MatchState& ms, u32 target) {
    u32 idx = ms.nextToUpdate;
    if (target - idx > 384) {
        row_insert_range(ms, idx, idx + 96);
        idx = target - 32;
    }
    row_insert_range(ms, idx, target);
    ms.nextToUpdate = target;
}

// The same for a LIST of positions, base0 + tab[k], k in [kFrom, kTo), ascending (what the finder inserts in lazy-skipping
// mode: the searched positions only).
SQ_DEV void row_insert_list(MatchState& ms, u32 base0, const u32* tab, u32 kFrom, u32 kTo) {
    const u8* const base = ms.base;
    u32 const hBits = ms.rowHashLog + 8;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 k0 = kFrom; k0 < kTo; k0 += 64) {
        u32 const k = k0 + lane;
        bool const active = k < kTo;
        u32 const idx = base0 + sq_gld32(tab + (active ? k : kTo - 1));
        u32 hash = 0;
        if (active) hash = hash5(sq_grd64(base + idx), hBits);
        u32 const relRow = (hash >> 8) << 6;
        WvGroup const g = wv_group(relRow, active, g_lds.grp);
        u32 hb = 0;
        wv_mem_sync(); // (as in row_insert_range)
        if (active) {
            u8* const tagRow = ms.tagTable + relRow;
            u32* const row = ms.hashTable + relRow;
            hb = sq_tld8(tagRow);
            u32 const pos = row_pos_after(hb, g.rank + 1);
            if (g.rank + 63 >= g.gsize) {
                sq_gwr8(tagRow + pos, hash);
                sq_gst32(row + pos, idx);
            }
        }
        wv_fence();
        if (active && g.rank + 1 == g.gsize) sq_gwr8(ms.tagTable + relRow, row_pos_after(hb, g.gsize));
    }
#else
    for (u32 k = kFrom; k < kTo; k++) {
        u32 const idx = base0 + tab[k];
        u32 const hash = hash5(sq_rd64(base + idx), hBits);
        u32 const relRow = (hash >> 8) << 6;
        u8* const tagRow = ms.tagTable + relRow;
        u32 const pos = row_pos_after(tagRow[0], 1);
        tagRow[0] = (u8)pos;
        tagRow[pos] = (u8)hash;
        ms.hashTable[relRow + pos] = idx;
    }
#endif
}

// A block in which every search of the lazy parser fails (stage 1's certificate for its pair of blocks, or the repcode
// re-check below), WITHOUT the searches: what is left of ZSTD_compressBlock_lazy_generic + ZSTD_RowFindBestMatch then is
// their effect on the tables.  The searched positions are the no-match schedule s_0 < s_1 < ... (step =
// ((s - blockStart) >> 8) + 1, the constant tables of sched_tab_fill, up to 16 bytes before the block's end).  While the
// step is <= 8 a search first inserts every position up to itself (ZSTD_row_update_internal, with its rule for gaps
// above 384 on the first one) and then itself; the search after which the step exceeds 8 switches the finder to lazy
// skipping, where only the searched positions are inserted.  So: everything up to s_m (the first search at or beyond
// 2 KiB into the block), then the later s_k alone, 64 per step instead of one search each.
#if !SQ_ON_DEVICE
static u32 g_sim_replayed_blocks = 0; // (simulator build: lets the tests see that the path was taken)
#endif
SQ_DEVFN void row_replay_nomatch(MatchState& ms, u32 istartPos, u32 srcSize) {
#if !SQ_ON_DEVICE
    g_sim_replayed_blocks++;
#endif
    bool const first = istartPos == SQZ_WINDOW_START;
    const u32* const tab = ms.schedTab + 2 + (first ? 0u : SQZ_SCHEDTAB_N);
    u32 const full = gld32(ms.schedTab + (first ? 0u : 1u));
    u32 const n = (srcSize <= 16) ? 0u : sched_count(tab, full, srcSize - 16);
    if (n == 0) return;
    u32 const m1 = sched_count(tab, n, 2048); // entries below 2 KiB: searches 0 .. m1 run before the switch (m1 itself if it exists)
    u32 const m = m1 < n ? m1 : n - 1;
    u32 const s0 = istartPos + gld32(tab), sm = istartPos + gld32(tab + m);
    row_update(ms, s0);
    row_insert_range(ms, s0, sm + 1);
    if (m + 1 < n) row_insert_list(ms, istartPos, tab, m + 1, n);
    ms.nextToUpdate = istartPos + gld32(tab + n - 1) + 1;
    wv_mem_sync();
}
// ... and the one thing the certificate assumed about the parser's state: the repcode test at s + 1 with offset 1.
// For another offset the test is repeated here: true = no scheduled search of the block has a repcode match.
SQ_DEVFN bool row_sched_rep_clear(MatchState& ms, u32 istartPos, u32 srcSize, u32 off1) {
    bool const first = istartPos == SQZ_WINDOW_START;
    const u32* const tab = ms.schedTab + 2 + (first ? 0u : SQZ_SCHEDTAB_N);
    u32 const full = gld32(ms.schedTab + (first ? 0u : 1u));
    u32 const n = (srcSize <= 16) ? 0u : sched_count(tab, full, srcSize - 16);
    const u8* const base = ms.base;
    bool hit = false;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 k0 = 0; k0 < n; k0 += 256) {
        u32 a[4], b[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            u32 const k = k0 + 64 * g + lane;
            u32 const sp = istartPos + sq_gld32(tab + (k < n ? k : n - 1));
            bool const ok = sp + 1 >= off1 + SQZ_WINDOW_START;
            a[g] = ok ? sq_grd32(base + sp + 1 - off1) : 0u;
            b[g] = ok ? sq_grd32(base + sp + 1) : 1u;
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) hit = hit || (k0 + 64 * g + lane < n && a[g] == b[g]);
    }
    return wv_ballot(hit) == 0;
#else
    for (u32 k = 0; k < n; k++) {
        u32 const sp = istartPos + tab[k];
        if (sp + 1 >= off1 + SQZ_WINDOW_START && sq_rd32(base + sp + 1 - off1) == sq_rd32(base + sp + 1)) hit = true;
    }
    return !hit;
#endif
}

// ZSTD_RowFindBestMatch (noDict, mls 5, rowLog 6)
SQ_DEV u32 row_find(MatchState& ms, const u8* const ip, const u8* const iLimit, u32* offsetPtr) {
    const u8* const base = ms.base;
    u32 const curr = (u32)(ip - base);
    u32 const lowLimit = SQZ_WINDOW_START;
    u32 ml = 4 - 1;
    PROF_T0(t_ru);
    if (!ms.lazySkipping) {
        row_update(ms, curr);
    } else {
        ms.nextToUpdate = curr;
    }
    PROF_ADD(4, t_ru);
    PROF_T0(t_rf);
    u32 const hash = hash5(srd64(ip), ms.rowHashLog + 8);
    u32 const relRow = (hash >> 8) << 6;
    u32 const tag = hash & 0xFF;
    u32* const row = ms.hashTable + relRow;
    u8* const tagRow = ms.tagTable + relRow;
    u32 const limitLen = (u32)(iLimit - ip);
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    // (earlier table stores -- this search's row_update, the previous search's insertion -- land first)
    wv_mem_sync();
    // the whole row in ONE round trip: lane l reads tag byte l and entry l; byte 0 is the head, and the row is
    // then rotated to start at the head with two wave shuffles (no second, dependent trip for "head + lane")
    u32 const tl = sq_tld8(tagRow + lane);
    u32 const cl = sq_tld32(row + lane);
    u32 const hb = wv_readlane(tl, 0);
    u32 const head = hb & 63;
    u32 const matchPos = (head + lane) & 63;
    u32 const t = (u32)__shfl((int)tl, (int)matchPos, 64);
    u32 const cand = (u32)__shfl((int)cl, (int)matchPos, 64);
    bool const hit = (t == tag) && (matchPos != 0);
    u64 const M = wv_ballot(hit);
    if (wv_ballot(hit && cand >= curr)) ms.fault = 1;
    u64 const B = wv_ballot(hit && cand < lowLimit);
    u64 valid = M;
    if (B) {
        u32 const jb = (u32)__builtin_ctzll(B);
        valid &= (jb == 0) ? 0ull : (~0ull >> (64 - jb));
    }
    wv_fence();
    { // insert the current position (ZSTD_row_nextIndex)
        u32 const pos = row_pos_after(hb, 1);
        if (lane == 0) {
            sq_gwr8(tagRow, pos);
            sq_gwr8(tagRow + pos, tag);
            sq_gst32(row + pos, ms.nextToUpdate);
        }
        ms.nextToUpdate++;
    }
    // (no drain here: nothing reads the tables before the next row_insert_range / row_find, which drain first)
    if (valid) {
        // one candidate per lane: full match length, then first-maximum selection
        u32 L = 0;
        if ((valid >> lane) & 1) {
            const u8* const m = base + cand;
            u32 o = 0;
            bool done = false;
            if (limitLen >= 32) { // the first 32 bytes of both sides in one round trip
                sq_v4u const a0 = sq_grd128(ip), a1 = sq_grd128(ip + 16), b0 = sq_grd128(m), b1 = sq_grd128(m + 16);
                u32 const x[8] = {a0.x ^ b0.x, a0.y ^ b0.y, a0.z ^ b0.z, a0.w ^ b0.w, a1.x ^ b1.x, a1.y ^ b1.y, a1.z ^ b1.z, a1.w ^ b1.w};
                o = 32;
#pragma unroll
                for (int k = 7; k >= 0; k--)
                    if (x[k]) o = 4 * (u32)k + ((u32)__builtin_ctz(x[k]) >> 3);
                done = o < 32;
            }
            while (!done && o + 8 <= limitLen) {
                u64 x = sq_grd64(ip + o) ^ sq_grd64(m + o);
                if (x) {
                    o += (u32)(__builtin_ctzll(x) >> 3);
                    done = true;
                } else
                    o += 8;
            }
            while (!done && o < limitLen) {
                if (sq_grd8(ip + o) != sq_grd8(m + o))
                    done = true;
                else
                    o++;
            }
            L = o;
        }
        u32 mx = L;
        for (int s = 32; s >= 1; s >>= 1) {
            u32 y = (u32)__shfl_xor((int)mx, s, 64);
            mx = y > mx ? y : mx;
        }
        mx = wv_uni(mx);
        if (mx > ml) {
            u64 const who = wv_ballot(L == mx && ((valid >> lane) & 1));
            u32 const w = (u32)__builtin_ctzll(who);
            u32 const idx = wv_readlane(cand, w);
            ml = mx;
            *offsetPtr = OFFSET_TO_OFFBASE(curr - idx);
        }
    }
#else
    {
        u32 const head = tagRow[0] & 63;
        u32 matchBuffer[64];
        u32 numMatches = 0;
        u32 nbAttempts = 64;
        for (u32 j = 0; j < 64 && nbAttempts > 0; j++) {
            u32 const matchPos = (head + j) & 63;
            if (tagRow[matchPos] != (u8)tag) continue;
            u32 const matchIndex = row[matchPos];
            if (matchPos == 0) continue;
            if (matchIndex < lowLimit) break;
            matchBuffer[numMatches++] = matchIndex;
            --nbAttempts;
        }
        {
            u32 const pos = row_pos_after(tagRow[0], 1);
            tagRow[0] = (u8)pos;
            tagRow[pos] = (u8)tag;
            row[pos] = ms.nextToUpdate++;
        }
        for (u32 c = 0; c < numMatches; ++c) {
            u32 const matchIndex = matchBuffer[c];
            u32 currentMl = 0;
            const u8* const match = base + matchIndex;
            if (sq_rd32(match + ml - 3) == sq_rd32(ip + ml - 3)) currentMl = wv_count(ip, match, limitLen);
            if (currentMl > ml) {
                ml = currentMl;
                *offsetPtr = OFFSET_TO_OFFBASE(curr - matchIndex);
                if (currentMl == limitLen) break;
            }
        }
    }
#endif
    PROF_ADD(5, t_rf);
    return ml;
}

// ================================================================ lazy parser (depth 2)
template <int ROW>
SQ_DEV u32 search_max(MatchState& ms, const u8* ip, const u8* iend, u32* offBase) {
    if (ROW) return row_find(ms, ip, iend, offBase);
    return bt_find_best(ms, ip, iend, offBase);
}

// ZSTD_compressBlock_lazy_generic (noDict, depth 2); returns last literals size
template <int ROW>
SQ_DEV u32 compressBlock_lazy_impl(MatchState& ms, SeqStore& ss, u32* rep, const u8* src, u32 srcSize) {
    const u8* const istart = src;
    const u8* ip = istart;
    const u8* anchor = istart;
    const u8* const iend = istart + srcSize;
    const u8* const ilimit = ROW ? iend - 8 - 8 : iend - 8;
    const u8* const base = ms.base;
    const u8* const prefixLowest = base + SQZ_WINDOW_START;
    u32 offset_1 = rep[0], offset_2 = rep[1];
    u32 offsetSaved1 = 0, offsetSaved2 = 0;

    u32 failStreak = 0, exactAt = 0; // speculative epochs start after 16 plain failures in a row
    ip += (ip == prefixLowest);
    {
        u32 const curr = (u32)(ip - base);
        u32 const maxRep = curr - SQZ_WINDOW_START;
        if (offset_2 > maxRep) offsetSaved2 = offset_2, offset_2 = 0;
        if (offset_1 > maxRep) offsetSaved1 = offset_1, offset_1 = 0;
    }
    ms.lazySkipping = 0;
    ms.tailSeqs = 0; // sequences are counted per block
    ms.tailBlkTried = 0;

    if (!ROW && (ms.certEnd | ms.certEnd1)) {
        // No-match certificate (cert_compute): the scheduled searches of this block at positions
        // below its certEnd certainly fail.  Skip them: what remains of them is the parser state
        // (ip, nextToUpdate) and their deferred tree updates (cert_mark_deferred).
        u32 const blk = (u32)(istart - base), ipPos = (u32)(ip - base), ilimPos = (u32)(ilimit - base);
        bool const second = blk != SQZ_WINDOW_START;
        u32 const predEnd = (ms.chunkEndPos - blk < SQZ_BLOCKSIZE_MAX) ? ms.chunkEndPos : blk + SQZ_BLOCKSIZE_MAX;
        bool const asPredicted = (blk == SQZ_WINDOW_START || blk == SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX) &&
                                 (blk + srcSize == predEnd) && (ss.nseq == 0) && (second || offset_1 == 1);
        bool const live = ms.eager || ms.nLive; // the finder holds tables that later searches will read
        u32 ce = second ? ms.certEnd1 : ms.certEnd;
        if (asPredicted && second && ce > ipPos && offset_1 > 1) // block 0 left another repcode behind
            ce = cert_rep_recheck(ms, offset_1, ce);
        if (!asPredicted || ce <= ipPos || ilimit <= ip || (live && ce < ilimPos)) {
            // nothing to skip here (a partial certificate is only used while the finder is still
            // asleep: otherwise the exact code simply does the whole block)
            if (!asPredicted || ce <= ipPos || live) {
                if (second)
                    ms.certEnd1 = 0;
                else
                    ms.certEnd = 0;
            }
        } else if (live) {
            // every search of this (last) block fails and nothing after it can observe the tables:
            // no search, no insertion, no bookkeeping
            ip = ilimit;
            SQ_TRACE(38, 1);
        } else {
            PROF_T0(t_bw);
            // this block's part of the schedule is a prefix of one of the two tables
            const u32* const ent = ms.schedTab + 2 + (second ? SQZ_SCHEDTAB_N : 0u);
            u32 const nEnt = second ? ms.nCert - ms.certCnt0 : ms.certCnt0;
            u32 const cnt = (ce >= ilimPos) ? nEnt : sched_count(ent, nEnt, ce - blk);
            u32 const last = cnt ? blk + gld32(ent + cnt - 1) : 0u;
            u32 const pz = (cnt < nEnt) ? blk + gld32(ent + cnt) : ilimPos;
            PROF_ADD(28, t_bw);
            ms.nCertDone += cnt;
            if (cnt && last + 1 > ms.nextToUpdate) ms.nextToUpdate = last + 1;
            ip = base + pz;
            failStreak = SQZ_EPOCH_AFTER_FAILS;
            if (pz < ilimPos) { // the exact code takes over at the first position that is not certified
                exactAt = pz;
                if (second)
                    ms.certEnd1 = 0;
                else
                    ms.certEnd = 0;
            }
        }
    }

    while (ip < ilimit) {
        u32 matchLength = 0;
        u32 offBase = 1; // REPCODE1_TO_OFFBASE
        const u8* start = ip + 1;
        ms.anchorPos = (u32)(anchor - base);
        if (!ROW && ms.tailTried < SQZ_TAIL_TRIES && ms.tailAt != (u32)(ip - base) &&
            (ss.nseq > ms.tailSeqs || (!ms.tailBlkTried && failStreak >= SQZ_EPOCH_AFTER_FAILS)) && (ms.eager || ms.nLive)) {
            // The finder was woken by a hand-over and the exact code has dealt with what stopped the
            // certificate (a sequence has been stored, or the search found nothing after all): before
            // the next search (and the schedule repair it would trigger), ask the tail certificate whether the rest of this block can
            // find anything at all.  Only where no later code can observe the tables we would leave
            // behind: in the last block, or when the next (last) block is certified whatever repcode
            // offset this block passes on (offset 1 if it ends raw, offset_1 if it is compressed).
            u32 const blkEnd = (u32)(iend - base);
            bool lastOrCovered = blkEnd >= ms.chunkEndPos;
            if (!lastOrCovered && blkEnd == SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX && ms.certEnd1 &&
                ms.chunkEndPos - blkEnd <= SQZ_BLOCKSIZE_MAX) {
                u32 const lim1 = ms.chunkEndPos - 8;
                lastOrCovered = ms.certEnd1 >= lim1 && (offset_1 <= 1 || cert_rep_recheck(ms, offset_1, lim1) >= lim1);
            }
            ms.tailTried++;
            SQ_TRACE(43, ms.tailTried);
            ms.tailAt = (u32)(ip - base); // not again before the parser has moved on ...
            ms.tailSeqs = ss.nseq;        // ... and stored another sequence
            ms.tailBlkTried = 1;
            if (lastOrCovered) {
                PROF_T0(t_tailc);
                bool const tailOk = tail_certify(ms, (u32)(ip - base), (u32)(anchor - base), offset_1, (u32)(ilimit - base));
                PROF_ADD(25, t_tailc);
                PROF_CNT(26, 100000); // count of tail certificates, in the time slots' unit (1 "ms" = 1)
                if (tailOk) {
                    SQ_TRACE(39, 1);
                    ip = ilimit; // every remaining search of the block fails; nothing observes the tables any more
                    break;
                }
                if (!ms.eager) bloom_rebuild(ms);
            }
        }
        if (!ROW && !ms.specOff && failStreak >= SQZ_EPOCH_AFTER_FAILS && (u32)(ip - base) != exactAt) {
            u32 nextPos = 0, stopped = 1;
            u32 const f = spec_epoch(ms, (u32)(ip - base), ms.anchorPos, offset_1, (u32)(ilimit - base), &nextPos, &stopped);
            if (ms.fault) break;
            // a "maybe" stopped the epoch (or it was too short to run): the exact code does that position
            exactAt = stopped ? (f ? nextPos : (u32)(ip - base)) : 0;
            ms.nEpoch++;
            ms.nSpecFail += f;
            if (f) {
                failStreak += f;
                ip = base + nextPos;
                continue;
            }
        }

        if ((offset_1 > 0) && (srd32(ip + 1 - offset_1) == srd32(ip + 1))) {
            matchLength = wv_count(ip + 1 + 4, ip + 1 + 4 - offset_1, (u32)(iend - (ip + 1 + 4))) + 4;
        }
        {
            u32 offbaseFound = 999999999;
            u32 const ml2 = search_max<ROW>(ms, ip, iend, &offbaseFound);
            if (ml2 > matchLength) matchLength = ml2, start = ip, offBase = offbaseFound;
        }
        if (matchLength < 4) {
            u32 const step = ((u32)(ip - anchor) >> 8) + 1;
            ip += step;
            ms.lazySkipping = step > 8;
            failStreak++;
            continue;
        }
        failStreak = 0;
        while (ip < ilimit) {
            ip++;
            if ((offset_1 > 0) && (srd32(ip) == srd32(ip - offset_1))) {
                u32 const mlRep = wv_count(ip + 4, ip + 4 - offset_1, (u32)(iend - (ip + 4))) + 4;
                int const gain2 = (int)(mlRep * 3);
                int const gain1 = (int)(matchLength * 3 - sq_highbit32(offBase) + 1);
                if ((mlRep >= 4) && (gain2 > gain1)) matchLength = mlRep, offBase = 1, start = ip;
            }
            {
                u32 ofbCandidate = 999999999;
                u32 const ml2 = search_max<ROW>(ms, ip, iend, &ofbCandidate);
                int const gain2 = (int)(ml2 * 4 - sq_highbit32(ofbCandidate));
                int const gain1 = (int)(matchLength * 4 - sq_highbit32(offBase) + 4);
                if ((ml2 >= 4) && (gain2 > gain1)) {
                    matchLength = ml2, offBase = ofbCandidate, start = ip;
                    continue;
                }
            }
            if (ip < ilimit) {
                ip++;
                if ((offset_1 > 0) && (srd32(ip) == srd32(ip - offset_1))) {
                    u32 const mlRep = wv_count(ip + 4, ip + 4 - offset_1, (u32)(iend - (ip + 4))) + 4;
                    int const gain2 = (int)(mlRep * 4);
                    int const gain1 = (int)(matchLength * 4 - sq_highbit32(offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1)) matchLength = mlRep, offBase = 1, start = ip;
                }
                {
                    u32 ofbCandidate = 999999999;
                    u32 const ml2 = search_max<ROW>(ms, ip, iend, &ofbCandidate);
                    int const gain2 = (int)(ml2 * 4 - sq_highbit32(ofbCandidate));
                    int const gain1 = (int)(matchLength * 4 - sq_highbit32(offBase) + 7);
                    if ((ml2 >= 4) && (gain2 > gain1)) {
                        matchLength = ml2, offBase = ofbCandidate, start = ip;
                        continue;
                    }
                }
            }
            break;
        }
        if (OFFBASE_IS_OFFSET(offBase)) {
            u32 const off = OFFBASE_TO_OFFSET(offBase);
            while ((start > anchor) && (start - off > prefixLowest) && (srd8(start - 1) == srd8(start - off - 1))) {
                start--;
                matchLength++;
            }
            offset_2 = offset_1;
            offset_1 = off;
        }
        {
            u32 const litLength = (u32)(start - anchor);
            storeSeq(ss, litLength, anchor, offBase, matchLength);
            anchor = ip = start + matchLength;
        }
        if (ms.lazySkipping) ms.lazySkipping = 0;

        while ((ip <= ilimit) && (offset_2 > 0) && (srd32(ip) == srd32(ip - offset_2))) {
            matchLength = wv_count(ip + 4, ip + 4 - offset_2, (u32)(iend - (ip + 4))) + 4;
            offBase = offset_2;
            offset_2 = offset_1;
            offset_1 = offBase;
            storeSeq(ss, 0, anchor, 1, matchLength);
            ip += matchLength;
            anchor = ip;
        }
    }
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    return (u32)(iend - anchor);
}

// The parser works on a LOCAL copy of the finder's state.  Through the caller's reference every field access
// is a flat load (a full vmcnt/lgkmcnt drain in the middle of the parser's loop, since any global store might
// alias it).  The row-hash variant (2 MiB chunks) inlines everything it calls, so its copy lives in registers;
// the btlazy2 variant still hands the copy to its out-of-line helpers, but it is a private object now: hipcc
// keeps its fields in registers between those calls (the tree walk of a search makes none).
template <int ROW>
SQ_DEVFN u32 compressBlock_lazy(MatchState& ms, SeqStore& ss, u32* rep, const u8* src, u32 srcSize) {
#if SQ_ON_DEVICE
    MatchState lm = ms;
    u32 const r = compressBlock_lazy_impl<ROW>(lm, ss, rep, src, srcSize);
    ms = lm;
    return r;
#else
    return compressBlock_lazy_impl<ROW>(ms, ss, rep, src, srcSize);
#endif
}

#include "sq_zstd_rowbatch.h"

} // namespace sqz
