// sq_zstd_kernel.h -- zstd level-12 frame compressor kernel for gfx950: one wavefront per chunk,
// persistent waves pulling chunk indices from a device work queue.
//
// Replaces (bit-exact) the call at /root/reference/src/util/chunk.rs:89-90
//     compress(chunk, COMPRESSION_LEVEL)        // zstd::bulk::compress(chunk, 12), libzstd 1.5.7
// i.e. one self-contained zstd frame per chunk from a fresh context: magic, single-segment frame
// header with content size, <=128 KiB blocks (1.5.7 pre-block splitter), no checksum.
//
// HBM layout: each wave owns one workspace (ws_layout): hash table, chain/tag table, 3-byte hash
// table (btopt), sequence store (sequences, literals, code bytes), btopt tables, pre-split
// fingerprints.  Entropy tables and statistics live in LDS (sq_zstd_entropy.h).  Frames are
// written straight into per-chunk output slots.
#pragma once
#include "sq_zstd_opt.h"
#include "sq_zstd_fast.h"

namespace sqz {

#define SQZ_MAX_NBSEQ (SQZ_BLOCKSIZE_MAX / 3 + 2)

struct Fingerprint {
    u32 events[1024];
    u32 nbEvents;
    u32 pad[3];
};

struct WsLayout {
    size_t hashTable, table2, hashTable3, seqs, lits, llCode, ofCode, mlCode, stBits, optState, fp, sched, liveList,
        skipList, blockTab, certTab, dummy, total;
};

#if SQ_ON_DEVICE
#define SQ_HD __host__ __device__ inline
#else
#define SQ_HD static inline
#endif

SQ_HD size_t ws_align(size_t x) { return (x + 255) & ~(size_t)255; }

SQ_HD WsLayout ws_layout(u32 max_len) {
    // table areas sized for the largest parameters of any chunk length <= max_len
    size_t hashBytes = 0, t2Bytes = 0, certBytes = 0;
    const u32 bps[9] = {16u << 10, 32u << 10, 64u << 10, 128u << 10, 256u << 10, 512u << 10, 1u << 20, 2u << 20, max_len};
    for (int i = 0; i < 9; i++) {
        u32 n = bps[i] < max_len ? bps[i] : max_len;
        if (n == 0) n = 1;
        // get_cparams, restated for host+device use
        u32 hashLog, chainLog, strategy;
        if (n <= (16u << 10)) {
            u32 w = 0;
            while ((1u << w) < n) w++;
            chainLog = (w + 1 < 7) ? 7 : w + 1;
            hashLog = chainLog > 14 ? 14 : chainLog;
            strategy = 7;
        } else if (n <= (256u << 10)) {
            u32 w = 15;
            while ((1u << w) < n) w++;
            chainLog = w + 1;
            hashLog = (w == 17) ? 17 : w + 1;
            strategy = 6;
        } else {
            u32 w = 19;
            while ((1u << w) < n && w < 21) w++;
            chainLog = w;
            hashLog = w + 1;
            strategy = 5;
        }
        size_t hb = ((size_t)4) << hashLog;
        size_t tb = (strategy == 5) ? ((size_t)1 << hashLog) : (((size_t)4) << chainLog);
        if (hb > hashBytes) hashBytes = hb;
        if (tb > t2Bytes) t2Bytes = tb;
        // table of a tail certificate (tail_certify): 8 bytes x 2^(chainLog-1), btlazy2 only
        if (strategy == 6 && ((size_t)8 << (chainLog - 1)) > certBytes) certBytes = (size_t)8 << (chainLog - 1);
    }
    WsLayout L;
    size_t o = 0;
    L.hashTable = o;
    o += ws_align(hashBytes);
    L.table2 = o;
    o += ws_align(t2Bytes);
    L.hashTable3 = o;
    o += ws_align((size_t)4 << 14);
    L.seqs = o;
    o += ws_align((size_t)SQZ_MAX_NBSEQ * 8);
    L.lits = o;
    o += ws_align(SQZ_BLOCKSIZE_MAX + 64);
    L.llCode = o;
    o += ws_align(SQZ_MAX_NBSEQ);
    L.ofCode = o;
    o += ws_align(SQZ_MAX_NBSEQ);
    L.mlCode = o;
    o += ws_align(SQZ_MAX_NBSEQ);
    L.stBits = o;
    o += ws_align((size_t)SQZ_MAX_NBSEQ * 8);
    L.optState = o;
    o += ws_align(sizeof(opt::optstate));
    L.fp = o;
    o += ws_align(2 * sizeof(Fingerprint));
    L.sched = o;
    o += ws_align((size_t)SQZ_SCHED_CAP * 4);
    L.liveList = o;
    o += ws_align((size_t)SQZ_LIVE_CAP * 4);
    L.skipList = o;
    o += ws_align((size_t)SQZ_SKIP_CAP * 8);
    L.blockTab = o;
    o += ws_align((size_t)SQZ_BLOCKTAB_CAP * 8);
    L.certTab = o;
    o += ws_align(certBytes);
    L.dummy = o;
    o += 256;
    L.total = o;
    return L;
}
SQ_HD size_t workspace_bytes(u32 max_len) { return ws_layout(max_len).total + 1024; } // + debug / profiling tail

// The constant no-match schedule tables (layout: see "no-match schedule tables" in sq_zstd_match.h);
// filled once by the host (SQZ_SCHEDTAB_WORDS words).
SQ_HD void sched_tab_fill(u32* tab) {
    for (u32 variant = 0; variant < 2; variant++) {
        u32* const e = tab + 2 + variant * SQZ_SCHEDTAB_N;
        u32 n = 0, p = (variant == 0) ? 1u : 0u;
        while (p < SQZ_BLOCKSIZE_MAX - 8 && n < SQZ_SCHEDTAB_N) {
            e[n++] = p;
            p += (p >> 8) + 1;
        }
        tab[variant] = n;
        // level words: smallest scheduled position >= 256 k | ceil(65536 / (k + 1)) << 17
        u32* const lv = tab + SQZ_SCHEDLVL_OFF + variant * SQZ_SCHEDLVL_N;
        u32 j = 0;
        for (u32 k = 0; k < SQZ_SCHEDLVL_N; k++) {
            while (j < n && e[j] < (k << 8)) j++;
            u32 const first = (j < n) ? e[j] : 0x1FFFFu;
            u32 const R = (65536u + k) / (k + 1);
            lv[k] = first | ((R > 0x7FFFu ? 0x7FFFu : R) << 17);
        }
        // the table as a bitmap: bit (p & 15) of the u16 at window (p >> 4)
        u16* const bits = (u16*)(tab + SQZ_SCHEDBITS_OFF) + variant * SQZ_SCHEDBITS_N;
        for (u32 w = 0; w < SQZ_SCHEDBITS_N; w++) bits[w] = 0;
        for (u32 i = 0; i < n; i++) bits[e[i] >> 4] = (u16)(bits[e[i] >> 4] | (1u << (e[i] & 15)));
    }
}

// ---------------------------------------------------------------- pre-block splitter (1.5.7)
SQ_DEV u32 ps_hash2_10(const u8* p) { return ((u32)sq_grd16(p) * 0x9e3779b9U) >> 22; }

// recordFingerprint_generic(samplingRate 5, hashLog 10) over 8 KiB
SQ_DEV void ps_record(Fingerprint* fp, const u8* src) {
    const u32 limit = 8192 - 2 + 1;
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    SQ_G u32* const ev = (SQ_G u32*)fp->events;
    for (u32 i = lane; i < 1024; i += 64) ev[i] = 0;
    wv_mem_sync();
    for (u32 n = lane * 5; n < limit; n += 64 * 5) __hip_atomic_fetch_add(&ev[ps_hash2_10(src + n)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) sq_gst32(&fp->nbEvents, limit / 5);
    wv_mem_sync();
#else
    for (u32 i = 0; i < 1024; i++) fp->events[i] = 0;
    for (u32 n = 0; n < limit; n += 5) fp->events[ps_hash2_10(src + n)]++;
    fp->nbEvents = limit / 5;
#endif
}
// compareFingerprints: 1 when "too different"
SQ_DEV u32 ps_compare(const Fingerprint* ref, const Fingerprint* nw, int penalty) {
    u64 distance = 0;
    const s64 nbA = (s64)gld32(&ref->nbEvents), nbB = (s64)gld32(&nw->nbEvents);
#if SQ_ON_DEVICE
    for (u32 n = wv_lane(); n < 1024; n += 64) {
        s64 d = (s64)sq_gld32(&ref->events[n]) * nbB - (s64)sq_gld32(&nw->events[n]) * nbA;
        distance += (u64)(d < 0 ? -d : d);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        u32 lo = (u32)__shfl_xor((int)(u32)distance, m, 64);
        u32 hi = (u32)__shfl_xor((int)(u32)(distance >> 32), m, 64);
        distance += ((u64)hi << 32) | lo;
    }
    distance = wv_uni64(distance);
#else
    for (u32 n = 0; n < 1024; n++) {
        s64 d = (s64)ref->events[n] * nbB - (s64)nw->events[n] * nbA;
        distance += (u64)(d < 0 ? -d : d);
    }
#endif
    u64 const p50 = (u64)nbA * (u64)nbB;
    u64 const threshold = p50 * (u64)(16 - 2 + penalty) / 16;
    return distance >= threshold;
}
SQ_DEV void ps_merge(Fingerprint* acc, const Fingerprint* nw) {
#if SQ_ON_DEVICE
    const u32 lane = wv_lane();
    for (u32 n = lane; n < 1024; n += 64) sq_gst32(&acc->events[n], sq_gld32(&acc->events[n]) + sq_gld32(&nw->events[n]));
    if (lane == 0) sq_gst32(&acc->nbEvents, sq_gld32(&acc->nbEvents) + sq_gld32(&nw->nbEvents));
    wv_mem_sync();
#else
    for (u32 n = 0; n < 1024; n++) acc->events[n] += nw->events[n];
    acc->nbEvents += nw->nbEvents;
#endif
}
// ZSTD_splitBlock_byChunks(level 2) on a full 128 KiB block
SQ_NOINLINE u32 splitBlock_byChunks(const u8* p, Fingerprint* fps) {
    Fingerprint* const past = fps;
    Fingerprint* const nw = fps + 1;
    int penalty = 3;
    ps_record(past, p);
    for (u32 pos = 8192; pos <= SQZ_BLOCKSIZE_MAX - 8192; pos += 8192) {
        ps_record(nw, p + pos);
        if (ps_compare(past, nw, penalty)) return pos;
        ps_merge(past, nw);
        if (penalty > 0) penalty--;
    }
    return SQZ_BLOCKSIZE_MAX;
}
// ZSTD_optimalBlockSize for strategies lazy2 / btlazy2 (split level 3) / btopt (never >= 128 KiB)
SQ_DEV u32 optimalBlockSize(const u8* src, u32 srcSize, s64 savings, Fingerprint* fps) {
    if (srcSize < SQZ_BLOCKSIZE_MAX) return srcSize;
    if (savings < 3) return SQZ_BLOCKSIZE_MAX;
    return splitBlock_byChunks(src, fps);
}

// ---------------------------------------------------------------- one frame, one wave
struct BlockState { // entropy tables are g_lds.ent[idx]
    u32 rep[2][SQZ_REP_NUM];
    u32 prevIdx;
    u32 isFirstBlock;
};

// ZSTD_buildSeqStore: the match finder + parser of one block (at least 7 bytes); fills ss, leaves the block's
// repcodes in bs.rep[bs.prevIdx ^ 1]; returns the size of the last literals.  LAZY2_ONLY: instantiated by the
// match-finding kernel of the lazy2 size class (k_find_batch), which must not carry the other strategies' code.
template <bool LAZY2_ONLY>
SQ_DEV u32 cb_find(MatchState& ms, SeqStore& ss, BlockState& bs, void* optState, const u8* src, u32 srcSize) {
    u32 const p = bs.prevIdx, nx = p ^ 1;
    ss.nseq = 0;
    ss.nlits = 0;
    ss.longLengthType = 0;
    ss.longLengthPos = 0;
    {
        u32 const curr = (u32)(src - ms.base);
        ms.blockEndPos = curr + srcSize;
        if constexpr (!LAZY2_ONLY) {
            if (ms.cp.strategy == STRAT_BTLAZY2) {
                if (ms.nBlocks < SQZ_BLOCKTAB_CAP) {
                    gst32(ms.blockTab + 2 * ms.nBlocks, curr);
                    gst32(ms.blockTab + 2 * ms.nBlocks + 1, curr + srcSize);
                    ms.nBlocks++;
                    wv_mem_sync();
                } else
                    ms.specOff = 1; // no room to remember this block's end: no deferred searches in it
            }
            if (!ms.eager && !bs.isFirstBlock) bloom_rebuild(ms); // the entropy stage reused the LDS
            ms.sbMax = 256; // live buckets are dense right after a block start; grows back as they thin out
        }
        if (curr > ms.nextToUpdate + 384) {
            u32 const d = curr - ms.nextToUpdate - 384;
            u32 const n2u = curr - (192 < d ? 192 : d);
            if constexpr (!LAZY2_ONLY) filter_add_skip(ms, ms.nextToUpdate, n2u);
            ms.nextToUpdate = n2u;
        }
    }
    ms.nxEnt = nx;
    for (int i = 0; i < SQZ_REP_NUM; ++i) bs.rep[nx][i] = bs.rep[p][i];
    u32 lastLLSize;
    SQ_TRACE(0, 6);
    if (LAZY2_ONLY || ms.cp.strategy == STRAT_LAZY2) {
        lastLLSize = compressBlock_lazy_rowbatch(ms, ss, bs.rep[nx], src, srcSize);
    } else if constexpr (!LAZY2_ONLY) {
        if (ms.cp.strategy == STRAT_BTLAZY2) {
            lastLLSize = compressBlock_lazy<0>(ms, ss, bs.rep[nx], src, srcSize);
        } else {
            // btopt: the whole parse is one serial region on lane 0
            u32 r0 = bs.rep[nx][0], r1 = bs.rep[nx][1], r2 = bs.rep[nx][2];
            WV_SERIAL {
                opt::mstate om;
                om.base = ms.base;
                om.nextToUpdate = ms.nextToUpdate;
                om.hashTable = ms.hashTable;
                om.chainTable = ms.chainTable;
                om.hashTable3 = ms.hashTable3;
                om.hashLog3 = ms.hashLog3;
                om.cp = ms.cp;
                om.opt = optState;
                opt::seqstore os;
                os.seqs = ss.seqs;
                os.lits = ss.lits;
                os.nseq = 0;
                os.nlits = 0;
                os.longLengthType = 0;
                os.longLengthPos = 0;
                u32 rr[3] = {r0, r1, r2};
                u32 const last = (u32)opt::compressBlock_btopt(&om, &os, rr, src, srcSize);
                g_lds.ret[0] = last;
                g_lds.ret[1] = os.nseq;
                g_lds.ret[2] = os.nlits;
                g_lds.ret[3] = os.longLengthType;
                g_lds.ret[4] = os.longLengthPos;
                g_lds.ret[5] = rr[0];
                g_lds.ret[6] = rr[1];
                g_lds.ret[7] = rr[2];
            }
            wv_mem_sync();
            lastLLSize = g_lds.ret[0];
            ss.nseq = g_lds.ret[1];
            ss.nlits = g_lds.ret[2];
            ss.longLengthType = g_lds.ret[3];
            ss.longLengthPos = g_lds.ret[4];
            bs.rep[nx][0] = g_lds.ret[5];
            bs.rep[nx][1] = g_lds.ret[6];
            bs.rep[nx][2] = g_lds.ret[7];
            wv_fence();
        }
    }
    SQ_TRACE(0, 7);
    SQ_TRACE(4, ss.nseq);
    return lastLLSize;
}

// ZSTD_compressBlock_internal after the sequence store is built: the entropy stage.  Returns the body size; 0 => raw;
// 1 => RLE; confirms the block's repcodes and entropy tables (bs.prevIdx) when it is emitted compressed.
SQ_DEV u32 cb_emit(u32 strategy, SeqStore& ss, BlockState& bs, u8* dst, u32 dstCap, const u8* src, u32 srcSize, u32 lastLLSize) {
    u32 const p = bs.prevIdx, nx = p ^ 1;
    u32 cSize;
    if (ss.nseq == 0 && ss.nlits == 0 && lastLLSize == srcSize) {
        // no sequence at all: the literals ARE the block; the entropy stage reads them in place
        SeqStore whole = ss;
        whole.lits = const_cast<u8*>(src);
        whole.nlits = srcSize;
        ss.nlits = srcSize;
        wv_mem_sync_inv();
        cSize = entropy_compress_seqstore(whole, &g_lds.ent[p], &g_lds.ent[nx], strategy, dst, dstCap, srcSize);
    } else {
        PROF_T0(t_ll);
        if (lastLLSize) wv_memcpy(ss.lits + ss.nlits, src + srcSize - lastLLSize, lastLLSize);
        PROF_ADD(7, t_ll);
        ss.nlits += lastLLSize;
        wv_mem_sync_inv(); // sequence records / literals written by other lanes have landed
        cSize = entropy_compress_seqstore(ss, &g_lds.ent[p], &g_lds.ent[nx], strategy, dst, dstCap, srcSize);
    }
    SQ_TRACE(0, 9);
    SQ_TRACE(5, cSize);
    if (is_err(cSize)) return cSize;
    if (!bs.isFirstBlock && cSize < 25 && wv_is_rle(src, srcSize)) {
        cSize = 1;
        gst8(dst, srd8(src));
    }
    if (cSize > 1) bs.prevIdx ^= 1; // ZSTD_blockState_confirmRepcodesAndEntropyTables
    // offcode_repeatMode valid -> check: REP_VALID never occurs without a dictionary
    return cSize;
}

// ZSTD_buildSeqStore + ZSTD_compressBlock_internal: returns body size; 0 => raw; 1 => RLE
SQ_DEVFN u32 compressBlock_internal(MatchState& ms, SeqStore& ss, BlockState& bs, void* optState, u8* dst,
                                    u32 dstCap, const u8* src, u32 srcSize) {
    if (srcSize < 1 + 1 + 3 + 1 + 1) return 0; // ZSTDbss_noCompress
    u32 const lastLLSize = cb_find<false>(ms, ss, bs, optState, src, srcSize);
    return cb_emit(ms.cp.strategy, ss, bs, dst, dstCap, src, srcSize, lastLLSize);
}

// State of a wave's hash-table area across the chunks of one launch.  gen == 0: the area holds
// unknown words (start of a launch, or the previous chunk had another workspace layout / strategy).
// Consecutive btlazy2 chunks of the same size class (same layout) tag their words with gen
// instead of clearing the table (SQZ_HT_TAG in sq_zstd_match.h).
struct TableGen {
    u32 gen;
    u32 windowLog; // size class of the chunk that last cleared / tagged the table
    u32 tailTabClean; // 1 once the tail-certificate table of this size class has been cleared in this launch
};

// Returns the frame length, 0 on error.
// certHint: two words left by k_incompressible_batch for this chunk (certificate ends, see
// fastpath_frame_wave) or nullptr.
SQ_DEVFN u32 compress_frame_wave(const u8* src, u32 n, u8* dst, u32 cap, u8* ws, const WsLayout& L, TableGen& tg,
                                 const u32* schedTab, const u32* certHint) {
    MatchState ms;
    SeqStore ss;
    BlockState bs;
    CParams const cp = get_cparams(n);
    u8* op = dst;
    u8* const oend = dst + cap;
    WV_SERIAL { g_lds.trace = (u32*)(ws + L.dummy) + 8; }
    wv_mem_sync_inv(); // previous chunk's traffic fully landed; no L1 line survives from it
    SQ_TRACE(0, 1);
    SQ_TRACE(1, n);
    if (cap < 18) return 0;
    // ---- frame header (ZSTD_writeFrameHeader)
    u32 const fcsCode = (n >= 256) + (n >= 65536 + 256);
    u32 hdr;
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
    hdr = 5 + (fcsCode == 0 ? 1 : (fcsCode == 1 ? 2 : 4));
    op += hdr;
    // ---- fresh context
    ms.cp = cp;
    ms.base = src - SQZ_WINDOW_START;
    ms.nextToUpdate = SQZ_WINDOW_START;
    ms.lazySkipping = 0;
    ms.fault = 0;
    ms.hashTable = (u32*)(ws + L.hashTable);
    ms.chainTable = (u32*)(ws + L.table2);
    ms.tagTable = ws + L.table2;
    ms.hashTable3 = (u32*)(ws + L.hashTable3);
    ms.dummy = (u32*)(ws + L.dummy);
    ms.hashLog3 = 0;
    ms.rowHashLog = 0;
    ms.eager = (cp.strategy == STRAT_BTLAZY2) ? 0 : 1;
    ms.nRepairs = 0;
    ms.anchorPos = SQZ_WINDOW_START;
    ms.chunkEndPos = n + SQZ_WINDOW_START;
    ms.blockEndPos = SQZ_WINDOW_START + (n < SQZ_BLOCKSIZE_MAX ? n : SQZ_BLOCKSIZE_MAX);
    ms.sched = (u32*)(ws + L.sched);
    ms.liveList = (u32*)(ws + L.liveList);
    ms.skipList = (u32*)(ws + L.skipList);
    ms.blockTab = (u32*)(ws + L.blockTab);
    ms.schedTab = schedTab;
    ms.certCnt0 = 0;
    ms.tailTab = (u32*)(ws + L.certTab);
    ms.tailTabClean = &tg.tailTabClean;
    ms.tailTried = 0;
    ms.tailAt = 0;
    ms.tailSeqs = 0;
    ms.nCert = ms.nCertDone = ms.certEnd = ms.certEnd1 = 0;
    ms.nLive = ms.nSkip = ms.nBlocks = 0;
    ms.specOff = 0;
    ms.sbMax = 256;
    ms.nxEnt = 1;
    ms.nEpoch = ms.nSpecFail = ms.nReplay = 0;
    SQ_TRACE(0, 2);
    SQ_TRACE(38, 0);
    SQ_TRACE(39, 0);
    PROF_T0(t_clr);
    ms.genBits = 0;
    if (cp.strategy == STRAT_BTLAZY2) {
        if (tg.gen == 0 || tg.gen >= SQZ_HT_GEN_MAX || tg.windowLog != cp.windowLog) {
            wv_fill32(ms.hashTable, 0, (size_t)1 << cp.hashLog);
            // the chain-table area hosts the no-match certificate's tagged entries (cert_compute),
            // unless stage 1 computes the certificates
            if (!certHint) wv_fill32(ms.chainTable, 0, (size_t)1 << cp.chainLog);
            tg.gen = 0;
            tg.windowLog = cp.windowLog;
            tg.tailTabClean = 0;
        }
        tg.gen++;
        ms.genBits = tg.gen << SQZ_HT_GEN_SHIFT;
    } else {
        wv_fill32(ms.hashTable, 0, (size_t)1 << cp.hashLog);
        tg.gen = 0;
    }
    PROF_ADD(1, t_clr);
    SQ_TRACE(0, 3);
    if (cp.strategy == STRAT_LAZY2) {
        ms.rowHashLog = cp.hashLog - 6;
    } else if (cp.strategy == STRAT_BTOPT) {
        ms.hashLog3 = cp.windowLog < 17 ? cp.windowLog : 17;
        wv_fill32(ms.hashTable3, 0, (size_t)1 << ms.hashLog3);
        // optstate: only litLengthSum == 0 matters for the first (only) block
        WV_SERIAL { ((opt::optstate*)(ws + L.optState))->litLengthSum = 0; }
    }
    ss.seqs = (SeqDef*)(ws + L.seqs);
    ss.lits = ws + L.lits;
    ss.llCode = ws + L.llCode;
    ss.ofCode = ws + L.ofCode;
    ss.mlCode = ws + L.mlCode;
    ss.stBits = (u16*)(ws + L.stBits);
    ss.nseq = ss.nlits = 0;
    ss.longLengthType = ss.longLengthPos = 0;
    ss.recordsOnly = 0;
    WV_SERIAL {
        entropy_reset(&g_lds.ent[0]);
        entropy_reset(&g_lds.ent[1]);
    }
    for (int k = 0; k < 2; k++) {
        bs.rep[k][0] = 1;
        bs.rep[k][1] = 4;
        bs.rep[k][2] = 8;
    }
    bs.prevIdx = 0;
    bs.isFirstBlock = 1;
    wv_mem_sync();
    if (!ms.eager && n >= 9) { // btlazy2: predict the no-match search schedule, make its buckets live
        PROF_T0(t_fs);
        // whole-chunk no-match certificate first (its table lives in the chain-table area, which
        // the finder only ever reads through heads it wrote itself)
        PROF_T0(t_pr);
        // a chunk of this strategy has at most two blocks; their schedules are table prefixes
        u32 const len0 = n < SQZ_BLOCKSIZE_MAX ? n : SQZ_BLOCKSIZE_MAX;
        u32 const len1 = n - len0;
        u32 const full0 = gld32(schedTab), full1 = gld32(schedTab + 1);
        ms.certCnt0 = (len0 == SQZ_BLOCKSIZE_MAX) ? full0 : sched_count(schedTab + 2, full0, len0 - 8);
        u32 const cnt1 = (len1 < 9) ? 0u : (len1 == SQZ_BLOCKSIZE_MAX) ? full1 : sched_count(schedTab + 2 + SQZ_SCHEDTAB_N, full1, len1 - 8);
        ms.nCert = ms.certCnt0 + cnt1;
        PROF_ADD(24, t_pr);
        u32 const hint1 = certHint ? gld32(certHint + 1) : 0u;
        if (hint1 & SQZ_CERT_HINT_VALID) { // already computed by the incompressible-chunk kernel
            ms.certEnd = gld32(certHint);
            ms.certEnd1 = hint1 & ~SQZ_CERT_HINT_VALID;
        } else if (ms.nCert > 0 && !certHint) {
            cert_compute(ms);
        }
        wv_zero_lds(g_lds.u.bloom, SQZ_BLOOM_WORDS);
        if (ms.certEnd <= SQZ_WINDOW_START + 1) {
            // nothing certified: predicted buckets live from the start (cheaper than the first repair)
            ms.certEnd = 0;
            filter_repair(ms, SQZ_WINDOW_START + 1, SQZ_WINDOW_START, true);
        } // else: the filter comes to life at the first exact search (dubt_find -> filter_repair)
        PROF_ADD(8, t_fs);
    }

    if (n == 0) { // ZSTD_writeLastEmptyBlock
        WV_SERIAL {
            op[0] = 1;
            op[1] = 0;
            op[2] = 0;
        }
        return hdr + 3;
    }
    const u8* ip = src;
    u32 remaining = n;
    s64 savings = 0;
    while (remaining) {
        SQ_TRACE(0, 4);
        SQ_TRACE(2, (u32)(ip - src));
        PROF_T0(t_ps);
        u32 const blockSize = optimalBlockSize(ip, remaining, savings, (Fingerprint*)(ws + L.fp));
        PROF_ADD(8, t_ps);
        SQ_TRACE(0, 5);
        SQ_TRACE(3, blockSize);
        u32 const lastBlock = (blockSize == remaining);
        if ((u32)(oend - op) < 3 + 2 + 1) return 0;
        if (ms.nextToUpdate < SQZ_WINDOW_START) ms.nextToUpdate = SQZ_WINDOW_START;
        u32 cSize = compressBlock_internal(ms, ss, bs, ws + L.optState, op + 3, (u32)(oend - op) - 3, ip, blockSize);
        if (is_err(cSize) || ms.fault) return 0;
        if (cSize == 0) { // ZSTD_noCompressBlock
            u32 const h = lastBlock + (0u << 1) + (blockSize << 3);
            if (blockSize + 3 > (u32)(oend - op)) return 0;
            WV_SERIAL {
                op[0] = (u8)h;
                op[1] = (u8)(h >> 8);
                op[2] = (u8)(h >> 16);
            }
            PROF_T0(t_raw);
            wv_memcpy(op + 3, ip, blockSize);
            PROF_ADD(7, t_raw);
            cSize = 3 + blockSize;
        } else {
            u32 const h = (cSize == 1) ? lastBlock + (1u << 1) + (blockSize << 3) : lastBlock + (2u << 1) + (cSize << 3);
            WV_SERIAL {
                op[0] = (u8)h;
                op[1] = (u8)(h >> 8);
                op[2] = (u8)(h >> 16);
            }
            cSize += 3;
        }
        savings += (s64)blockSize - (s64)cSize;
        ip += blockSize;
        remaining -= blockSize;
        op += cSize;
        bs.isFirstBlock = 0;
    }
    wv_fence();
    SQ_TRACE(0, 10);
    PROF_CNT(21, ms.nLive);
    PROF_CNT(22, ms.nRepairs * 1000 + ms.eager * 1000000);
    SQ_TRACE(30, ms.eager);
    SQ_TRACE(31, ms.nRepairs);
    SQ_TRACE(32, ms.nLive);
    SQ_TRACE(33, ms.nSkip);
    SQ_TRACE(34, ms.nEpoch);
    SQ_TRACE(35, ms.nSpecFail);
    SQ_TRACE(36, ms.nReplay);
    SQ_TRACE(37, ms.nCertDone);
    return (u32)(op - dst);
}

// ---------------------------------------------------------------- the lazy2 size class cut in two kernels
// Chunks above 256 KiB (lazy2 + row hash, sq_zstd_rowbatch.h) spend > 90 % of their time in the match finder, whose
// throughput grows with the number of resident waves (it waits on dependent random reads); the entropy stage is what
// pins k_compress_batch to 8 waves per CU (19.5 KB of LDS, 256 registers).  So the path is cut where the reference's
// ZSTD_compressBlock_internal is cut (ZSTD_buildSeqStore | ZSTD_entropyCompressSeqStore):
//   k_find_batch  (12 waves per CU: SQZ_LDS_FIND_BYTES of LDS, <= 168 registers) runs the parser over ALL blocks of a
//                 chunk and leaves one record per block + the sequence records (8 B each) in the chunk's find area;
//   k_emit_batch  (8 waves per CU) turns the records into the frame: literals gathered from the source, entropy
//                 stage, block headers -- byte for byte what compress_frame_wave does after its finder call.
// What the finder cannot know is what the entropy stage decides, and two of those decisions feed back into it:
//   (1) a block emitted raw / RLE does not confirm its repcodes (ZSTD_blockState_confirmRepcodesAndEntropyTables):
//       the finder assumes that a block of equal bytes (other than the first) is emitted RLE and every other block with
//       sequences compressed;
//   (2) the pre-block splitter only runs once the frame has saved 3 bytes (ZSTD_optimalBlockSize, `savings`): the
//       finder estimates the savings from what it knows (sequences, literal bytes) and records what it assumed.
// k_emit_batch knows the truth and checks both per block; a chunk where an assumption mattered and was wrong is left
// without a frame (frame length 0) and k_compress_batch compresses it the one-kernel way afterwards.  Results are
// therefore identical by construction; the assumptions only decide how often the work is done twice.
#define SQZ_FIND_MAXBLOCKS 264u // 2 MiB / 8 KiB (the splitter's smallest cut) + slack
#define SQZ_FIND_HDR_WORDS 8u
#define SQZ_FIND_REC_WORDS 8u   // blockSize, nseq, lastLL, longType | longPos << 2, seqStart, flags, splitSize, -
#define SQZ_FIND_F_SPLIT_ON 1u  // the block was sized by the splitter (savings >= 3 assumed)
#define SQZ_FIND_F_RLE 2u       // all bytes equal, not the first block: assumed to be emitted as an RLE block
SQ_HD size_t find_seq_cap(u32 n) { return (size_t)n / 4 + SQZ_FIND_MAXBLOCKS + 64; }
SQ_HD size_t find_area_bytes(u32 n) {
    return ws_align(((size_t)SQZ_FIND_HDR_WORDS + (size_t)SQZ_FIND_REC_WORDS * SQZ_FIND_MAXBLOCKS) * 4 + find_seq_cap(n) * 8);
}
SQ_DEV SeqDef* find_area_seqs(u32* area) { return (SeqDef*)(area + SQZ_FIND_HDR_WORDS + SQZ_FIND_REC_WORDS * SQZ_FIND_MAXBLOCKS); }

// The finder half.  area[0] = 1 when every block of the chunk has its record, 2 when the chunk must take the
// one-kernel path (too many blocks, a finder fault).
SQ_DEVFN void find_frame_wave(const u8* src, u32 n, u8* ws, const WsLayout& L, u32* area) {
    MatchState ms;
    SeqStore ss;
    BlockState bs;
    CParams const cp = get_cparams(n);
    WV_SERIAL { g_lds.trace = (u32*)(ws + L.dummy) + 8; }
    wv_mem_sync_inv();
    ms.cp = cp;
    ms.base = src - SQZ_WINDOW_START;
    ms.nextToUpdate = SQZ_WINDOW_START;
    ms.lazySkipping = 0;
    ms.fault = 0;
    ms.hashTable = (u32*)(ws + L.hashTable);
    ms.chainTable = (u32*)(ws + L.table2);
    ms.tagTable = ws + L.table2;
    ms.hashTable3 = nullptr;
    ms.dummy = (u32*)(ws + L.dummy);
    ms.hashLog3 = 0;
    ms.rowHashLog = cp.hashLog - 6;
    ms.eager = 1;
    ms.nRepairs = 0;
    ms.anchorPos = SQZ_WINDOW_START;
    ms.chunkEndPos = n + SQZ_WINDOW_START;
    ms.blockEndPos = SQZ_WINDOW_START + (n < SQZ_BLOCKSIZE_MAX ? n : SQZ_BLOCKSIZE_MAX);
    ms.sched = nullptr;
    ms.liveList = ms.skipList = ms.blockTab = nullptr;
    ms.schedTab = nullptr;
    ms.certCnt0 = 0;
    ms.tailTab = nullptr;
    ms.tailTabClean = nullptr;
    ms.tailTried = ms.tailAt = ms.tailSeqs = 0;
    ms.nCert = ms.nCertDone = ms.certEnd = ms.certEnd1 = 0;
    ms.nLive = ms.nSkip = ms.nBlocks = 0;
    ms.specOff = 0;
    ms.sbMax = 256;
    ms.nxEnt = 1;
    ms.nEpoch = ms.nSpecFail = ms.nReplay = 0;
    ms.genBits = 0;
    wv_fill32(ms.hashTable, 0, (size_t)1 << cp.hashLog);
    ss.seqs = find_area_seqs(area);
    ss.lits = nullptr;
    ss.llCode = ss.ofCode = ss.mlCode = nullptr;
    ss.stBits = nullptr;
    ss.nseq = ss.nlits = 0;
    ss.longLengthType = ss.longLengthPos = 0;
    ss.recordsOnly = 1;
    for (int k = 0; k < 2; k++) {
        bs.rep[k][0] = 1;
        bs.rep[k][1] = 4;
        bs.rep[k][2] = 8;
    }
    bs.prevIdx = 0;
    bs.isFirstBlock = 1;
    wv_mem_sync();
    const u8* ip = src;
    u32 remaining = n, nBlocks = 0, seqStart = 0, status = 1;
    s64 estSavings = 0; // a lower bound in spirit: literals priced as incompressible, three bytes per sequence
    // (every value that steers this loop is made wave-uniform explicitly -- wv_uni -- so that hipcc keeps the loop and
    // its calls under scalar control: ms / ss / bs live in per-lane scratch across the out-of-line parser call)
    while (remaining) {
        if (nBlocks >= SQZ_FIND_MAXBLOCKS) {
            status = 2;
            break;
        }
        SQ_TRACE(0, 100 + nBlocks);
        SQ_TRACE(2, (u32)(ip - src));
        u32 blockSize = remaining < SQZ_BLOCKSIZE_MAX ? remaining : SQZ_BLOCKSIZE_MAX, flags = 0, splitSize = 0;
        if (remaining >= SQZ_BLOCKSIZE_MAX && estSavings >= 3) {
            splitSize = wv_uni(splitBlock_byChunks(ip, (Fingerprint*)(ws + L.fp)));
            blockSize = splitSize;
            flags |= SQZ_FIND_F_SPLIT_ON;
        }
        if (ms.nextToUpdate < SQZ_WINDOW_START) ms.nextToUpdate = SQZ_WINDOW_START;
        u32 lastLL = blockSize, nseq = 0, lng = 0, nlits = 0;
        if (blockSize >= 1 + 1 + 3 + 1 + 1) {
            if (!bs.isFirstBlock && wv_uni(wv_is_rle(ip, blockSize))) flags |= SQZ_FIND_F_RLE;
            ss.seqs = find_area_seqs(area) + seqStart;
            lastLL = wv_uni(cb_find<true>(ms, ss, bs, nullptr, ip, blockSize));
            SQ_TRACE(9, 1);
            if (wv_uni(ms.fault)) {
                status = 2;
                break;
            }
            SQ_TRACE(9, 2);
            nseq = wv_uni(ss.nseq);
            nlits = wv_uni(ss.nlits);
            lng = wv_uni(ss.longLengthType | (ss.longLengthPos << 2));
            // assumption (1): emitted compressed -- the repcodes it leaves are confirmed -- unless it is a run of one byte
            // (irrelevant when it has no sequence)
            if (!(flags & SQZ_FIND_F_RLE)) bs.prevIdx ^= 1;
        }
        SQ_TRACE(9, 3);
        u32* const rec = area + SQZ_FIND_HDR_WORDS + SQZ_FIND_REC_WORDS * nBlocks;
        gst32(rec + 0, blockSize);
        gst32(rec + 1, nseq);
        gst32(rec + 2, lastLL);
        gst32(rec + 3, lng);
        gst32(rec + 4, seqStart);
        gst32(rec + 5, flags);
        gst32(rec + 6, splitSize);
        gst32(rec + 7, 0);
        estSavings += nseq ? (s64)blockSize - (s64)(nlits + lastLL) - 3 * (s64)nseq - 3 : -3;
        seqStart += nseq;
        nBlocks++;
        ip += blockSize;
        remaining -= blockSize;
        bs.isFirstBlock = 0;
        SQ_TRACE(9, 4);
        SQ_TRACE(10, remaining);
    }
    SQ_TRACE(9, 5);
    wv_mem_sync();
    SQ_TRACE(9, 6);
    gst32(area + 1, nBlocks);
    gst32(area + 0, status);
    wv_mem_sync();
    SQ_TRACE(9, 7);
}

// One block's literal position list (ss.stBits) and literal buffer from its sequence records: what storeSeq_deferred
// and gather_literals produce inside the parser in the one-kernel path.  blockPos: index of the block's first byte.
SQ_DEVFN void emit_prepare_block(SeqStore& ss, const u8* base, u32 blockPos) {
    u32* const posList = (u32*)ss.stBits;
    u32 pos = blockPos, lit = 0;
#if SQ_ON_DEVICE
    u32 const lane = wv_lane();
    for (u32 s0 = 0; s0 < ss.nseq; s0 += 64) {
        u32 const s = s0 + lane;
        u32 ll = 0, ml = 0;
        if (s < ss.nseq) {
            u32 const w = sq_gld32((const u32*)&ss.seqs[s] + 1);
            ll = w & 0xFFFF;
            ml = (w >> 16) + SQZ_MINMATCH;
            if (ss.longLengthPos == s) {
                if (ss.longLengthType == 1) ll += 0x10000;
                if (ss.longLengthType == 2) ml += 0x10000;
            }
        }
        u32 totL, totA;
        u32 const exL = wv_excl_scan(ll, &totL), exA = wv_excl_scan(ll + ml, &totA);
        if (s < ss.nseq) sq_gst64v(posList + 2 * (size_t)s, pos + exA, lit + exL);
        pos += totA;
        lit += totL;
    }
#else
    for (u32 s = 0; s < ss.nseq; s++) {
        u32 ll = ss.seqs[s].litLength, ml = (u32)ss.seqs[s].mlBase + SQZ_MINMATCH;
        if (ss.longLengthPos == s) {
            if (ss.longLengthType == 1) ll += 0x10000;
            if (ss.longLengthType == 2) ml += 0x10000;
        }
        posList[2 * (size_t)s] = pos;
        posList[2 * (size_t)s + 1] = lit;
        pos += ll + ml;
        lit += ll;
    }
#endif
    ss.nlits = lit;
    gather_literals(ss, base);
}

// The emitting half: the frame of a chunk whose blocks the finder left in `area`.  Returns the frame length; 0 when an
// assumption of the finder did not hold (or on any error): the chunk then takes the one-kernel path.
#if SQ_ON_DEVICE
#define EMIT_FAIL(code) do { SQ_TRACE(40, code); return 0; } while (0)
#else
#define EMIT_FAIL(code) do { if (getenv("SQ_SIM_STATS")) fprintf(stderr, "[sim] emit refuses: reason %d at block %u of %u (savings %lld)\n", code, b_, nBlocks_, (long long)sav_); return 0; } while (0)
#endif
SQ_DEVFN u32 emit_frame_wave(const u8* src, u32 n, u8* dst, u32 cap, u8* ws, const WsLayout& L, const u32* area) {
    SeqStore ss;
    BlockState bs;
    CParams const cp = get_cparams(n);
    u8* op = dst;
    u8* const oend = dst + cap;
#if !SQ_ON_DEVICE
    u32 b_ = ~0u, nBlocks_ = 0; // (for EMIT_FAIL's message)
    s64 sav_ = 0;
#endif
    WV_SERIAL { g_lds.trace = (u32*)(ws + L.dummy) + 8; }
    wv_mem_sync_inv();
    if (cap < 18 || gld32(area) != 1) EMIT_FAIL(1);
    u32 const nBlocks = gld32(area + 1);
    u32 const hdr = fast_frame_header(dst, n);
    op += hdr;
    ss.lits = ws + L.lits;
    ss.llCode = ws + L.llCode;
    ss.ofCode = ws + L.ofCode;
    ss.mlCode = ws + L.mlCode;
    ss.stBits = (u16*)(ws + L.stBits);
    ss.recordsOnly = 0;
    WV_SERIAL {
        entropy_reset(&g_lds.ent[0]);
        entropy_reset(&g_lds.ent[1]);
    }
    bs.prevIdx = 0;
    bs.isFirstBlock = 1;
    wv_mem_sync();
    const u8* ip = src;
    u32 remaining = n;
    s64 savings = 0;
    for (u32 b = 0; b < nBlocks; b++) {
        const u32* const rec = area + SQZ_FIND_HDR_WORDS + SQZ_FIND_REC_WORDS * b;
        u32 const blockSize = gld32(rec), nseq = gld32(rec + 1), lastLL = gld32(rec + 2), lng = gld32(rec + 3),
                  seqStart = gld32(rec + 4), flags = gld32(rec + 5);
#if !SQ_ON_DEVICE
        b_ = b, nBlocks_ = nBlocks, sav_ = savings;
#endif
        SQ_TRACE(0, 200 + b);
        SQ_TRACE(2, (u32)(ip - src));
        if (blockSize == 0 || blockSize > remaining) EMIT_FAIL(2);
        // assumption (2): was the block sized the way ZSTD_optimalBlockSize sizes it with the true savings?
        if (remaining >= SQZ_BLOCKSIZE_MAX) {
            bool const splitOn = savings >= 3;
            if (splitOn != ((flags & SQZ_FIND_F_SPLIT_ON) != 0)) {
                // the finder took the other branch: fine when both give the same size
                u32 const want = splitOn ? wv_uni(splitBlock_byChunks(ip, (Fingerprint*)(ws + L.fp))) : SQZ_BLOCKSIZE_MAX;
                if (want != blockSize) EMIT_FAIL(3);
            }
        } else if (blockSize != remaining)
            EMIT_FAIL(4);
        u32 const lastBlock = (blockSize == remaining);
        if ((u32)(oend - op) < 3 + 2 + 1) EMIT_FAIL(5);
        u32 cSize = 0;
        if (blockSize >= 1 + 1 + 3 + 1 + 1) {
            ss.seqs = find_area_seqs(const_cast<u32*>(area)) + seqStart;
            ss.nseq = nseq;
            ss.nlits = 0;
            ss.longLengthType = lng & 3;
            ss.longLengthPos = lng >> 2;
            if (nseq) emit_prepare_block(ss, src - SQZ_WINDOW_START, (u32)(ip - src) + SQZ_WINDOW_START);
            cSize = wv_uni(cb_emit(cp.strategy, ss, bs, op + 3, (u32)(oend - op) - 3, ip, blockSize, lastLL));
            if (is_err(cSize)) EMIT_FAIL(6);
        }
        // assumption (1): a block with sequences that is NOT emitted compressed leaves other repcodes to the next block
        if (nseq && !lastBlock && (cSize > 1) != !(flags & SQZ_FIND_F_RLE)) EMIT_FAIL(7);
        if (cSize == 0) { // ZSTD_noCompressBlock
            u32 const h = lastBlock + (0u << 1) + (blockSize << 3);
            if (blockSize + 3 > (u32)(oend - op)) EMIT_FAIL(8);
            WV_SERIAL {
                op[0] = (u8)h;
                op[1] = (u8)(h >> 8);
                op[2] = (u8)(h >> 16);
            }
            wv_memcpy(op + 3, ip, blockSize);
            cSize = 3 + blockSize;
        } else {
            u32 const h = (cSize == 1) ? lastBlock + (1u << 1) + (blockSize << 3) : lastBlock + (2u << 1) + (cSize << 3);
            WV_SERIAL {
                op[0] = (u8)h;
                op[1] = (u8)(h >> 8);
                op[2] = (u8)(h >> 16);
            }
            cSize += 3;
        }
        savings += (s64)blockSize - (s64)cSize;
        ip += blockSize;
        remaining -= blockSize;
        op += cSize;
        bs.isFirstBlock = 0;
        SQ_TRACE(7, 300 + b);
    }
    if (remaining) EMIT_FAIL(9);
    SQ_TRACE(7, 400);
    wv_fence();
    return (u32)(op - dst);
}

#if SQ_ON_DEVICE
// One chunk -> one frame in its output slot (kept out of line: keeps the queue loop simple).
__device__ __noinline__ void compress_one(const u8* data, const u64* off, const u32* len, u32 idx, u8* slots,
                                          const u64* slot_off, u32* frame_len, u8* ws, u32* err, TableGen& tg, const u32* schedTab,
                                          const u32* certHint) {
    u32 const L = len[idx];
    WsLayout const lay = ws_layout(L); // offsets for this length are <= the pool's
    u32 const cap = (u32)(slot_off[idx + 1] - slot_off[idx]);
    u32 const r = compress_frame_wave(data + off[idx], L, slots + slot_off[idx], cap, ws, lay, tg, schedTab, certHint ? certHint + 2 * (size_t)idx : nullptr);
    if (wv_lane() == 0) {
        frame_len[idx] = r;
        if (r == 0) atomicOr(err, 1u);
    }
}

// Persistent compressor: grid = number of workspaces, block = one wave.  Each wave pulls chunk
// indices from *queue until exhausted; every wave reaches the exit condition (idx >= n).
__global__ void __launch_bounds__(64, 2) // 2 waves per SIMD: <= 256 registers, 8 waves per CU (LDS: 8 x 19 KiB)
k_compress_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* slots,
                 const u64* slot_off, u32* frame_len, u8* ws_pool, u64 ws_each, u32* queue, u32* err,
                 const u32* schedTab, const u32* certHint, u32 persist, const u32* work, const u64* find_off) {
    u8* const ws = ws_pool + (u64)blockIdx.x * ws_each;
    u32* const dbg = (u32*)(ws + ws_each - 256) + 8; // wave-level progress words for the watchdog
    u32 iter = 0;
    // State of this wave's table areas.  `persist`: the host vouches that this workspace is exactly as
    // the previous launch of this kernel left it, so the state saved at its end is still true and the
    // tables need no clearing (generation tags go on counting); otherwise they hold unknown words.
    u32* const saved = (u32*)(ws + ws_each - 704);
    TableGen tg = {0, 0, 0};
    if (persist) {
        tg.gen = gld32(saved);
        tg.windowLog = gld32(saved + 1);
        tg.tailTabClean = gld32(saved + 2);
    }
#ifdef SQ_PROF
    if (wv_lane() == 0) atomicAdd(queue + 44, 1u); // b_misc[44]
    unsigned long long const t_kernel0 = __builtin_amdgcn_s_memrealtime();
    if (wv_lane() == 0)
        for (int k = 0; k < 32; k++) g_lds.prof[k] = 0;
    wv_fence();
#endif
    // When stage 1 ran it left two lists of the chunks it did not decide: first the "heavy" ones (certified
    // in neither block: the whole exact path, several times the cost of the others), then the rest --
    // longest jobs first, one chunk per claim, nothing to skip.  Without stage 1: every index in order.
    // queue[0] / queue[1]: the two passes' counters.
    for (u32 pass = work ? 0u : 1u; pass < 2; pass++) {
        u32 const count = work ? gld32(work + pass) : n;
        const u32* const list = work ? work + 2 + (pass ? n : 0u) : nullptr;
        for (;;) {
            u32 q = 0;
            if (wv_lane() == 0) q = atomicAdd(queue + pass, 1u);
            q = wv_uni(q);
            if (wv_lane() == 0) {
                dbg[12] = 1000 + iter;
                dbg[13] = q;
            }
            iter++;
            if (q >= count) break;
            u32 const idx = list ? gld32(list + q) : q;
            // (the lists hold what stage 1 left undecided: of the selected chunks in the split form, of ALL chunks in the
            // fused form, where stage 1 runs before the dedup decision -- duplicates are skipped here)
            bool take = (select == nullptr) || (select[idx] != 0);
            if (find_off && take && find_off[idx] != ~0ull && frame_len[idx] != 0) take = false; // k_emit_batch made its frame
            PROF_T0(t_all);
#ifdef SQ_PROF
            if (wv_lane() == 0)
                for (int k = 0; k < 29; k++) g_lds.prof0[k] = g_lds.prof[k];
#endif
            if (take) compress_one(data, off, len, idx, slots, slot_off, frame_len, ws, err, tg, schedTab, certHint);
            PROF_ADD(0, t_all);
#ifdef SQ_PROF
            {   // the wave's longest chunk (ticks, index, list it came from) and when the wave finished it
                unsigned long long const e_ = __builtin_amdgcn_s_memrealtime();
                if (wv_lane() == 0) {
                    if (e_ - t_all > g_lds.prof[29]) {
                        g_lds.prof[29] = e_ - t_all;
                        g_lds.prof[30] = ((unsigned long long)pass << 32) | idx;
                        for (int k = 0; k < 29; k++) g_lds.profLong[k] = g_lds.prof[k] - g_lds.prof0[k];
                    }
                    g_lds.prof[31] = e_ - t_kernel0;
                }
            }
#endif
        }
    }
    if (wv_lane() == 0) {
#ifdef SQ_PROF
        atomicAdd(queue + 45, 1u); // b_misc[45]
#endif
        dbg[11] = 777;
        saved[0] = tg.gen;
        saved[1] = tg.windowLog;
        saved[2] = tg.tailTabClean;
    }
#ifdef SQ_PROF
    wv_fence();
    if (wv_lane() == 0) {
        unsigned long long* o = (unsigned long long*)(ws + ws_each - 1024);
        for (int k = 0; k < 32; k++) o[k] = g_lds.prof[k];
        o[32] = iter;
        for (int k = 0; k < 29; k++) o[42 + k] = g_lds.profLong[k];
    }
#endif
}

// (out of line, like compress_one: the queue loops below stay small)
__device__ __noinline__ void find_one(const u8* data, const u64* off, const u32* len, u32 idx, u8* ws, u8* area) {
    u32 const L = len[idx];
    WsLayout const lay = ws_layout(L);
    find_frame_wave(data + off[idx], L, ws, lay, (u32*)area);
}
__device__ __noinline__ void emit_one(const u8* data, const u64* off, const u32* len, u32 idx, u8* slots, const u64* slot_off,
                                      u32* frame_len, u8* ws, const u8* area, u32* stats) {
    u32 const L = len[idx];
    WsLayout const lay = ws_layout(L);
    u64 const so = slot_off[idx];
    u32 const cap = (u32)(slot_off[idx + 1] - so);
    u32 const r = emit_frame_wave(data + off[idx], L, slots + so, cap, ws, lay, (const u32*)area);
    if (wv_lane() == 0) {
        frame_len[idx] = r; // 0: the chunk stays on k_compress_batch's plate
        atomicAdd(stats + (r ? 0 : 1), 1u);
    }
}

// The two kernels of the lazy2 size class (see "the lazy2 size class cut in two kernels" above).  Both walk the same
// lists as k_compress_batch (stage 1's work lists, or every index) with counters of their own and touch only chunks
// that have a find area (find_off[idx] != ~0: selected by the host for this path).
__global__ void __launch_bounds__(64, 3) // 12 waves per CU: <= 168 registers, SQZ_LDS_FIND_BYTES of LDS each
k_find_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* ws_pool, u64 ws_each,
             u32* queue, const u32* work, const u64* find_off, u8* find_pool) {
    u8* const ws = ws_pool + (u64)blockIdx.x * ws_each;
#ifdef SQ_PROF
    if (wv_lane() == 0) atomicAdd(queue + 36, 1u); // b_misc[40]: waves that entered
#endif
    for (u32 pass = work ? 0u : 1u; pass < 2; pass++) {
        u32 const count = work ? gld32(work + pass) : n;
        const u32* const list = work ? work + 2 + (pass ? n : 0u) : nullptr;
        for (;;) {
            u32 q = 0;
            if (wv_lane() == 0) q = atomicAdd(queue + pass, 1u);
            q = wv_uni(q);
            if (q >= count) break;
            u32 const idx = list ? gld32(list + q) : q;
            u64 const fo = find_off[idx];
            if (fo == ~0ull || (select != nullptr && select[idx] == 0)) continue;
            find_one(data, off, len, idx, ws, find_pool + fo);
        }
    }
#ifdef SQ_PROF
    if (wv_lane() == 0) atomicAdd(queue + 37, 1u); // b_misc[41]: waves that left
#endif
}

__global__ void __launch_bounds__(64, 2)
k_emit_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* slots, const u64* slot_off,
             u32* frame_len, u8* ws_pool, u64 ws_each, u32* queue, const u32* work, const u64* find_off,
             const u8* find_pool, u32* stats) {
    u8* const ws = ws_pool + (u64)blockIdx.x * ws_each;
#ifdef SQ_PROF
    if (wv_lane() == 0) atomicAdd(queue + 36, 1u); // b_misc[42]
#endif
    for (u32 pass = work ? 0u : 1u; pass < 2; pass++) {
        u32 const count = work ? gld32(work + pass) : n;
        const u32* const list = work ? work + 2 + (pass ? n : 0u) : nullptr;
        for (;;) {
            u32 q = 0;
            if (wv_lane() == 0) q = atomicAdd(queue + pass, 1u);
            q = wv_uni(q);
            if (q >= count) break;
            u32 const idx = list ? gld32(list + q) : q;
            u64 const fo = find_off[idx];
            if (fo == ~0ull || (select != nullptr && select[idx] == 0)) continue;
            emit_one(data, off, len, idx, slots, slot_off, frame_len, ws, find_pool + fo, stats);
        }
    }
#ifdef SQ_PROF
    if (wv_lane() == 0) atomicAdd(queue + 37, 1u); // b_misc[43]
#endif
}
#endif

} // namespace sqz
