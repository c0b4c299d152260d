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

#define SQZ_STAGE2_WAVES_PER_CU 11u // resident compressor waves per CU: 14.4 KB of LDS and <= 168 registers each (k_compress_batch)

struct WsLayout {
    size_t hashTable, table2, hashTable3, seqs, lits, llCode, ofCode, mlCode, stBits, optState, fp, sched, liveList,
        skipList, blockTab, certTab, ent, dummy, total;
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
    L.ent = o; // the two sets of entropy tables (previous / next block's)
    o += ws_align(2 * sizeof(Entropy));
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
SQ_DEVFN u32 splitBlock_byChunks(const u8* p, Fingerprint* fps) {
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
struct BlockState { // entropy tables are g_lds.entp[idx]
    u32 rep[2][SQZ_REP_NUM];
    u32 prevIdx;
    u32 isFirstBlock;
};

// ZSTD_buildSeqStore + ZSTD_compressBlock_internal: returns body size; 0 => raw; 1 => RLE
SQ_DEVFN u32 compressBlock_internal(MatchState& ms, SeqStore& ss, BlockState& bs, void* optState, u8* dst,
                                    u32 dstCap, const u8* src, u32 srcSize) {
    u32 const p = bs.prevIdx, nx = p ^ 1;
    u32 cSize;
    if (srcSize < 1 + 1 + 3 + 1 + 1) { // ZSTDbss_noCompress
        cSize = 0;
    } else {
        ss.nseq = 0;
        ss.nlits = 0;
        ss.longLengthType = 0;
        ss.longLengthPos = 0;
        {
            u32 const curr = (u32)(src - ms.base);
            ms.blockEndPos = curr + srcSize;
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
            if (curr > ms.nextToUpdate + 384) {
                u32 const d = curr - ms.nextToUpdate - 384;
                u32 const n2u = curr - (192 < d ? 192 : d);
                filter_add_skip(ms, ms.nextToUpdate, n2u);
                ms.nextToUpdate = n2u;
            }
        }
        ms.nxEnt = nx;
        for (int i = 0; i < SQZ_REP_NUM; ++i) bs.rep[nx][i] = bs.rep[p][i];
        u32 lastLLSize;
        SQ_TRACE(0, 6);
        if (ms.cp.strategy == STRAT_LAZY2) {
            lastLLSize = compressBlock_lazy_rowbatch(ms, ss, bs.rep[nx], src, srcSize);
        } else if (ms.cp.strategy == STRAT_BTLAZY2) {
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
                u32 const last = (u32)opt::compressBlock_btopt(&om, &os, rr, src, srcSize, (void*)&g_lds.u);
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
        SQ_TRACE(0, 7);
        SQ_TRACE(4, ss.nseq);
        if (ss.nseq == 0 && ss.nlits == 0 && lastLLSize == srcSize) {
            // no sequence at all: the literals ARE the block; the entropy stage reads them in place
            SeqStore whole = ss;
            whole.lits = const_cast<u8*>(src);
            whole.nlits = srcSize;
            ss.nlits = srcSize;
            wv_mem_sync_inv();
            cSize = entropy_compress_seqstore(whole, &g_lds.entp[p], &g_lds.entp[nx], ms.cp.strategy, dst, dstCap, srcSize);
        } else {
            PROF_T0(t_ll);
            if (lastLLSize) wv_memcpy(ss.lits + ss.nlits, src + srcSize - lastLLSize, lastLLSize);
            PROF_ADD(7, t_ll);
            ss.nlits += lastLLSize;
            wv_mem_sync_inv(); // sequence records / literals written by other lanes have landed
            cSize = entropy_compress_seqstore(ss, &g_lds.entp[p], &g_lds.entp[nx], ms.cp.strategy, dst, dstCap, srcSize);
        }
        SQ_TRACE(0, 9);
        SQ_TRACE(5, cSize);
        if (is_err(cSize)) return cSize;
        if (!bs.isFirstBlock && cSize < 25 && wv_is_rle(src, srcSize)) {
            cSize = 1;
            gst8(dst, srd8(src));
        }
    }
    if (cSize > 1) bs.prevIdx ^= 1; // ZSTD_blockState_confirmRepcodesAndEntropyTables
    // offcode_repeatMode valid -> check: REP_VALID never occurs without a dictionary
    return cSize;
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
// pairInfo (chunks above 256 KiB): stage 1's word for the chunk -- pairs of blocks finished (bits 0..7), how many failed
// (8..15), which (16..23) -- or 0.
SQ_DEVFN u32 compress_frame_wave(const u8* src, u32 n, u8* dst, u32 cap, u8* ws, const WsLayout& L, TableGen& tg,
                                 const u32* schedTab, const u32* certHint, u32 pairInfo = 0) {
    MatchState ms;
    SeqStore ss;
    BlockState bs;
    CParams const cp = get_cparams(n);
    u8* op = dst;
    u8* const oend = dst + cap;
    WV_SERIAL {
        g_lds.trace = (u32*)(ws + L.dummy) + 8;
        g_lds.entp = (Entropy*)(ws + L.ent);
    }
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
    ms.certPairs = 0;
    if (cp.strategy == STRAT_LAZY2 && schedTab) {
        u32 const nBlocks = (n + SQZ_BLOCKSIZE_MAX - 1) / SQZ_BLOCKSIZE_MAX, nPairs = (nBlocks + 1) / 2;
        if (nPairs <= 8 && (pairInfo & 0xFFu) == nPairs) ms.certPairs = ~(pairInfo >> 16) & ((1u << nPairs) - 1u);
    }
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
    WV_SERIAL {
        entropy_reset(&g_lds.entp[0]);
        entropy_reset(&g_lds.entp[1]);
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
            ms.certEnd = gld32(certHint) & SQZ_CERT_HINT_POS_MASK;
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
    // chunk bytes that stage 1's scan has already written as raw-block payload where a frame of full raw blocks has them
    // (block b at dst + hdr + 3 (b + 1) + 128 KiB b): a raw block found there is not copied again
    u32 const rawHave = (certHint && cp.strategy == STRAT_BTLAZY2) ? (gld32(certHint) >> SQZ_CERT_HINT_POS_BITS) << SQZ_CERT_HINT_RAW_LOG : 0u;
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
        // Room for this block's attempt.  While stage 1's raw payload of LATER blocks sits in the slot (rawHave reaches beyond
        // this block), an attempt must not write over it: the room ends where the next block's header goes.  That changes no
        // result -- an attempt that would need more than blockSize bytes is emitted raw whatever it costs (compressLiterals /
        // entropy_compress_seqstore fall back to "raw" when the room runs out, as libzstd's cSize >= maxCSize test does) -- but it
        // makes the in-place shortcut below safe by construction instead of by the arithmetic of sequence costs.
        u32 room = (u32)(oend - op) - 3;
        if (rawHave > (u32)(ip - src) + blockSize && room > blockSize) room = blockSize;
        u32 cSize = compressBlock_internal(ms, ss, bs, ws + L.optState, op + 3, room, ip, blockSize);
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
            {
                u32 const o = (u32)(ip - src);
                // in place already?  (this block and all before it full raw blocks, stage 1 got that far, and the entropy
                // stage's attempt wrote nothing over it: g_lds.ret[7], set by compressLiterals / entropy_compress_seqstore)
                bool const inPlace = blockSize >= 7 && o + blockSize <= rawHave && op == dst + hdr + 3 * (o >> 17) + o &&
                                     (o & (SQZ_BLOCKSIZE_MAX - 1)) == 0 && g_lds.ret[7] == 0;
                if (!inPlace) wv_memcpy(op + 3, ip, blockSize);
            }
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

#if SQ_ON_DEVICE
// One chunk -> one frame in its output slot.  Kept OUT OF LINE on purpose: it keeps the persistent queue loop small, and an
// experiment of round 3 that inlined a per-chunk body of this size (with calls inside) into such a loop hung on the device --
// the wave never returned to the queue's atomicAdd.
__device__ __noinline__ void compress_one(const u8* data, const u64* off, const u32* len, u32 idx, u8* slots,
                                          const u64* slot_off, u32* frame_len, u8* ws, u32* err, TableGen& tg, const u32* schedTab,
                                          const u32* certHint, const u32* pair_state) {
    u32 const L = len[idx];
    WsLayout const lay = ws_layout(L); // offsets for this length are <= the pool's
    u32 const cap = (u32)(slot_off[idx + 1] - slot_off[idx]);
    u32 const r = compress_frame_wave(data + off[idx], L, slots + slot_off[idx], cap, ws, lay, tg, schedTab, certHint ? certHint + 2 * (size_t)idx : nullptr,
                                      pair_state ? pair_state[idx] : 0u);
    if (wv_lane() == 0) {
        frame_len[idx] = r;
        if (r == 0) atomicOr(err, 1u);
    }
}

// Persistent compressor: grid = number of workspaces, block = one wave.  Each wave pulls chunk
// indices from *queue until exhausted; every wave reaches the exit condition (idx >= n).
// (the kernel's symbol, for the copy of it that sq_api.hip loads from squish_s2_w3.co; tests/test_abi_cpu.py checks it against the build)
#define SQZ_K_COMPRESS_BATCH_SYMBOL "_ZN3sqz16k_compress_batchEPKhPKmPKjS1_jPhS3_PjS6_mS7_S7_S5_S5_jS5_S5_"
#ifndef SQZ_STAGE2_WAVES_PER_SIMD
#define SQZ_STAGE2_WAVES_PER_SIMD 2 // the library's copy: 256 registers, 8 waves per CU; squish_s2_w3.co is built with 3 (168, 11 per CU)
#endif
// Two builds of this kernel exist (sq_ctx::f_s2w3 in sq_api.hip says which batch takes which).  Measured (round 4,
// profiles/round4f_tree_step_loads.txt, round4f_stage2_register_bound_ab.txt): at 168 registers and 8 waves per CU the code runs as fast as
// at 256, and 11 per CU are +13 % on 64 KiB text, +9 % on 2 MiB mixed chunks -- but beside stage 1 on the headline the 168-register
// build costs the step 5 %
__global__ void __launch_bounds__(64, SQZ_STAGE2_WAVES_PER_SIMD)
k_compress_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* slots,
                 const u64* slot_off, u32* frame_len, u8* ws_pool, u64 ws_each, u32* queue, u32* err,
                 const u32* schedTab, const u32* certHint, u32 persist, const u32* work, const u32* pair_state) {
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
            bool const take = (select == nullptr) || (select[idx] != 0);
            PROF_T0(t_all);
#ifdef SQ_PROF
            if (wv_lane() == 0)
                for (int k = 0; k < 29; k++) g_lds.prof0[k] = g_lds.prof[k];
#endif
            if (take) compress_one(data, off, len, idx, slots, slot_off, frame_len, ws, err, tg, schedTab, certHint, pair_state);
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
#endif

} // namespace sqz
