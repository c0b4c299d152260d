// sq_zstd_rowbatch.h -- lazy2 parser (chunks > 256 KiB: row-hash finder) over SPECULATIVE SEARCH BATCHES.
// Included at the end of sq_zstd_match.h, inside namespace sqz.
//
// libzstd's lazy parser asks the row-hash finder one position at a time (ZSTD_RowFindBestMatch, reached from
// /root/reference/src/util/chunk.rs:89-90), and on a GPU every such search is a chain of dependent memory round
// trips (source bytes -> tag row -> entries -> candidate bytes) during which a wave does nothing else: round 1
// measured ~1 us per search and one wave 50x slower than one CPU core.  Here ONE chain of round trips prepares the
// searches of 64 consecutive positions, one per lane, before the parser needs them:
//
//   * Why that is exact.  While the parser is not in "lazy skipping" mode, ZSTD_row_update inserts EVERY position
//     between two searches (distances inside a batch are < 384, the skip rule of ZSTD_row_update_internal never
//     fires), so the table a search at q = B0 + i sees is: the table as it stands when the batch is made (everything
//     below B0 inserted) plus the positions B0 .. q-1.  Lane i therefore has as candidates (a) the earlier positions
//     of the batch that share its row AND tag (found by in-wave grouping of the hashes), most recent first, then
//     (b) the row's entries in recency order, the r oldest of which -- r = earlier batch positions in the same row,
//     whatever their tag -- have been overwritten by then and are left out.  First longest candidate wins, a slot
//     that was never written ends the scan: ZSTD_RowFindBestMatch's rules.  The physical insertion of the batch's
//     positions is deferred until the parser leaves the batch (or needs the exact finder), so the rows are read in
//     the state every lane assumes.
//   * rb_prepare (one chain of three trips per batch): source bytes, hashes, grouping, the 64-byte tag row -> every
//     lane's candidate masks; and the repcode-1 match length at every lane for the parser's current offset_1
//     (refreshed in one trip when a sequence changes the offset; the stale values then ARE the new offset_2's, which
//     is what the parser checks right after a match).
//   * rb_eval (two trips): the candidates are evaluated for a WINDOW of lanes starting at the position the parser
//     asks for -- as many lanes as fit one round of 256 (owner, candidate) pairs, at least the three positions the
//     lazy parser is about to look at, all 64 where candidates are few -- balanced over the wave through LDS, with
//     no lane-serial loop: on repetitive data a search has dozens of TRUE candidates and the kernel runs at the
//     memory system's rate of small random requests, so pairs of positions the parser skips are not to be spent.
//     Per lane it leaves the best match and the backward-extension count of that candidate.
//   * The parser runs on these registers (v_readlane) and touches memory only to store sequences; runs of positions
//     known to give nothing are skipped in one step.
//   * A lane whose answer cannot be had this way (a candidate that matches 32 bytes or more, positions within 32
//     bytes of the chunk end) is flagged; the parser then inserts up to that position and calls the exact finder
//     (row_find).  Lazy-skipping stretches (no match for 2 KiB) use the exact finder as before.
//   * Literals are not copied sequence by sequence: storeSeq records where they are, and gather_literals copies
//     them once per block, one sequence per lane.
//
// The SQ_SIM build runs the same parser with the batch evaluated lane after lane by plain scalar code
// (tests/test_sim_cpu.py diffs it against the oracle); the device code below is checked by the GPU parity tests.

#define SQZ_RB_N 64u
#define SQZ_RB_EXACT 1u
#define SQZ_RB_DONE 2u // the lane's candidates have been evaluated (device; the simulator evaluates every lane at once)
#define SQZ_RB_ROUND 256u // candidate pairs evaluated per round
#ifndef SQZ_RB_SPEC
#define SQZ_RB_SPEC 256u  // ... of which speculative ones (beyond the first three lanes) up to this total (48 / 96 / 160 measured: fewer pairs, more rounds, no gain)
#endif
#define SQZ_RB_RLCAP 16u
#define SQZ_RB_BECAP 8u

struct RowBatch {
#if SQ_ON_DEVICE
    u32 ml, off, rl, be, flags; // per lane: results
    u64 imask, tmask;           // per lane: candidates still to be evaluated (in-batch lanes; row entries by recency rank)
    u32 head, rowId, cnt;       // per lane
    sq_v4u own0, own1;          // per lane: the 32 bytes at the lane's position
    u64 ownPrev;                // per lane: the 8 bytes in front of it
#else
    u32 ml[64], off[64], rl[64], be[64], flags[64];
    u64 imask[64], tmask[64];
    u32 head[64], rowId[64], cnt[64];
#endif
    u32 B0, n, valid, off1; // wave-uniform
    u32 ldsOk;              // wave-uniform: the lanes' source bytes are in LDS (g_lds.grp; clobbered by an insertion sweep)
};
#if SQ_ON_DEVICE
#define RB_GET(bt, f, i) wv_readlane((bt).f, (i))
#else
#define RB_GET(bt, f, i) ((bt).f[(i)])
#endif

// bit j: the search at lane j is known to find nothing (evaluated, not delegated, no match) / its repcode length is >= 4
#if SQ_ON_DEVICE
SQ_DEV u64 rb_nomatch_mask(const RowBatch& bt) {
    return wv_ballot(wv_lane() < bt.n && (bt.flags & SQZ_RB_DONE) && !(bt.flags & SQZ_RB_EXACT) && bt.ml <= 3);
}
SQ_DEV u64 rb_rep_mask(const RowBatch& bt) { return wv_ballot(wv_lane() < bt.n && bt.rl >= 4); }
#else
SQ_DEV u64 rb_nomatch_mask(const RowBatch& bt) {
    u64 m = 0;
    for (u32 j = 0; j < bt.n; j++)
        if ((bt.flags[j] & SQZ_RB_DONE) && !(bt.flags[j] & SQZ_RB_EXACT) && bt.ml[j] <= 3) m |= 1ull << j;
    return m;
}
SQ_DEV u64 rb_rep_mask(const RowBatch& bt) {
    u64 m = 0;
    for (u32 j = 0; j < bt.n; j++)
        if (bt.rl[j] >= 4) m |= 1ull << j;
    return m;
}
#endif

// recency order of a row's 63 usable slots (slot 0 holds the head byte): rank 0 = the head's slot
SQ_DEV u32 rb_slot_of(u32 head, u32 rank) {
    u32 const j0 = (64u - head) & 63u; // where slot 0 sits in the scan order (head + j) & 63
    u32 const j = rank < j0 ? rank : rank + 1;
    return (head + j) & 63u;
}

#if SQ_ON_DEVICE
SQ_DEV u32 rb_prefix16(sq_v4u a, sq_v4u b) { // common prefix of two 16-byte strings
    u32 const x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    u32 r = 16;
    if (x3) r = 12 + ((u32)__builtin_ctz(x3) >> 3);
    if (x2) r = 8 + ((u32)__builtin_ctz(x2) >> 3);
    if (x1) r = 4 + ((u32)__builtin_ctz(x1) >> 3);
    if (x0) r = (u32)__builtin_ctz(x0) >> 3;
    return r;
}

// repcode-1 lengths of every lane for offset off1 (0 = none): one round trip
SQ_DEV void rb_refresh(const MatchState& ms, RowBatch& bt, u32 off1, u32 iendPos) {
    u32 const lane = wv_lane();
    u32 const q = bt.B0 + (lane < bt.n ? lane : 0u);
    bt.off1 = off1;
    u32 rl = 0;
    bool const ok = off1 && q >= off1 + SQZ_WINDOW_START; // (lanes the parser has left behind may lie before the offset's reach)
    if (ok && q + 16 <= ms.chunkEndPos) {
        sq_v4u const a = sq_grd128(ms.base + q), b = sq_grd128(ms.base + q - off1);
        rl = rb_prefix16(a, b);
    } else if (ok) { // the last bytes of the chunk: byte by byte
        u32 const lim = iendPos - q;
        while (rl < lim && rl < SQZ_RB_RLCAP && sq_grd8(ms.base + q + rl) == sq_grd8(ms.base + q + rl - off1)) rl++;
    }
    u32 const cap = iendPos - q;
    bt.rl = rl < cap ? rl : cap;
}

// The batch, part 1 (rb_prepare): for positions B0 .. B0+n-1 the source bytes, hashes, in-batch grouping, tag rows
// -> per lane the candidate masks, and the repcode lengths.  Precondition: every position below B0 that libzstd
// would have inserted is in the tables, none at or above B0.  Part 2 (rb_eval) evaluates candidates -- not of all
// 64 positions at once: on repetitive data a search has dozens of true candidates (a frequent 5-gram fills its row)
// and the parser skips most positions inside matches, so candidates are evaluated for a WINDOW of lanes starting
// at the position the parser asks for, as many lanes as fit one round of 256 (owner, candidate) pairs: at least
// four positions (the lazy parser looks at ip, ip+1, ip+2), all 64 where candidates are few.
SQ_DEVFN void rb_prepare(MatchState& ms, RowBatch& bt, u32 B0, u32 n, u32 off1, u32 iendPos) {
    PROF_T0(t_rb);
    u32 const lane = wv_lane();
    const u8* const base = ms.base;
    bool const valid = lane < n;
    u32 const q = B0 + (valid ? lane : 0u);
    bt.B0 = B0;
    bt.n = n;
    bt.valid = 1;
    u32 flags = 0;
    // positions whose 32-byte windows would reach beyond the chunk go to the exact finder
    bool const nearEnd = q + 32 > ms.chunkEndPos;
    if (nearEnd) flags = SQZ_RB_EXACT;
    u32 const qs = nearEnd ? ms.chunkEndPos - 32 : q; // a safe address for the unconditional loads (chunks here are > 256 KiB)
    sq_v4u const own0 = sq_grd128(base + qs), own1 = sq_grd128(base + qs + 16);
    bt.own0 = own0;
    bt.own1 = own1;
    bt.ownPrev = (q >= 8 + SQZ_WINDOW_START) ? sq_grd64(base + q - 8) : 0ull;
    // ---- repcode lengths for the parser's current offset
    bt.off1 = off1;
    {
        u32 rl = 0;
        bool const ok = off1 && q >= off1 + SQZ_WINDOW_START;
        if (ok && !nearEnd) rl = rb_prefix16(own0, sq_grd128(base + q - off1));
        else if (ok) {
            u32 const lim = iendPos - q;
            while (rl < lim && rl < SQZ_RB_RLCAP && sq_grd8(base + q + rl) == sq_grd8(base + q + rl - off1)) rl++;
        }
        u32 const cap = iendPos - q;
        bt.rl = rl < cap ? rl : cap;
    }
    // ---- hashes, rows, in-batch grouping
    u32 const hash = hash5((u64)own0.x | ((u64)own0.y << 32), ms.rowHashLog + 8);
    u32 const rowId = hash >> 8, tag = hash & 0xFF;
    u32 r = 0;       // earlier positions of the batch in the same row
    u64 imask = 0;   // earlier lanes with the same row AND tag
    {
        SQ_L u8* const ls = (SQ_L u8*)g_lds.grp;
        u32 const slot = (rowId * 2654435761u) >> 21;
        if (valid) ls[slot] = (u8)lane;
        wv_fence();
        bool const loser = valid && (ls[slot] != (u8)lane);
        u64 todo = wv_ballot(loser);
        u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        while (todo) { // one iteration per row that occurs more than once (or shares an LDS slot)
            u32 const l = (u32)__builtin_ctzll(todo);
            u32 const k = wv_readlane(rowId, l);
            u64 const G = wv_ballot(valid && rowId == k);
            bool const mine = valid && rowId == k;
            if (mine) r = (u32)__popcll(G & below);
            u64 rest = G;
            while (rest) { // ... and per distinct tag inside it
                u32 const l2 = (u32)__builtin_ctzll(rest);
                u32 const t2 = wv_readlane(tag, l2);
                u64 const m = wv_ballot(mine && tag == t2);
                if (mine && tag == t2) imask = m & below;
                rest &= ~m;
            }
            todo &= ~G;
        }
        wv_fence();
    }
    // ---- the tag row (64 bytes per lane), as of "everything below B0 inserted"
    wv_mem_sync_inv(); // earlier table stores have landed; no stale L1 line
    const u8* const tagRow = ms.tagTable + ((size_t)rowId << 6);
    u64 M = 0;
    u32 head;
    {
        sq_v4u const t0 = sq_grd128(tagRow), t1 = sq_grd128(tagRow + 16), t2 = sq_grd128(tagRow + 32), t3 = sq_grd128(tagRow + 48);
        u32 const d[16] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w, t3.x, t3.y, t3.z, t3.w};
        u32 const pat = tag * 0x01010101u;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 const x = d[k] ^ pat;
            u32 y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
            y = ~(y | x | 0x7F7F7F7Fu);                     // 0x80 in every byte of x that is zero
            u32 const nib = ((y >> 7) * 0x01020408u) >> 24; // bit b = byte b
            M |= (u64)(nib & 0xFu) << (4 * k);
        }
        head = d[0] & 63u;
    }
    M &= ~1ull; // slot 0 is the head byte
    u64 tmask; // candidates of the row by recency rank (bit rank)
    {
        u64 const rot = head ? ((M >> head) | (M << (64 - head))) : M; // bit j <-> slot (head + j) & 63
        u32 const j0 = (64u - head) & 63u;
        u64 const low = rot & ((1ull << j0) - 1);
        u64 const high = (j0 == 63) ? 0ull : ((rot >> (j0 + 1)) << j0);
        tmask = low | high;
        u32 const keep = 63u - r; // the r oldest entries are overwritten by then
        tmask &= (keep >= 64) ? ~0ull : ((1ull << keep) - 1);
    }
    if (!valid || nearEnd) imask = tmask = 0;
    bt.imask = imask;
    bt.tmask = tmask;
    bt.head = head;
    bt.rowId = rowId;
    bt.cnt = (u32)__popcll(imask) + (u32)__popcll(tmask); // <= 63
    bt.ml = 3;
    bt.off = 0;
    bt.be = 0;
    bt.flags = flags;
    { // what rb_eval reads of OTHER lanes goes to LDS once per batch
        SQ_L u32* const Lrow = (SQ_L u32*)g_lds.u.rb.rowId;
        SQ_L u32* const Lprev = (SQ_L u32*)g_lds.u.rb.prev;
        SQ_L u32* const Lown = (SQ_L u32*)g_lds.grp;
        Lrow[lane] = rowId;
        Lprev[2 * lane] = (u32)bt.ownPrev;
        Lprev[2 * lane + 1] = (u32)(bt.ownPrev >> 32);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Lown[8 * lane + k] = own0[k];
            Lown[8 * lane + 4 + k] = own1[k];
        }
        bt.ldsOk = 1;
        wv_fence();
    }
    PROF_ADD(4, t_rb);
    PROF_CNT(16, 1);
}

// Part 2: the candidates of lanes i0 .. i0+W-1 (W as large as one round of pairs allows), balanced over the wave.
// No lane-serial loop anywhere: the (owner, candidate) pairs are listed in LDS by one pass per OWNER (its two
// candidate masks are wave-uniform there: lane j places candidate j at the rank a popcount gives it), every lane then
// evaluates up to four pairs, and the per-owner result -- libzstd's "first longest candidate, stop at a slot that
// was never written" -- is an LDS atomic min (first stopper) followed by an atomic max over (length, -order).
SQ_DEVFN void rb_eval(MatchState& ms, RowBatch& bt, u32 i0, u32 iendPos) {
    PROF_T0(t_ev);
    u32 const lane = wv_lane();
    const u8* const base = ms.base;
    u32 const B0 = bt.B0;
    u32 const q = B0 + (lane < bt.n ? lane : 0u);
    u32 flags = bt.flags;
    bool const cand_lane = lane >= i0 && lane < bt.n && !(flags & SQZ_RB_EXACT);
    u32 const mycnt = cand_lane ? bt.cnt : 0u;
    u32 total;
    u32 const pbase = wv_excl_scan(mycnt, &total);
    // a contiguous run of lanes from i0: the three positions the lazy parser is about to look at in any case, more
    // only while they are cheap (SQZ_RB_SPEC pairs): candidate evaluation runs at the HBM's rate of random 64-byte
    // sectors (measured: 8 pairs per input byte on mixed data), so pairs of positions the parser may skip are the
    // one thing not to spend
    bool const sel = lane >= i0 && lane < bt.n && pbase + mycnt <= SQZ_RB_ROUND && (lane < i0 + 3 || pbase + mycnt <= SQZ_RB_SPEC);
    u32 const cnt = (sel && cand_lane) ? mycnt : 0u;
    u64 const owners = wv_ballot(cnt != 0);
    SQ_L u16* const Ldesc = (SQ_L u16*)g_lds.u.rb.desc;
    SQ_L u32* const Lcand = (SQ_L u32*)g_lds.u.rb.cand;
    SQ_L u32* const Lrow = (SQ_L u32*)g_lds.u.rb.rowId;
    SQ_L u32* const Lpb = (SQ_L u32*)g_lds.u.rb.pbase;
    SQ_L u32* const Lstop = (SQ_L u32*)g_lds.u.rb.stop;
    SQ_L u32* const Lbest = (SQ_L u32*)g_lds.u.rb.best;
    SQ_L u32* const Lflag = (SQ_L u32*)g_lds.u.rb.flag;
    SQ_L u32* const Lown = (SQ_L u32*)g_lds.grp; // the 32 bytes at every lane's position
    Lpb[lane] = pbase;
    Lstop[lane] = 0xFFFFFFFFu;
    Lbest[lane] = 0;
    Lflag[lane] = 0;
    if (!bt.ldsOk) { // an insertion sweep (exact finder) has used the scratch in between
#pragma unroll
        for (int k = 0; k < 4; k++) {
            Lown[8 * lane + k] = bt.own0[k];
            Lown[8 * lane + 4 + k] = bt.own1[k];
        }
        bt.ldsOk = 1;
    }
    PROF_ADD(2, t_ev); // setup: scan + LDS cells
    PROF_T0(t_ol);
    u32 P = 0;
    {
        u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        u64 const above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
        u64 todo = owners;
        while (todo) { // one pass per owner: lane j holds its candidate number j
            u32 const o = (u32)__builtin_ctzll(todo);
            todo &= todo - 1;
            u64 const im = wv_readlane64(bt.imask, o), tm = wv_readlane64(bt.tmask, o);
            u32 const pb = wv_readlane(pbase, o), hd = wv_readlane(bt.head, o);
            u32 const nIn = (u32)__popcll(im);
            if ((im >> lane) & 1) Ldesc[pb + (u32)__popcll(im & above)] = (u16)((o << 7) | 64u | lane); // most recent (highest lane) first
            if ((tm >> lane) & 1) Ldesc[pb + nIn + (u32)__popcll(tm & below)] = (u16)((o << 7) | rb_slot_of(hd, lane));
            P = pb + nIn + (u32)__popcll(tm);
        }
    }
    wv_fence();
    PROF_ADD(3, t_ol); // owner loop
    PROF_T0(t_tr);
    if (P) {
        PROF_CNT(19, 1);
        PROF_CNT(22, P);
        u32 c[4], own[4], len[4], ord[4];
        bool has[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { // the candidates' positions: in-batch ones are known, the row's are loaded
            u32 const p = 64u * (u32)k + lane;
            has[k] = p < P;
            u32 const d = has[k] ? (u32)Ldesc[p] : 0u;
            own[k] = (d >> 7) & 63u;
            ord[k] = p - Lpb[own[k]];
            c[k] = B0 + (d & 63u);
            if (has[k] && !(d & 64u)) c[k] = sq_tld32(ms.hashTable + ((size_t)Lrow[own[k]] << 6) + (d & 63u));
        }
        // Requests are what this kernel is short of: per pair only the FIRST 16 bytes of the candidate are
        // fetched here; the second 16 only for the pairs whose first 16 all match (a second, sparse trip), and the 8 bytes
        // in front of a candidate -- the backward extension -- only for each owner's winner (below).
        sq_v4u b0[4];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // an unwritten slot (reads 0) ends its owner's scan, as does an impossible index: neither is dereferenced
            bool const ok = has[k] && c[k] >= SQZ_WINDOW_START && c[k] < B0 + own[k];
            if (has[k] && !ok) {
                __hip_atomic_fetch_min(&Lstop[own[k]], ord[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                bad |= c[k] >= B0 + own[k];
                has[k] = false;
            }
            u32 const cc = ok ? c[k] : SQZ_WINDOW_START;
            b0[k] = sq_grd128(base + cc);
        }
        if (wv_ballot(bad)) ms.fault = 1;
        bool more[4];
        bool anyMore = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            len[k] = 0;
            more[k] = false;
            if (has[k]) {
                sq_v4u const o0 = *(const SQ_L sq_v4u*)(Lown + 8 * own[k]);
                len[k] = rb_prefix16(o0, b0[k]);
                more[k] = len[k] == 16;
                anyMore |= more[k];
            }
        }
        if (wv_ballot(anyMore)) { // the sparse second trip
            sq_v4u b1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) b1[k] = more[k] ? sq_grd128(base + c[k] + 16) : (sq_v4u){0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (more[k]) {
                    sq_v4u const o1 = *(const SQ_L sq_v4u*)(Lown + 8 * own[k] + 4);
                    len[k] = 16 + rb_prefix16(o1, b1[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (has[k]) {
                u32 const p = 64u * (u32)k + lane;
                u32 l = len[k];
                u32 const lim = iendPos - (B0 + own[k]);
                if (l >= 32 && lim > 32) l = 255; // long: the exact finder extends and decides
                else if (l > lim) l = lim;
                len[k] = l;
                Lcand[p] = c[k];
            }
        }
        wv_fence(); // (the stoppers are in before anybody reads them: LDS operations of a wave execute in order)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (has[k] && ord[k] < Lstop[own[k]]) {
                if (len[k] == 255) __hip_atomic_fetch_or(&Lflag[own[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_max(&Lbest[own[k]], (len[k] << 8) | (255u - ord[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    wv_fence();
    PROF_ADD(13, t_tr); // the two trips + compares + atomics
    u32 ml = 3, bestC = 0, be = 0;
    if (cnt) {
        // (a long candidate anywhere in front of the first stopper sends the position to the exact finder, even where
        // libzstd's scan would have ended before reaching it: asking the exact finder is always right)
        if (Lflag[lane]) flags |= SQZ_RB_EXACT;
        else {
            u32 const b = Lbest[lane];
            if ((b >> 8) > 3) {
                u32 const t = 255u - (b & 255u);
                ml = b >> 8;
                bestC = Lcand[pbase + t];
            }
        }
    }
    {   // backward extension (catch-up) of each owner's winner: equal bytes in front of both positions, at most 8; only
        // meaningful (and only used) when both have 8 bytes in front of them -- one 8-byte load per owner with a match
        bool const want = ml > 3 && bestC >= 8 + SQZ_WINDOW_START && q >= 8 + SQZ_WINDOW_START;
        if (wv_ballot(want)) {
            u64 const bp = want ? sq_grd64(base + bestC - 8) : 0ull;
            u64 const x = bt.ownPrev ^ bp;
            if (want) be = x ? ((u32)__builtin_clzll(x) >> 3) : SQZ_RB_BECAP;
        }
    }
    if (sel) {
        bt.ml = ml;
        bt.off = OFFSET_TO_OFFBASE(q - bestC);
        bt.be = be;
        bt.flags = flags | SQZ_RB_DONE;
    }
    PROF_ADD(5, t_ev);
}
#else
// ---- simulator twin: the same batch, lane after lane, in plain scalar code
SQ_DEV void rb_refresh(const MatchState& ms, RowBatch& bt, u32 off1, u32 iendPos) {
    bt.off1 = off1;
    for (u32 i = 0; i < 64; i++) {
        u32 const q = bt.B0 + (i < bt.n ? i : 0u);
        u32 rl = 0;
        u32 const lim = iendPos - q;
        if (off1 && q >= off1 + SQZ_WINDOW_START)
            while (rl < lim && rl < SQZ_RB_RLCAP && ms.base[q + rl] == ms.base[q + rl - off1]) rl++;
        bt.rl[i] = rl;
    }
}
// rb_prepare: per lane the candidate masks exactly as the device computes them (in-batch lanes with the same row and
// tag; the row's tag matches by recency rank minus the r entries the batch itself will have overwritten) -- nothing is
// evaluated here.
SQ_DEVFN void rb_prepare(MatchState& ms, RowBatch& bt, u32 B0, u32 n, u32 off1, u32 iendPos) {
    const u8* const base = ms.base;
    bt.B0 = B0;
    bt.n = n;
    bt.valid = 1;
    u32 hashes[64];
    for (u32 i = 0; i < 64; i++) {
        bt.ml[i] = 3;
        bt.off[i] = bt.be[i] = 0;
        bt.flags[i] = 0;
        bt.imask[i] = bt.tmask[i] = 0;
        bt.head[i] = bt.rowId[i] = bt.cnt[i] = 0;
        hashes[i] = 0;
    }
    rb_refresh(ms, bt, off1, iendPos);
    for (u32 i = 0; i < n; i++) hashes[i] = hash5(sq_rd64(base + B0 + i), ms.rowHashLog + 8);
    for (u32 i = 0; i < n; i++) {
        u32 const q = B0 + i;
        if (q + 32 > ms.chunkEndPos) { // positions whose 32-byte windows would reach beyond the chunk go to the exact finder
            bt.flags[i] = SQZ_RB_EXACT;
            continue;
        }
        u32 const hash = hashes[i], rowId = hash >> 8, tag = hash & 0xFF;
        u32 r = 0;
        u64 imask = 0;
        for (u32 j = 0; j < i; j++)
            if ((hashes[j] >> 8) == rowId) {
                r++;
                if ((hashes[j] & 0xFF) == tag) imask |= 1ull << j;
            }
        const u8* const tagRow = ms.tagTable + ((size_t)rowId << 6);
        u32 const head = tagRow[0] & 63u;
        u64 tmask = 0;
        for (u32 rank = 0; rank + r < 63; rank++) { // the r oldest entries are overwritten by then
            u32 const slot = rb_slot_of(head, rank);
            if (slot != 0 && tagRow[slot] == (u8)tag) tmask |= 1ull << rank;
        }
        bt.imask[i] = imask;
        bt.tmask[i] = tmask;
        bt.head[i] = head;
        bt.rowId[i] = rowId;
        bt.cnt[i] = (u32)__builtin_popcountll(imask) + (u32)__builtin_popcountll(tmask);
    }
}

// rb_eval: the device routine statement by statement over the SAME arrays (g_lds.u.rb: descriptors, candidate
// positions, per-owner pair bases / first stoppers / best / "long candidate" flags), scalar loops in place of the lanes:
// the window of owners one round of SQZ_RB_ROUND pairs allows, the descriptor listing (in-batch candidates most recent
// first, then the row's by recency rank), the pairs' lengths from at most 32 bytes, first stopper = minimum order,
// winner = maximum of (length, -order) in front of it, backward extension of the winner only.  The bounds the device
// relies on are asserted here, so a sanitizer build of the simulator sees an out-of-range descriptor.
#define SQZ_SIM_CHECK(c)                                                                                    \
    do {                                                                                                    \
        if (!(c)) {                                                                                         \
            ms.fault = 1;                                                                                   \
            __builtin_trap();                                                                               \
        }                                                                                                   \
    } while (0)
// (measurement, simulator build only: rounds, pairs, pairs with a live candidate, distinct diagonals owner - candidate per
// round, and how many pairs a diagonal's FIRST pair would also decide (the later pairs of a run of consecutive owners on one
// diagonal, inside the first one's match) -- the round-3 review's question whether one fetch per diagonal instead of one
// per pair would pay, tools/diag_stats.py)
static unsigned long long g_sim_rb_stats[5];
SQ_DEVFN void rb_eval(MatchState& ms, RowBatch& bt, u32 i0, u32 iendPos) {
    const u8* const base = ms.base;
    u32 const B0 = bt.B0;
    RbLds& R = g_lds.u.rb;
    u32 pbase[64], cnt[64];
    bool sel[64];
    u32 run = 0;
    for (u32 lane = 0; lane < 64; lane++) {
        bool const cand_lane = lane >= i0 && lane < bt.n && !(bt.flags[lane] & SQZ_RB_EXACT);
        u32 const mycnt = cand_lane ? bt.cnt[lane] : 0u;
        pbase[lane] = run;
        sel[lane] = lane >= i0 && lane < bt.n && run + mycnt <= SQZ_RB_ROUND && (lane < i0 + 3 || run + mycnt <= SQZ_RB_SPEC);
        cnt[lane] = (sel[lane] && cand_lane) ? mycnt : 0u;
        run += mycnt;
        R.pbase[lane] = pbase[lane];
        R.stop[lane] = 0xFFFFFFFFu;
        R.best[lane] = 0;
        R.flag[lane] = 0;
        R.rowId[lane] = bt.rowId[lane];
    }
    u32 P = 0;
    for (u32 o = 0; o < 64; o++) { // one pass per owner
        if (!cnt[o]) continue;
        u64 const im = bt.imask[o], tm = bt.tmask[o];
        u32 const pb = pbase[o], nIn = (u32)__builtin_popcountll(im);
        for (u32 lane = 0; lane < 64; lane++) {
            u64 const above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
            u64 const below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            if ((im >> lane) & 1) {
                u32 const at = pb + (u32)__builtin_popcountll(im & above);
                SQZ_SIM_CHECK(at < SQZ_RB_PAIRS);
                R.desc[at] = (u16)((o << 7) | 64u | lane);
            }
            if ((tm >> lane) & 1) {
                u32 const at = pb + nIn + (u32)__builtin_popcountll(tm & below);
                SQZ_SIM_CHECK(at < SQZ_RB_PAIRS);
                R.desc[at] = (u16)((o << 7) | rb_slot_of(bt.head[o], lane));
            }
        }
        P = pb + nIn + (u32)__builtin_popcountll(tm);
    }
    SQZ_SIM_CHECK(P <= SQZ_RB_ROUND);
    u32 len[SQZ_RB_PAIRS], ord[SQZ_RB_PAIRS], own[SQZ_RB_PAIRS];
    bool has[SQZ_RB_PAIRS];
    for (u32 p = 0; p < P; p++) { // the pairs: positions, stoppers, lengths
        u32 const d = R.desc[p];
        own[p] = (d >> 7) & 63u;
        SQZ_SIM_CHECK(p >= R.pbase[own[p]]);
        ord[p] = p - R.pbase[own[p]];
        u32 c = B0 + (d & 63u);
        if (!(d & 64u)) c = ms.hashTable[((size_t)R.rowId[own[p]] << 6) + (d & 63u)];
        bool const ok = c >= SQZ_WINDOW_START && c < B0 + own[p];
        has[p] = ok;
        len[p] = 0;
        if (!ok) { // an unwritten slot (reads 0) ends its owner's scan, as does an impossible index
            if (ord[p] < R.stop[own[p]]) R.stop[own[p]] = ord[p];
            if (c >= B0 + own[p]) ms.fault = 1;
            continue;
        }
        u32 const q = B0 + own[p];
        u32 l = 0;
        while (l < 32 && base[q + l] == base[c + l]) l++; // (q + 32 <= chunk end for every owner: rb_prepare)
        u32 const lim = iendPos - q;
        if (l >= 32 && lim > 32) l = 255; // long: the exact finder extends and decides
        else if (l > lim) l = lim;
        len[p] = l;
        R.cand[p] = c;
    }
    { // measurement only (see g_sim_rb_stats)
        g_sim_rb_stats[0]++;
        g_sim_rb_stats[1] += P;
        s64 dg[SQZ_RB_PAIRS];
        u32 nd = 0, live = 0, covered = 0;
        for (u32 p = 0; p < P; p++) {
            if (!has[p]) continue;
            live++;
            s64 const d = (s64)(B0 + own[p]) - (s64)R.cand[p];
            bool seen = false;
            for (u32 k = 0; k < nd; k++) seen = seen || dg[k] == d;
            if (!seen) dg[nd++] = d;
            // is this pair inside the match of an earlier owner's pair on the same diagonal?
            for (u32 r = 0; r < P; r++)
                if (has[r] && own[r] < own[p] && (s64)(B0 + own[r]) - (s64)R.cand[r] == d && len[r] != 255 && own[r] + len[r] > own[p]) {
                    covered++;
                    break;
                }
        }
        g_sim_rb_stats[2] += live;
        g_sim_rb_stats[3] += nd;
        g_sim_rb_stats[4] += covered;
    }
    for (u32 p = 0; p < P; p++)
        if (has[p] && ord[p] < R.stop[own[p]]) {
            if (len[p] == 255) R.flag[own[p]] |= 1u;
            else {
                u32 const v = (len[p] << 8) | (255u - ord[p]);
                if (v > R.best[own[p]]) R.best[own[p]] = v;
            }
        }
    for (u32 lane = 0; lane < 64; lane++) {
        if (!sel[lane]) continue;
        u32 const q = B0 + lane;
        u32 ml = 3, bestC = 0, be = 0, flags = bt.flags[lane];
        if (cnt[lane]) {
            if (R.flag[lane]) flags |= SQZ_RB_EXACT;
            else {
                u32 const b = R.best[lane];
                if ((b >> 8) > 3) {
                    u32 const t = 255u - (b & 255u);
                    SQZ_SIM_CHECK(pbase[lane] + t < P);
                    ml = b >> 8;
                    bestC = R.cand[pbase[lane] + t];
                }
            }
        }
        if (ml > 3 && bestC >= 8 + SQZ_WINDOW_START && q >= 8 + SQZ_WINDOW_START)
            while (be < SQZ_RB_BECAP && base[q - 1 - be] == base[bestC - 1 - be]) be++;
        bt.ml[lane] = ml;
        bt.off[lane] = OFFSET_TO_OFFBASE(q - bestC);
        bt.be[lane] = be;
        bt.flags[lane] = flags | SQZ_RB_DONE;
    }
}
#endif

// ZSTD_storeSeq without the literal copy: the sequence record, plus (source position, offset in the literal
// buffer) of its literals in the block's position list (ss.stBits, free until the entropy stage).
SQ_DEV void storeSeq_deferred(SeqStore& ss, u32 litLength, u32 litPos, u32 offBase, u32 matchLength) {
    if (litLength > 0xFFFF) {
        ss.longLengthType = 1;
        ss.longLengthPos = ss.nseq;
    }
    u32 const mlBase = matchLength - SQZ_MINMATCH;
    if (mlBase > 0xFFFF) {
        ss.longLengthType = 2;
        ss.longLengthPos = ss.nseq;
    }
    u32* const posList = (u32*)ss.stBits;
#if SQ_ON_DEVICE
    if (wv_lane() == 0) {
        sq_gst64v((u32*)&ss.seqs[ss.nseq], offBase, (litLength & 0xFFFF) | (mlBase << 16));
        sq_gst64v(posList + 2 * (size_t)ss.nseq, litPos, ss.nlits);
    }
#else
    ss.seqs[ss.nseq].offBase = offBase;
    ss.seqs[ss.nseq].litLength = (u16)litLength;
    ss.seqs[ss.nseq].mlBase = (u16)mlBase;
    posList[2 * (size_t)ss.nseq] = litPos;
    posList[2 * (size_t)ss.nseq + 1] = ss.nlits;
#endif
    ss.nlits += litLength;
    ss.nseq++;
}

// Copies every sequence's literals into the literal buffer: one sequence per lane, short runs by the lane itself,
// long runs (> 64 bytes) by the whole wave.
SQ_DEVFN void gather_literals(const SeqStore& ss, const u8* base) {
    const u32* const posList = (const u32*)ss.stBits;
#if SQ_ON_DEVICE
    wv_mem_sync(); // the records are in memory
    u32 const lane = wv_lane();
    for (u32 s0 = 0; s0 < ss.nseq; s0 += 64) {
        u32 const s = s0 + lane;
        bool const act = s < ss.nseq;
        u32 srcPos = 0, dstOff = 0, ll = 0;
        if (act) {
            sq_v2u const p = sq_gld64v(posList + 2 * (size_t)s);
            srcPos = p.x;
            dstOff = p.y;
            ll = ((const SQ_G SeqDef*)ss.seqs)[s].litLength;
            if (ss.longLengthType == 1 && ss.longLengthPos == s) ll += 0x10000;
        }
        bool const big = ll > 64;
        if (act && !big) {
            const u8* sp = base + srcPos;
            u8* dp = ss.lits + dstOff;
            u32 o = 0;
            for (; o + 16 <= ll; o += 16) sq_gwr128(dp + o, sq_grd128(sp + o));
            if (o + 8 <= ll) {
                *(SQ_G sq_u64_ua*)(dp + o) = sq_grd64(sp + o);
                o += 8;
            }
            if (o + 4 <= ll) {
                *(SQ_G sq_u32_ua*)(dp + o) = sq_grd32(sp + o);
                o += 4;
            }
            for (; o < ll; o++) sq_gwr8(dp + o, sq_grd8(sp + o));
        }
        u64 bm = wv_ballot(act && big);
        while (bm) {
            u32 const l = (u32)__builtin_ctzll(bm);
            bm &= bm - 1;
            wv_memcpy(ss.lits + wv_readlane(dstOff, l), base + wv_readlane(srcPos, l), wv_readlane(ll, l));
        }
    }
    wv_mem_sync();
#else
    for (u32 s = 0; s < ss.nseq; s++) {
        u32 ll = ss.seqs[s].litLength;
        if (ss.longLengthType == 1 && ss.longLengthPos == s) ll += 0x10000;
        memcpy(ss.lits + posList[2 * (size_t)s + 1], base + posList[2 * (size_t)s], ll);
    }
#endif
}

// ZSTD_compressBlock_lazy_generic (noDict, depth 2, row-hash finder) over search batches; returns last literals size
SQ_DEV u32 compressBlock_lazy_rb_impl(MatchState& ms, SeqStore& ss, u32* rep, const u8* src, u32 srcSize) {
    const u8* const base = ms.base;
    u32 const istartPos = (u32)(src - base), iendPos = istartPos + srcSize;
    u32 const ilimitPos = iendPos - 8 - 8;
    u32 const prefixLowestPos = SQZ_WINDOW_START;
    u32 ip = istartPos, anchor = istartPos;
    u32 offset_1 = rep[0], offset_2 = rep[1];
    u32 offsetSaved1 = 0, offsetSaved2 = 0;
    // A block of a pair that stage 1 certified (no search of it finds a match, ms.certPairs): no searches, only their
    // effect on the tables (row_replay_nomatch).  The certificate assumed full 128 KiB blocks at their places and the
    // repcode offset 1; another offset is re-checked over the block's schedule.
    if (ms.certPairs) {
        u32 const o = istartPos - SQZ_WINDOW_START;
        if ((o & (SQZ_BLOCKSIZE_MAX - 1)) == 0 && ((ms.certPairs >> (o >> 18)) & 1) &&
            (srcSize == SQZ_BLOCKSIZE_MAX || iendPos == ms.chunkEndPos) && offset_1 != 0 && offset_1 <= istartPos - SQZ_WINDOW_START + (o == 0 ? 1u : 0u) &&
            (offset_1 == 1 || row_sched_rep_clear(ms, istartPos, srcSize, offset_1))) {
            row_replay_nomatch(ms, istartPos, srcSize);
            return srcSize; // no sequence: the repcodes stay as they came
        }
    }
    ip += (ip == prefixLowestPos);
    {
        u32 const maxRep = ip - SQZ_WINDOW_START;
        if (offset_2 > maxRep) offsetSaved2 = offset_2, offset_2 = 0;
        if (offset_1 > maxRep) offsetSaved1 = offset_1, offset_1 = 0;
    }
    ms.lazySkipping = 0;
    RowBatch bt;
    bt.valid = 0;
    bt.B0 = bt.n = bt.off1 = bt.ldsOk = 0;
    u32 physNext = ms.nextToUpdate; // positions below it are in the tables (as far as libzstd inserts them)

    // everything the finder logically holds (positions < ms.nextToUpdate) goes into the tables
    auto flush = [&]() {
        PROF_T0(t_fl);
        if (physNext < ms.nextToUpdate) row_insert_range(ms, physNext, ms.nextToUpdate);
        physNext = ms.nextToUpdate;
        PROF_ADD(9, t_fl);
    };
    // ZSTD_searchMax at position q
    auto search = [&](u32 q, u32* offOut) -> u32 {
        if (ms.lazySkipping) {
            flush();
            bt.valid = 0;
            u32 const r = row_find(ms, base + q, base + iendPos, offOut);
            physNext = ms.nextToUpdate;
            return r;
        }
        if (!bt.valid || q < bt.B0 || q >= bt.B0 + bt.n) {
            flush();
            if (ms.nextToUpdate < q) row_update(ms, q); // tables = everything libzstd has inserted before this search
            physNext = ms.nextToUpdate;
            if (ms.nextToUpdate != q) { // (cannot happen: nextToUpdate never runs ahead of the parser)
                bt.valid = 0;
                u32 const r = row_find(ms, base + q, base + iendPos, offOut);
                physNext = ms.nextToUpdate;
                return r;
            }
            u32 const left = ilimitPos + 1 - q;
            rb_prepare(ms, bt, q, left < SQZ_RB_N ? left : SQZ_RB_N, offset_1, iendPos);
        }
        u32 const i = q - bt.B0;
        if (!(RB_GET(bt, flags, i) & (SQZ_RB_EXACT | SQZ_RB_DONE))) rb_eval(ms, bt, i, iendPos);
        if (RB_GET(bt, flags, i) & SQZ_RB_EXACT) {
            PROF_CNT(17, 1);
            bt.ldsOk = 0; // (row_insert_range ranks equal rows in the same LDS scratch)
            if (physNext < q) row_insert_range(ms, physNext, q);
            ms.nextToUpdate = q;
            u32 const r = row_find(ms, base + q, base + iendPos, offOut);
            physNext = ms.nextToUpdate;
            return r;
        }
        ms.nextToUpdate = q + 1;
        PROF_CNT(20, 1);
        u32 const ml = RB_GET(bt, ml, i);
        if (ml > 3) *offOut = RB_GET(bt, off, i);
        return ml;
    };
    // The parser's state is wave-uniform by construction; saying so (v_readfirstlane) where values come back from the
    // finder keeps it in SGPRs: scalar ALU and scalar branches instead of VALU arithmetic under exec masks.
    auto searchU = [&](u32 q, u32* offOut) -> u32 {
        u32 o = *offOut;
        u32 const r = wv_uni(search(wv_uni(q), &o));
        *offOut = wv_uni(o);
        return r;
    };
    // length of the repcode match at q for offset `off` (0 if its first 4 bytes differ)
    auto rep_len = [&](u32 q, u32 off, bool mayRefresh) -> u32 {
        if (bt.valid && q >= bt.B0 && q < bt.B0 + bt.n && (bt.off1 == off || mayRefresh)) {
            if (bt.off1 != off) {
                PROF_T0(t_rf);
                rb_refresh(ms, bt, off, iendPos);
                PROF_ADD(10, t_rf);
                PROF_CNT(18, 1);
            }
            u32 const rl = RB_GET(bt, rl, q - bt.B0);
            if (rl < 4) return 0;
            u32 const whole = iendPos - q;
            if (rl < SQZ_RB_RLCAP || rl >= whole) return rl;
            return SQZ_RB_RLCAP + wv_count(base + q + SQZ_RB_RLCAP, base + q + SQZ_RB_RLCAP - off, whole - SQZ_RB_RLCAP);
        }
        PROF_T0(t_rd);
        PROF_CNT(21, 1);
        u32 r = 0;
        if (srd32(base + q) == srd32(base + q - off)) r = wv_count(base + q + 4, base + q + 4 - off, iendPos - (q + 4)) + 4;
        PROF_ADD(12, t_rd);
        return r;
    };
    auto repU = [&](u32 q, u32 off, bool mayRefresh) -> u32 { return wv_uni(rep_len(wv_uni(q), wv_uni(off), mayRefresh)); };

    while (ip < ilimitPos) {
        ip = wv_uni(ip);
        anchor = wv_uni(anchor);
        offset_1 = wv_uni(offset_1);
        offset_2 = wv_uni(offset_2);
        u32 matchLength = 0;
        u32 offBase = 1; // REPCODE1_TO_OFFBASE
        u32 start = ip + 1;
        {
            // (the search first: it makes the batch that then answers the repcode test at ip + 1 as well; the two
            // are independent, the comparison below is libzstd's)
            u32 offbaseFound = 999999999;
            u32 const ml2 = searchU(ip, &offbaseFound);
            if (ms.fault) break;
            if (offset_1 > 0) matchLength = repU(ip + 1, offset_1, true);
            if (ml2 > matchLength) matchLength = ml2, start = ip, offBase = offbaseFound;
        }
        if (matchLength < 4) {
            u32 const step = ((ip - anchor) >> 8) + 1;
            ip += step;
            ms.lazySkipping = step > 8;
            // Fast-forward over positions that are known to give nothing: while the step is 1 the parser would visit
            // ip, ip+1, ... one by one, each time finding no match (the batch says so) and no repcode match at the
            // position after it; jump to the first position where one of these is not known to hold.
            if (step == 1 && bt.valid && ip > bt.B0 && ip < bt.B0 + bt.n && (offset_1 == 0 || bt.off1 == offset_1)) {
                u64 const nom = rb_nomatch_mask(bt);
                u64 const rep = offset_1 ? rb_rep_mask(bt) : 0ull;
                // lane j is skippable: no match at j, no repcode match at j+1 (which must be a lane of the batch)
                u64 skippable = nom & ~(rep >> 1);
                skippable &= (bt.n >= 64) ? (~0ull >> 1) : ((1ull << (bt.n - 1)) - 1);
                u32 const i = ip - bt.B0;
                u64 const stop = ~skippable & (~0ull << i); // first lane >= i that must be visited
                u32 j = stop ? (u32)__builtin_ctzll(stop) : 64u;
                u32 lim = anchor + 256 - bt.B0; // the step is 1 only while ip - anchor < 256
                if (ilimitPos - bt.B0 < lim) lim = ilimitPos - bt.B0;
                if (j > lim) j = lim;
                if (j > i) {
                    ip = bt.B0 + j;
                    ms.nextToUpdate = ip; // every position in between counts as searched (and inserted)
                }
            }
            continue;
        }
        while (ip < ilimitPos) {
            ip++;
            if (offset_1 > 0) {
                u32 const mlRep = repU(ip, offset_1, true);
                int const gain2 = (int)(mlRep * 3);
                int const gain1 = (int)(matchLength * 3 - sq_highbit32(offBase) + 1);
                if ((mlRep >= 4) && (gain2 > gain1)) matchLength = mlRep, offBase = 1, start = ip;
            }
            {
                u32 ofbCandidate = 999999999;
                u32 const ml2 = searchU(ip, &ofbCandidate);
                int const gain2 = (int)(ml2 * 4 - sq_highbit32(ofbCandidate));
                int const gain1 = (int)(matchLength * 4 - sq_highbit32(offBase) + 4);
                if ((ml2 >= 4) && (gain2 > gain1)) {
                    matchLength = ml2, offBase = ofbCandidate, start = ip;
                    continue;
                }
            }
            if (ip < ilimitPos) {
                ip++;
                if (offset_1 > 0) {
                    u32 const mlRep = repU(ip, offset_1, true);
                    int const gain2 = (int)(mlRep * 4);
                    int const gain1 = (int)(matchLength * 4 - sq_highbit32(offBase) + 1);
                    if ((mlRep >= 4) && (gain2 > gain1)) matchLength = mlRep, offBase = 1, start = ip;
                }
                {
                    u32 ofbCandidate = 999999999;
                    u32 const ml2 = searchU(ip, &ofbCandidate);
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
        if (ms.fault) break;
        if (OFFBASE_IS_OFFSET(offBase)) {
            u32 const off = OFFBASE_TO_OFFSET(offBase);
            // catch up: equal bytes in front of the match
            u32 const room = (start - anchor) < (start - off - prefixLowestPos) ? (start - anchor) : (start - off - prefixLowestPos);
            u32 back = 0;
            bool known = false;
            if (bt.valid && start >= bt.B0 && start < bt.B0 + bt.n && !(RB_GET(bt, flags, start - bt.B0) & SQZ_RB_EXACT) &&
                RB_GET(bt, ml, start - bt.B0) > 3 && RB_GET(bt, off, start - bt.B0) == offBase && start >= 8 + SQZ_WINDOW_START &&
                start - off >= 8 + SQZ_WINDOW_START) {
                u32 const be = RB_GET(bt, be, start - bt.B0);
                back = be < room ? be : room;
                known = be < SQZ_RB_BECAP || back == room;
            }
            if (!known)
                while (back < room && srd8(base + start - back - 1) == srd8(base + start - back - off - 1)) back++;
            start -= back;
            matchLength += back;
            offset_2 = offset_1;
            offset_1 = off;
        }
        storeSeq_deferred(ss, start - anchor, anchor, offBase, matchLength);
        anchor = ip = start + matchLength;
        if (ms.lazySkipping) ms.lazySkipping = 0;

        while ((ip <= ilimitPos) && (offset_2 > 0)) {
            u32 const ml2 = repU(ip, offset_2, false);
            if (ml2 < 4) break;
            u32 const t = offset_2;
            offset_2 = offset_1;
            offset_1 = t;
            storeSeq_deferred(ss, 0, anchor, 1, ml2);
            ip += ml2;
            anchor = ip;
        }
    }
    flush();
    offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
    rep[0] = offset_1 ? offset_1 : offsetSaved1;
    rep[1] = offset_2 ? offset_2 : offsetSaved2;
    PROF_T0(t_gl);
    gather_literals(ss, base);
    PROF_ADD(11, t_gl);
    return iendPos - anchor;
}

SQ_DEVFN u32 compressBlock_lazy_rowbatch(MatchState& ms, SeqStore& ss, u32* rep, const u8* src, u32 srcSize) {
#if SQ_ON_DEVICE
    MatchState lm = ms; // a private copy: its fields stay in registers (see compressBlock_lazy)
    u32 const r = compressBlock_lazy_rb_impl(lm, ss, rep, src, srcSize);
    ms = lm;
    return r;
#else
    return compressBlock_lazy_rb_impl(ms, ss, rep, src, srcSize);
#endif
}
