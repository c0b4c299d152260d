// tests/sim/sim_main.cpp -- TEST INFRASTRUCTURE.  Builds the product's wave-level compressor
// source (squishrs_amd/csrc/sq_zstd_*.h) with -DSQ_SIM, where every wave primitive has a scalar
// body, so the uniform control logic can be diffed against the oracle on a machine without a
// GPU.  Never linked into libsquish_gpu.so; not a fallback for anything.
#define SQ_SIM 1
#include "../../squishrs_amd/csrc/sq_zstd_kernel.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static const u32* sched_tab() {
    static u32 tab[SQZ_SCHEDTAB_WORDS];
    static bool done = false;
    if (!done) {
        sqz::sched_tab_fill(tab);
        done = true;
    }
    return tab;
}

extern "C" size_t sim_compress_chunk(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int poison) {
    sqz::WsLayout L = sqz::ws_layout((u32)n);
    u8* ws = (u8*)malloc(L.total + 256 + 1024);
    // poison the workspace: the kernel must not depend on anything it did not initialise itself
    memset(ws, poison ? 0xA5 : 0, L.total + 256 + 1024);
    memset(&sqz::g_lds, poison ? 0x5A : 0, sizeof(sqz::g_lds));
    u8* wsa = (u8*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    sqz::TableGen tg = {0, 0, 0};
    u32 r = sqz::compress_frame_wave(src, (u32)n, dst, (u32)cap, wsa, L, tg, sched_tab(), nullptr);
    if (getenv("SQ_SIM_STATS")) {
        const u32* t = (const u32*)(wsa + L.dummy) + 8;
        fprintf(stderr, "[sim] n=%zu eager=%u repairs=%u live=%u skips=%u epochs=%u specfail=%u replays=%u certified=%u lastblock_skipped_live=%u tail_certified=%u tail_tries=%u\n", n, t[30],
                t[31], t[32], t[33], t[34], t[35], t[36], t[37], t[38], t[39], t[43]);
    }
    free(ws);
    return r;
}
// Same, but with ONE workspace and ONE table-generation state kept across calls, as a persistent
// wave of k_compress_batch keeps them across the chunks it pulls from the queue.
static u8* g_ws = nullptr;
static sqz::TableGen g_tg = {0, 0, 0};
extern "C" void sim_persistent_reset(void) {
    free(g_ws);
    g_ws = nullptr;
    g_tg.gen = g_tg.windowLog = g_tg.tailTabClean = 0;
}
extern "C" unsigned sim_persistent_gen(void) { return g_tg.gen; }
extern "C" size_t sim_compress_persistent(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    if (!g_ws) {
        size_t const tot = sqz::workspace_bytes(2u << 20) + 512;
        g_ws = (u8*)malloc(tot);
        memset(g_ws, 0xC3, tot);
    }
    u8* wsa = (u8*)(((uintptr_t)g_ws + 255) & ~(uintptr_t)255);
    sqz::WsLayout L = sqz::ws_layout((u32)n); // per-chunk layout inside the pool slot, as compress_one does
    return sqz::compress_frame_wave(src, (u32)n, dst, (u32)cap, wsa, L, g_tg, sched_tab(), nullptr);
}
// The incompressible-chunk path (sq_zstd_fast.h) followed, when it leaves the chunk undecided, by the
// compressor with the certificate ends it computed -- the two-kernel sequence of sq_compress_batch.
// *decided = 1 when the fast path produced the frame.
extern "C" size_t sim_compress_two_stage(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int* decided) {
    static sqz::FastLds lds;
    static u32* etab = nullptr;
    static u32 gen = 0;
    if (!etab) {
        etab = (u32*)malloc(2u << 20);
        memset(etab, 0, 2u << 20);
    }
    if (++gen >= SQZ_HT_GEN_MAX) {
        memset(etab, 0, 2u << 20);
        gen = 1;
    }
    memset(&lds, 0x77, sizeof(lds));
    u32 hint[2] = {0x55555555u, 0x55555555u};
    u32 pairInfo = 0; // chunks above 256 KiB: stage 1's verdict per pair of blocks, as the kernel leaves it in pair_state
    u32 r = sqz::fastpath_frame_wave(src, (u32)n, dst, (u32)cap, etab, gen << SQZ_HT_GEN_SHIFT, sched_tab(), &lds, hint, 0, nullptr, nullptr, &pairInfo);
    *decided = r != 0;
    if (r) return r;
    sqz::WsLayout L = sqz::ws_layout((u32)n);
    u8* ws = (u8*)malloc(L.total + 256 + 1024);
    memset(ws, 0xA5, L.total + 256 + 1024);
    memset(&sqz::g_lds, 0x5A, sizeof(sqz::g_lds));
    u8* wsa = (u8*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    sqz::TableGen tg = {0, 0, 0};
    r = sqz::compress_frame_wave(src, (u32)n, dst, (u32)cap, wsa, L, tg, sched_tab(), hint, pairInfo);
    free(ws);
    return r;
}
extern "C" unsigned sim_one_match_blocks(void) { return sqz::g_sim_one_match_blocks; } // blocks stage 1's second pass decided so far
extern "C" unsigned sim_replayed_blocks(void) { return sqz::g_sim_replayed_blocks; } // blocks whose searches were replaced by row_replay_nomatch so far
// The arithmetic that tells the scan which positions are scheduled ones (fast_sched_mask) against the
// schedule tables, for a chunk of n bytes: returns the number of positions where they disagree and
// the first such position through *firstBad.
extern "C" unsigned sim_sched_mask_check(size_t n, unsigned* firstBad) {
    const u32* tab = sched_tab();
    u32 const blk1 = (n > SQZ_BLOCKSIZE_MAX) ? SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX : 0xFFFFFFFFu;
    std::vector<u8> want(n + 64, 0), got(n + 64, 0);
    u32 const len0 = n < SQZ_BLOCKSIZE_MAX ? (u32)n : SQZ_BLOCKSIZE_MAX, len1 = (u32)n - len0;
    for (u32 i = 0; i < tab[0]; i++)
        if (tab[2 + i] < len0 - 8) want[SQZ_WINDOW_START + tab[2 + i]] = 1;
    if (len1 >= 9)
        for (u32 i = 0; i < tab[1]; i++)
            if (tab[2 + SQZ_SCHEDTAB_N + i] < len1 - 8) want[SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX + tab[2 + SQZ_SCHEDTAB_N + i]] = 1;
    unsigned bad = 0;
    for (s32 align = 0; align < 16; align++) { // the scan's 16-byte windows sit at any alignment
        std::fill(got.begin(), got.end(), 0);
        for (s32 lo = align - 16; lo < (s32)(n + SQZ_WINDOW_START); lo += 16) {
            u32 const m = sqz::fast_sched_mask(tab + SQZ_SCHEDLVL_OFF, lo, blk1, (u32)n + SQZ_WINDOW_START, 0, 0, false);
            for (u32 j = 0; j < 16; j++)
                if (((m >> j) & 1) && lo + (s32)j >= 0) got[lo + j] = 1;
            // the loop-free variant the kernel uses for "plain" windows must agree with the general one
            bool const second = blk1 != 0xFFFFFFFFu && lo >= (s32)blk1;
            u32 const bs = second ? blk1 : SQZ_WINDOW_START;
            u32 const blkEnd = (!second && blk1 != 0xFFFFFFFFu) ? blk1 : (u32)n + SQZ_WINDOW_START;
            s32 const r = lo - (s32)bs;
            bool const plain = r >= 4096 && (u32)(lo + 16) <= blkEnd - 8 && !(blk1 != 0xFFFFFFFFu && lo < (s32)blk1 && lo + 16 > (s32)blk1);
            if (plain) {
                const u32* lv = tab + SQZ_SCHEDLVL_OFF + (second ? SQZ_SCHEDLVL_N : 0u);
                u32 const mp = sqz::fast_sched_mask_plain((u32)r, lv[(u32)r >> 8], lv[((u32)r + 15) >> 8]);
                if (mp != m) {
                    if (!bad) *firstBad = (unsigned)lo;
                    bad++;
                }
            }
        }
        for (u32 p = 0; p < n + SQZ_WINDOW_START; p++)
            if (want[p] != got[p]) {
                if (!bad) *firstBad = p;
                bad++;
            }
    }
    return bad;
}
extern "C" size_t sim_fast_lds_bytes(void) { return sizeof(sqz::FastLds); }
extern "C" size_t sim_workspace_bytes(size_t n) { return sqz::workspace_bytes((u32)n); }
extern "C" size_t sim_lds_bytes(void) { return sizeof(sqz::LdsState); }

// measurement: counters of the row-batch evaluation (sq_zstd_rowbatch.h g_sim_rb_stats); reset on read
extern "C" void sim_rb_stats(unsigned long long* out) {
    for (int k = 0; k < 5; k++) {
        out[k] = sqz::g_sim_rb_stats[k];
        sqz::g_sim_rb_stats[k] = 0;
    }
}
#ifdef SQ_SIM_TRIPS
// (tools/sim_trips.py) dependent trips to memory counted by the simulator since the last call: 8 load counts by phase, then 8 drain counts
extern "C" void sim_trip_counts(unsigned long long* out) {
    for (int i = 0; i < 16; i++) {
        out[i] = g_sim_trips[i];
        g_sim_trips[i] = 0;
    }
}
#endif
