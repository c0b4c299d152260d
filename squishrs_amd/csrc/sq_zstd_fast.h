// sq_zstd_fast.h -- the incompressible-chunk kernel of the zstd level-12 pack path.
//
// For a chunk on which zstd-12 (btlazy2: 16 KiB < len <= 256 KiB, /root/reference/src/util/chunk.rs:89-90
// -> libzstd 1.5.7) provably finds nothing, the frame is known without running the compressor:
//   * the no-match certificate (cert_compute_core, sq_zstd_match.h) proves that every search of the
//     lazy parser fails in every block: no sequences;
//   * a block without sequences goes to ZSTD_compressLiterals with "suspect uncompressible" set;
//     for srcSize >= 40960 HUF_compress first looks at the byte histograms of the block's first and
//     last 4 KiB: largest1 + largest2 <= (8192 >> 7) + 4 = 68 means "not compressible" -> raw
//     literals -> section + 1 > block -> ZSTD_noCompressBlock: the block is emitted raw
//     (and it is not an RLE block, since no byte fills 4 KiB);
// so the frame is header + raw blocks.  Everything else is left to k_compress_batch (frame length 0),
// together with the certificate ends already computed here.
//
// Why a separate kernel: k_compress_batch needs 248 VGPRs and 20 KB of LDS per wave (8 waves/CU)
// and is latency-bound at that occupancy; this path needs ~11 KB of LDS and half the registers, so
// twice as many waves hide the same memory latency.
#pragma once
#include "sq_zstd_match.h"

namespace sqz {

struct FastLds {
    u32 bloom[SQZ_BLOOM_WORDS]; // certificate Bloom filter; afterwards the 256-bin sample histograms
    u32 cand[256];
    u32 ovf[2 * SQZ_CERT_OVF_CAP];
    u32 lvl[2 * SQZ_SCHEDLVL_N]; // level words of the two schedules (copied once per wave)
};

#define SQZ_FAST_MIN_BLOCK (4096u * 10u) // HUF's sampled pre-check needs srcSize >= 10 samples
#define SQZ_CERT_HINT_VALID 0x80000000u   // flag in certOut[1]: the two words are a certificate

// ZSTD_writeFrameHeader for a single-segment frame with content size n; returns the header size.
SQ_DEV u32 fast_frame_header(u8* dst, u32 n) {
    u32 const fcsCode = (n >= 256) + (n >= 65536 + 256);
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
    return 5 + (fcsCode == 0 ? 1 : (fcsCode == 1 ? 2 : 4));
}

// 1 if HUF's sampled pre-check calls this block's literals uncompressible
SQ_DEVFN u32 fast_literals_uncompressible(const u8* blk, u32 size, u32* count /*256 words, LDS*/) {
    u32 l1, l2, dummy;
    wv_zero_lds(count, 256);
    wv_hist_bytes(count, blk, 4096);
    wv_max_and_last(count, 256, &l1, &dummy);
    wv_zero_lds(count, 256);
    wv_hist_bytes(count, blk + size - 4096, 4096);
    wv_max_and_last(count, 256, &l2, &dummy);
    return l1 + l2 <= ((2 * 4096) >> 7) + 4;
}

// Returns the frame length when the chunk is decided here, 0 when it is left to k_compress_batch.
// certOut[0..1]: the certificate ends for k_compress_batch (certOut[1] carries SQZ_CERT_HINT_VALID),
// or 0, 0 when no certificate was computed.
SQ_DEVFN u32 fastpath_frame_wave(const u8* src, u32 n, u8* dst, u32 cap, u32* E, u32 genBits, const u32* schedTab,
                                 FastLds* L, u32* certOut, u32 dbg = 0) {
    CParams const cp = get_cparams(n);
    u32 c0 = 0, c1 = 0;
    u32 frame = 0;
    if (cp.strategy == STRAT_BTLAZY2 && n >= 9) {
        u32 const len0 = n < SQZ_BLOCKSIZE_MAX ? n : SQZ_BLOCKSIZE_MAX;
        u32 const len1 = n - len0;
        CertCtx cx;
        cx.base = src - SQZ_WINDOW_START;
        cx.E = E;
        cx.eLog = cp.chainLog - 1;
        cx.genBits = genBits;
        cx.schedTab = schedTab;
        u32 const full0 = gld32(schedTab), full1 = gld32(schedTab + 1);
        cx.certCnt0 = (len0 == SQZ_BLOCKSIZE_MAX) ? full0 : sched_count(schedTab + 2, full0, len0 - 8);
        u32 const cnt1 = (len1 < 9) ? 0u : (len1 == SQZ_BLOCKSIZE_MAX) ? full1 : sched_count(schedTab + 2 + SQZ_SCHEDTAB_N, full1, len1 - 8);
        cx.nCert = cx.certCnt0 + cnt1;
        cx.base1Pos = SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX;
        cx.tabOff1 = 0;
        cx.repOff = 1;
        cx.chunkEndPos = n + SQZ_WINDOW_START;
        cx.bloom = (SQ_L u32*)L->bloom;
        cx.cand = (SQ_L u32*)L->cand;
        cx.ovf = (SQ_L u32*)L->ovf;
        cx.lvl = (SQ_L u32*)L->lvl;
        cx.certEnd = cx.certEnd1 = 0;
        if (cx.nCert > 0) {
            cx.dbg = dbg;
            cert_compute_core(cx);
            c0 = cx.certEnd;
            c1 = cx.certEnd1 | SQZ_CERT_HINT_VALID;
            // every search of every block certainly fails?
            bool whole = cx.certEnd >= SQZ_WINDOW_START + len0 - 8;
            if (len1) whole = whole && (cx.certEnd1 >= SQZ_WINDOW_START + SQZ_BLOCKSIZE_MAX + len1 - 8);
            bool const sized = len0 >= SQZ_FAST_MIN_BLOCK && (len1 == 0 || len1 >= SQZ_FAST_MIN_BLOCK);
            u32 const nBlocks = len1 ? 2u : 1u;
            if (dbg & 4) whole = false; // timing experiment: certificate only
            if (whole && sized && cap >= n + 9 + 3 * nBlocks) {
                u32 ok = fast_literals_uncompressible(src, len0, L->bloom);
                if (ok && len1) ok = fast_literals_uncompressible(src + len0, len1, L->bloom);
                if (ok) {
                    u32 const hdr = fast_frame_header(dst, n);
                    u8* op = dst + hdr;
                    for (u32 b = 0; b < nBlocks; b++) {
                        u32 const bs = b ? len1 : len0;
                        u32 const h = ((b + 1 == nBlocks) ? 1u : 0u) + (0u << 1) + (bs << 3); // ZSTD_noCompressBlock
                        WV_SERIAL {
                            op[0] = (u8)h;
                            op[1] = (u8)(h >> 8);
                            op[2] = (u8)(h >> 16);
                        }
                        wv_memcpy(op + 3, src + (b ? len0 : 0u), bs);
                        op += 3 + bs;
                    }
                    frame = (u32)(op - dst);
                }
            }
        }
    }
    WV_SERIAL {
        certOut[0] = c0;
        certOut[1] = c1;
    }
    wv_mem_sync();
    return frame;
}

#if SQ_ON_DEVICE
// Persistent waves, one certificate table each; every wave pulls chunk indices from *queue until
// exhausted (every wave reaches idx >= n).  frame_len[idx] = frame size, or 0 = not decided here.
__global__ void __launch_bounds__(64, 4)
k_incompressible_batch(const u8* data, const u64* off, const u32* len, const u8* select, u32 n, u8* slots,
                       const u64* slot_off, u32* frame_len, u32* cert_out, u8* etab_pool, u64 etab_bytes,
                       u32* queue, const u32* schedTab, u32 dbg, u32 persist) { // queue[1]: number of chunks decided here
    __shared__ FastLds lds;
    // a wave's slice: its table, then 256 bytes that keep its generation counter between launches
    u32* const E = (u32*)(etab_pool + (u64)blockIdx.x * (etab_bytes + 256));
    u32* const saved = (u32*)((u8*)E + etab_bytes);
    for (u32 i = wv_lane(); i < 2 * SQZ_SCHEDLVL_N; i += 64) lds.lvl[i] = schedTab[SQZ_SCHEDLVL_OFF + i];
    __builtin_amdgcn_s_waitcnt(0);
    // 0: the table holds unknown words; `persist`: the host vouches that the pool is as the previous
    // launch left it, so the saved counter is still true and the tags go on counting
    u32 gen = persist ? gld32(saved) : 0u;
    for (;;) {
        u32 idx = 0;
        if (wv_lane() == 0) idx = atomicAdd(queue, 1u);
        idx = wv_uni(idx);
        if (idx >= n) break;
        u32 r = 0;
        if (select == nullptr || select[idx] != 0) {
            if (gen == 0 || gen >= SQZ_HT_GEN_MAX) {
                wv_fill32(E, 0, etab_bytes / 4);
                gen = 0;
            }
            gen++;
            u32 const cap = (u32)(slot_off[idx + 1] - slot_off[idx]);
            wv_mem_sync_inv(); // no L1 line from the previous chunk's table / data survives
            r = fastpath_frame_wave(data + off[idx], len[idx], slots + slot_off[idx], cap, E, gen << SQZ_HT_GEN_SHIFT,
                                    schedTab, &lds, cert_out + 2 * (size_t)idx, dbg);
        } else if (wv_lane() == 0) {
            cert_out[2 * (size_t)idx] = 0;
            cert_out[2 * (size_t)idx + 1] = 0;
        }
        if (wv_lane() == 0) {
            frame_len[idx] = r;
            if (r) atomicAdd(queue + 1, 1u);
        }
    }
    if (wv_lane() == 0) saved[0] = gen;
}
#endif

} // namespace sqz
