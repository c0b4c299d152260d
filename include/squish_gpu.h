/*
 * squish_gpu.h -- C ABI of libsquish_gpu.so: the MI355X (gfx950) implementation of squishRS's
 * pack-side hot path (per-chunk XXH3-128 content hash, global dedup set, zstd level-12 frame
 * compressor).  Plain pointers and sizes only; no C++/torch types; never throws or aborts across
 * the boundary.  Every entry point returns 0 (SQ_OK) or a negative sq_err.
 *
 * The reference (SamB032/squishRS v1.2.0) has no FFI for this path; the seam these symbols replace
 * is the Rust API of /root/reference/src/util/chunk.rs, called from
 * /root/reference/src/archive/writer.rs:249 on Rayon workers.  INTEGRATION.md shows the
 * `extern "C"` block a maintainer would add to src/util/chunk.rs to bind them.
 *
 * Pointer arguments documented "host or device" are classified with hipPointerGetAttributes:
 * device pointers are used in place (nothing crosses PCIe); host pointers are staged through the
 * context's device buffers.
 */
#ifndef SQUISH_GPU_H
#define SQUISH_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_ABI_VERSION 1
#define SQ_CHUNK_SIZE (2048u * 1024u) /* CHUNK_SIZE, src/util/chunk.rs:11 */
#define SQ_LEVEL 12                   /* COMPRESSION_LEVEL, src/util/chunk.rs:12 */
#define SQ_HASH_BYTES 16              /* ChunkHash = [u8; 16], src/util/chunk.rs:9 */

typedef enum sq_err {
    SQ_OK = 0,
    SQ_E_INVAL = -1,    /* bad argument */
    SQ_E_NOMEM = -2,    /* host or device allocation failed */
    SQ_E_CAP = -3,      /* output capacity too small */
    SQ_E_DEVICE = -4,   /* HIP runtime error / no gfx950 device / kernel fault */
    SQ_E_COMPRESS = -5, /* == AppError::Compression (src/util/chunk.rs:90, errors.rs:25-26) */
    SQ_E_NODEVICE = -6, /* library built without a usable GPU at run time: fails loudly, no CPU path */
    SQ_E_DECOMPRESS = -7 /* sq_decompress_batch: at least one frame did not decode (per-frame status in h_status); the reference
                            reports zstd's message through AppError::ReaderError (src/archive/reader.rs:302-303) */
} sq_err;

typedef struct sq_ctx sq_ctx;

typedef struct sq_config {
    uint32_t abi_version;      /* SQ_ABI_VERSION */
    int32_t device;            /* HIP device ordinal */
    uint32_t max_chunk;        /* largest chunk length in bytes (<= SQ_CHUNK_SIZE; 0 = SQ_CHUNK_SIZE) */
    uint32_t max_batch_chunks; /* chunks per sq_pack_batch call (0 = 4096) */
    uint64_t max_batch_bytes;  /* input bytes per call when staged from host (0 = 1 GiB) */
    uint64_t dedup_capacity;   /* distinct chunks the dedup set can hold (0 = 1<<22) */
    uint32_t compress_waves;   /* persistent compressor wavefronts (0 = auto) */
    uint32_t flags;            /* SQ_CFG_*; 0 = defaults */
} sq_config;
/* How sq_pack_batch / sq_pack_submit order the hash and the first compress stage of a batch (same results either way):
 * by default chosen per batch from the duplicate ratio of the batch before -- few duplicates: "fused", the first
 * compress stage runs before the dedup decision and computes the hash of every chunk it streams (one read of the chunk
 * instead of two); many duplicates: "split", hash -> dedup -> compress only the new chunks. */
#define SQ_CFG_NEVER_FUSE_HASH 1u
#define SQ_CFG_ALWAYS_FUSE_HASH 2u

/* Lifecycle.  One ctx per driver thread (a ctx is not re-entrant); the dedup set inside is the
 * counterpart of ChunkStore.primary_store (src/util/chunk.rs:20-24,52-56). */
int sq_ctx_create(const sq_config* cfg, sq_ctx** out);
void sq_ctx_destroy(sq_ctx* ctx);
/* Run all subsequent work of this ctx on an existing hipStream_t (e.g. torch's current stream). */
int sq_ctx_set_stream(sq_ctx* ctx, void* hip_stream);
int sq_ctx_synchronize(sq_ctx* ctx);
/* Gives the context's cached device buffers back to the device (the compressor's workspace pools -- up to 65 GB per lane for
 * 2 MiB chunks --, staging and slot buffers); the dedup set and everything observable stay.  They are allocated again on
 * demand; the next compress stage clears its tables once.  SQ_E_INVAL while batches of sq_pack_submit are in flight. */
int sq_ctx_trim(sq_ctx* ctx);
/* Device memory for a driver that keeps its batches on the device (the sq_shard_* calls take device pointers only) without
 * linking the HIP runtime itself: allocation on the context's device (NULL on failure), release, and a synchronous copy
 * (to_device != 0: host -> device, else device -> host).  `squishrs pack --gpus N` (host/squishrs_main.cpp) is built on them. */
void* sq_dev_alloc(sq_ctx* ctx, uint64_t bytes);
void sq_dev_free(sq_ctx* ctx, void* p);
int sq_dev_copy(sq_ctx* ctx, void* dst, const void* src, uint64_t bytes, int to_device);

/* == hash_chunk (src/util/chunk.rs:46-49): out = xxh3_128(chunk).to_le_bytes().
 * Stateless convenience form (host pointers, runs one tiny launch on device 0).  Thread-safe: the
 * reference calls hash_chunk / insert from up to 25 Rayon workers (src/util/chunk.rs:80,
 * src/cmd/mod.rs:16); sq_hash_chunk and sq_compress_chunk share one internal context and are
 * serialised by a mutex (safe from any thread, not concurrent -- use one sq_ctx per thread and
 * the batch forms for throughput). */
int sq_hash_chunk(const uint8_t* chunk, size_t len, uint8_t out[SQ_HASH_BYTES]);

/* == ZSTD_compressBound: capacity that always suffices for one frame of n input bytes. */
size_t sq_compress_bound(size_t n);

/* == zstd::bulk::compress(chunk, 12) (src/util/chunk.rs:89-90): one complete zstd frame.
 * Stateless convenience form (host pointers). */
int sq_compress_chunk(const uint8_t* chunk, size_t len, uint8_t* dst, size_t cap, size_t* out_len);

/* == n x ChunkStore::insert applied in index order (src/util/chunk.rs:80-100): chunk i is
 * data[off[i] .. off[i]+len[i]) (chunks may be padded apart; 16-byte aligned starts are fastest).
 * For every chunk hashes[16*i..] receives its hash; is_new[i] = 1 iff
 * this is the first occurrence of that hash in index order across all batches of this ctx
 * (Entry::Vacant), and only those chunks are compressed: frame i occupies
 * comp[comp_off[i] .. comp_off[i+1]) (empty for duplicates).  comp_cap bytes must be available at
 * comp; SQ_E_CAP if the frames do not fit.  Whenever the call fails, nothing stays committed: the
 * dedup set, sq_unique_count and the global chunk index are exactly as before the call, so the
 * same batch can be retried (e.g. with a larger comp buffer) and reports the same is_new flags.
 * data / hashes / is_new / comp: host or device, ALL on the same side (SQ_E_INVAL otherwise).
 * off, len (n entries each) and comp_off (n+1 entries): host. */
int sq_pack_batch(sq_ctx* ctx, const uint8_t* data, const uint64_t* off, const uint32_t* len,
                  uint32_t n, uint8_t* hashes, uint8_t* is_new, uint8_t* comp, uint64_t comp_cap,
                  uint64_t* comp_off);

/* == ChunkStore::len (src/util/chunk.rs:116-118). */
uint64_t sq_unique_count(const sq_ctx* ctx);

/* == ChunkStore::new on an existing ctx (src/util/chunk.rs:52-56): empties the dedup set, keeping
 * every device buffer.  Asynchronous on the ctx stream. */
int sq_dedup_reset(sq_ctx* ctx);

/* ---- stage-level entry points (same kernels sq_pack_batch chains; used by the multi-GPU
 * driver, the parity tests and bench.py).  d_* are DEVICE pointers, h_* HOST pointers. */

/* hash stage: d_hashes[16*i..] = hash_chunk(chunk i). */
int sq_hash_batch(sq_ctx* ctx, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len,
                  uint32_t n, uint8_t* d_hashes);

/* dedup stage over an already gathered, globally ordered hash list of n_total entries (multi-GPU:
 * the RCCL all-gather result; single GPU: the batch itself).  Entries [my_first, my_first+my_n)
 * belong to this rank; d_is_new[my_n] receives 1 where the entry is the first occurrence of its
 * hash in global index order and was not seen in any earlier call.  All n_total hashes are
 * committed to the ctx's set, so every rank ends with the same set and unique count. */
int sq_dedup_global(sq_ctx* ctx, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first,
                    uint32_t my_n, uint8_t* d_is_new);

/* compress stage: frames for chunks with d_select[i] != 0 (NULL = all), compacted into d_comp in
 * index order; h_comp_off (host, n+1 entries) receives the offsets.  Returns SQ_E_CAP if
 * comp_cap is too small. */
int sq_compress_batch(sq_ctx* ctx, const uint8_t* d_data, const uint64_t* h_off,
                      const uint32_t* h_len, uint32_t n, const uint8_t* d_select, uint8_t* d_comp,
                      uint64_t comp_cap, uint64_t* h_comp_off);

/* compress stage without the packing pass: every frame stays in the slot it was produced in.
 * Frame i = d_comp[h_comp_off[i] .. h_comp_off[i] + h_comp_len[i]); h_comp_len[i] == 0 for chunks
 * that were not selected.  Slots are disjoint, ascending and 16-byte aligned; slot i holds
 * sq_compress_bound(h_len[i]) + 32 bytes rounded up to 16, so d_comp needs sq_slots_bound(h_len, n)
 * bytes (SQ_E_CAP otherwise).  The reference hands each frame to its writer as a separate buffer
 * (InsertReturn.compressed_data, src/util/chunk.rs:60-64; writer.rs:255-262), which interleaves
 * them with 32-byte chunk-record headers, so nothing downstream needs them back to back; this saves
 * one read and one write of every frame. */
int sq_compress_batch_slots(sq_ctx* ctx, const uint8_t* d_data, const uint64_t* h_off,
                            const uint32_t* h_len, uint32_t n, const uint8_t* d_select,
                            uint8_t* d_comp, uint64_t comp_cap, uint64_t* h_comp_off,
                            uint32_t* h_comp_len);
uint64_t sq_slots_bound(const uint32_t* h_len, uint32_t n);

/* ---- pipelined form: hash -> dedup -> compress of one batch is ENQUEUED and the call returns; up to two batches may be in flight on one context (its two internal "lanes": own stream, events,
 * per-batch buffers and compressor table pools; the dedup set is shared and its updates stay in batch order).  The
 * thin tail of one batch's compress stage -- a few long chunks -- then runs under the next batch's hash and first
 * compress stage.  This is the shape of the reference's own overlap: Rayon workers keep hashing / compressing the
 * next files while the writer drains earlier results (src/archive/writer.rs:153-170, src/fsutil/writer.rs:17-39).
 *   sq_pack_submit: == sq_pack_batch's semantics for the batch (n x ChunkStore::insert in index order), with the
 *     frames left in their slots of d_comp (see sq_compress_batch_slots; SQ_E_CAP unless comp_cap >=
 *     sq_slots_bound(h_len, n)).  flags bit 0: empty the dedup set first (== sq_dedup_reset, ordered correctly
 *     behind the batch before).  Every in-flight batch needs its OWN hashes / is_new / comp buffers.
 *     SQ_E_INVAL when two batches are already in flight.  Failure: every allocation, capacity and argument check of
 *     the batch (SQ_E_NOMEM, SQ_E_CAP, SQ_E_INVAL) runs BEFORE the shared dedup set is touched -- such a failure commits
 *     nothing (not even the reset of flags bit 0) and the batch may be submitted again; SQ_E_DEVICE (a HIP call failed
 *     after the commit) leaves the batch's hashes in the set: call sq_dedup_reset before reusing the context.
 *     data / hashes / is_new / comp may also be HOST pointers (all four on the same side): the batch is then staged
 *     through the lane's device buffers -- one H2D copy on the lane's stream, which overlaps the other lane's
 *     kernels -- the frames are packed back to back on the device and sq_pack_wait copies hashes, flags and
 *     exactly the frames' bytes back (comp_cap bytes must be available at comp; SQ_E_CAP from sq_pack_wait
 *     otherwise).  Page-locked buffers (sq_host_alloc) make the copies asynchronous.
 *   sq_pack_wait: waits for the OLDEST batch in flight and returns its frames' offsets in comp and their lengths (n
 *     entries each; length 0 = duplicate).  If it fails (SQ_E_COMPRESS / SQ_E_DEVICE) the context's dedup set holds that
 *     batch's hashes: call sq_dedup_reset before reusing the context.
 * The synchronous entry points above return SQ_E_INVAL while batches are in flight. */
int sq_pack_submit(sq_ctx* ctx, const uint8_t* data, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                   uint8_t* hashes, uint8_t* is_new, uint8_t* comp, uint64_t comp_cap, uint32_t flags);
int sq_pack_wait(sq_ctx* ctx, uint64_t* h_comp_off, uint32_t* h_comp_len);

/* ---- sharded pipelined form (one process per GPU): sq_pack_submit cut in two where the ranks exchange
 * their hashes, so that with two batches in flight the exchange of batch k+1 -- the caller's collective, e.g. one RCCL
 * all-gather -- happens while batch k compresses.  Device pointers only.
 *   sq_shard_submit_hash: enqueue the hash of the batch (this rank's n chunks) on the next free lane; d_hashes = n x 16.
 *   sq_shard_wait_hash:   block the host until those hashes are in d_hashes (the collective may start).
 *   sq_shard_submit_rest: given the hashes of ALL ranks in global chunk-index order (this rank's at my_first .. my_first
 *     + n), enqueue the global dedup (== sq_dedup_global: the set is replicated on every rank and every rank resolves
 *     "first occurrence in global index order" identically) and the compression of this rank's new chunks into slots
 *     (== sq_compress_batch_slots with select = is_new) behind the hash; flags bit 0 as for sq_pack_submit.  The batch
 *     is then in flight exactly as after sq_pack_submit and is collected by sq_pack_wait.  d_all_hashes, d_is_new and
 *     d_comp must stay valid until then.  The reference's counterpart of the exchange is the one DashMap all Rayon
 *     workers share (src/util/chunk.rs:52-56,83).
 * SQ_E_INVAL: no lane free, or the calls out of order.  A sq_shard_submit_rest[_idx] that finds a pending hash step
 * consumes it WHATEVER it returns -- every failure, SQ_E_CAP and SQ_E_INVAL for a bad argument included: start the
 * batch again with sq_shard_submit_hash (a second sq_shard_submit_rest returns SQ_E_INVAL: nothing is pending).  SQ_E_CAP (dedup capacity or comp_cap too small), SQ_E_NOMEM and
 * SQ_E_INVAL are detected before anything is enqueued and commit nothing -- neither the hashes nor the reset of flags
 * bit 0 -- so this rank's replicated set stays equal to the other ranks'; SQ_E_DEVICE (a HIP call failed after the
 * commit) leaves the set in an undefined relation to theirs: every rank must sq_dedup_reset. */
int sq_shard_submit_hash(sq_ctx* ctx, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                         uint8_t* d_hashes);
int sq_shard_wait_hash(sq_ctx* ctx);
int sq_shard_submit_rest(sq_ctx* ctx, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first, uint8_t* d_is_new,
                         uint8_t* d_comp, uint64_t comp_cap, uint32_t flags);
/* The same for ANY assignment of the chunks of the global order to ranks (block-cyclic stripes spread the first occurrences,
 * i.e. the compression work, evenly -- with contiguous ranges the first rank holds first occurrences only): this rank's
 * chunk i is entry d_my_idx[i] (device, n x u32, ascending) of d_all_hashes, which holds all ranks' hashes in GLOBAL
 * chunk-index order (the caller permutes the rank-major result of its all-gather); d_my_idx may be NULL when this rank
 * holds no chunk of the batch (n == 0 at sq_shard_submit_hash).
 *
 * Environment switches of the library (SQ_DEBUG, SQ_FUSE_HASH, SQ_NO_FASTPATH, SQ_FAST_DBG, SQ_STAGE*_WAVES_PER_CU ...:
 * experiment / bring-up switches, none needed in production) are read ONCE PER PROCESS, by the first call that creates
 * a context (thread-safe); changing the environment afterwards has no effect on that process. */
int sq_shard_submit_rest_idx(sq_ctx* ctx, const uint8_t* d_all_hashes, uint64_t n_total, const uint32_t* d_my_idx, uint8_t* d_is_new,
                             uint8_t* d_comp, uint64_t comp_cap, uint32_t flags);

/* ---- the unpack side: GPU zstd frame decoder -------------------------------------------------------
 * == n x `zstd::bulk::decompress(&compressed_data, orig_size)`, the loop of ArchiveReader::read_chunks
 * (/root/reference/src/archive/reader.rs:259-314; zstd-sys 2.0.15 = libzstd 1.5.7).  Record i's compressed bytes
 * frames[h_off[i] .. h_off[i] + h_len[i]) -- one or more zstd frames, skippable frames allowed, no dictionary -- are decoded
 * into out[h_out_off[i] .. h_out_off[i] + h_out_cap[i]) (h_out_cap[i] = the record's original_size, the capacity the
 * reference hands to zstd).  h_out_len[i] = bytes produced (0 when the record failed), h_status[i] = 0 or an SQ_DZ_* code whose
 * text (sq_decompress_strerror) is libzstd's for the same condition.  frames / out: host or device pointers, both on the same
 * side (SQ_E_INVAL otherwise); the h_* arrays are host memory.  One wavefront per record; a record may be up to 4 GiB - 1.
 * Returns SQ_OK when every record decoded, SQ_E_DECOMPRESS when some did not (the others are complete and valid), or
 * SQ_E_INVAL / SQ_E_NOMEM / SQ_E_DEVICE.  Synchronous; not to be called while batches of sq_pack_submit are in flight. */
enum sq_dz_status {
    SQ_DZ_OK = 0,
    SQ_DZ_CORRUPT = 1,     /* "Data corruption detected" */
    SQ_DZ_SRCSIZE = 2,     /* "Src size is incorrect" */
    SQ_DZ_UNKNOWN = 3,     /* "Unknown frame descriptor" */
    SQ_DZ_UNSUPPORTED = 4, /* "Unsupported frame parameter" */
    SQ_DZ_DICT = 5,        /* "Dictionary mismatch" */
    SQ_DZ_DSTSIZE = 6,     /* "Destination buffer is too small" */
    SQ_DZ_CHECKSUM = 7     /* "Restored data doesn't match checksum" */
};
int sq_decompress_batch(sq_ctx* ctx, const uint8_t* frames, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                        uint8_t* out, const uint64_t* h_out_off, const uint32_t* h_out_cap, uint32_t* h_out_len,
                        uint32_t* h_status);
const char* sq_decompress_strerror(uint32_t status);

/* Page-locked host buffers for the batch forms (hipHostMalloc).  The reference's packer reads every
 * chunk into a heap buffer (BufReader + chunk_buf, src/archive/writer.rs:233-246) and hands the frames on
 * as Vec<u8> (chunk.rs:60-64); a host driver of this library reads into / writes from buffers it gets
 * here instead, so that the H2D / D2H copies of sq_pack_batch run at DMA speed.  sq_pack_batch copies a
 * host batch with ONE transfer when the chunks lie back to back with their starts padded to 256 bytes
 * (off[i+1] == off[i] + ((len[i] + 255) & ~255)).  Returns NULL when no device is present or the
 * allocation fails (callers may fall back to ordinary memory). */
void* sq_host_alloc(size_t bytes);
void sq_host_free(void* p);

/* HBM read-bandwidth calibration (a trivial 16 B/lane streaming read over d_buf): returns the
 * kernel time in ms through *ms.  bench.py reports it beside the 8 TB/s nominal peak. */
int sq_calibrate_read_bw(sq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, float* ms);

/* Random-access calibration (16-byte loads at pseudo-random places of d_buf, 8 in flight per lane, 8 waves of 256
 * lanes... per CU): kernel time in ms and the number of loads issued.  The compressor's candidate evaluation on
 * compressible data is bound by this rate, not by streaming bandwidth; bench.py reports it beside configs[3]. */
int sq_calibrate_gather(sq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint32_t per_lane, float* ms, uint64_t* requests);
/* The same with the shape of the requests chosen: req_bytes 16 / 32 / 64 per request, depth 1 / 2 / 4 / 8 / 16 requests in flight
 * per lane, waves_per_cu 4 / 8 / 16 / 32 resident waves (0 = 16 bytes, 8 in flight, 32 waves).  tools/gather_sweep.py. */
int sq_calibrate_gather2(sq_ctx* ctx, const uint8_t* d_buf, uint64_t nbytes, uint32_t per_lane, uint32_t req_bytes, uint32_t depth,
                         uint32_t waves_per_cu, float* ms, uint64_t* requests);

/* Timing of the most recent stage launches of this ctx, in milliseconds, measured with HIP
 * events on the ctx stream (0 if the stage did not run). */
typedef struct sq_timings {
    float hash_ms, dedup_ms, compress_ms, compact_ms;
    float fast_ms; /* incompressible-chunk kernel (stage 1 of the compress stage); compress_ms = stage 2 */
    uint32_t fast_decided; /* chunks whose frame stage 1 produced in the most recent compress stage */
    uint32_t fused;        /* 1: the batch took the fused form (stage 1 before the dedup decision, computing the hashes'
                              block sums; hash_ms is then the kernel that finishes them), 0: the split form */
} sq_timings;
int sq_last_timings(sq_ctx* ctx, sq_timings* out);

/* Test / measurement hook: indices (within the most recent compress launch of this ctx) of the chunks that its
 * first stage (the incompressible-chunk kernel) left to the compressor proper, heavy ones first; at most cap
 * entries, *n_out = entries written (0 when that stage did not run).  bench.py draws its oracle sample from it. */
int sq_debug_stage2_list(sq_ctx* ctx, uint32_t* h_out, uint32_t cap, uint32_t* n_out);

/* Test / measurement hook: *loaded = the second build of the compressor kernel (squish_s2_w3.co beside the library: 3 waves per
 * SIMD, 11 per CU) is available; *used_last = the most recent compress stage ran it (chosen per batch: the batch before it left
 * more than a quarter of its chunks to stage 2). */
int sq_debug_stage2_kernel(sq_ctx* ctx, uint32_t* loaded, uint32_t* used_last);

const char* sq_strerror(int err);
uint32_t sq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
