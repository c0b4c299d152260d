/*
 * sq_oracle.h -- CPU ORACLE for the squishRS pack hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * library, and only as the checker.  The product (squishrs_amd/, libsquish_gpu.so) never links,
 * loads or executes anything under oracle/.
 *
 * What it restates (plain scalar C, one thread):
 *   - hash_chunk           /root/reference/src/util/chunk.rs:46-49   (xxh3_128 -> u128 LE bytes)
 *   - compress(chunk, 12)  /root/reference/src/util/chunk.rs:89-90   (one zstd frame per chunk)
 *   - ChunkStore::insert   /root/reference/src/util/chunk.rs:80-100  (exactly-once dedup)
 *   - chunk record / manifest / preamble bytes
 *                          /root/reference/src/fsutil/writer.rs:17-39,
 *                          /root/reference/src/archive/writer.rs:66-107,151-195,229-271,292-329,
 *                          /root/reference/src/util/header.rs:10-14,35-63,192-233
 *
 * The arithmetic of the first two lives in third-party crates that are NOT vendored under
 * /root/reference:  xxhash-rust 0.8.15 (Cargo.lock:1593-1596) and zstd 0.13.3 -> zstd-safe 7.2.4
 * -> zstd-sys 2.0.15+zstd.1.5.7 (Cargo.lock:1619-1640), i.e. C libzstd 1.5.7.  This file set
 * restates their published algorithms (XXH3 spec; RFC 8878 + libzstd 1.5.7 encoder decisions).
 *
 * Pinning: the reference's own tests hold no known-answer bytes for this path
 * (src/util/tests.rs:88-166 check only determinism / round trip).  The oracle is therefore pinned
 * against outputs of the real libraries run in the build container -- libzstd 1.5.7 (the same C
 * library zstd-sys wraps) and xxHash via python-xxhash -- see tools/gen_golden.py and
 * tests/golden/.  The script that made the vectors is committed beside them.
 */
#ifndef SQ_ORACLE_H
#define SQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQO_CHUNK_SIZE (2048u * 1024u) /* src/util/chunk.rs:11 */
#define SQO_LEVEL 12                   /* src/util/chunk.rs:12 */

/* chunk.rs:46-49 : out = xxh3_128(chunk).to_le_bytes() = low64 LE || high64 LE */
void sqo_hash_chunk(const uint8_t* chunk, size_t len, uint8_t out[16]);

/* ZSTD_compressBound */
size_t sqo_compress_bound(size_t n);

/* chunk.rs:89-90 : zstd::bulk::compress(chunk, 12).  Returns frame length, 0 on error
 * (cap too small / unsupported). */
size_t sqo_compress_chunk(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

/* Level-12 parameters after ZSTD_adjustCParams for a source of n bytes. */
typedef struct {
    uint32_t windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
} sqo_cparams;
sqo_cparams sqo_get_cparams(size_t n);

/* Debug/diff hooks: per-block sequences produced by the match finder, in the format of
 * ZSTD_generateSequences (offset, litLength, matchLength, rep) with a (0, lastLits, 0) delimiter
 * per block.  Returns number of entries written (<= cap). */
typedef struct {
    uint32_t offset, litLength, matchLength, rep;
} sqo_sequence;
size_t sqo_generate_sequences(const uint8_t* src, size_t n, sqo_sequence* out, size_t cap);

/* Minimal zstd frame decoder (RFC 8878; raw/RLE/compressed blocks, Huffman + FSE) used to
 * round-trip what both the oracle and the HIP path emit.  Returns decoded size or (size_t)-1. */
size_t sqo_decompress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);

/* ---- ChunkStore + archive bytes (serial, canonical "-j 1" order) ---- */
typedef struct sqo_store sqo_store;
sqo_store* sqo_store_new(void);
void sqo_store_free(sqo_store*);
/* chunk.rs:80-100.  Returns 1 if the chunk was new (and *out_len holds the frame length written
 * to dst), 0 if duplicate, -1 on error. */
int sqo_store_insert(sqo_store*, const uint8_t* chunk, size_t len, uint8_t hash[16], uint8_t* dst,
                     size_t cap, size_t* out_len);
uint64_t sqo_store_len(const sqo_store*);

/* Whole-archive restatement: files in walk order -> malloc'd .squish image (timestamp given). */
typedef struct {
    const char* rel_path;
    const uint8_t* data;
    uint64_t size;
} sqo_file;
size_t sqo_pack_archive(const sqo_file* files, uint32_t nfiles, uint64_t timestamp, uint8_t** out);
void sqo_free(void* p);

/* bench.py cpu_baseline: `threads` pthreads hash + compress chunks of `data` for `seconds` (bench_threads.c) */
uint64_t sqo_bench_insert_threads(const uint8_t* data, size_t n_chunks, size_t chunk_len, unsigned threads, double seconds);

/* bench.py cpu_baseline: the reference's whole pack pipeline on host threads (pool over files, sharded dedup map
 * with the lock held during compression, one writer thread); see bench_threads.c.  The caller measures wall time
 * around the call (it includes draining the writer).  stats: chunks written, bytes written, duplicates, max backlog. */
int sqo_bench_pipeline(const uint8_t* data, size_t n_chunks, size_t chunk_len, unsigned threads, double seconds,
                       uint64_t stats[4]);

/* the same over chunks of different lengths, reporting (frame length, XXH3-128 of the frame) per chunk of the first
 * pass; and a threaded decode-and-verify of frames (bench.py's parity checks of the configs[3] leg) */
int sqo_bench_pipeline2(const uint8_t* data, const uint64_t* off, const uint32_t* len, size_t n_chunks, unsigned threads,
                        double seconds, uint64_t stats[4], uint32_t* out_len, uint8_t* out_digest);
int64_t sqo_bench_decode_verify(const uint8_t* frames, const uint64_t* foff, const uint32_t* flen, const uint32_t* want_len,
                                const uint8_t* want_hash, size_t n, unsigned threads, double seconds, uint64_t* done);

#ifdef __cplusplus
}
#endif
#endif
