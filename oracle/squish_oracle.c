/*
 * squish_oracle.c -- CPU ORACLE (test infrastructure, see sq_oracle.h).
 * ChunkStore semantics and .squish v1.2 bytes in the canonical "-j 1" order:
 *   ChunkStore::insert      /root/reference/src/util/chunk.rs:80-100
 *   ChunkStore::len         /root/reference/src/util/chunk.rs:116-118
 *   writer_thread record    /root/reference/src/fsutil/writer.rs:17-39
 *   ArchiveWriter::new      /root/reference/src/archive/writer.rs:66-107 (preamble)
 *   process_file chunker    /root/reference/src/archive/writer.rs:229-271 (original_size quirk :255)
 *   pack / patch count      /root/reference/src/archive/writer.rs:151-195
 *   write_files_metadata    /root/reference/src/archive/writer.rs:292-329
 *   magic + timestamp       /root/reference/src/util/header.rs:10-14,35-63,192-233
 */
#include "sq_oracle.h"
#include <stdlib.h>
#include <string.h>

struct sqo_store {
    uint8_t* keys; /* cap * 16 */
    uint8_t* used;
    size_t cap, n;
};

sqo_store* sqo_store_new(void) {
    sqo_store* s = (sqo_store*)calloc(1, sizeof(*s));
    s->cap = 1024;
    s->keys = (uint8_t*)malloc(s->cap * 16);
    s->used = (uint8_t*)calloc(s->cap, 1);
    return s;
}
void sqo_store_free(sqo_store* s) {
    if (!s) return;
    free(s->keys);
    free(s->used);
    free(s);
}
static size_t slot_of(const uint8_t* key, size_t cap) {
    uint64_t v;
    memcpy(&v, key, 8);
    return (size_t)(v * 0x9E3779B97F4A7C15ULL >> 20) & (cap - 1);
}
/* returns 1 if inserted (vacant), 0 if already present (DashMap::entry, chunk.rs:83) */
static int set_insert(sqo_store* s, const uint8_t key[16]) {
    if ((s->n + 1) * 2 > s->cap) {
        size_t ocap = s->cap;
        uint8_t *ok = s->keys, *ou = s->used;
        s->cap *= 2;
        s->keys = (uint8_t*)malloc(s->cap * 16);
        s->used = (uint8_t*)calloc(s->cap, 1);
        for (size_t i = 0; i < ocap; i++)
            if (ou[i]) {
                size_t p = slot_of(ok + 16 * i, s->cap);
                while (s->used[p]) p = (p + 1) & (s->cap - 1);
                s->used[p] = 1;
                memcpy(s->keys + 16 * p, ok + 16 * i, 16);
            }
        free(ok);
        free(ou);
    }
    size_t p = slot_of(key, s->cap);
    while (s->used[p]) {
        if (!memcmp(s->keys + 16 * p, key, 16)) return 0;
        p = (p + 1) & (s->cap - 1);
    }
    s->used[p] = 1;
    memcpy(s->keys + 16 * p, key, 16);
    s->n++;
    return 1;
}

int sqo_store_insert(sqo_store* s, const uint8_t* chunk, size_t len, uint8_t hash[16], uint8_t* dst,
                     size_t cap, size_t* out_len) {
    sqo_hash_chunk(chunk, len, hash);           /* chunk.rs:81 */
    if (!set_insert(s, hash)) return 0;         /* Entry::Occupied -> compressed_data: None */
    size_t n = sqo_compress_chunk(chunk, len, dst, cap); /* Entry::Vacant -> compress (chunk.rs:89) */
    if (n == 0) return -1;                      /* AppError::Compression */
    *out_len = n;
    return 1;
}
uint64_t sqo_store_len(const sqo_store* s) { return (uint64_t)s->n; }

/* ---- whole-archive restatement ---- */
typedef struct {
    uint8_t* p;
    size_t n, cap;
} buf;
static void put(buf* b, const void* src, size_t n) {
    if (b->n + n > b->cap) {
        while (b->n + n > b->cap) b->cap = b->cap ? b->cap * 2 : 1 << 16;
        b->p = (uint8_t*)realloc(b->p, b->cap);
    }
    memcpy(b->p + b->n, src, n);
    b->n += n;
}
static void put_u64(buf* b, uint64_t v) { put(b, &v, 8); } /* little-endian host */
static void put_u32(buf* b, uint32_t v) { put(b, &v, 4); }

/* Packs files (already in walk order) into a malloc'd archive image; returns its size. */
size_t sqo_pack_archive(const sqo_file* files, uint32_t nfiles, uint64_t timestamp, uint8_t** out) {
    buf b = {0};
    buf meta = {0};
    sqo_store* store = sqo_store_new();
    size_t const cap = sqo_compress_bound(SQO_CHUNK_SIZE) + 64;
    uint8_t* frame = (uint8_t*)malloc(cap);
    put(&b, "squish", 6);  /* header.rs:10 */
    put(&b, "1.2.0", 5);   /* CARGO_PKG_VERSION, Cargo.toml:3 */
    put_u64(&b, timestamp); /* header.rs:55-63 */
    put_u64(&b, 0);         /* placeholder, writer.rs:83-84 */
    put_u32(&meta, nfiles); /* writer.rs:300-303 */
    for (uint32_t f = 0; f < nfiles; f++) {
        uint64_t const size = files[f].size;
        uint32_t const nchunks = (uint32_t)((size + SQO_CHUNK_SIZE - 1) / SQO_CHUNK_SIZE);
        uint32_t const plen = (uint32_t)strlen(files[f].rel_path);
        put_u32(&meta, plen);
        put(&meta, files[f].rel_path, plen);
        put_u64(&meta, size);
        put_u32(&meta, nchunks);
        for (uint32_t c = 0; c < nchunks; c++) { /* writer.rs:240-268 */
            uint64_t const off = (uint64_t)c * SQO_CHUNK_SIZE;
            size_t const len = (size_t)((size - off < SQO_CHUNK_SIZE) ? size - off : SQO_CHUNK_SIZE);
            uint8_t hash[16];
            size_t clen = 0;
            int r = sqo_store_insert(store, files[f].data + off, len, hash, frame, cap, &clen);
            if (r < 0) {
                free(b.p);
                b.p = NULL;
                b.n = 0;
                goto out;
            }
            if (r == 1) { /* fsutil/writer.rs:21-36 */
                put(&b, hash, 16);
                put_u64(&b, SQO_CHUNK_SIZE); /* original_size = chunk_buf.len(), writer.rs:255 */
                put_u64(&b, (uint64_t)clen);
                put(&b, frame, clen);
            }
            put(&meta, hash, 16);
        }
    }
    {
        uint64_t const count = sqo_store_len(store); /* writer.rs:177-184 */
        memcpy(b.p + 19, &count, 8);
    }
    put(&b, meta.p, meta.n);
out:
    free(meta.p);
    free(frame);
    sqo_store_free(store);
    *out = b.p;
    return b.n;
}
void sqo_free(void* p) { free(p); }
