/* bench_threads.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; see sq_oracle.h).
 *
 * The reference packs with a Rayon pool: one chunk per task, each task = ChunkStore::insert
 * (/root/reference/src/util/chunk.rs:80-100: hash, then compress when the hash is new;
 * /root/reference/src/archive/writer.rs:151-195 fans the files out).  bench.py's `cpu_baseline` leg times
 * that shape on the GPU box's host cores with plain pthreads (no Python in the timed loop): thread t
 * cycles through chunks t, t + T, t + 2T, ... of the buffer it is given, hashing and compressing each,
 * until the time is up. */
#include "sq_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    const uint8_t* data;
    size_t n_chunks, chunk_len;
    unsigned t, T;
    double seconds;
    uint64_t done;
} bt_arg;

static double bt_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* bt_worker(void* p) {
    bt_arg* a = (bt_arg*)p;
    size_t const cap = sqo_compress_bound(a->chunk_len) + 64;
    uint8_t* dst = (uint8_t*)malloc(cap);
    uint8_t h[16];
    double const end = bt_now() + a->seconds;
    size_t i = a->t % a->n_chunks;
    a->done = 0;
    if (!dst) return NULL;
    while (bt_now() < end) {
        const uint8_t* c = a->data + i * a->chunk_len;
        sqo_hash_chunk(c, a->chunk_len, h);
        if (sqo_compress_chunk(c, a->chunk_len, dst, cap) == 0) break;
        a->done++;
        i += a->T;
        if (i >= a->n_chunks) i = a->t % a->n_chunks;
    }
    free(dst);
    return NULL;
}

/* returns the number of chunks hashed + compressed by `threads` threads in `seconds` of wall time */
uint64_t sqo_bench_insert_threads(const uint8_t* data, size_t n_chunks, size_t chunk_len, unsigned threads, double seconds) {
    if (!data || !n_chunks || !threads) return 0;
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    bt_arg* args = (bt_arg*)calloc(threads, sizeof(bt_arg));
    uint64_t total = 0;
    if (!th || !args) {
        free(th);
        free(args);
        return 0;
    }
    for (unsigned t = 0; t < threads; t++) {
        args[t] = (bt_arg){data, n_chunks, chunk_len, t, threads, seconds, 0};
        if (pthread_create(&th[t], NULL, bt_worker, &args[t]) != 0) {
            threads = t;
            break;
        }
    }
    for (unsigned t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        total += args[t].done;
    }
    free(th);
    free(args);
    return total;
}

/* ---------------------------------------------------------------------------------------------------------
 * The reference's WHOLE pack pipeline on host threads: a pool of `threads` workers over
 * the files (/root/reference/src/archive/writer.rs:153-165, Rayon par_iter; here: an atomic file counter), each
 * file = one chunk of chunk_len bytes -> ChunkStore::insert (/root/reference/src/util/chunk.rs:80-100): hash,
 * then the entry of a SHARDED map (DashMap: the shard's lock is held while the new chunk is compressed,
 * chunk.rs:83-95), new frames go through a channel to ONE writer thread (/root/reference/src/fsutil/writer.rs:17-39)
 * that serialises `hash || original_size || compressed_size || frame` into the archive (here: a memory sink that
 * wraps around, standing in for a page-cache file).  The buffer of n_chunks chunks is cycled until the time is up;
 * the pass number is mixed into the dedup key so that later passes count as new chunks (same work per chunk).
 * stats[0] = chunks written, [1] = bytes written, [2] = duplicates seen, [3] = longest channel backlog. */
#define BP_SHARDS 64u
typedef struct bp_msg {
    struct bp_msg* next;
    uint8_t hash[16];
    uint64_t orig, clen;
    uint8_t* frame;
} bp_msg;
typedef struct {
    pthread_mutex_t mu;
    uint64_t* keys; /* 2 x u64 per slot, 0/0 = empty */
    size_t cap, used;
} bp_shard;
typedef struct {
    const uint8_t* data;
    size_t n_chunks, chunk_len;
    const uint64_t* off; /* ragged form (sqo_bench_pipeline2): chunk i = data + off[i], len[i]; NULL = uniform chunk_len */
    const uint32_t* len;
    uint32_t* out_len;    /* per chunk of the FIRST pass over the buffer: its frame's length ... */
    uint8_t* out_digest;  /* ... and XXH3-128 (16 bytes): what bench.py compares with the GPU's frames */
    double t_end;
    uint64_t next_file; /* atomic */
    bp_shard shards[BP_SHARDS];
    pthread_mutex_t qmu;
    pthread_cond_t qcv;
    bp_msg *qhead, *qtail;
    int closed;
    uint64_t backlog, max_backlog, dups;
    uint8_t* sink;
    size_t sink_cap, sink_pos;
    uint64_t written, bytes;
} bp_state;

static int bp_shard_insert(bp_shard* s, uint64_t k0, uint64_t k1) { /* 1 = vacant (inserted), 0 = occupied */
    if ((s->used + 1) * 2 > s->cap) {
        size_t ncap = s->cap ? s->cap * 2 : 1024;
        uint64_t* nk = (uint64_t*)calloc(ncap * 2, 8);
        if (!nk) return -1;
        for (size_t i = 0; i < s->cap; i++)
            if (s->keys[2 * i] | s->keys[2 * i + 1]) {
                size_t j = (size_t)(s->keys[2 * i + 1] & (ncap - 1));
                while (nk[2 * j] | nk[2 * j + 1]) j = (j + 1) & (ncap - 1);
                nk[2 * j] = s->keys[2 * i];
                nk[2 * j + 1] = s->keys[2 * i + 1];
            }
        free(s->keys);
        s->keys = nk;
        s->cap = ncap;
    }
    size_t j = (size_t)(k1 & (s->cap - 1));
    while (s->keys[2 * j] | s->keys[2 * j + 1]) {
        if (s->keys[2 * j] == k0 && s->keys[2 * j + 1] == k1) return 0;
        j = (j + 1) & (s->cap - 1);
    }
    s->keys[2 * j] = k0;
    s->keys[2 * j + 1] = k1;
    s->used++;
    return 1;
}

static void* bp_worker(void* p) {
    bp_state* st = (bp_state*)p;
    size_t const cap = sqo_compress_bound(st->chunk_len) + 64; /* (chunk_len = the longest chunk) */
    while (bt_now() < st->t_end) {
        uint64_t const f = __atomic_fetch_add(&st->next_file, 1, __ATOMIC_RELAXED);
        size_t const ci = (size_t)(f % st->n_chunks);
        const uint8_t* c = st->off ? st->data + st->off[ci] : st->data + ci * st->chunk_len;
        size_t const clen = st->len ? st->len[ci] : st->chunk_len;
        uint64_t const pass = f / st->n_chunks;
        uint8_t h[16];
        uint64_t k0, k1;
        sqo_hash_chunk(c, clen, h);
        memcpy(&k0, h, 8);
        memcpy(&k1, h + 8, 8);
        k0 ^= pass * 0x9E3779B97F4A7C15ull; /* a later pass over the buffer = other files */
        if (!(k0 | k1)) k0 = 1;
        bp_shard* s = &st->shards[(k0 >> 7) % BP_SHARDS];
        pthread_mutex_lock(&s->mu);
        int const vac = bp_shard_insert(s, k0, k1);
        bp_msg* m = NULL;
        if (vac == 1) { /* Entry::Vacant: compress under the shard lock, as DashMap's entry API does */
            m = (bp_msg*)malloc(sizeof(bp_msg));
            uint8_t* fr = (uint8_t*)malloc(cap);
            if (m && fr) {
                m->clen = sqo_compress_chunk(c, clen, fr, cap);
                m->frame = fr;
                if (pass == 0 && st->out_len) { /* (outside the reference's work: a 16-byte hash of the frame) */
                    st->out_len[ci] = (uint32_t)m->clen;
                    sqo_hash_chunk(fr, (size_t)m->clen, st->out_digest + 16 * ci);
                }
                m->orig = SQO_CHUNK_SIZE; /* writer.rs:255: always the buffer size */
                memcpy(m->hash, h, 16);
                m->next = NULL;
            }
        }
        pthread_mutex_unlock(&s->mu);
        if (vac == 1 && m) {
            pthread_mutex_lock(&st->qmu);
            if (st->qtail)
                st->qtail->next = m;
            else
                st->qhead = m;
            st->qtail = m;
            if (++st->backlog > st->max_backlog) st->max_backlog = st->backlog;
            pthread_cond_signal(&st->qcv);
            pthread_mutex_unlock(&st->qmu);
        } else if (vac == 0)
            __atomic_fetch_add(&st->dups, 1, __ATOMIC_RELAXED);
    }
    return NULL;
}

static void bp_sink(bp_state* st, const void* p, size_t n) {
    if (st->sink_pos + n > st->sink_cap) st->sink_pos = 0;
    if (n <= st->sink_cap) memcpy(st->sink + st->sink_pos, p, n);
    st->sink_pos += n;
    st->bytes += n;
}

static void* bp_writer(void* p) {
    bp_state* st = (bp_state*)p;
    for (;;) {
        pthread_mutex_lock(&st->qmu);
        while (!st->qhead && !st->closed) pthread_cond_wait(&st->qcv, &st->qmu);
        bp_msg* m = st->qhead;
        if (m) {
            st->qhead = m->next;
            if (!st->qhead) st->qtail = NULL;
            st->backlog--;
        }
        pthread_mutex_unlock(&st->qmu);
        if (!m) break;
        uint8_t hdr[32];
        memcpy(hdr, m->hash, 16);
        memcpy(hdr + 16, &m->orig, 8);
        memcpy(hdr + 24, &m->clen, 8);
        bp_sink(st, hdr, 32);
        bp_sink(st, m->frame, (size_t)m->clen);
        st->written++;
        free(m->frame);
        free(m);
    }
    return NULL;
}

static int bp_run(const uint8_t* data, const uint64_t* off, const uint32_t* len, size_t n_chunks, size_t chunk_len, unsigned threads,
                  double seconds, uint64_t stats[4], uint32_t* out_len, uint8_t* out_digest);

int sqo_bench_pipeline(const uint8_t* data, size_t n_chunks, size_t chunk_len, unsigned threads, double seconds,
                       uint64_t stats[4]) {
    return bp_run(data, NULL, NULL, n_chunks, chunk_len, threads, seconds, stats, NULL, NULL);
}

/* The same pipeline over chunks of different lengths (chunk i = data + off[i], len[i] bytes: the 2 MiB chunks of
 * large files and their ragged tails), which also reports, for every chunk its first pass over the buffer reached,
 * the frame's length (0 = not reached or a duplicate) and the XXH3-128 of the frame.  bench.py compares these with
 * the GPU's frames: every chunk the CPU baseline compresses is a parity check. */
int sqo_bench_pipeline2(const uint8_t* data, const uint64_t* off, const uint32_t* len, size_t n_chunks, unsigned threads,
                        double seconds, uint64_t stats[4], uint32_t* out_len, uint8_t* out_digest) {
    if (!off || !len || !out_len || !out_digest) return -1;
    size_t mx = 0;
    for (size_t i = 0; i < n_chunks; i++) {
        if (len[i] > mx) mx = len[i];
        out_len[i] = 0;
    }
    return bp_run(data, off, len, n_chunks, mx, threads, seconds, stats, out_len, out_digest);
}

static int bp_run(const uint8_t* data, const uint64_t* off, const uint32_t* len, size_t n_chunks, size_t chunk_len, unsigned threads,
                  double seconds, uint64_t stats[4], uint32_t* out_len, uint8_t* out_digest) {
    if (!data || !n_chunks || !threads || !stats) return -1;
    bp_state* st = (bp_state*)calloc(1, sizeof(bp_state));
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    pthread_t wr;
    if (!st || !th) return -1;
    st->data = data;
    st->n_chunks = n_chunks;
    st->chunk_len = chunk_len;
    st->off = off;
    st->len = len;
    st->out_len = out_len;
    st->out_digest = out_digest;
    st->sink_cap = (size_t)256 << 20;
    st->sink = (uint8_t*)malloc(st->sink_cap);
    if (!st->sink) return -1;
    for (unsigned i = 0; i < BP_SHARDS; i++) pthread_mutex_init(&st->shards[i].mu, NULL);
    pthread_mutex_init(&st->qmu, NULL);
    pthread_cond_init(&st->qcv, NULL);
    st->t_end = bt_now() + seconds;
    if (pthread_create(&wr, NULL, bp_writer, st) != 0) return -1;
    unsigned started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, bp_worker, st) != 0) break;
    for (unsigned t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_mutex_lock(&st->qmu);
    st->closed = 1; /* drop(tx): the writer drains the channel and ends (writer.rs:167-170) */
    pthread_cond_signal(&st->qcv);
    pthread_mutex_unlock(&st->qmu);
    pthread_join(wr, NULL);
    stats[0] = st->written;
    stats[1] = st->bytes;
    stats[2] = st->dups;
    stats[3] = st->max_backlog;
    for (unsigned i = 0; i < BP_SHARDS; i++) free(st->shards[i].keys);
    free(st->sink);
    free(st);
    free(th);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------
 * bench.py, configs[3] leg: decode-and-verify of frames the GPU produced, on `threads` host threads for at most
 * `seconds`: frame k = frames + foff[k], flen[k] bytes must decode (zstd_dec_oracle.c, an RFC 8878 decoder written
 * independently of the encoder) to exactly want_len[k] bytes whose XXH3-128 is want_hash + 16 k.  *done = frames
 * verified; returns the number of frames that failed (0 = all good), -1 on bad arguments. */
typedef struct {
    const uint8_t* frames;
    const uint64_t* foff;
    const uint32_t *flen, *want_len;
    const uint8_t* want_hash;
    size_t n;
    double t_end;
    uint64_t next, done, bad;
} dv_state;

static void* dv_worker(void* p) {
    dv_state* st = (dv_state*)p;
    size_t const cap = SQO_CHUNK_SIZE + 64;
    uint8_t* out = (uint8_t*)malloc(cap);
    if (!out) return NULL;
    while (bt_now() < st->t_end) {
        uint64_t const k = __atomic_fetch_add(&st->next, 1, __ATOMIC_RELAXED);
        if (k >= st->n) break;
        uint8_t h[16];
        size_t const got = sqo_decompress_frame(st->frames + st->foff[k], st->flen[k], out, cap);
        int ok = got == (size_t)st->want_len[k];
        if (ok) {
            sqo_hash_chunk(out, got, h);
            ok = memcmp(h, st->want_hash + 16 * k, 16) == 0;
        }
        if (!ok) __atomic_fetch_add(&st->bad, 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&st->done, 1, __ATOMIC_RELAXED);
    }
    free(out);
    return NULL;
}

int64_t sqo_bench_decode_verify(const uint8_t* frames, const uint64_t* foff, const uint32_t* flen, const uint32_t* want_len,
                                const uint8_t* want_hash, size_t n, unsigned threads, double seconds, uint64_t* done) {
    if (!frames || !foff || !flen || !want_len || !want_hash || !threads || !done) return -1;
    dv_state st = {frames, foff, flen, want_len, want_hash, n, bt_now() + seconds, 0, 0, 0};
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    if (!th) return -1;
    unsigned started = 0;
    for (; started < threads; started++)
        if (pthread_create(&th[started], NULL, dv_worker, &st) != 0) break;
    for (unsigned t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    *done = st.done;
    return (int64_t)st.bad;
}
