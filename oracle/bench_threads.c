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
