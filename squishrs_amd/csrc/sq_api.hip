// sq_api.hip -- host side of libsquish_gpu.so: implements include/squish_gpu.h on top of the HIP
// kernels in this directory.  There is deliberately NO CPU implementation behind these entry
// points: without a usable gfx950 device every call fails with SQ_E_NODEVICE / SQ_E_DEVICE.
#include "../../include/squish_gpu.h"
#include "sq_common.h"
#include "sq_dedup.h"
#include "sq_zstd_decode.h"
#include "sq_xxh3.h"
#ifndef SQ_NO_COMPRESS
#include "sq_zstd_kernel.h"
#endif
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
#include <dlfcn.h>
#include <string>
#include <mutex>
#include <new>
#include <vector>

// Experiment / bring-up switches of the environment: read ONCE (the first sq_ctx_create), never on the per-batch path.
struct sq_env {
    bool debug = false, debug_stats = false, hash_wave_per_chunk = false, no_fastpath = false;
    int stage1_wgs_per_cu = 0, stage2_waves_per_cu = 0, fast_dbg = 0;
    bool s1_after_s2 = true; // SQ_S1_AFTER_S2=0: experiment switch of the launch order in pack_enqueue
    bool s2_early = false;   // SQ_S2_EARLY=1: fused form, stage 2 launched BEFORE the hash finish / dedup kernels, which then run beside it on
                             // the lane's second stream (pack_enqueue).  Built and measured in round 4: the step is the same 18.0-18.1 ms
                             // either way (profiles/round4c_s2_early_ab.txt) -- it is the sum of the kernels' work, not a gap -- so the
                             // plain order stays the default
    bool aux_prio = false;   // SQ_AUX_PRIO=1 (experiment, with SQ_S2_EARLY=1): the lanes' second streams are created at the device's highest
                             // stream priority, so that the hash finish / dedup kernels on them are dispatched ahead of the next batch's
                             // stage 1 when both become ready at the end of this batch's stage 1
    int stage1_grid = -1; // SQ_STAGE1_GRID: 1 = one workgroup per chunk instead of persistent workgroups (experiment)
    int fuse_hash = -1; // SQ_FUSE_HASH: 0 never, 1 always, unset: chosen per batch from the previous batch's duplicate ratio
    bool allow_any_arch = false;
};
static sq_env g_env;
static std::once_flag g_env_once;
static void sq_env_read_once();
// read once per PROCESS (squish_gpu.h): the first sq_ctx_create or stateless call of any thread fills g_env, later changes
// of the environment are not seen
static void sq_env_read() { std::call_once(g_env_once, sq_env_read_once); }
static void sq_env_read_once() {
    g_env.debug = getenv("SQ_DEBUG") != nullptr;
    g_env.debug_stats = getenv("SQ_DEBUG_STATS") != nullptr;
    g_env.hash_wave_per_chunk = getenv("SQ_HASH_WAVE_PER_CHUNK") != nullptr;
    g_env.no_fastpath = getenv("SQ_NO_FASTPATH") != nullptr;
    g_env.allow_any_arch = getenv("SQ_ALLOW_ANY_ARCH") != nullptr;
    if (const char* e = getenv("SQ_STAGE1_WAVES_PER_CU")) g_env.stage1_wgs_per_cu = atoi(e);
    if (const char* e = getenv("SQ_STAGE2_WAVES_PER_CU")) g_env.stage2_waves_per_cu = atoi(e);
    if (const char* e = getenv("SQ_FAST_DBG")) g_env.fast_dbg = atoi(e);
    if (const char* e = getenv("SQ_FUSE_HASH")) g_env.fuse_hash = atoi(e) ? 1 : 0;
    if (const char* e = getenv("SQ_STAGE1_GRID")) g_env.stage1_grid = atoi(e);
    if (const char* e = getenv("SQ_S1_AFTER_S2")) g_env.s1_after_s2 = atoi(e) != 0;
    if (const char* e = getenv("SQ_S2_EARLY")) g_env.s2_early = atoi(e) != 0;
    if (const char* e = getenv("SQ_AUX_PRIO")) g_env.aux_prio = atoi(e) != 0;
}

#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            if (g_env.debug)                                                                       \
                fprintf(stderr, "[squish_gpu] %s:%d %s -> %s\n", __FILE__, __LINE__, #x,           \
                        hipGetErrorString(e_));                                                    \
            return (e_ == hipErrorOutOfMemory) ? SQ_E_NOMEM : SQ_E_DEVICE;                         \
        }                                                                                          \
    } while (0)

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return SQ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + (n >> 3) + 4096; // grow with slack so steady state does not reallocate
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            e = hipMalloc(&p, n);
            if (e != hipSuccess) {
                p = nullptr;
                return SQ_E_NOMEM;
            }
            want = n;
        }
        cap = want;
        return SQ_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

// Everything ONE batch in flight needs: a stream, events, the per-batch device buffers, the two table pools of the
// compressor and the small pinned mirror.  A context has two of them ("lanes"): the synchronous entry points use
// lane 0; sq_pack_submit alternates, so that the thin tail of one batch's compress stage runs under the next
// batch's hash and stage-1 kernels (and, with host buffers, under its transfers).
struct sq_lane {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[10] = {};
    hipEvent_t dedup_done = nullptr; // recorded after this lane's dedup kernels: the other lane's dedup waits for it
    hipEvent_t pre_s2 = nullptr;     // recorded right before this lane's stage 2 is launched (fused form, see pack_enqueue)
    hipStream_t aux = nullptr;       // fused form of sq_pack_submit: hash finish + dedup run here, beside stage 2 (pack_enqueue)
    hipEvent_t s1_done = nullptr;    // recorded behind this lane's stage 1 (what `aux` waits for)
    DevBuf b_off, b_len, b_slot_off, b_frame_len, b_out_off, b_isnew, b_hashes, b_stage_in,
        b_stage_out, b_slots, b_ws, b_misc, b_cert, b_etab, b_work, b_hashed, b_hpool, b_jobs, b_pair;
    DevBuf b_dz_in, b_dz_out, b_dz_ooff, b_dz_ocap, b_dz_olen, b_dz_stat, b_dz_lit; // sq_decompress_batch
    // shapes of the two table pools at the last launch that completed (persistent generation tags)
    void* ws_ptr_last = nullptr;
    size_t ws_each_last = 0;
    u32 ws_waves_last = 0;
    void* etab_ptr_last = nullptr;
    size_t etab_bytes_last = 0;
    u32 etab_waves_last = 0;
    void* h_pinned = nullptr; // small pinned host mirror (error words etc.)
    sq_timings last = {0, 0, 0, 0, 0, 0, 0};
    u32 work_n = 0; // chunks of the launch whose stage-2 work lists b_work holds (0: none)
    // a batch enqueued by sq_pack_submit and not yet collected by sq_pack_wait
    bool busy = false;
    u32 job_n = 0;
    std::vector<u64> job_slot_off;
    void* h_len_pinned = nullptr; // pinned mirror of the frame lengths of the batch in flight
    size_t h_len_cap = 0;
    u32 job_max_len = 0;
    u32 job_waves = 0, job_fwaves = 0;
    bool job_s2w3 = false; // this batch's stage 2 runs the 3-waves-per-SIMD build of the kernel (sq_ctx::f_s2w3)
    size_t job_ws_each = 0, job_etab_bytes = 0;
    bool job_fast = false, job_in_place = false, job_fused = false;
    bool job_sharded = false; // enqueued through sq_shard_submit_rest[_idx] (sq_pack_wait: no duplicate-ratio note)
    u8* job_slots = nullptr;
    std::vector<u32> job_list; // stage 1's jobs when the batch holds chunks of more than two blocks (one job per pair of blocks)
    u32 job_count = 0;
    u32 job_ws_persist = 0, job_etab_persist = 0;
    // host-pointer batch of sq_pack_submit: where the results go at sq_pack_wait
    bool job_host = false;
    u8 *job_h_hashes = nullptr, *job_h_isnew = nullptr, *job_h_comp = nullptr;
    u64 job_comp_cap = 0;
    std::vector<u64> job_out_off;
    // sharded form: the batch whose hash was enqueued by sq_shard_submit_hash and waits for sq_shard_submit_rest
    const u8* shard_data = nullptr;
    std::vector<u32> shard_len;
};

struct sq_ctx {
    sq_config cfg;
    int device = 0;
    int num_cu = 256;

    // dedup state (persistent)
    u64* d_keys = nullptr; // 2 x u64 per chunk ever seen
    u32* d_table = nullptr;
    unsigned long long* d_unique = nullptr;
    u64 table_mask = 0;
    u64 key_capacity = 0;
    u64 n_seen = 0; // global chunk index of the next chunk

    u32* d_sched_tab = nullptr; // constant no-match schedule tables of the compressor (sched_tab_fill)
    sq_lane ln[2];
    u32 last_lane = 0; // lane of the batch finished most recently (sq_debug_stage2_list)
    u32 submitted = 0, collected = 0; // sq_pack_submit / sq_pack_wait tickets (submitted - collected <= 2)
    float dup_ratio_last = 0.f;       // duplicates / chunks of the batch collected last (chooses the fused form)
    bool shard_pending = false;       // sq_shard_submit_hash called, sq_shard_submit_rest not yet
    // k_compress_batch exists twice: in this library at 2 waves per SIMD (256 registers, 8 waves per CU) and in the code
    // object squish_s2_w3.co beside it at 3 (168 registers, 11 per CU; same source, SQZ_STAGE2_WAVES_PER_SIMD=3).  Compressible
    // batches run 5-13 % faster on the second (more chains of dependent trips in flight); batches that stage 1 mostly decides
    // run 5 % SLOWER with it (its spills take bandwidth from stage 1's stream: profiles/round4f_stage2_register_bound_ab.txt).
    // s2_dense: the batch collected last left more than a quarter of its new chunks to stage 2.
    hipModule_t mod_s2w3 = nullptr;
    hipFunction_t f_s2w3 = nullptr;
    bool s2_dense = false, s2_w3_last = false, s2w3_tried = false;
    bool s2w3_avail = false; // the file is there (checked at sq_ctx_create; the module itself is loaded when a batch first turns out compressible)
};

static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

extern "C" {

uint32_t sq_abi_version(void) { return SQ_ABI_VERSION; }

const char* sq_strerror(int err) {
    switch (err) {
    case SQ_OK: return "ok";
    case SQ_E_INVAL: return "invalid argument";
    case SQ_E_NOMEM: return "out of memory";
    case SQ_E_CAP: return "output capacity too small";
    case SQ_E_DEVICE: return "HIP device error";
    case SQ_E_COMPRESS: return "Compression error";
    case SQ_E_NODEVICE: return "no gfx950 device available (libsquish_gpu has no CPU path)";
    case SQ_E_DECOMPRESS: return "Decompression error";
    default: return "unknown error";
    }
}

size_t sq_compress_bound(size_t n) {
    return n + (n >> 8) + ((n < (128u << 10)) ? (((128u << 10) - n) >> 11) : 0);
}

#ifndef SQ_NO_COMPRESS
// The second build of the compressor kernel (squish_s2_w3.co beside the library; optional: without it every batch takes the
// first).  Loaded when a batch first wants it, not in sq_ctx_create: hipModuleLoad costs ~0.1 s, which a small pack never
// needs to pay.
static std::string s2w3_path() {
    Dl_info di;
    if (getenv("SQ_NO_S2_W3") || !dladdr((const void*)&sq_ctx_create, &di) || !di.dli_fname) return std::string();
    std::string path(di.dli_fname);
    size_t const slash = path.find_last_of('/');
    return (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/squish_s2_w3.co";
}
static void load_s2w3(sq_ctx* c) {
    if (c->s2w3_tried) return;
    c->s2w3_tried = true;
    std::string const path = s2w3_path();
    if (path.empty()) return;
    if (hipModuleLoad(&c->mod_s2w3, path.c_str()) == hipSuccess) {
        if (hipModuleGetFunction(&c->f_s2w3, c->mod_s2w3, SQZ_K_COMPRESS_BATCH_SYMBOL) != hipSuccess) c->f_s2w3 = nullptr;
    } else
        c->mod_s2w3 = nullptr;
    (void)hipGetLastError();
    if (!c->f_s2w3 && g_env.debug) fprintf(stderr, "[squish_gpu] %s not loaded: compressible batches use the 8-waves-per-CU kernel\n", path.c_str());
}
#endif

int sq_ctx_create(const sq_config* cfg_in, sq_ctx** out) {
    if (!out) return SQ_E_INVAL;
    *out = nullptr;
    sq_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    if (cfg_in) cfg = *cfg_in;
    if (cfg_in && cfg.abi_version != SQ_ABI_VERSION) return SQ_E_INVAL;
    if (cfg.max_chunk == 0 || cfg.max_chunk > SQ_CHUNK_SIZE) cfg.max_chunk = SQ_CHUNK_SIZE;
    if (cfg.max_batch_chunks == 0) cfg.max_batch_chunks = 4096;
    if (cfg.max_batch_bytes == 0) cfg.max_batch_bytes = 1ull << 30;
    if (cfg.dedup_capacity == 0) cfg.dedup_capacity = 1ull << 22;

    sq_env_read();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return SQ_E_NODEVICE;
    }
    if (cfg.device < 0 || cfg.device >= ndev) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(cfg.device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg.device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !g_env.allow_any_arch) return SQ_E_NODEVICE;

    sq_ctx* c = new (std::nothrow) sq_ctx();
    if (!c) return SQ_E_NOMEM;
    c->cfg = cfg;
    c->device = cfg.device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int prio_least = 0, prio_greatest = 0;
    if (g_env.aux_prio) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (sq_lane& ln : c->ln) {
        sq_lane* const L = &ln;
        if (hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking) != hipSuccess) {
            sq_ctx_destroy(c);
            return SQ_E_DEVICE;
        }
        L->own_stream = true;
        for (auto& e : L->ev)
            if (hipEventCreate(&e) != hipSuccess) {
                sq_ctx_destroy(c);
                return SQ_E_DEVICE;
            }
        if (hipEventCreateWithFlags(&L->dedup_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L->pre_s2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L->s1_done, hipEventDisableTiming) != hipSuccess ||
            (g_env.aux_prio ? hipStreamCreateWithPriority(&L->aux, hipStreamNonBlocking, prio_greatest)
                            : hipStreamCreateWithFlags(&L->aux, hipStreamNonBlocking)) != hipSuccess ||
            hipHostMalloc(&L->h_pinned, 4096, hipHostMallocDefault) != hipSuccess) {
            sq_ctx_destroy(c);
            return SQ_E_NOMEM;
        }
    }
    sq_lane* const L = &c->ln[0];
    // dedup set: table of 2x capacity (power of two), keys for `capacity` chunks
    u64 slots = 1024;
    while (slots < cfg.dedup_capacity * 2) slots <<= 1;
    c->table_mask = slots - 1;
    c->key_capacity = cfg.dedup_capacity;
    if (hipMalloc((void**)&c->d_table, slots * sizeof(u32)) != hipSuccess ||
        hipMalloc((void**)&c->d_keys, c->key_capacity * 16) != hipSuccess ||
        hipMalloc((void**)&c->d_unique, 64) != hipSuccess) {
        sq_ctx_destroy(c);
        return SQ_E_NOMEM;
    }
    if (hipMemsetAsync(c->d_table, 0, slots * sizeof(u32), L->stream) != hipSuccess ||
        hipMemsetAsync(c->d_unique, 0, 64, L->stream) != hipSuccess ||
        hipStreamSynchronize(L->stream) != hipSuccess) {
        sq_ctx_destroy(c);
        return SQ_E_DEVICE;
    }
#ifndef SQ_NO_COMPRESS
    {
        std::vector<u32> tab(SQZ_SCHEDTAB_WORDS + 4, 0); // (+ padding: stage 1 reads the last 2-byte bitmap word with a 4-byte load)
        sqz::sched_tab_fill(tab.data());
        if (hipMalloc((void**)&c->d_sched_tab, tab.size() * 4) != hipSuccess ||
            hipMemcpy(c->d_sched_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            sq_ctx_destroy(c);
            return SQ_E_NOMEM;
        }
    }
#endif
#ifndef SQ_NO_COMPRESS
    {
        std::string const path = s2w3_path();
        c->s2w3_avail = !path.empty() && access(path.c_str(), R_OK) == 0;
    }
#endif
    *out = c;
    return SQ_OK;
}

void sq_ctx_destroy(sq_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (sq_lane& ln : c->ln) {
        sq_lane* const L = &ln;
        if (L->stream) (void)hipStreamSynchronize(L->stream);
        if (L->aux) (void)hipStreamSynchronize(L->aux);
        for (DevBuf* b : {&L->b_off, &L->b_len, &L->b_slot_off, &L->b_frame_len, &L->b_out_off, &L->b_isnew,
                          &L->b_hashes, &L->b_stage_in, &L->b_stage_out, &L->b_slots, &L->b_ws, &L->b_misc, &L->b_cert, &L->b_etab, &L->b_work,
                          &L->b_hashed, &L->b_hpool, &L->b_jobs, &L->b_pair, &L->b_dz_in, &L->b_dz_out, &L->b_dz_ooff, &L->b_dz_ocap,
                          &L->b_dz_olen, &L->b_dz_stat, &L->b_dz_lit})
            b->release();
        if (L->h_pinned) (void)hipHostFree(L->h_pinned);
        if (L->h_len_pinned) (void)hipHostFree(L->h_len_pinned);
        for (auto& e : L->ev)
            if (e) (void)hipEventDestroy(e);
        if (L->dedup_done) (void)hipEventDestroy(L->dedup_done);
        if (L->pre_s2) (void)hipEventDestroy(L->pre_s2);
        if (L->s1_done) (void)hipEventDestroy(L->s1_done);
        if (L->aux) (void)hipStreamDestroy(L->aux);
        if (L->own_stream && L->stream) (void)hipStreamDestroy(L->stream);
    }
    if (c->d_table) (void)hipFree(c->d_table);
    if (c->d_keys) (void)hipFree(c->d_keys);
    if (c->d_unique) (void)hipFree(c->d_unique);
    if (c->d_sched_tab) (void)hipFree(c->d_sched_tab);
    if (c->mod_s2w3) (void)hipModuleUnload(c->mod_s2w3);
    delete c;
}

int sq_ctx_set_stream(sq_ctx* c, void* s) {
    if (!c) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(L->stream));
    if (L->own_stream) (void)hipStreamDestroy(L->stream);
    L->stream = (hipStream_t)s;
    L->own_stream = false;
    return SQ_OK;
}

int sq_ctx_synchronize(sq_ctx* c) {
    if (!c) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(c->device));
    for (sq_lane& ln : c->ln) HIPCHK(hipStreamSynchronize(ln.stream));
    return SQ_OK;
}

void* sq_dev_alloc(sq_ctx* c, uint64_t bytes) {
    if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void sq_dev_free(sq_ctx* c, void* p) {
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(p);
}
int sq_dev_copy(sq_ctx* c, void* dst, const void* src, uint64_t bytes, int to_device) {
    if (!c || (bytes && (!dst || !src))) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
    return SQ_OK;
}

int sq_ctx_trim(sq_ctx* c) {
    if (!c) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches in flight hold these buffers
    HIPCHK(hipSetDevice(c->device));
    for (sq_lane& ln : c->ln) {
        HIPCHK(hipStreamSynchronize(ln.stream));
        if (ln.aux) HIPCHK(hipStreamSynchronize(ln.aux));
        for (DevBuf* b : {&ln.b_ws, &ln.b_slots, &ln.b_stage_in, &ln.b_stage_out, &ln.b_hpool, &ln.b_etab, &ln.b_dz_in, &ln.b_dz_out, &ln.b_dz_lit})
            b->release();
        ln.ws_ptr_last = nullptr; // (the saved table states went with the pool: the next launch clears its tables)
        ln.ws_waves_last = 0;
        ln.etab_ptr_last = nullptr;
    }
    return SQ_OK;
}

uint64_t sq_unique_count(const sq_ctx* c) {
    if (!c) return 0;
    unsigned long long v = 0;
    (void)hipSetDevice(c->device);
    for (const sq_lane& ln : c->ln)
        if (hipStreamSynchronize(ln.stream) != hipSuccess) return 0;
    if (hipMemcpy(&v, c->d_unique, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (uint64_t)v;
}

int sq_dedup_reset(sq_ctx* c) {
    if (!c) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(c->d_table, 0, (c->table_mask + 1) * sizeof(u32), L->stream));
    HIPCHK(hipMemsetAsync(c->d_unique, 0, 64, L->stream));
    c->n_seen = 0;
    return SQ_OK;
}

int sq_last_timings(sq_ctx* c, sq_timings* out) {
    if (!c || !out) return SQ_E_INVAL;
    sq_lane* const L = &c->ln[0];
    *out = L->last;
    return SQ_OK;
}

// ---- helpers --------------------------------------------------------------------------------
static int upload_off_len(sq_ctx* c, sq_lane* L, const u64* h_off, const u32* h_len, u32 n) {
    int r;
    if ((r = L->b_off.ensure((size_t)n * 8))) return r;
    if ((r = L->b_len.ensure((size_t)n * 4))) return r;
    HIPCHK(hipMemcpyAsync(L->b_off.p, h_off, (size_t)n * 8, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemcpyAsync(L->b_len.p, h_len, (size_t)n * 4, hipMemcpyHostToDevice, L->stream));
    return SQ_OK;
}

// d_summed / d_sums (or null): chunks whose block sums the fused form of compress stage 1 has left in its pool
// st (or null = the lane's stream): the stream the kernel is enqueued on
static int launch_hash(sq_ctx* c, sq_lane* L, const u8* d_data, u32 n, u8* d_hashes, const u8* d_summed = nullptr,
                       const u8* d_sums = nullptr, hipStream_t st = nullptr) {
    if (n == 0) return SQ_OK;
    if (!st) st = L->stream;
    const u32 wpb = 4;
    HIPCHK(hipEventRecord(L->ev[0], st));
    if (g_env.hash_wave_per_chunk && !d_summed) { // the first mapping: one wave per chunk
        dim3 grid((n + wpb - 1) / wpb), block(64 * wpb);
        hipLaunchKernelGGL(sqx::k_xxh3_128_batch, grid, block, 0, st, d_data, (const u64*)L->b_off.p,
                           (const u32*)L->b_len.p, n, d_hashes);
    } else { // 16 lanes per chunk, four chunks per wave
        dim3 grid((n + 4 * wpb - 1) / (4 * wpb)), block(64 * wpb);
        hipLaunchKernelGGL(sqx::k_xxh3_128_rows, grid, block, 0, st, d_data, (const u64*)L->b_off.p,
                           (const u32*)L->b_len.p, n, d_hashes, d_summed, d_sums, (u64)SQZ_FAST_HPOOL_BYTES);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->ev[1], st));
    return SQ_OK;
}

int sq_hash_batch(sq_ctx* c, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len,
                  uint32_t n, uint8_t* d_hashes) {
    if (!c || (n && (!d_data || !h_off || !h_len || !d_hashes))) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r;
    if ((r = upload_off_len(c, L, h_off, h_len, n))) return r;
    if ((r = launch_hash(c, L, d_data, n, d_hashes))) return r;
    HIPCHK(hipStreamSynchronize(L->stream));
    if (n) HIPCHK(hipEventElapsedTime(&L->last.hash_ms, L->ev[0], L->ev[1]));
    return SQ_OK;
}

static int dedup_async(sq_ctx* c, sq_lane* L, const u8* d_all_hashes, u64 n_total, u64 my_first, u32 my_n,
                       u8* d_is_new, hipStream_t st = nullptr) {
    if (c->n_seen + n_total > c->key_capacity) return SQ_E_CAP;
    if (c->n_seen + n_total >= 0xFFFFFFFFull) return SQ_E_CAP;
    if (n_total == 0) return SQ_OK;
    if (!st) st = L->stream;
    const u64 g0 = c->n_seen;
    HIPCHK(hipEventRecord(L->ev[2], st));
    HIPCHK(hipMemcpyAsync(c->d_keys + 2 * g0, d_all_hashes, n_total * 16, hipMemcpyDeviceToDevice,
                          st));
    {
        dim3 block(256), grid((unsigned)((n_total + 255) / 256));
        hipLaunchKernelGGL(sqd::k_dedup_insert, grid, block, 0, st, (const u64*)c->d_keys, g0,
                           n_total, c->d_table, c->table_mask, c->d_unique);
        HIPCHK(hipGetLastError());
    }
    if (my_n) {
        dim3 block(256), grid((my_n + 255) / 256);
        hipLaunchKernelGGL(sqd::k_dedup_resolve, grid, block, 0, st, (const u64*)c->d_keys,
                           g0 + my_first, my_n, (const u32*)c->d_table, c->table_mask, d_is_new);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(L->ev[3], st));
    c->n_seen += n_total;
    return SQ_OK;
}

int sq_dedup_global(sq_ctx* c, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first,
                    uint32_t my_n, uint8_t* d_is_new) {
    if (!c || (n_total && !d_all_hashes) || (my_n && !d_is_new) || my_first + my_n > n_total)
        return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r = dedup_async(c, L, d_all_hashes, n_total, my_first, my_n, d_is_new);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(L->stream));
    if (n_total) HIPCHK(hipEventElapsedTime(&L->last.dedup_ms, L->ev[2], L->ev[3]));
    return SQ_OK;
}

int sq_calibrate_read_bw(sq_ctx* c, const uint8_t* d_buf, uint64_t nbytes, float* ms) {
    if (!c || !d_buf || !ms || ((uintptr_t)d_buf & 15)) return SQ_E_INVAL;
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r = L->b_misc.ensure(256);
    if (r) return r;
    dim3 block(256), grid((unsigned)(c->num_cu * 8));
    HIPCHK(hipEventRecord(L->ev[6], L->stream));
    hipLaunchKernelGGL(sqd::k_read_bw, grid, block, 0, L->stream, (const uint4*)d_buf, nbytes / 16,
                       (u32*)L->b_misc.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->ev[7], L->stream));
    HIPCHK(hipStreamSynchronize(L->stream));
    HIPCHK(hipEventElapsedTime(ms, L->ev[6], L->ev[7]));
    return SQ_OK;
}

// req_bytes: 16, 32 or 64 (one lane's piece: 1, 2 or 4 consecutive 16-byte loads); depth: 1, 2, 4, 8 or 16 pieces in flight per
// lane; waves_per_cu: 4, 8, 16 or 32 resident waves per CU.  0 = the defaults (16 bytes, 8 in flight, 32 waves per CU).
int sq_calibrate_gather2(sq_ctx* c, const uint8_t* d_buf, uint64_t nbytes, uint32_t per_lane, uint32_t req_bytes, uint32_t depth,
                         uint32_t waves_per_cu, float* ms, uint64_t* requests) {
    if (!c || !d_buf || !ms || !requests || ((uintptr_t)d_buf & 15) || nbytes < 4096 || per_lane == 0) return SQ_E_INVAL;
    if (c->submitted != c->collected) return SQ_E_INVAL;
    if (req_bytes == 0) req_bytes = 16;
    if (depth == 0) depth = 8;
    if (waves_per_cu == 0) waves_per_cu = 32;
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r = L->b_misc.ensure(256);
    if (r) return r;
    per_lane = (per_lane + 15) & ~15u;
    dim3 block(256), grid((unsigned)(c->num_cu * waves_per_cu / 4));
    const uint4* const p = (const uint4*)d_buf;
    u64 const n16 = nbytes / 16;
    u32* const sink = (u32*)L->b_misc.p;
    HIPCHK(hipEventRecord(L->ev[6], L->stream));
#define SQ_GB(R, D) hipLaunchKernelGGL((sqd::k_gather_bw<R, D>), grid, block, 0, L->stream, p, n16, per_lane, sink)
#define SQ_GBD(R)                                                                        \
    switch (depth) {                                                                     \
    case 1: SQ_GB(R, 1); break;                                                          \
    case 2: SQ_GB(R, 2); break;                                                          \
    case 4: SQ_GB(R, 4); break;                                                          \
    case 8: SQ_GB(R, 8); break;                                                          \
    case 16: SQ_GB(R, 16); break;                                                        \
    default: return SQ_E_INVAL;                                                          \
    }
    switch (req_bytes) {
    case 16: SQ_GBD(1); break;
    case 32: SQ_GBD(2); break;
    case 64: SQ_GBD(4); break;
    default: return SQ_E_INVAL;
    }
#undef SQ_GBD
#undef SQ_GB
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->ev[7], L->stream));
    HIPCHK(hipStreamSynchronize(L->stream));
    HIPCHK(hipEventElapsedTime(ms, L->ev[6], L->ev[7]));
    *requests = (uint64_t)grid.x * 256 * per_lane;
    return SQ_OK;
}

int sq_calibrate_gather(sq_ctx* c, const uint8_t* d_buf, uint64_t nbytes, uint32_t per_lane, float* ms, uint64_t* requests) {
    return sq_calibrate_gather2(c, d_buf, nbytes, per_lane, 16, 8, 32, ms, requests);
}

#ifndef SQ_NO_COMPRESS
// Frames are produced into per-chunk slots (bound(len) bytes, 16-byte aligned).  h_comp_len ==
// nullptr: the slots are a scratch buffer and the frames are then packed back to back into d_comp
// (h_comp_off: n + 1 offsets).  Otherwise the slots ARE d_comp and the frames stay where they were
// produced (h_comp_off: n slot offsets, h_comp_len: n frame lengths): no packing pass.
// Three steps, all of them only ENQUEUE on the lane's stream: compress_prepare sizes and allocates everything a batch
// needs (every fallible allocation and argument check of the compress stage happens here, BEFORE a caller commits the
// batch to the shared dedup set), compress_stage1 launches the incompressible-chunk kernel, compress_stage2 the
// compressor proper and the small result copies.  compress_finish waits for them, hands out offsets / lengths and
// (packed layout) runs the packing pass.
// fused: stage 1 will run on every chunk before the dedup decision and also hash what it streams (sq_zstd_fast.h).
static int compress_prepare(sq_ctx* c, sq_lane* L, const u32* h_len, u32 n, u8* d_comp, u64 comp_cap, bool in_place, bool fused) {
    int r;
    L->job_n = n;
    L->job_in_place = in_place;
    L->job_fused = false;
    L->job_fast = false;
    if (n == 0) return SQ_OK;
    // slot offsets on the host (prefix sum of bounds)
    std::vector<u64>& slot_off = L->job_slot_off;
    slot_off.resize((size_t)n + 1);
    u64 acc = 0;
    u32 max_len = 0;
    for (u32 i = 0; i < n; i++) {
        slot_off[i] = acc;
        acc += (sq_compress_bound(h_len[i]) + 32 + 15) & ~(u64)15;
        max_len = std::max(max_len, h_len[i]);
    }
    slot_off[n] = acc;
    if (max_len > c->cfg.max_chunk) return SQ_E_INVAL;
    if ((r = L->b_slot_off.ensure(((size_t)n + 1) * 8))) return r;
    if ((r = L->b_frame_len.ensure((size_t)n * 4))) return r;
    if ((r = L->b_out_off.ensure(((size_t)n + 1) * 8))) return r;
    if (in_place) {
        if (acc > comp_cap) return SQ_E_CAP;
    } else if ((r = L->b_slots.ensure(acc)))
        return r;
    L->job_slots = in_place ? d_comp : (u8*)L->b_slots.p;

    // persistent compressor waves, one private workspace each
    if (c->s2_dense && !c->cfg.compress_waves && !g_env.stage2_waves_per_cu) load_s2w3(c);
    bool const w3 = c->f_s2w3 != nullptr && c->s2_dense && !c->cfg.compress_waves && !g_env.stage2_waves_per_cu;
    L->job_s2w3 = w3;
    u32 waves = c->cfg.compress_waves ? c->cfg.compress_waves : (u32)c->num_cu * (w3 ? SQZ_STAGE2_WAVES_PER_CU : 8u);
    if (g_env.stage2_waves_per_cu >= 1 && g_env.stage2_waves_per_cu <= 8 && !c->cfg.compress_waves) // experiment: leave wave slots to the other lane's kernels
        waves = (u32)c->num_cu * (u32)g_env.stage2_waves_per_cu;
    waves = std::min(waves, n);
    const size_t ws_each = sqz::workspace_bytes(max_len);
    // The pool is sized for the larger of the two builds' grids, so that a change of build between batches does not
    // reallocate it (12 GB and more); keep it within ~96 GiB
    u32 pool_waves = ((c->s2w3_avail && (!c->s2w3_tried || c->f_s2w3 != nullptr)) && !c->cfg.compress_waves && !g_env.stage2_waves_per_cu)
                         ? std::min((u32)c->num_cu * SQZ_STAGE2_WAVES_PER_CU, n) : waves;
    while (pool_waves > 64 && (u64)pool_waves * ws_each > (96ull << 30)) pool_waves /= 2;
    waves = std::min(waves, pool_waves);
    // Stage 1, the incompressible-chunk kernel (sq_zstd_fast.h): as many workgroups as fit (its state is all
    // in LDS).  It only applies to chunks of 16 KiB < len <= 256 KiB.
    const bool fast = max_len > (16u << 10) && !g_env.no_fastpath;
    u32 fwaves = 0;
    size_t etab_bytes = 0;
    if (fast) {
        etab_bytes = 0; // the certificate's set lives in LDS now; a workgroup's slice only keeps its 256 bytes of saved state
        int occ = 0; // resident workgroups (of SQZ_FAST_WG_WAVES waves, one chunk each) per CU
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sqz::k_incompressible_batch, 64 * SQZ_FAST_WG_WAVES, 0);
        if (occ < 1) occ = 4;
        if (g_env.stage1_wgs_per_cu >= 1 && g_env.stage1_wgs_per_cu < occ) occ = g_env.stage1_wgs_per_cu; // experiment: leave room for another lane's kernels
        fwaves = c->cfg.compress_waves ? c->cfg.compress_waves * 2 : (u32)c->num_cu * (u32)occ;
        // Stage 1's jobs: a chunk of at most two blocks is one; a larger one (lazy2 class: the reference's 2 MiB chunks)
        // is one per pair of its 128 KiB blocks (sq_zstd_fast.h fastpath_blocks).  The list only exists when needed.
        L->job_list.clear();
        L->job_count = n;
        if (max_len > (256u << 10) && n < (1u << 28)) {
            for (u32 i = 0; i < n; i++) {
                u32 const nb = (h_len[i] + (128u << 10) - 1) / (128u << 10), np = nb > 2 ? (nb + 1) / 2 : 1u;
                for (u32 p = 0; p < np; p++) L->job_list.push_back(i | (p << 28));
            }
            L->job_count = (u32)L->job_list.size();
            if ((r = L->b_jobs.ensure((size_t)L->job_count * 4))) return r;
            if ((r = L->b_pair.ensure((size_t)n * 4))) return r;
        }
        fwaves = std::min(fwaves, L->job_count);
        if (g_env.stage1_grid == 1) fwaves = n;
        if ((r = L->b_cert.ensure((size_t)n * 8))) return r;
        if ((r = L->b_work.ensure(((size_t)2 * n + 2) * 4))) return r;
        if (fused) {
            if ((r = L->b_hashed.ensure((size_t)n + 16))) return r;
            if ((r = L->b_hpool.ensure((size_t)n * SQZ_FAST_HPOOL_BYTES))) return r;
        }
    }
    if ((r = L->b_ws.ensure((size_t)pool_waves * ws_each + 256))) return r;
    if (fast && (r = L->b_etab.ensure((size_t)fwaves * (etab_bytes + 256)))) return r;
    if ((r = L->b_misc.ensure(256))) return r;
    if (in_place && L->h_len_cap < (size_t)n * 4) {
        if (L->h_len_pinned) (void)hipHostFree(L->h_len_pinned);
        L->h_len_pinned = nullptr;
        L->h_len_cap = 0;
        size_t const want = (size_t)n * 4 + ((size_t)n >> 1) + 4096;
        if (hipHostMalloc(&L->h_len_pinned, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return SQ_E_NOMEM;
        }
        L->h_len_cap = want;
    }
    // the pools keep their generation tags from launch to launch as long as nothing about them changes
    // (the saved table states of slots [0, ws_waves_last) are good: a launch of at most that many waves continues from them)
    L->job_ws_persist = (L->b_ws.p == L->ws_ptr_last && ws_each == L->ws_each_last && waves <= L->ws_waves_last && fast) ? 1u : 0u;
    L->job_etab_persist = (fast && L->b_etab.p == L->etab_ptr_last && etab_bytes == L->etab_bytes_last && fwaves == L->etab_waves_last) ? 1u : 0u;
    L->ws_ptr_last = nullptr; // becomes valid again when this launch has completed
    L->etab_ptr_last = nullptr;
    L->job_max_len = max_len;
    L->job_waves = waves;
    L->job_fwaves = fwaves;
    L->job_ws_each = ws_each;
    L->job_etab_bytes = etab_bytes;
    L->job_fast = fast;
    L->job_fused = fused && fast;
    L->last.fast_ms = 0;
    // nothing below can fail for lack of memory
    HIPCHK(hipMemcpyAsync(L->b_slot_off.p, slot_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemsetAsync(L->b_frame_len.p, 0, (size_t)n * 4, L->stream));
    if (fast) HIPCHK(hipMemsetAsync(L->b_work.p, 0, 8, L->stream));
    if (fast && !L->job_list.empty()) {
        HIPCHK(hipMemcpyAsync(L->b_jobs.p, L->job_list.data(), (size_t)L->job_count * 4, hipMemcpyHostToDevice, L->stream));
        HIPCHK(hipMemsetAsync(L->b_pair.p, 0, (size_t)n * 4, L->stream));
    }
    if (L->job_fused) HIPCHK(hipMemsetAsync(L->b_hashed.p, 0, (size_t)n, L->stream));
    HIPCHK(hipMemsetAsync(L->b_misc.p, 0, 256, L->stream));
    if (g_env.debug) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sqz::k_compress_batch, 64, 0);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void*)sqz::k_compress_batch);
        fprintf(stderr, "[squish_gpu] compress kernel: occupancy %d blocks/CU, regs %d, lds %zu, scratch %zu; waves %u, ws_each %zu\n",
                nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, waves, ws_each);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sqz::k_incompressible_batch, 64 * SQZ_FAST_WG_WAVES, 0);
        (void)hipFuncGetAttributes(&fa, (const void*)sqz::k_incompressible_batch);
        fprintf(stderr, "[squish_gpu] incompressible-chunk kernel: occupancy %d blocks/CU, regs %d, lds %zu, scratch %zu; workgroups %u, fused hash %d\n",
                nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fwaves, (int)L->job_fused);
    }
    return SQ_OK;
}

// Stage 1.  d_select: the chunks to look at (null = all).  fused (L->job_fused must hold): the kernel also leaves the
// XXH3 block sums of every chunk it streams in L->b_hpool, L->b_hashed says which.
static int compress_stage1(sq_ctx* c, sq_lane* L, const u8* d_data, const u8* d_select, bool fused) {
    u32 const n = L->job_n;
    if (n == 0 || !L->job_fast) return SQ_OK;
    fused = fused && L->job_fused;
    HIPCHK(hipEventRecord(L->ev[8], L->stream));
    hipLaunchKernelGGL(sqz::k_incompressible_batch, dim3(L->job_fwaves), dim3(64 * SQZ_FAST_WG_WAVES), 0, L->stream, d_data,
                       (const u64*)L->b_off.p, (const u32*)L->b_len.p, d_select, n, L->job_slots,
                       (const u64*)L->b_slot_off.p, (u32*)L->b_frame_len.p, (u32*)L->b_cert.p, (u8*)L->b_etab.p,
                       (u64)L->job_etab_bytes, (u32*)L->b_misc.p + 32 /*queue*/, (const u32*)c->d_sched_tab,
                       (u32)g_env.fast_dbg, L->job_etab_persist, (u32*)L->b_work.p,
                       fused ? (u8*)L->b_hashed.p : (u8*)nullptr, fused ? (u8*)L->b_hpool.p : (u8*)nullptr,
                       L->job_list.empty() ? (const u32*)nullptr : (const u32*)L->b_jobs.p, L->job_count,
                       L->job_list.empty() ? (u32*)nullptr : (u32*)L->b_pair.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->ev[9], L->stream));
    return SQ_OK;
}

// Stage 2, the compressor proper, for everything stage 1 left undecided (of the selected chunks), then the small
// result copies.  Fused form: stage 1 ran before the dedup decision, so the frames it wrote for duplicates are dropped
// here (their lengths zeroed) and the compressor skips the duplicates on its work lists.
// flags_ready (fused form of sq_pack_submit, or null): the dedup flags d_select are produced on ANOTHER stream and this
// event says when; stage 2 is then launched at once, on every chunk stage 1 left undecided, and the duplicates' frames --
// stage 2's first (no count to correct), then stage 1's -- are dropped behind it.
static int compress_stage2(sq_ctx* c, sq_lane* L, const u8* d_data, const u8* d_select, hipEvent_t flags_ready = nullptr) {
    u32 const n = L->job_n;
    if (n == 0) return SQ_OK;
    bool const fast = L->job_fast, in_place = L->job_in_place;
    bool const late_drop = flags_ready != nullptr && L->job_fused && d_select != nullptr;
    const u8* const d_take = late_drop ? nullptr : d_select;
    if (L->job_fused && d_select && !late_drop) {
        hipLaunchKernelGGL(sqd::k_drop_unselected, dim3((n + 255) / 256), dim3(256), 0, L->stream, (u32*)L->b_frame_len.p, d_select, n,
                           (u32*)L->b_misc.p + 33 /* stage 1's count of decided chunks */);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(L->ev[4], L->stream));
    {
        dim3 block(64), grid(L->job_waves);
        const u64* a_off = (const u64*)L->b_off.p;
        const u32* a_len = (const u32*)L->b_len.p;
        u32 a_n = n;
        u8* a_slots = L->job_slots;
        const u64* a_slot_off = (const u64*)L->b_slot_off.p;
        u32* a_frame_len = (u32*)L->b_frame_len.p;
        u8* a_ws = (u8*)L->b_ws.p;
        u64 a_ws_each = (u64)L->job_ws_each;
        u32* a_queue = (u32*)L->b_misc.p;
        u32* a_err = (u32*)L->b_misc.p + 16;
        const u32* a_sched = (const u32*)c->d_sched_tab;
        const u32* a_cert = fast ? (const u32*)L->b_cert.p : (const u32*)nullptr;
        u32 a_persist = L->job_ws_persist;
        const u32* a_work = fast ? (const u32*)L->b_work.p : (const u32*)nullptr;
        const u32* a_pair = (fast && !L->job_list.empty()) ? (const u32*)L->b_pair.p : (const u32*)nullptr;
        if (L->job_s2w3) { // the same kernel from squish_s2_w3.co (see sq_ctx::f_s2w3)
            void* args[] = {(void*)&d_data, &a_off, &a_len, (void*)&d_take, &a_n, &a_slots, &a_slot_off, &a_frame_len, &a_ws, &a_ws_each,
                            &a_queue, &a_err, &a_sched, &a_cert, &a_persist, &a_work, &a_pair};
            HIPCHK(hipModuleLaunchKernel(c->f_s2w3, grid.x, 1, 1, 64, 1, 1, 0, L->stream, args, nullptr));
        } else {
            hipLaunchKernelGGL(sqz::k_compress_batch, grid, block, 0, L->stream, d_data, a_off, a_len, d_take, a_n, a_slots, a_slot_off,
                               a_frame_len, a_ws, a_ws_each, a_queue, a_err, a_sched, a_cert, a_persist, a_work, a_pair);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipEventRecord(L->ev[5], L->stream));
    if (late_drop) {
        HIPCHK(hipStreamWaitEvent(L->stream, flags_ready, 0));
        hipLaunchKernelGGL(sqd::k_drop_listed, dim3((n + 255) / 256, 2), dim3(256), 0, L->stream, (u32*)L->b_frame_len.p, d_select, n,
                           (const u32*)L->b_work.p);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(sqd::k_drop_unselected, dim3((n + 255) / 256), dim3(256), 0, L->stream, (u32*)L->b_frame_len.p, d_select, n,
                           (u32*)L->b_misc.p + 33 /* stage 1's count of decided chunks */);
        HIPCHK(hipGetLastError());
    }
#ifdef SQ_PROF
    if (const char* wd = getenv("SQ_WATCHDOG_MS")) {
        // bring-up aid of the PROFILING build only (libsquish_gpu_prof.so; the release library never terminates
        // its host process): if the compressor does not finish in time, dump wave 0..3 progress markers
        const long limit_ms = atol(wd);
        long waited = 0;
        while (hipStreamQuery(L->stream) == hipErrorNotReady && waited < limit_ms) {
            struct timespec ts = {0, 5 * 1000 * 1000};
            nanosleep(&ts, nullptr);
            waited += 5;
        }
        if (hipStreamQuery(L->stream) == hipErrorNotReady) {
            hipStream_t s2;
            (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
            const sqz::WsLayout WL = sqz::ws_layout(L->job_max_len);
            for (u32 w = 0; w < std::min(L->job_waves, 4u); w++) {
                u32 tr[64];
                // per-chunk layouts differ from the pool layout: dump the dummy area of every
                // distinct layout used in this batch (bring-up batches are uniform)
                (void)hipMemcpyAsync(tr, (u8*)L->b_ws.p + (size_t)w * L->job_ws_each + WL.dummy, sizeof(tr),
                                     hipMemcpyDeviceToHost, s2);
                (void)hipStreamSynchronize(s2);
                fprintf(stderr, "[squish_gpu watchdog] wave %u trace:", w);
                for (int k = 8; k < 24; k++) fprintf(stderr, " %u", tr[k]);
                fprintf(stderr, "\n");
            }
            fflush(stderr);
            _exit(3);
        }
    }
#endif
    if (in_place) {
        HIPCHK(hipMemcpyAsync(L->h_len_pinned, L->b_frame_len.p, (size_t)n * 4, hipMemcpyDeviceToHost, L->stream));
    } else {
        hipLaunchKernelGGL(sqd::k_scan_lengths, dim3(1), dim3(1024), 0, L->stream, (const u32*)L->b_frame_len.p,
                           n, (u64*)L->b_out_off.p);
        HIPCHK(hipGetLastError());
    }
    u32* h_err = (u32*)L->h_pinned;
    HIPCHK(hipMemcpyAsync(h_err, (u32*)L->b_misc.p + 16, 4, hipMemcpyDeviceToHost, L->stream));
    HIPCHK(hipMemcpyAsync(h_err + 1, (u32*)L->b_misc.p + 33, 4, hipMemcpyDeviceToHost, L->stream));
    if (L->job_fast) HIPCHK(hipMemcpyAsync(h_err + 2, L->b_work.p, 8, hipMemcpyDeviceToHost, L->stream)); // lengths of stage 2's two lists
    return SQ_OK;
}

static int compress_enqueue(sq_ctx* c, sq_lane* L, const u8* d_data, const u32* h_len, u32 n, const u8* d_select,
                            u8* d_comp, u64 comp_cap, bool in_place) {
    int r;
    if ((r = compress_prepare(c, L, h_len, n, d_comp, comp_cap, in_place, false))) return r;
    if ((r = compress_stage1(c, L, d_data, d_select, false))) return r;
    return compress_stage2(c, L, d_data, d_select);
}

static int compress_finish(sq_ctx* c, sq_lane* L, u8* d_comp, u64 comp_cap, u64* h_comp_off, u32* h_comp_len, bool in_place) {
    (void)c;
    u32 const n = L->job_n;
    if (n == 0) {
        if (!in_place) h_comp_off[0] = 0;
        return SQ_OK;
    }
    u32 const max_len = L->job_max_len, waves = L->job_waves, fwaves = L->job_fwaves;
    size_t const ws_each = L->job_ws_each, etab_bytes = L->job_etab_bytes;
    bool const fast = L->job_fast;
    std::vector<u64> const& slot_off = L->job_slot_off;
    u32* h_err = (u32*)L->h_pinned;
    (void)max_len;
    if (!in_place)
        HIPCHK(hipMemcpyAsync(h_comp_off, L->b_out_off.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, L->stream));
    HIPCHK(hipStreamSynchronize(L->stream));
    if (in_place) {
        memcpy(h_comp_len, L->h_len_pinned, (size_t)n * 4);
        for (u32 i = 0; i < n; i++) h_comp_off[i] = slot_off[i];
    }
    HIPCHK(hipEventElapsedTime(&L->last.compress_ms, L->ev[4], L->ev[5]));
    if (fast) HIPCHK(hipEventElapsedTime(&L->last.fast_ms, L->ev[8], L->ev[9]));
    L->last.fast_decided = fast ? h_err[1] : 0;
    L->last.fused = L->job_fused ? 1u : 0u;
    c->s2_w3_last = L->job_s2w3;
    // (chooses the next batch's build of the kernel: stage 2's lists against what stage 1 decided -- duplicates are in neither)
    c->s2_dense = !fast || (u64)(h_err[2] + h_err[3]) * 3 > (u64)L->last.fast_decided;
    if (c->s2_dense && !c->cfg.compress_waves && !g_env.stage2_waves_per_cu) load_s2w3(c); // (here, behind a collected batch: not on the next batch's way)
    L->work_n = fast ? n : 0;
    if (!*h_err) { // both kernels ran to completion: their saved table states are good for the next launch
        L->ws_ptr_last = L->b_ws.p;
        L->ws_each_last = ws_each;
        if (!L->job_ws_persist) L->ws_waves_last = waves; // (a launch that continued from saved states left the slots above its own as they were)
        if (fast) {
            L->etab_ptr_last = L->b_etab.p;
            L->etab_bytes_last = etab_bytes;
            L->etab_waves_last = fwaves;
        }
    }
#ifdef SQ_PROF_S1 // (stage 1's timers are a switch of their own: with them the fast steps' scalar pointers do not compile)
    if (fast) { // stage 1 phase timers (one 256-byte slice per wave: its saved state, then 8 tick sums)
        std::vector<unsigned long long> tot(8, 0), one(8);
        for (u32 w = 0; w < fwaves; w++) {
            (void)hipMemcpy(one.data(), (u8*)L->b_etab.p + (size_t)w * (etab_bytes + 256) + etab_bytes + 32, 64, hipMemcpyDeviceToHost);
            for (int k = 0; k < 8; k++) tot[k] += one[k];
        }
        static const char* nm1[6] = {"build", "scan_loop", "lookups(in scan)", "hit_resolution", "precheck+frame", "chunk_total"};
        fprintf(stderr, "[squish_gpu prof] stage 1: waves=%u kernel=%.2f ms; per-wave mean ms:", fwaves, L->last.fast_ms);
        for (int k = 0; k < 6; k++) fprintf(stderr, " %s=%.3f", nm1[k], (double)tot[k] / fwaves / 1e5);
        fprintf(stderr, "\n");
    }
#endif
#ifdef SQ_PROF
    {   // profiling build: sum the per-wave phase timers (100 MHz ticks) and print them
        std::vector<unsigned long long> tot(33, 0), one(71), worst(71, 0), worst1(71, 0);
        std::vector<std::pair<unsigned long long, unsigned long long>> longest; // (ticks, pass<<32 | chunk) per wave
        std::vector<unsigned long long> fin;                                    // when each wave finished its last chunk
        for (u32 w = 0; w < waves; w++) {
            (void)hipMemcpy(one.data(), (u8*)L->b_ws.p + (size_t)(w + 1) * ws_each - 1024, 71 * 8, hipMemcpyDeviceToHost);
            for (int k = 0; k < 33; k++)
                if (k < 29 || k > 31) tot[k] += one[k];
            if (one[29] > worst[29]) worst = one;
            if ((one[30] >> 32) == 1 && one[29] > worst1[29]) worst1 = one;
            longest.push_back({one[29], one[30]});
            fin.push_back(one[31]);
        }
        std::sort(longest.begin(), longest.end());
        std::sort(fin.begin(), fin.end());
        fprintf(stderr, "[squish_gpu prof] longest chunk per wave, ms (pass:chunk):");
        for (size_t k = 0; k < 16 && k < longest.size(); k++) {
            auto const& e = longest[longest.size() - 1 - k];
            fprintf(stderr, " %.3f(%u:%u", (double)e.first / 1e5, (unsigned)(e.second >> 32), (unsigned)e.second);
            if (fast) { // where stage 1's certificates ended (positions; 2 = chunk start, 131074 = block 1)
                u32 ce[2] = {0, 0};
                (void)hipMemcpy(ce, (u32*)L->b_cert.p + 2 * (size_t)(u32)e.second, 8, hipMemcpyDeviceToHost);
                fprintf(stderr, " cert %u/%u", ce[0] & SQZ_CERT_HINT_POS_MASK, ce[1] & 0x7FFFFFFFu);
            }
            fprintf(stderr, ")");
        }
        fprintf(stderr, "; median %.3f\n", (double)longest[longest.size() / 2].first / 1e5);
        fprintf(stderr, "[squish_gpu prof] wave finish times ms: min %.3f p10 %.3f median %.3f p90 %.3f p99 %.3f max %.3f\n",
                (double)fin.front() / 1e5, (double)fin[fin.size() / 10] / 1e5, (double)fin[fin.size() / 2] / 1e5,
                (double)fin[fin.size() * 9 / 10] / 1e5, (double)fin[fin.size() * 99 / 100] / 1e5, (double)fin.back() / 1e5);
        static const char* nm[33] = {"frame_total", "table_clear", "dubt_update|eval_setup", "dubt_find|eval_ownerloop", "row_update|rb_prepare", "row_find|rb_eval+exact",
                                     "entropy", "copies", "presplit+filterinit", "epoch_insert|rb_flush", "epoch_predicate|rb_refresh", "epoch_rollback|gather_lits",
                                     "ins_phase1|rep_direct", "ins_phase23|eval_trips", "ins_tail", "bloom_rebuild",
                                     "#tail_iters|#batches", "#bloom_overflow|#exact_searches", "#superbatches|#refreshes", "#candidates|#cand_rounds", "#survivors|#batch_answers", "#nLive_sum|#rep_direct", "#repairs_x1000+eager_x1e6|#pairs", "#exact_finds",
                                     "cert_predict", "tail_certify", "#tail_certs", "cert_scan_p2", "cert_blockwalk", "-", "-", "-", "chunks+1"};
        static const char* nmL[29] = {"frame_total", "table_clear", "dubt_update", "dubt_find", "row_update", "row_find",
                                      "entropy", "copies", "presplit+filterinit", "epoch_insert", "epoch_predicate", "epoch_rollback",
                                      "ins_phase1", "ins_phase23", "ins_tail", "bloom_rebuild",
                                      "#tail_iters", "#bloom_overflow", "#superbatches", "#candidates", "#survivors", "#nLive_sum", "#repairs_x1000+eager_x1e6", "#exact_finds",
                                      "cert_predict", "tail_certify", "#tail_certs", "cert_scan_p2", "cert_blockwalk"};
        for (int which = 0; which < 2; which++) {
            auto const& wv = which ? worst1 : worst;
            fprintf(stderr, "[squish_gpu prof] the longest chunk%s (%u:%u), ms:", which ? " of the second list" : "", (unsigned)(wv[30] >> 32), (unsigned)wv[30]);
            for (int k = 0; k < 29; k++) fprintf(stderr, " %s=%.3f", nmL[k], (k >= 16 && k < 24) ? (double)wv[42 + k] : (double)wv[42 + k] / 1e5);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[squish_gpu prof] waves=%u kernel=%.2f ms; per-wave mean ms:", waves, L->last.compress_ms);
        for (int k = 0; k < 33; k++)
            if (nm[k][0] != '-') fprintf(stderr, " %s=%.3f", nm[k], (k >= 16 && k < 24) || k == 32 ? (double)tot[k] / waves : (double)tot[k] / waves / 1e5);
        fprintf(stderr, "\n");
    }
#endif
    if (g_env.debug_stats && fast) { // certificate ends of the first chunks, as stage 1 left them
        std::vector<u32> ce(2 * (size_t)std::min(n, 24u));
        (void)hipMemcpy(ce.data(), L->b_cert.p, ce.size() * 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[squish_gpu stats] cert ends:");
        for (size_t i = 0; i < ce.size(); i += 2) fprintf(stderr, " %u/%u", ce[i] & SQZ_CERT_HINT_POS_MASK, ce[i + 1] & 0x7FFFFFFFu);
        fprintf(stderr, "\n");
    }
    if (g_env.debug_stats) { // per-wave statistics words of the last chunk each wave compressed
        const sqz::WsLayout WL = sqz::ws_layout(max_len);
        for (u32 w = 0; w < waves && w < 6; w++) {
            u32 tr[64];
            (void)hipMemcpy(tr, (u8*)L->b_ws.p + (size_t)w * ws_each + WL.dummy, sizeof(tr), hipMemcpyDeviceToHost);
            fprintf(stderr, "[squish_gpu stats] wave %u: n=%u eager=%u repairs=%u live=%u skips=%u epochs=%u specfail=%u replays=%u certified=%u dbg=%u,%u,%u,%u\n",
                    w, tr[8 + 1], tr[8 + 30], tr[8 + 31], tr[8 + 32], tr[8 + 33], tr[8 + 34], tr[8 + 35], tr[8 + 36], tr[8 + 37],
                    tr[8 + 40], tr[8 + 41], tr[8 + 42], tr[8 + 43]);
        }
    }
    if (*h_err && g_env.debug) {
        const sqz::WsLayout WL = sqz::ws_layout(max_len);
        int shown = 0;
        for (u32 w = 0; w < waves && shown < 8; w++) {
            u32 tr[64];
            (void)hipMemcpy(tr, (u8*)L->b_ws.p + (size_t)w * ws_each + WL.dummy, sizeof(tr), hipMemcpyDeviceToHost);
            if (tr[8 + 16]) {
                fprintf(stderr, "[squish_gpu fault] wave %u:", w);
                for (int k = 8; k < 8 + 24; k++) fprintf(stderr, " %u", tr[k]);
                fprintf(stderr, "\n");
                shown++;
            }
        }
    }
    if (*h_err) return SQ_E_COMPRESS;
    if (in_place) {
        L->last.compact_ms = 0;
        return SQ_OK;
    }
    if (h_comp_off[n] > comp_cap) return SQ_E_CAP;
    if (L->job_host) { // the packing pass was enqueued with the kernels (sq_pack_submit, host buffers)
        HIPCHK(hipEventElapsedTime(&L->last.compact_ms, L->ev[6], L->ev[7]));
        return SQ_OK;
    }
    HIPCHK(hipEventRecord(L->ev[6], L->stream));
    {
        const u32 wpb = 4;
        dim3 grid((n + wpb - 1) / wpb), block(64 * wpb);
        hipLaunchKernelGGL(sqd::k_compact_frames, grid, block, 0, L->stream, (const u8*)L->b_slots.p,
                           (const u64*)L->b_slot_off.p, (const u32*)L->b_frame_len.p,
                           (const u64*)L->b_out_off.p, n, d_comp);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(L->ev[7], L->stream));
    HIPCHK(hipStreamSynchronize(L->stream));
    HIPCHK(hipEventElapsedTime(&L->last.compact_ms, L->ev[6], L->ev[7]));
    return SQ_OK;
}

static int compress_async(sq_ctx* c, sq_lane* L, const u8* d_data, const u64* h_off, const u32* h_len, u32 n,
                          const u8* d_select, u8* d_comp, u64 comp_cap, u64* h_comp_off, u32* h_comp_len) {
    (void)h_off;
    c->last_lane = 0;
    L->job_host = false;
    bool const in_place = h_comp_len != nullptr;
    int const r = compress_enqueue(c, L, d_data, h_len, n, d_select, d_comp, comp_cap, in_place);
    if (r) return r;
    return compress_finish(c, L, d_comp, comp_cap, h_comp_off, h_comp_len, in_place);
}
#endif

int sq_compress_batch(sq_ctx* c, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len,
                      uint32_t n, const uint8_t* d_select, uint8_t* d_comp, uint64_t comp_cap,
                      uint64_t* h_comp_off) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)d_data; (void)h_off; (void)h_len; (void)n; (void)d_select; (void)d_comp;
    (void)comp_cap; (void)h_comp_off;
    return SQ_E_COMPRESS;
#else
    if (!c || !h_comp_off || (n && (!d_data || !h_off || !h_len || !d_comp))) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r;
    if ((r = upload_off_len(c, L, h_off, h_len, n))) return r;
    return compress_async(c, L, d_data, h_off, h_len, n, d_select, d_comp, comp_cap, h_comp_off, nullptr);
#endif
}

int sq_compress_batch_slots(sq_ctx* c, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len,
                            uint32_t n, const uint8_t* d_select, uint8_t* d_comp, uint64_t comp_cap,
                            uint64_t* h_comp_off, uint32_t* h_comp_len) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)d_data; (void)h_off; (void)h_len; (void)n; (void)d_select; (void)d_comp;
    (void)comp_cap; (void)h_comp_off; (void)h_comp_len;
    return SQ_E_COMPRESS;
#else
    if (!c || (n && (!d_data || !h_off || !h_len || !d_comp || !h_comp_off || !h_comp_len))) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    int r;
    if ((r = upload_off_len(c, L, h_off, h_len, n))) return r;
    return compress_async(c, L, d_data, h_off, h_len, n, d_select, d_comp, comp_cap, h_comp_off, h_comp_len);
#endif
}

int sq_debug_stage2_kernel(sq_ctx* c, uint32_t* loaded, uint32_t* used_last) {
    if (!c || !loaded || !used_last) return SQ_E_INVAL;
    load_s2w3(c);
    *loaded = c->f_s2w3 != nullptr;
    *used_last = c->s2_w3_last;
    return SQ_OK;
}

int sq_debug_stage2_list(sq_ctx* c, uint32_t* h_out, uint32_t cap, uint32_t* n_out) {
    if (!c || !n_out || (cap && !h_out)) return SQ_E_INVAL;
    sq_lane* const L = &c->ln[c->last_lane & 1];
    *n_out = 0;
    if (!L->work_n) return SQ_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(L->stream));
    u32 cnt[2] = {0, 0};
    HIPCHK(hipMemcpy(cnt, L->b_work.p, 8, hipMemcpyDeviceToHost));
    u32 k = 0;
    for (int pass = 0; pass < 2; pass++) {
        u32 const take = std::min(cnt[pass], cap - k);
        if (take)
            HIPCHK(hipMemcpy(h_out + k, (u32*)L->b_work.p + 2 + (pass ? L->work_n : 0u), (size_t)take * 4, hipMemcpyDeviceToHost));
        k += take;
    }
    *n_out = k;
    return SQ_OK;
}

uint64_t sq_slots_bound(const uint32_t* h_len, uint32_t n) {
    u64 acc = 0;
    for (u32 i = 0; i < n; i++) acc += (sq_compress_bound(h_len[i]) + 32 + 15) & ~(u64)15;
    return acc;
}

#ifndef SQ_NO_COMPRESS
// hash -> dedup -> compress of one batch of device-resident chunks, ENQUEUED on lane L (off / len already uploaded).
// Two forms, chosen per batch (DESIGN.md 4.2 "fused hash"):
//   split: XXH3 kernel -> dedup -> stage 1 and stage 2 on the new chunks              (two reads of every new chunk)
//   fused: stage 1 on EVERY chunk, hashing what it streams -> XXH3 kernel for the chunks stage 1 did not stream ->
//          dedup -> duplicates' frames dropped -> stage 2 on the new chunks stage 1 left undecided (one read; the
//          certificate and raw frame of a duplicate are wasted work, so this is for batches with few duplicates)
// Everything that can fail for lack of memory or capacity is checked before the shared dedup set is touched;
// *committed says whether it was (a caller that promises "nothing is committed on failure" rolls back then).
// other: the context's other lane when batches may be in flight on it (its dedup kernels come first), else null.
static int pack_enqueue(sq_ctx* c, sq_lane* L, sq_lane* other, const u8* d_data, const u32* h_len, u32 n, u8* d_hashes,
                        u8* d_isnew, u8* d_comp, u64 comp_cap, bool in_place, u32 flags, bool* committed) {
    int r;
    *committed = false;
    int const mode = (c->cfg.flags & SQ_CFG_ALWAYS_FUSE_HASH) ? 1 : (c->cfg.flags & SQ_CFG_NEVER_FUSE_HASH) ? 0 : g_env.fuse_hash;
    // (the crossover, measured: tools/bench_dupratio.py, profiles/round3_fused_vs_split_by_duplicates.txt)
    bool const fused = mode == 1 || (mode < 0 && c->dup_ratio_last < 0.20f);
    if ((r = compress_prepare(c, L, h_len, n, d_comp, comp_cap, in_place, fused))) return r;
    u64 const seen0 = (flags & 1u) ? 0 : c->n_seen;
    if (seen0 + n > c->key_capacity || seen0 + n >= 0xFFFFFFFFull) return SQ_E_CAP;
    // Experiment (SQ_S2_EARLY=1, off by default: measured, no gain): in the fused form with batches in flight stage 2
    // needs nothing of the dedup decision but which chunks to skip, and duplicates are few -- so it can be launched right
    // behind stage 1, on every chunk stage 1 left undecided, while the hash finish and the dedup kernels run beside it on
    // the lane's second stream; the duplicates' frames are then dropped behind stage 2 (compress_stage2).
    bool const early_s2 = L->job_fused && other != nullptr && g_env.s2_early;
    hipStream_t const side = early_s2 ? L->aux : L->stream;
    if (L->job_fused) {
        // Stage 1's persistent workgroups take every wave slot of the GPU the moment the previous batch's stage 1
        // retires -- before that batch's stage 2 has even been launched (its hashes and dedup come first), which then
        // runs alone AFTER this kernel.  Holding this launch back until the other lane's stage 2 is on its way lets
        // stage 2's waves become resident first and this kernel fill the slots as they retire.
        if (other && other->busy && other->job_fused && g_env.s1_after_s2) HIPCHK(hipStreamWaitEvent(L->stream, other->pre_s2, 0));
        if ((r = compress_stage1(c, L, d_data, nullptr, true))) return r;
        if (early_s2) {
            HIPCHK(hipEventRecord(L->s1_done, L->stream));
            HIPCHK(hipStreamWaitEvent(side, L->s1_done, 0));
        }
        if ((r = launch_hash(c, L, d_data, n, d_hashes, (const u8*)L->b_hashed.p, (const u8*)L->b_hpool.p, side))) return r;
    } else if ((r = launch_hash(c, L, d_data, n, d_hashes))) // (does not depend on the batch before)
        return r;
    // the dedup set is shared: this batch's reset / insert must follow the previous batch's dedup kernels
    if (other && other->busy) HIPCHK(hipStreamWaitEvent(side, other->dedup_done, 0));
    *committed = true;
    if (flags & 1u) {
        HIPCHK(hipMemsetAsync(c->d_table, 0, (c->table_mask + 1) * sizeof(u32), side));
        HIPCHK(hipMemsetAsync(c->d_unique, 0, 64, side));
        c->n_seen = 0;
    }
    if (!other) // (the synchronous form: what a roll-back restores)
        HIPCHK(hipMemcpyAsync(c->d_unique + 1, c->d_unique, sizeof(unsigned long long), hipMemcpyDeviceToDevice, L->stream));
    if ((r = dedup_async(c, L, d_hashes, n, 0, n, d_isnew, side))) return r;
    HIPCHK(hipEventRecord(L->dedup_done, side));
    if (!L->job_fused && (r = compress_stage1(c, L, d_data, d_isnew, false))) return r;
    if (L->job_fused) HIPCHK(hipEventRecord(L->pre_s2, L->stream));
    return compress_stage2(c, L, d_data, d_isnew, early_s2 ? L->dedup_done : nullptr);
}

static void note_dup_ratio(sq_ctx* c, u32 n, u32 n_new) {
    if (n) c->dup_ratio_last = 1.f - (float)n_new / (float)n;
}
#endif

int sq_pack_batch(sq_ctx* c, const uint8_t* data, const uint64_t* off, const uint32_t* len, uint32_t n,
                  uint8_t* hashes, uint8_t* is_new, uint8_t* comp, uint64_t comp_cap,
                  uint64_t* comp_off) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)data; (void)off; (void)len; (void)n; (void)hashes; (void)is_new; (void)comp;
    (void)comp_cap; (void)comp_off;
    return SQ_E_COMPRESS;
#else
    if (!c || !comp_off || (n && (!data || !off || !len || !hashes || !is_new || !comp))) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL; // batches of sq_pack_submit are in flight: collect them first
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) {
        comp_off[0] = 0;
        return SQ_OK;
    }
    // all four data-plane pointers must live on the same side
    const bool dev = is_device_ptr(data);
    if (is_device_ptr(hashes) != dev || is_device_ptr(is_new) != dev || is_device_ptr(comp) != dev) return SQ_E_INVAL;
    // every argument check that can fail runs BEFORE anything is committed to the dedup set
    for (u32 i = 0; i < n; i++)
        if (len[i] > c->cfg.max_chunk) return SQ_E_INVAL;
    int r;
    const u8* d_data = data;
    u8 *d_hashes = hashes, *d_isnew = is_new, *d_comp = comp;
    std::vector<u64> off2;
    const u64* use_off = off;
    if (!dev) {
        // stage: pack chunks at 256-byte aligned offsets into the ctx's input buffer
        off2.resize(n);
        u64 acc = 0;
        for (u32 i = 0; i < n; i++) {
            off2[i] = acc;
            acc += ((u64)len[i] + 255) & ~(u64)255;
        }
        if ((r = L->b_stage_in.ensure(acc + 256))) return r;
        if ((r = L->b_hashes.ensure((size_t)n * 16))) return r;
        if ((r = L->b_isnew.ensure(n))) return r;
        if ((r = L->b_stage_out.ensure(comp_cap ? comp_cap : 16))) return r;
        bool span = true; // the host batch already has the staging layout: one transfer
        for (u32 i = 0; i < n && span; i++) span = (off[i] - off[0]) == off2[i];
        if (span)
            HIPCHK(hipMemcpyAsync(L->b_stage_in.p, data + off[0], off2[n - 1] + len[n - 1], hipMemcpyHostToDevice, L->stream));
        else
            for (u32 i = 0; i < n; i++)
                HIPCHK(hipMemcpyAsync((u8*)L->b_stage_in.p + off2[i], data + off[i], len[i],
                                      hipMemcpyHostToDevice, L->stream));
        d_data = (const u8*)L->b_stage_in.p;
        d_hashes = (u8*)L->b_hashes.p;
        d_isnew = (u8*)L->b_isnew.p;
        d_comp = (u8*)L->b_stage_out.p;
        use_off = off2.data();
    }
    if ((r = upload_off_len(c, L, use_off, len, n))) return r;
    // The dedup commit is provisional until the frames are out: if the compress stage fails (SQ_E_CAP: the
    // frames do not fit comp_cap; SQ_E_COMPRESS; a launch failure), the set, its unique count and the global
    // chunk index are put back, so the caller can retry the same batch (squish_gpu.h: "nothing is committed").
    const u64 seen_before = c->n_seen;
    bool committed = false;
    c->last_lane = 0;
    L->job_host = false;
    r = pack_enqueue(c, L, nullptr, d_data, len, n, d_hashes, d_isnew, d_comp, comp_cap, false, 0, &committed);
    if (!r) r = compress_finish(c, L, d_comp, comp_cap, comp_off, nullptr, false);
    if (r) {
        if (committed) {
            const u64 nslots = c->table_mask + 1;
            hipLaunchKernelGGL(sqd::k_dedup_rollback, dim3((unsigned)std::min<u64>((nslots + 255) / 256, 4096)), dim3(256), 0,
                               L->stream, c->d_table, nslots, seen_before);
            (void)hipMemcpyAsync(c->d_unique, c->d_unique + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, L->stream);
        }
        (void)hipStreamSynchronize(L->stream);
        c->n_seen = seen_before;
        return r;
    }
    HIPCHK(hipEventElapsedTime(&L->last.hash_ms, L->ev[0], L->ev[1]));
    HIPCHK(hipEventElapsedTime(&L->last.dedup_ms, L->ev[2], L->ev[3]));
    {
        u32 n_new = 0;
        for (u32 i = 0; i < n; i++) n_new += comp_off[i + 1] != comp_off[i];
        note_dup_ratio(c, n, n_new);
    }
    if (!dev) {
        HIPCHK(hipMemcpyAsync(hashes, d_hashes, (size_t)n * 16, hipMemcpyDeviceToHost, L->stream));
        HIPCHK(hipMemcpyAsync(is_new, d_isnew, n, hipMemcpyDeviceToHost, L->stream));
        HIPCHK(hipMemcpyAsync(comp, d_comp, comp_off[n], hipMemcpyDeviceToHost, L->stream));
        HIPCHK(hipStreamSynchronize(L->stream));
    }
    return SQ_OK;
#endif
}

// ---- pipelined form: two batches in flight on the context's two lanes -------------------------------------------
int sq_pack_submit(sq_ctx* c, const uint8_t* data, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                   uint8_t* hashes, uint8_t* is_new, uint8_t* comp, uint64_t comp_cap, uint32_t flags) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)data; (void)h_off; (void)h_len; (void)n; (void)hashes; (void)is_new; (void)comp; (void)comp_cap; (void)flags;
    return SQ_E_COMPRESS;
#else
    if (!c || (n && (!data || !h_off || !h_len || !hashes || !is_new || !comp))) return SQ_E_INVAL;
    if (c->submitted - c->collected >= 2 || c->shard_pending) return SQ_E_INVAL; // both lanes busy: collect one with sq_pack_wait first
    HIPCHK(hipSetDevice(c->device));
    for (u32 i = 0; i < n; i++)
        if (h_len[i] > c->cfg.max_chunk) return SQ_E_INVAL;
    bool const dev = n ? is_device_ptr(data) : true;
    if (n && (is_device_ptr(hashes) != dev || is_device_ptr(is_new) != dev || is_device_ptr(comp) != dev)) return SQ_E_INVAL;
    if (dev && sq_slots_bound(h_len, n) > comp_cap) return SQ_E_CAP;
    sq_lane* const L = &c->ln[c->submitted & 1];
    sq_lane* const other = &c->ln[(c->submitted & 1) ^ 1];
    int r;
    const u8* d_data = data;
    u8 *d_hashes = hashes, *d_isnew = is_new, *d_comp = comp;
    std::vector<u64> off2;
    const u64* use_off = h_off;
    L->job_host = !dev;
    if (!dev) {
        // host buffers: one H2D into the lane's staging buffer (chunks at 256-byte aligned starts), frames produced
        // into the lane's slots and packed back to back on the device; sq_pack_wait brings the results back.  On
        // the lane's own stream all of this overlaps the other lane's kernels.
        off2.resize(n);
        u64 acc = 0, bound = 0;
        for (u32 i = 0; i < n; i++) {
            off2[i] = acc;
            acc += ((u64)h_len[i] + 255) & ~(u64)255;
            bound += sq_compress_bound(h_len[i]) + 16;
        }
        if ((r = L->b_stage_in.ensure(acc + 256))) return r;
        if ((r = L->b_hashes.ensure((size_t)n * 16 + 16))) return r;
        if ((r = L->b_isnew.ensure((size_t)n + 16))) return r;
        if ((r = L->b_stage_out.ensure(bound + 64))) return r;
        bool span = true;
        for (u32 i = 0; i < n && span; i++) span = (h_off[i] - h_off[0]) == off2[i];
        if (span && n)
            HIPCHK(hipMemcpyAsync(L->b_stage_in.p, data + h_off[0], off2[n - 1] + h_len[n - 1], hipMemcpyHostToDevice, L->stream));
        else
            for (u32 i = 0; i < n; i++)
                HIPCHK(hipMemcpyAsync((u8*)L->b_stage_in.p + off2[i], data + h_off[i], h_len[i], hipMemcpyHostToDevice, L->stream));
        d_data = (const u8*)L->b_stage_in.p;
        d_hashes = (u8*)L->b_hashes.p;
        d_isnew = (u8*)L->b_isnew.p;
        d_comp = (u8*)L->b_stage_out.p;
        use_off = off2.data();
        L->job_h_hashes = hashes;
        L->job_h_isnew = is_new;
        L->job_h_comp = comp;
        L->job_comp_cap = comp_cap;
    }
    if ((r = upload_off_len(c, L, use_off, h_len, n))) return r;
    bool committed = false;
    if ((r = pack_enqueue(c, L, other, d_data, h_len, n, d_hashes, d_isnew, d_comp, dev ? comp_cap : ~0ull, dev, flags, &committed))) {
        (void)hipStreamSynchronize(L->stream);
        return r; // (committed: squish_gpu.h -- sq_dedup_reset before the context is used again)
    }
    if (!dev && n) { // the packing pass, on the device: slots -> staging buffer, offsets from the scan
        HIPCHK(hipEventRecord(L->ev[6], L->stream));
        const u32 wpb = 4;
        hipLaunchKernelGGL(sqd::k_compact_frames, dim3((n + wpb - 1) / wpb), dim3(64 * wpb), 0, L->stream, (const u8*)L->b_slots.p,
                           (const u64*)L->b_slot_off.p, (const u32*)L->b_frame_len.p, (const u64*)L->b_out_off.p, n, d_comp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(L->ev[7], L->stream));
    }
    L->busy = true;
    L->job_sharded = false;
    c->submitted++;
    return SQ_OK;
#endif
}

// ---- sharded pipelined form: the caller's exchange of hashes sits between the hash and the rest of a batch --------
int sq_shard_submit_hash(sq_ctx* c, const uint8_t* d_data, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                         uint8_t* d_hashes) {
    if (!c || (n && (!d_data || !h_off || !h_len || !d_hashes))) return SQ_E_INVAL;
    if (c->shard_pending || c->submitted - c->collected >= 2) return SQ_E_INVAL;
    if (n && (!is_device_ptr(d_data) || !is_device_ptr(d_hashes))) return SQ_E_INVAL;
    for (u32 i = 0; i < n; i++)
        if (h_len[i] > c->cfg.max_chunk) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(c->device));
    sq_lane* const L = &c->ln[c->submitted & 1];
    int r;
    if ((r = upload_off_len(c, L, h_off, h_len, n))) return r;
    if ((r = launch_hash(c, L, d_data, n, d_hashes))) return r;
    if (n == 0) HIPCHK(hipEventRecord(L->ev[1], L->stream));
    L->shard_data = d_data;
    L->shard_len.assign(h_len, h_len + n);
    L->job_host = false;
    c->shard_pending = true;
    return SQ_OK;
}

int sq_shard_wait_hash(sq_ctx* c) {
    if (!c || !c->shard_pending) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ln[c->submitted & 1].ev[1]));
    return SQ_OK;
}

#ifndef SQ_NO_COMPRESS
static int shard_submit_rest(sq_ctx* c, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first, bool by_idx,
                             const uint32_t* d_my_idx, uint8_t* d_is_new, uint8_t* d_comp, uint64_t comp_cap, uint32_t flags);
#endif

int sq_shard_submit_rest(sq_ctx* c, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first, uint8_t* d_is_new,
                         uint8_t* d_comp, uint64_t comp_cap, uint32_t flags) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)d_all_hashes; (void)n_total; (void)my_first; (void)d_is_new; (void)d_comp; (void)comp_cap; (void)flags;
    return SQ_E_COMPRESS;
#else
    return shard_submit_rest(c, d_all_hashes, n_total, my_first, false, nullptr, d_is_new, d_comp, comp_cap, flags);
#endif
}

int sq_shard_submit_rest_idx(sq_ctx* c, const uint8_t* d_all_hashes, uint64_t n_total, const uint32_t* d_my_idx, uint8_t* d_is_new,
                             uint8_t* d_comp, uint64_t comp_cap, uint32_t flags) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)d_all_hashes; (void)n_total; (void)d_my_idx; (void)d_is_new; (void)d_comp; (void)comp_cap; (void)flags;
    return SQ_E_COMPRESS;
#else
    return shard_submit_rest(c, d_all_hashes, n_total, 0, true, d_my_idx, d_is_new, d_comp, comp_cap, flags);
#endif
}

#ifndef SQ_NO_COMPRESS
// by_idx false: this rank's chunks are the contiguous range [my_first, my_first + n) of the global order; true: chunk i
// of this rank is entry d_my_idx[i] of it (any assignment of chunks to ranks, e.g. block-cyclic; d_my_idx may be null
// when the rank holds no chunk of the batch).
// A call that finds a pending hash step CONSUMES it whatever it returns (squish_gpu.h): after a failure the caller starts
// the batch again with sq_shard_submit_hash.
static int shard_submit_rest(sq_ctx* c, const uint8_t* d_all_hashes, uint64_t n_total, uint64_t my_first, bool by_idx,
                             const uint32_t* d_my_idx, uint8_t* d_is_new, uint8_t* d_comp, uint64_t comp_cap, uint32_t flags) {
    if (!c || !c->shard_pending) return SQ_E_INVAL;
    c->shard_pending = false; // (before the first check that can fail)
    sq_lane* const L = &c->ln[c->submitted & 1];
    sq_lane* const other = &c->ln[(c->submitted & 1) ^ 1];
    u32 const n = (u32)L->shard_len.size();
    if (by_idx && n && (!d_my_idx || !is_device_ptr(d_my_idx))) return SQ_E_INVAL;
    if ((n_total && !d_all_hashes) || my_first + n > n_total || (n && (!d_is_new || !d_comp))) return SQ_E_INVAL;
    if (n && (!is_device_ptr(d_all_hashes) || !is_device_ptr(d_is_new) || !is_device_ptr(d_comp))) return SQ_E_INVAL;
    if (by_idx && n_total >= 0xFFFFFFFFull) return SQ_E_INVAL;
    if (sq_slots_bound(L->shard_len.data(), n) > comp_cap) return SQ_E_CAP;
    HIPCHK(hipSetDevice(c->device));
    int r;
    // every allocation and capacity check of the batch comes BEFORE the replicated dedup set is touched: a failure
    // here leaves this rank's set as the other ranks' (squish_gpu.h)
    {
        u64 const seen0 = (flags & 1u) ? 0 : c->n_seen;
        if (seen0 + n_total > c->key_capacity || seen0 + n_total >= 0xFFFFFFFFull) return SQ_E_CAP;
    }
    if ((r = compress_prepare(c, L, L->shard_len.data(), n, d_comp, comp_cap, true, false))) return r;
    if (by_idx && (r = L->b_isnew.ensure((size_t)n_total + 16))) return r;
    // the dedup set is shared: this batch's reset / insert must follow the previous batch's dedup kernels
    if (other->busy) HIPCHK(hipStreamWaitEvent(L->stream, other->dedup_done, 0));
    if (flags & 1u) {
        HIPCHK(hipMemsetAsync(c->d_table, 0, (c->table_mask + 1) * sizeof(u32), L->stream));
        HIPCHK(hipMemsetAsync(c->d_unique, 0, 64, L->stream));
        c->n_seen = 0;
    }
    if (by_idx) { // flags of ALL chunks of the global order, then this rank's picked out of them
        if ((r = dedup_async(c, L, d_all_hashes, n_total, 0, (u32)n_total, (u8*)L->b_isnew.p))) return r;
        if (n) {
            hipLaunchKernelGGL(sqd::k_pick_flags, dim3((n + 255) / 256), dim3(256), 0, L->stream, (const u8*)L->b_isnew.p, d_my_idx, n,
                               (u32)n_total, d_is_new);
            HIPCHK(hipGetLastError());
        }
    } else if ((r = dedup_async(c, L, d_all_hashes, n_total, my_first, n, d_is_new))) // (SQ_E_CAP: fails before touching the set)
        return r;
    HIPCHK(hipEventRecord(L->dedup_done, L->stream));
    if ((r = compress_stage1(c, L, L->shard_data, d_is_new, false)) || (r = compress_stage2(c, L, L->shard_data, d_is_new))) {
        (void)hipStreamSynchronize(L->stream);
        return r;
    }
    L->busy = true;
    L->job_sharded = true; // (this rank's share of a global batch: its new-ratio says nothing about the fused form's crossover)
    c->submitted++;
    return SQ_OK;
}
#endif

int sq_pack_wait(sq_ctx* c, uint64_t* h_comp_off, uint32_t* h_comp_len) {
#ifdef SQ_NO_COMPRESS
    (void)c; (void)h_comp_off; (void)h_comp_len;
    return SQ_E_COMPRESS;
#else
    if (!c || c->submitted == c->collected) return SQ_E_INVAL;
    sq_lane* const L = &c->ln[c->collected & 1];
    u32 const n = L->job_n;
    if (n && (!h_comp_off || !h_comp_len)) return SQ_E_INVAL;
    HIPCHK(hipSetDevice(c->device));
    int r;
    if (!L->job_host) {
        r = compress_finish(c, L, nullptr, 0, h_comp_off, h_comp_len, true);
    } else {
        L->job_out_off.resize((size_t)n + 1);
        r = compress_finish(c, L, (u8*)L->b_stage_out.p, L->job_comp_cap, L->job_out_off.data(), nullptr, false);
        if (!r && n) {
            u64 const total = L->job_out_off[n];
            r = [&]() -> int {
                HIPCHK(hipMemcpyAsync(L->job_h_hashes, L->b_hashes.p, (size_t)n * 16, hipMemcpyDeviceToHost, L->stream));
                HIPCHK(hipMemcpyAsync(L->job_h_isnew, L->b_isnew.p, n, hipMemcpyDeviceToHost, L->stream));
                if (total) HIPCHK(hipMemcpyAsync(L->job_h_comp, L->b_stage_out.p, total, hipMemcpyDeviceToHost, L->stream));
                HIPCHK(hipStreamSynchronize(L->stream));
                return SQ_OK;
            }();
            for (u32 i = 0; i < n; i++) {
                h_comp_off[i] = L->job_out_off[i];
                h_comp_len[i] = (u32)(L->job_out_off[i + 1] - L->job_out_off[i]);
            }
        }
    }
    // Also for an empty batch: the lane may still have dedup work queued for it (the sharded form commits every rank's
    // hashes; a reset clears the table), and the next submit on the other lane only orders itself behind a BUSY lane.
    if (hipStreamSynchronize(L->stream) != hipSuccess && !r) r = SQ_E_DEVICE;
    L->busy = false;
    c->collected++;
    if (r) return r;
    if (n && !L->job_sharded) {
        u32 n_new = 0;
        for (u32 i = 0; i < n; i++) n_new += h_comp_len[i] != 0;
        note_dup_ratio(c, n, n_new);
    }
    if (n) {
        HIPCHK(hipEventElapsedTime(&L->last.hash_ms, L->ev[0], L->ev[1]));
        HIPCHK(hipEventElapsedTime(&L->last.dedup_ms, L->ev[2], L->ev[3]));
    }
    c->last_lane = (u32)(L - c->ln);
    c->ln[0].last = L->last; // sq_last_timings reports the batch collected last
    return SQ_OK;
#endif
}

// ---- unpack side: n x zstd::bulk::decompress (reader.rs:259-314) ----------------------------------------------------------
const char* sq_decompress_strerror(uint32_t status) {
    switch (status) {
    case SQ_DZ_OK: return "ok";
    case SQ_DZ_CORRUPT: return "Data corruption detected";
    case SQ_DZ_SRCSIZE: return "Src size is incorrect";
    case SQ_DZ_UNKNOWN: return "Unknown frame descriptor";
    case SQ_DZ_UNSUPPORTED: return "Unsupported frame parameter";
    case SQ_DZ_DICT: return "Dictionary mismatch";
    case SQ_DZ_DSTSIZE: return "Destination buffer is too small";
    case SQ_DZ_CHECKSUM: return "Restored data doesn't match checksum";
    default: return "unknown error";
    }
}

int sq_decompress_batch(sq_ctx* c, const uint8_t* frames, const uint64_t* h_off, const uint32_t* h_len, uint32_t n,
                        uint8_t* out, const uint64_t* h_out_off, const uint32_t* h_out_cap, uint32_t* h_out_len,
                        uint32_t* h_status) {
    if (!c || (n && (!frames || !h_off || !h_len || !out || !h_out_off || !h_out_cap || !h_out_len || !h_status))) return SQ_E_INVAL;
    if (c->submitted != c->collected || c->shard_pending) return SQ_E_INVAL;
    if (n == 0) return SQ_OK;
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    bool const dev = is_device_ptr(frames);
    if (dev != is_device_ptr(out)) return SQ_E_INVAL;
    u64 in_end = 0, out_end = 0;
    for (u32 i = 0; i < n; i++) {
        in_end = std::max<u64>(in_end, h_off[i] + h_len[i]);
        out_end = std::max<u64>(out_end, h_out_off[i] + h_out_cap[i]);
    }
    int r;
    const u8* d_in = frames;
    u8* d_out = out;
    if (!dev) { // staged: one copy in, one copy out (the spans the records cover)
        if ((r = L->b_dz_in.ensure((size_t)in_end + 64))) return r;
        if ((r = L->b_dz_out.ensure((size_t)out_end + 64))) return r;
        HIPCHK(hipMemcpyAsync(L->b_dz_in.p, frames, (size_t)in_end, hipMemcpyHostToDevice, L->stream));
        d_in = (const u8*)L->b_dz_in.p;
        d_out = (u8*)L->b_dz_out.p;
    }
    u32 const waves = std::min<u32>(n, (u32)c->num_cu * 12u);
    if ((r = L->b_off.ensure((size_t)n * 8)) || (r = L->b_len.ensure((size_t)n * 4)) || (r = L->b_dz_ooff.ensure((size_t)n * 8)) ||
        (r = L->b_dz_ocap.ensure((size_t)n * 4)) || (r = L->b_dz_olen.ensure((size_t)n * 4)) || (r = L->b_dz_stat.ensure((size_t)n * 4)) ||
        (r = L->b_dz_lit.ensure((size_t)waves * SQDZ_LITBUF_BYTES)) || (r = L->b_misc.ensure(256)))
        return r;
    HIPCHK(hipMemcpyAsync(L->b_off.p, h_off, (size_t)n * 8, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemcpyAsync(L->b_len.p, h_len, (size_t)n * 4, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemcpyAsync(L->b_dz_ooff.p, h_out_off, (size_t)n * 8, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemcpyAsync(L->b_dz_ocap.p, h_out_cap, (size_t)n * 4, hipMemcpyHostToDevice, L->stream));
    HIPCHK(hipMemsetAsync(L->b_misc.p, 0, 256, L->stream));
    HIPCHK(hipEventRecord(L->ev[6], L->stream));
    hipLaunchKernelGGL(sqdz::k_decompress_batch, dim3(waves), dim3(64), 0, L->stream, d_in, (const u64*)L->b_off.p, (const u32*)L->b_len.p, n,
                       d_out, (const u64*)L->b_dz_ooff.p, (const u32*)L->b_dz_ocap.p, (u32*)L->b_dz_olen.p, (u32*)L->b_dz_stat.p,
                       (u8*)L->b_dz_lit.p, (u32*)L->b_misc.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(L->ev[7], L->stream));
    HIPCHK(hipMemcpyAsync(h_out_len, L->b_dz_olen.p, (size_t)n * 4, hipMemcpyDeviceToHost, L->stream));
    HIPCHK(hipMemcpyAsync(h_status, L->b_dz_stat.p, (size_t)n * 4, hipMemcpyDeviceToHost, L->stream));
    if (!dev) HIPCHK(hipMemcpyAsync(out, L->b_dz_out.p, (size_t)out_end, hipMemcpyDeviceToHost, L->stream));
    HIPCHK(hipStreamSynchronize(L->stream));
    HIPCHK(hipEventElapsedTime(&L->last.compact_ms, L->ev[6], L->ev[7])); // (sq_last_timings: the decode kernel's time is reported in compact_ms)
    for (u32 i = 0; i < n; i++)
        if (h_status[i]) return SQ_E_DECOMPRESS;
    return SQ_OK;
}

void* sq_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void sq_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// ---- stateless convenience forms --------------------------------------------------------------
// The reference calls hash_chunk / insert from up to 25 Rayon workers (src/util/chunk.rs:80, cmd/mod.rs:16): the
// convenience forms share ONE lazily created context (its staging buffers, stream and events), so they are
// serialised by a mutex -- thread-safe, not concurrent.  Callers that want concurrency use one sq_ctx per thread.
static sq_ctx* g_default_ctx = nullptr;
static std::mutex g_default_mu;
static int default_ctx(sq_ctx** out) {
    if (!g_default_ctx) {
        sq_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.abi_version = SQ_ABI_VERSION;
        cfg.dedup_capacity = 1024;
        cfg.compress_waves = 1;
        int r = sq_ctx_create(&cfg, &g_default_ctx);
        if (r) return r;
    }
    *out = g_default_ctx;
    return SQ_OK;
}

int sq_hash_chunk(const uint8_t* chunk, size_t len, uint8_t out[SQ_HASH_BYTES]) {
    if ((!chunk && len) || !out || len > SQ_CHUNK_SIZE) return SQ_E_INVAL;
    std::lock_guard<std::mutex> lk(g_default_mu);
    sq_ctx* c;
    int r = default_ctx(&c);
    if (r) return r;
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    if ((r = L->b_stage_in.ensure(len + 256))) return r;
    if ((r = L->b_hashes.ensure(16))) return r;
    if (len) HIPCHK(hipMemcpyAsync(L->b_stage_in.p, chunk, len, hipMemcpyHostToDevice, L->stream));
    u64 off = 0;
    u32 l32 = (u32)len;
    if ((r = sq_hash_batch(c, (const u8*)L->b_stage_in.p, &off, &l32, 1, (u8*)L->b_hashes.p))) return r;
    HIPCHK(hipMemcpy(out, L->b_hashes.p, 16, hipMemcpyDeviceToHost));
    return SQ_OK;
}

int sq_compress_chunk(const uint8_t* chunk, size_t len, uint8_t* dst, size_t cap, size_t* out_len) {
#ifdef SQ_NO_COMPRESS
    (void)chunk; (void)len; (void)dst; (void)cap; (void)out_len;
    return SQ_E_COMPRESS;
#else
    if ((!chunk && len) || !dst || !out_len || len > SQ_CHUNK_SIZE) return SQ_E_INVAL;
    std::lock_guard<std::mutex> lk(g_default_mu);
    sq_ctx* c;
    int r = default_ctx(&c);
    if (r) return r;
    sq_lane* const L = &c->ln[0];
    HIPCHK(hipSetDevice(c->device));
    const size_t bound = sq_compress_bound(len) + 64;
    if ((r = L->b_stage_in.ensure(len + 256))) return r;
    if ((r = L->b_stage_out.ensure(bound))) return r;
    if (len) HIPCHK(hipMemcpyAsync(L->b_stage_in.p, chunk, len, hipMemcpyHostToDevice, L->stream));
    u64 off = 0, coff[2] = {0, 0};
    u32 l32 = (u32)len;
    if ((r = sq_compress_batch(c, (const u8*)L->b_stage_in.p, &off, &l32, 1, nullptr,
                               (u8*)L->b_stage_out.p, bound, coff)))
        return r;
    if (coff[1] > cap) return SQ_E_CAP;
    HIPCHK(hipMemcpy(dst, L->b_stage_out.p, coff[1], hipMemcpyDeviceToHost));
    *out_len = (size_t)coff[1];
    return SQ_OK;
#endif
}

} // extern "C"
