// squishrs_main.cpp -- C++ host driver above the C ABI (include/squish_gpu.h): the counterpart of
// the reference's pack driver, keeping the CLI surface and the .squish v1.2 bytes identical.
// (Rust is not available in the build image, so the host side is C++; INTEGRATION.md shows the
// Rust binding a maintainer would add instead.)
//
//   ArchiveWriter::new / pack / process_file / write_files_metadata
//                          /root/reference/src/archive/writer.rs:66-107,151-195,229-271,292-329
//   writer_thread record   /root/reference/src/fsutil/writer.rs:17-39
//   header helpers         /root/reference/src/util/header.rs:10-14,35-63,192-233
//   walk_dir               /root/reference/src/fsutil/directory.rs:39-73
//   ArchiveReader::unpack / read_chunks / rebuild_files
//                          /root/reference/src/archive/reader.rs:232-244,259-313,315-413
//   CLI flags / messages   /root/reference/src/cmd/mod.rs:11-58, /root/reference/src/lib.rs:19-111,
//                          /root/reference/src/main.rs:5-10
//
// Chunk records are emitted in canonical "-j 1" order (first occurrence in manifest order,
// SURVEY.md Appendix E).  Every chunk's hash / dedup decision / frame comes from sq_pack_batch:
// there is no CPU hashing or compression in this program.
#include "../../include/squish_gpu.h"
#include "zstd_frame_decoder.hpp"

#include <dirent.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <filesystem>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

static const char* kVersion = "1.2.0"; // CARGO_PKG_VERSION, Cargo.toml:3
static bool g_tty = false;

struct AppError {
    std::string msg;
};
static std::string green(const std::string& s) { return g_tty ? "\x1b[32m" + s + "\x1b[0m" : s; }
static std::string blue(const std::string& s) { return g_tty ? "\x1b[34m" + s + "\x1b[0m" : s; }
static std::string red(const std::string& s) { return isatty(2) ? "\x1b[31m" + s + "\x1b[0m" : s; }

// cmd/mod.rs:161-165 -- byte-unit decimal units, "{:.2} {}"
static std::string format_bytes(uint64_t bytes) {
    static const char* units[] = {"B", "KB", "MB", "GB", "TB", "PB", "EB"};
    double v = (double)bytes;
    int u = 0;
    while (v >= 1000.0 && u < 6) {
        v /= 1000.0;
        u++;
    }
    char buf[64];
    snprintf(buf, sizeof(buf), "%.2f %s", v, units[u]);
    return buf;
}

static std::string with_commas(uint64_t v) { // num_format Locale::en
    std::string s = std::to_string(v), o;
    int c = 0;
    for (size_t i = s.size(); i-- > 0;) {
        o.insert(o.begin(), s[i]);
        if (++c % 3 == 0 && i) o.insert(o.begin(), ',');
    }
    return o;
}

// fsutil/directory.rs:39-73: iterative DFS with stack.pop(); OS read_dir order (not sorted);
// is_dir() follows symlinks.
static std::vector<std::string> walk_dir(const std::string& root) {
    std::vector<std::string> stack{root}, files;
    while (!stack.empty()) {
        std::string dir = stack.back();
        stack.pop_back();
        DIR* d = opendir(dir.c_str());
        if (!d) throw AppError{"Failed to read directory " + dir + ": " + strerror(errno)};
        std::vector<std::string> dirs;
        while (struct dirent* e = readdir(d)) {
            if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
            std::string p = dir + "/" + e->d_name;
            struct stat st;
            if (stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode))
                dirs.push_back(p);
            else
                files.push_back(p);
        }
        closedir(d);
        for (auto& x : dirs) stack.push_back(x);
    }
    return files;
}

static void put_u64(std::string& b, uint64_t v) { b.append((const char*)&v, 8); } // LE host
static void put_u32(std::string& b, uint32_t v) { b.append((const char*)&v, 4); }

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void check(int rc, const char* what) {
    if (rc != SQ_OK) {
        if (rc == SQ_E_COMPRESS) throw AppError{"Compression error"}; // errors.rs:25-26
        throw AppError{std::string("Unknown error: ") + what + ": " + sq_strerror(rc)};
    }
}

struct FileMeta {
    std::string rel;
    uint64_t size;
    std::vector<std::string> hashes; // 16 bytes each
};

static std::string strip_prefix(const std::string& path, const std::string& prefix) {
    // Path::strip_prefix works on components: tolerate trailing slashes in the prefix
    std::string pre = prefix;
    while (pre.size() > 1 && pre.back() == '/') pre.pop_back();
    if (path.compare(0, pre.size(), pre) != 0) throw AppError{"Unknown error: prefix not found"};
    size_t i = pre.size();
    while (i < path.size() && path[i] == '/') i++;
    return path.substr(i);
}

static uint64_t pack(const std::string& input, const std::string& output, int device, uint64_t batch_bytes, long max_threads) {
    double const t_wall0 = now_s();
    std::string trimmed = input;
    while (!trimmed.empty() && (trimmed.back() == '/' || trimmed.back() == '\\')) trimmed.pop_back();
    std::vector<std::string> files = walk_dir(trimmed); // lib.rs:36

    FILE* out = fopen(output.c_str(), "wb"); // writer.rs:72
    if (!out) throw AppError{std::string("I/O error: ") + strerror(errno)};
    {
        std::string pre = "squish"; // header.rs:10-14
        pre += kVersion;
        put_u64(pre, (uint64_t)time(nullptr)); // header.rs:55-63
        put_u64(pre, 0);                       // placeholder, writer.rs:83-84
        fwrite(pre.data(), 1, pre.size(), out);
    }
    sq_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = SQ_ABI_VERSION;
    cfg.device = device;
    cfg.max_batch_bytes = batch_bytes;
    uint64_t total_chunks = 0;
    std::vector<uint64_t> sizes(files.size());
    for (size_t i = 0; i < files.size(); i++) {
        struct stat st;
        if (stat(files[i].c_str(), &st) != 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
        sizes[i] = (uint64_t)st.st_size;
        total_chunks += (sizes[i] + SQ_CHUNK_SIZE - 1) / SQ_CHUNK_SIZE;
    }
    cfg.dedup_capacity = total_chunks + 1024;
    sq_ctx* ctx = nullptr;
    double const t_ctx0 = now_s(), t_walk = t_ctx0 - t_wall0;
    check(sq_ctx_create(&cfg, &ctx), "sq_ctx_create");
    double t_ctx = now_s() - t_ctx0, t_read = 0, t_gpu = 0, t_write = 0; // SQUISH_TIMING=1 prints them

    std::vector<FileMeta> metas(files.size());
    // Three batches rotate through three stages, each stage on its own thread and every batch through the
    // stages in order (so the archive's bytes do not depend on the overlap):
    //   reader : files -> chunks, read straight into the batch's page-locked input buffer at 256-byte
    //            aligned starts (process_file's read loop, writer.rs:233-246); one H2D per batch then
    //   gpu    : sq_pack_batch (this thread)
    //   writer : chunk records + per-file hash lists (writer_thread, fsutil/writer.rs:17-39)
    // The reference gets the same overlap from its Rayon workers and its writer thread.
    struct HostBuf {
        uint8_t* p = nullptr;
        size_t cap = 0;
        bool pinned = false;
        double alloc_s = 0;
        bool want_pinned = true;
        void ensure(size_t n) {
            if (n <= cap) return;
            release();
            double const t0 = now_s();
            p = want_pinned ? (uint8_t*)sq_host_alloc(n) : nullptr;
            pinned = p != nullptr;
            if (!p) p = (uint8_t*)malloc(n);
            if (!p) throw AppError{"I/O error: out of memory"};
            cap = n;
            alloc_s += now_s() - t0;
        }
        void release() {
            if (p) {
                if (pinned) sq_host_free(p); else free(p);
            }
            p = nullptr;
            cap = 0;
        }
        ~HostBuf() { release(); }
    };
    struct Batch {
        HostBuf data, comp;
        size_t used = 0; // bytes of `data` in use (the next chunk starts at the next multiple of 256)
        std::vector<uint64_t> off, coff;
        std::vector<uint32_t> len;
        std::vector<size_t> owner; // file index of each chunk
        std::vector<uint8_t> hashes, is_new;
        int state = 0; // 0 free (reader's), 1 filled (gpu's), 2 packed (writer's)
        bool last = false;
    };
    uint64_t total_bytes = 0;
    for (uint64_t z : sizes) total_bytes += z;
    // batch size: the configured one, but no larger than an eighth of the input (four batches rotate through the
    // stages, two of them in flight on the GPU: enough batches for the stages to overlap) -- and not below 1 GiB:
    // the compressor runs one wave per chunk, a compressible chunk takes a wave the better part of a second (a 256 KiB
    // text chunk: two), and the compressor launches of two batches do not fit the GPU side by side -- so the GPU stage
    // of a compressible tree takes (batches) x (its slowest chunk), and a small input must reach the GPU in ONE batch
    // (256 MiB until the end of round 4: a 388 MiB tree went in two batches and its GPU stage took 3.3 s instead of
    // one batch's 1.8; profiles/round4h_cli_soak_batch_floor.txt)
    uint64_t floor_bytes = 1ull << 30;
    if (const char* e = getenv("SQUISH_BATCH_FLOOR")) floor_bytes = std::max<uint64_t>(strtoull(e, nullptr, 10), 8u << 20); // (experiments)
    uint64_t eff_batch = std::max<uint64_t>(std::min<uint64_t>(batch_bytes, std::max<uint64_t>(total_bytes / 8 + SQ_CHUNK_SIZE, floor_bytes)), 8u << 20);
    {   // ... and the four batches' input + output buffers (2 x 4 x eff_batch) within half of the memory the host has
        // available (MemAvailable): a small machine packs a large tree in more, smaller batches instead of being
        // killed for lack of memory
        uint64_t avail = 0;
        if (FILE* mi = fopen("/proc/meminfo", "r")) {
            char line[256];
            while (fgets(line, sizeof line, mi)) {
                unsigned long long kb = 0;
                if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) avail = (uint64_t)kb << 10;
            }
            fclose(mi);
        }
        if (avail && !getenv("SQUISH_BATCH_BYTES"))
            while (eff_batch > (64ull << 20) && 8 * eff_batch > avail / 2) eff_batch /= 2;
    }
    int const NB = 4; // reader: 1, gpu: 2 in flight (sq_pack_submit / sq_pack_wait), writer: 1
    Batch bt[4];
    {   // Page-locking costs ~0.2 s per GiB here and only pays when the GPU stage (two PCIe crossings) is the
        // slowest of the three, i.e. on large inputs; SQUISH_PINNED=0/1 overrides.
        bool pin = total_bytes >= (16ull << 30);
        if (const char* e = getenv("SQUISH_PINNED")) pin = atoi(e) != 0;
        for (auto& b : bt) b.data.want_pinned = b.comp.want_pinned = pin;
    }
    std::mutex mu;
    std::condition_variable cv;
    std::string err; // first failure of any stage
    auto wait_state = [&](Batch& b, int want) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return b.state == want || !err.empty(); });
        return err.empty();
    };
    auto set_state = [&](Batch& b, int st) {
        { std::lock_guard<std::mutex> lk(mu); b.state = st; }
        cv.notify_all();
    };
    auto fail = [&](const std::string& m) {
        { std::lock_guard<std::mutex> lk(mu); if (err.empty()) err = m; }
        cv.notify_all();
    };

    // The reader: the reference reads every file on its Rayon workers (writer.rs:153-165, process_file's read
    // loop 233-246).  Here the chunk plan -- file, offset, length, batch, position in the batch buffer -- follows
    // from the file sizes alone (files in slice order, 2 MiB chunks, the batch layout of sq_pack_submit: starts padded
    // to 256 bytes), so up to `-j` reader threads (cmd/mod.rs:16) fill a batch's chunks in parallel with pread();
    // the archive's bytes do not depend on how many there are.
    struct PlanChunk {
        size_t file;
        uint64_t foff;
        uint32_t len;
        size_t pos;
    };
    std::vector<std::vector<PlanChunk>> plan(1);
    {
        size_t used = 0;
        for (size_t f = 0; f < files.size(); f++)
            for (uint64_t o = 0; o < sizes[f]; o += SQ_CHUNK_SIZE) { // an empty file has no chunk (writer.rs:241-245)
                uint32_t const len = (uint32_t)std::min<uint64_t>(SQ_CHUNK_SIZE, sizes[f] - o);
                size_t const pos = (used + 255) & ~(size_t)255;
                plan.back().push_back({f, o, len, pos});
                used = pos + len;
                if (used >= eff_batch || plan.back().size() >= 4096) {
                    plan.emplace_back();
                    used = 0;
                }
            }
    }
    unsigned const hw = std::max(1u, std::thread::hardware_concurrency());
    size_t const n_readers = (size_t)std::max<long>(1, std::min<long>(max_threads, (long)hw));
    std::thread reader([&] {
        try {
            for (size_t bi = 0; bi < plan.size(); bi++) {
                Batch& b = bt[bi % NB];
                if (!wait_state(b, 0)) return;
                std::vector<PlanChunk> const& pc = plan[bi];
                b.data.ensure((size_t)eff_batch + 2 * (size_t)SQ_CHUNK_SIZE + 4096);
                double const tr0 = now_s();
                std::atomic<size_t> next{0};
                std::string rerr;
                std::mutex rmu;
                // bytes actually read per planned chunk.  The plan comes from the sizes stat() reported before packing;
                // the reference reads every file to its END (process_file, writer.rs:233-246) and records the chunks it
                // got.  A file that has SHRUNK since is therefore packed as what is there: the chunk that runs into the
                // end is shortened, chunks beyond it are dropped (the manifest keeps metadata.len() as the reference's
                // does, writer.rs:231,270).  A file that has GROWN is packed up to its planned size (divergence: the
                // reference would read on).
                std::vector<uint32_t> actual(pc.size(), 0);
                auto work = [&]() {
                    int fd = -1, fd_errno = 0;
                    size_t cur = (size_t)-1;
                    for (;;) {
                        size_t const i = next.fetch_add(1);
                        if (i >= pc.size()) break;
                        PlanChunk const& c = pc[i];
                        if (c.file != cur) {
                            if (fd >= 0) close(fd);
                            fd = open(files[c.file].c_str(), O_RDONLY);
                            fd_errno = fd < 0 ? errno : 0; // (captured at the failing call: other threads may set errno's meaning aside by then)
                            cur = c.file;
                        }
                        ssize_t got = 0;
                        int rd_errno = 0;
                        size_t done = 0;
                        if (fd >= 0)
                            for (; done < c.len; done += (size_t)got) {
                                got = pread(fd, b.data.p + c.pos + done, c.len - done, (off_t)(c.foff + done));
                                if (got < 0) rd_errno = errno;
                                if (got <= 0) break;
                            }
                        if (fd < 0 || got < 0) {
                            std::lock_guard<std::mutex> lk(rmu);
                            if (rerr.empty()) rerr = std::string("I/O error: ") + strerror(fd < 0 ? fd_errno : rd_errno) + " (" + files[c.file] + ")";
                            break;
                        }
                        actual[i] = (uint32_t)done; // < c.len: the file ends here now (0: it ends before this chunk)
                    }
                    if (fd >= 0) close(fd);
                };
                size_t const nt = std::min(n_readers, std::max<size_t>(1, pc.size()));
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
                work();
                for (auto& t : pool) t.join();
                if (!rerr.empty()) throw AppError{rerr};
                for (size_t i = 0; i < pc.size(); i++) {
                    if (actual[i] == 0) continue; // (only for a file that shrank: an empty read is the loop's end, writer.rs:241-245)
                    b.off.push_back(pc[i].pos);
                    b.len.push_back(actual[i]);
                    b.owner.push_back(pc[i].file);
                }
                b.used = pc.empty() ? 0 : pc.back().pos + pc.back().len;
                t_read += now_s() - tr0;
                b.last = bi + 1 == plan.size();
                set_state(b, 1);
            }
        } catch (const AppError& e) {
            fail(e.msg);
        }
    });

    // The writer (fsutil/writer.rs:17-39): ONE stage emits the chunk records in canonical order.  Every record's
    // place in the file follows from the frame lengths of the records before it, so the stage lays a batch's
    // records out with a prefix sum and lets up to 8 threads pwritev() them (hash || original_size ||
    // compressed_size || frame per record); the bytes are those of the reference's single writer thread.
    fflush(out);
    int const out_fd = fileno(out);
    uint64_t file_pos = 27; // behind the preamble
    // (measured on tmpfs, same box: one writer 1.4-1.6 s for 8.6 GB, two 2.8 s, eight 1.7-3.2 s -- page allocation in the
    // kernel serialises them; so ONE by default, like the reference; SQUISH_WRITERS=n for file systems that scale)
    size_t n_writers = 1;
    if (const char* e = getenv("SQUISH_WRITERS")) n_writers = (size_t)std::max(1, std::min(atoi(e), (int)hw));
    std::thread writer([&] {
        try {
            for (int k = 0;; k = (k + 1) % NB) {
                Batch& b = bt[k];
                if (!wait_state(b, 2)) return;
                double const tw0 = now_s();
                uint32_t const n = (uint32_t)b.len.size();
                std::vector<uint64_t> at((size_t)n, 0); // file offset of record i (new chunks only)
                std::vector<uint32_t> recs;
                for (uint32_t i = 0; i < n; i++) {
                    metas[b.owner[i]].hashes.emplace_back((const char*)&b.hashes[(size_t)16 * i], 16); // writer.rs:267
                    if (b.is_new[i]) { // fsutil/writer.rs:21-36
                        at[i] = file_pos;
                        file_pos += 32 + (b.coff[i + 1] - b.coff[i]);
                        recs.push_back(i);
                    }
                }
                std::atomic<size_t> next{0};
                std::string werr;
                std::mutex wmu;
                auto work = [&]() {
                    for (;;) {
                        size_t const r = next.fetch_add(1);
                        if (r >= recs.size()) return;
                        uint32_t const i = recs[r];
                        uint64_t const osz = SQ_CHUNK_SIZE /* chunk_buf.len(), writer.rs:255 */, csz = b.coff[i + 1] - b.coff[i];
                        uint8_t hdr[32];
                        memcpy(hdr, &b.hashes[(size_t)16 * i], 16);
                        memcpy(hdr + 16, &osz, 8);
                        memcpy(hdr + 24, &csz, 8);
                        struct iovec iov[2] = {{hdr, 32}, {b.comp.p + b.coff[i], (size_t)csz}};
                        uint64_t off = at[i];
                        size_t left = 32 + (size_t)csz;
                        int cnt = 2;
                        struct iovec* v = iov;
                        while (left) {
                            ssize_t const w = pwritev(out_fd, v, cnt, (off_t)off);
                            if (w <= 0) {
                                std::lock_guard<std::mutex> lk(wmu);
                                if (werr.empty()) werr = std::string("I/O error: ") + strerror(errno);
                                return;
                            }
                            left -= (size_t)w;
                            off += (uint64_t)w;
                            size_t adv = (size_t)w;
                            while (cnt && adv >= v->iov_len) {
                                adv -= v->iov_len;
                                v++;
                                cnt--;
                            }
                            if (cnt && adv) {
                                v->iov_base = (char*)v->iov_base + adv;
                                v->iov_len -= adv;
                            }
                        }
                    }
                };
                size_t const nt = std::min(n_writers, std::max<size_t>(1, recs.size()));
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
                work();
                for (auto& t : pool) t.join();
                if (!werr.empty()) throw AppError{werr};
                t_write += now_s() - tw0;
                bool const last = b.last;
                b.used = 0;
                b.off.clear();
                b.len.clear();
                b.owner.clear();
                b.last = false;
                set_state(b, 0);
                if (last) return;
            }
        } catch (const AppError& e) {
            fail(e.msg);
        }
    });

    try { // the gpu stage: two batches in flight on the context's two lanes -- batch k+1's H2D copy and kernels
          // run while batch k's frames travel back (the read-ahead of writer.rs:233-246 + Rayon, on PCIe)
        std::vector<int> inflight;
        std::vector<uint64_t> woff;
        std::vector<uint32_t> wlen;
        auto collect = [&]() {
            Batch& p = bt[inflight.front()];
            inflight.erase(inflight.begin());
            uint32_t const n = (uint32_t)p.len.size();
            woff.resize((size_t)n + 1);
            wlen.resize((size_t)n + 1);
            double const tg0 = now_s();
            check(sq_pack_wait(ctx, woff.data(), wlen.data()), "sq_pack_wait");
            t_gpu += now_s() - tg0;
            p.coff.resize((size_t)n + 1);
            for (uint32_t i = 0; i < n; i++) p.coff[i] = woff[i];
            p.coff[n] = n ? woff[n - 1] + wlen[n - 1] : 0;
            set_state(p, 2);
        };
        auto ready = [&](Batch& x) {
            std::lock_guard<std::mutex> lk(mu);
            return x.state == 1 || !err.empty();
        };
        for (int k = 0;; k = (k + 1) % NB) {
            Batch& b = bt[k];
            // hand finished batches to the writer while the reader is still filling the next one
            while (!inflight.empty() && (inflight.size() == 2 || !ready(b))) collect();
            if (!wait_state(b, 1)) break;
            uint32_t const n = (uint32_t)b.len.size();
            b.hashes.resize((size_t)n * 16 + 16);
            b.is_new.resize((size_t)n + 16);
            uint64_t cap = 64;
            for (uint32_t i = 0; i < n; i++) cap += sq_compress_bound(b.len[i]) + 64;
            b.comp.ensure((size_t)cap);
            double const tg0 = now_s();
            check(sq_pack_submit(ctx, b.data.p, b.off.data(), b.len.data(), n, b.hashes.data(), b.is_new.data(), b.comp.p, cap, 0),
                  "sq_pack_submit");
            t_gpu += now_s() - tg0;
            inflight.push_back(k);
            if (b.last) {
                while (!inflight.empty()) collect();
                break;
            }
        }
    } catch (const AppError& e) {
        fail(e.msg);
    }
    reader.join();
    writer.join();
    double const t_stages_end = now_s();
    if (!err.empty()) {
        fclose(out);
        sq_ctx_destroy(ctx);
        throw AppError{err};
    }
    for (size_t f = 0; f < files.size(); f++) {
        metas[f].rel = strip_prefix(files[f], input);
        metas[f].size = sizes[f];
    }
    if (getenv("SQUISH_TIMING"))
    {
        double t_alloc = 0;
        for (auto& b : bt) t_alloc += b.data.alloc_s + b.comp.alloc_s;
        fprintf(stderr, "[squishrs timing] ctx_create %.3f s; overlapped stages: file reads %.3f s, sq_pack_submit/wait %.3f s, record writes %.3f s; page-locked allocations %.3f s (inside the stages)\n", t_ctx, t_read, t_gpu, t_write, t_alloc);
    }
    {
        uint64_t count = sq_unique_count(ctx); // writer.rs:177-184
        fseek(out, 19, SEEK_SET);
        fwrite(&count, 8, 1, out);
        fseek(out, (long)file_pos, SEEK_SET); // behind the last chunk record
    }
    { // writer.rs:292-329
        std::string m;
        put_u32(m, (uint32_t)metas.size());
        for (auto& fm : metas) {
            put_u32(m, (uint32_t)fm.rel.size());
            m += fm.rel;
            put_u64(m, fm.size);
            put_u32(m, (uint32_t)fm.hashes.size());
            for (auto& h : fm.hashes) m += h;
        }
        fwrite(m.data(), 1, m.size(), out);
    }
    fflush(out);
    uint64_t total = (uint64_t)ftell(out);
    fclose(out);
    double const t_fin = now_s();
    sq_ctx_destroy(ctx);
    if (getenv("SQUISH_TIMING"))
        fprintf(stderr, "[squishrs timing] wall inside pack(): walk + stat %.3f s, ctx_create %.3f s, the three stages %.3f s, count + manifest %.3f s, ctx_destroy %.3f s "
                        "(the batch buffers are released after this line)\n", t_walk, t_ctx, t_stages_end - (t_wall0 + t_walk + t_ctx), t_fin - t_stages_end, now_s() - t_fin);
    return total;
}

// ---- pack on N GPUs (the reference's counterpart is the Rayon fan-out of ArchiveWriter::pack,
// writer.rs:153-195, with the one DashMap all workers share, chunk.rs:52-56,83, and the single ordered writer thread,
// fsutil/writer.rs:17-39).  One PROCESS per GPU, forked before anything touches a GPU.  The chunk plan (files in walk order,
// 2 MiB chunks) is cut into super-batches; inside one the chunks go to the ranks in block-cyclic stripes of 16 (the
// first occurrences -- the compression work -- spread evenly).  Per super-batch every rank
//   1. reads and hashes ITS chunks (sq_shard_submit_hash), puts the hashes into a table all ranks share (host memory:
//      the exchange -- 16 bytes per chunk -- is the path's only communication),
//   2. after a barrier hands the hashes of ALL ranks, in global order, to sq_shard_submit_rest_idx: every rank's replica of
//      the dedup set resolves "first occurrence in global order" identically, and the rank compresses only its own new chunks,
//   3. publishes its records' sizes; after a second barrier every rank knows where each of its records goes (a prefix sum
//      over the global order) and writes them itself with pwritev: no funnel, the archive is byte-identical to the one-GPU one.
// Rank 0 writes the chunk count and the manifest at the end.  (Synchronous per super-batch; the pipelined form -- two
// super-batches in flight -- is what squishrs_amd/sharded_pack.py does with the same calls over RCCL.)
struct MultiShared {
    pthread_barrier_t bar;
    std::atomic<int> failed;
    char msg[480];
};
static uint64_t pack_multi(const std::string& input, const std::string& output, const std::vector<int>& devices, uint32_t batch_chunks,
                           long max_threads) {
    std::string trimmed = input;
    while (!trimmed.empty() && (trimmed.back() == '/' || trimmed.back() == '\\')) trimmed.pop_back();
    std::vector<std::string> files = walk_dir(trimmed);
    std::vector<uint64_t> sizes(files.size());
    struct Chunk {
        uint32_t file;
        uint32_t len;
        uint64_t foff;
    };
    std::vector<Chunk> plan;
    std::vector<uint32_t> first_chunk(files.size() + 1, 0);
    for (size_t f = 0; f < files.size(); f++) {
        struct stat st;
        if (stat(files[f].c_str(), &st) != 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
        sizes[f] = (uint64_t)st.st_size;
        first_chunk[f] = (uint32_t)plan.size();
        for (uint64_t o = 0; o < sizes[f]; o += SQ_CHUNK_SIZE)
            plan.push_back({(uint32_t)f, (uint32_t)std::min<uint64_t>(SQ_CHUNK_SIZE, sizes[f] - o), o});
    }
    first_chunk[files.size()] = (uint32_t)plan.size();
    uint64_t const total = plan.size();
    int const world = (int)devices.size();
    uint32_t const stripe = 16, B = std::max<uint32_t>(batch_chunks, stripe * (uint32_t)world);

    {   // preamble (header.rs:10-14,55-63; writer.rs:83-84), by the parent: one timestamp
        FILE* out = fopen(output.c_str(), "wb");
        if (!out) throw AppError{std::string("I/O error: ") + strerror(errno)};
        std::string pre = "squish";
        pre += kVersion;
        put_u64(pre, (uint64_t)time(nullptr));
        put_u64(pre, 0);
        fwrite(pre.data(), 1, pre.size(), out);
        fclose(out);
    }
    // shared between the ranks: barrier + failure word, every chunk's hash, every chunk's record size (0: duplicate)
    size_t const sh_bytes = sizeof(MultiShared) + (size_t)total * 16 + (size_t)total * 4 + 64;
    void* const shm = mmap(nullptr, sh_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (shm == MAP_FAILED) throw AppError{std::string("I/O error: ") + strerror(errno)};
    MultiShared* const sh = new (shm) MultiShared();
    uint8_t* const all_hashes = (uint8_t*)shm + sizeof(MultiShared);
    uint32_t* const csize = (uint32_t*)(all_hashes + (size_t)total * 16);
    sh->failed = 0;
    sh->msg[0] = 0;
    {
        pthread_barrierattr_t ba;
        pthread_barrierattr_init(&ba);
        pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&sh->bar, &ba, (unsigned)world);
        pthread_barrierattr_destroy(&ba);
    }
    auto rank_main = [&](int rank) -> int {
        auto fail = [&](const std::string& m) {
            int expect = 0;
            if (sh->failed.compare_exchange_strong(expect, 1)) snprintf(sh->msg, sizeof(sh->msg), "%s", m.c_str());
        };
        sq_ctx* ctx = nullptr;
        double t_ctx = 0, t_alloc = 0, t_read = 0, t_h2d = 0, t_hash = 0, t_rest = 0, t_d2h = 0, t_write = 0, t_bar = 0; // SQUISH_TIMING=1
        // (page-locking costs ~0.2 s per GiB and only pays on large inputs: as in the one-GPU path)
        uint64_t total_bytes = 0;
        for (uint64_t z : sizes) total_bytes += z;
        bool pin = total_bytes / (uint64_t)world >= (16ull << 30);
        if (const char* e = getenv("SQUISH_PINNED")) pin = atoi(e) != 0;
        auto host_alloc = [&](size_t n) -> uint8_t* { return pin ? (uint8_t*)sq_host_alloc(n) : (uint8_t*)malloc(n); };
        auto host_free = [&](uint8_t* q) { if (!q) return; if (pin) sq_host_free(q); else free(q); };
        uint8_t *h_in = nullptr, *h_comp = nullptr;
        uint8_t *d_data = nullptr, *d_hashes = nullptr, *d_all = nullptr, *d_isnew = nullptr, *d_comp = nullptr;
        uint32_t* d_idx = nullptr;
        size_t cap_in = 0, cap_comp = 0, cap_n = 0;
        int out_fd = -1;
        uint64_t file_pos = 27, unique = 0; // (every rank computes the same values)
        try {
            sq_config cfg;
            memset(&cfg, 0, sizeof(cfg));
            cfg.abi_version = SQ_ABI_VERSION;
            cfg.device = devices[(size_t)rank];
            cfg.dedup_capacity = total + 1024;
            double const tc0 = now_s();
            check(sq_ctx_create(&cfg, &ctx), "sq_ctx_create");
            t_ctx = now_s() - tc0;
            out_fd = open(output.c_str(), O_WRONLY);
            if (out_fd < 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
        } catch (const AppError& e) {
            fail(e.msg);
        }
        std::vector<uint32_t> mine;           // global indices of this rank's chunks of the super-batch, ascending
        std::vector<uint64_t> off, woff;      // their places in the input buffer; their slots' offsets
        std::vector<uint32_t> len, wlen;
        std::vector<uint8_t> isnew, hashes;
        for (uint64_t g0 = 0; g0 < total; g0 += B) {
            uint32_t const nb = (uint32_t)std::min<uint64_t>(B, total - g0);
            try {
                if (sh->failed) throw AppError{""};
                mine.clear(); off.clear(); len.clear();
                size_t used = 0;
                uint64_t comp_need = 64;
                for (uint32_t j = 0; j < nb; j++)
                    if ((int)((j / stripe) % (uint32_t)world) == rank) {
                        Chunk const& c = plan[(size_t)(g0 + j)];
                        size_t const pos = (used + 255) & ~(size_t)255;
                        mine.push_back(j);
                        off.push_back(pos);
                        len.push_back(c.len);
                        used = pos + c.len;
                        comp_need += sq_compress_bound(c.len) + 64;
                    }
                uint32_t const n = (uint32_t)mine.size();
                double const ta0 = now_s();
                if (used + 4096 > cap_in) { // (buffers grow to the largest super-batch and stay)
                    host_free(h_in);
                    sq_dev_free(ctx, d_data);
                    cap_in = used + (used >> 3) + 4096;
                    h_in = host_alloc(cap_in);
                    d_data = (uint8_t*)sq_dev_alloc(ctx, cap_in);
                    if (!h_in || !d_data) throw AppError{"I/O error: out of memory"};
                }
                if (comp_need > cap_comp) {
                    host_free(h_comp);
                    sq_dev_free(ctx, d_comp);
                    cap_comp = comp_need + (comp_need >> 3);
                    h_comp = host_alloc(cap_comp);
                    d_comp = (uint8_t*)sq_dev_alloc(ctx, cap_comp);
                    if (!h_comp || !d_comp) throw AppError{"I/O error: out of memory"};
                }
                if ((size_t)n + 16 > cap_n || !d_all) {
                    sq_dev_free(ctx, d_hashes); sq_dev_free(ctx, d_isnew); sq_dev_free(ctx, d_idx); sq_dev_free(ctx, d_all);
                    cap_n = (size_t)n + (n >> 3) + 64;
                    d_hashes = (uint8_t*)sq_dev_alloc(ctx, cap_n * 16);
                    d_isnew = (uint8_t*)sq_dev_alloc(ctx, cap_n);
                    d_idx = (uint32_t*)sq_dev_alloc(ctx, cap_n * 4);
                    d_all = (uint8_t*)sq_dev_alloc(ctx, (size_t)B * 16);
                    if (!d_hashes || !d_isnew || !d_idx || !d_all) throw AppError{"I/O error: out of memory"};
                }
                t_alloc += now_s() - ta0;
                double const tr0 = now_s();
                {   // this rank's chunks: up to -j reader threads, pread straight into the input buffer
                    std::atomic<size_t> next{0};
                    std::string rerr;
                    std::mutex rmu;
                    auto work = [&]() {
                        int fd = -1;
                        uint32_t cur = 0xFFFFFFFFu;
                        for (;;) {
                            size_t const i = next.fetch_add(1);
                            if (i >= n) break;
                            Chunk const& c = plan[(size_t)(g0 + mine[i])];
                            if (c.file != cur) {
                                if (fd >= 0) close(fd);
                                fd = open(files[c.file].c_str(), O_RDONLY);
                                cur = c.file;
                            }
                            size_t done = 0;
                            ssize_t got = fd < 0 ? -1 : 0;
                            if (fd >= 0)
                                for (; done < c.len; done += (size_t)got) {
                                    got = pread(fd, h_in + off[i] + done, c.len - done, (off_t)(c.foff + done));
                                    if (got <= 0) break;
                                }
                            if (fd < 0 || got < 0 || done != c.len) { // (a file that changed size under the pack: the one-GPU path packs what is there, this one stops)
                                std::lock_guard<std::mutex> lk(rmu);
                                if (rerr.empty()) rerr = std::string("I/O error: ") + (fd < 0 || got < 0 ? strerror(errno) : "file changed while packing") + " (" + files[c.file] + ")";
                                break;
                            }
                        }
                        if (fd >= 0) close(fd);
                    };
                    size_t const nt = (size_t)std::max<long>(1, std::min<long>(std::min<long>(max_threads, 16), (long)n));
                    std::vector<std::thread> pool;
                    for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
                    work();
                    for (auto& t : pool) t.join();
                    if (!rerr.empty()) throw AppError{rerr};
                }
                t_read += now_s() - tr0;
                double const th0 = now_s();
                check(sq_dev_copy(ctx, d_data, h_in, used, 1), "copy to the device");
                t_h2d += now_s() - th0;
                double const tk0 = now_s();
                check(sq_shard_submit_hash(ctx, d_data, off.data(), len.data(), n, d_hashes), "sq_shard_submit_hash");
                check(sq_shard_wait_hash(ctx), "sq_shard_wait_hash");
                t_hash += now_s() - tk0;
                hashes.resize((size_t)n * 16 + 16);
                check(sq_dev_copy(ctx, hashes.data(), d_hashes, (size_t)n * 16, 0), "copy from the device");
                for (uint32_t i = 0; i < n; i++) memcpy(all_hashes + (size_t)(g0 + mine[i]) * 16, &hashes[(size_t)16 * i], 16);
            } catch (const AppError& e) {
                if (!e.msg.empty()) fail(e.msg);
            }
            { double const tb0 = now_s(); pthread_barrier_wait(&sh->bar); t_bar += now_s() - tb0; } // every rank's hashes of the super-batch are in the table
            uint32_t const n = (uint32_t)mine.size();
            bool submitted = false;
            try {
                if (sh->failed) throw AppError{""};
                double const tr1 = now_s();
                check(sq_dev_copy(ctx, d_all, all_hashes + (size_t)g0 * 16, (size_t)nb * 16, 1), "copy to the device");
                check(sq_dev_copy(ctx, d_idx, mine.data(), (size_t)n * 4, 1), "copy to the device");
                check(sq_shard_submit_rest_idx(ctx, d_all, nb, n ? d_idx : nullptr, d_isnew, d_comp, cap_comp, 0), "sq_shard_submit_rest_idx");
                submitted = true;
                woff.resize((size_t)n + 1);
                wlen.resize((size_t)n + 1);
                check(sq_pack_wait(ctx, woff.data(), wlen.data()), "sq_pack_wait");
                t_rest += now_s() - tr1;
                double const td0 = now_s();
                isnew.resize((size_t)n + 16);
                check(sq_dev_copy(ctx, isnew.data(), d_isnew, n, 0), "copy from the device");
                uint64_t end = 0;
                for (uint32_t i = 0; i < n; i++) {
                    csize[(size_t)(g0 + mine[i])] = isnew[i] ? wlen[i] : 0u;
                    if (isnew[i]) end = std::max<uint64_t>(end, woff[i] + wlen[i]);
                }
                check(sq_dev_copy(ctx, h_comp, d_comp, end, 0), "copy from the device");
                t_d2h += now_s() - td0;
            } catch (const AppError& e) {
                if (!e.msg.empty()) fail(e.msg);
                (void)submitted;
            }
            { double const tb0 = now_s(); pthread_barrier_wait(&sh->bar); t_bar += now_s() - tb0; } // every record's size is known to every rank
            double const tw0 = now_s();
            try {
                if (sh->failed) throw AppError{""};
                // where the records go: a prefix sum over the global order (fsutil/writer.rs:21-36: hash || original_size ||
                // compressed_size || frame); this rank writes its own
                uint32_t k = 0;
                for (uint32_t j = 0; j < nb; j++) {
                    uint32_t const cs = csize[(size_t)(g0 + j)];
                    bool const is_mine = k < n && mine[k] == j;
                    if (cs) {
                        if (is_mine) {
                            uint64_t const osz = SQ_CHUNK_SIZE /* chunk_buf.len(), writer.rs:255 */, csz = cs;
                            uint8_t hdr[32];
                            memcpy(hdr, all_hashes + (size_t)(g0 + j) * 16, 16);
                            memcpy(hdr + 16, &osz, 8);
                            memcpy(hdr + 24, &csz, 8);
                            struct iovec iov[2] = {{hdr, 32}, {h_comp + woff[k], (size_t)csz}};
                            struct iovec* v = iov;
                            int cnt = 2;
                            uint64_t at = file_pos;
                            size_t left = 32 + (size_t)csz;
                            while (left) {
                                ssize_t const w = pwritev(out_fd, v, cnt, (off_t)at);
                                if (w <= 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
                                left -= (size_t)w;
                                at += (uint64_t)w;
                                size_t adv = (size_t)w;
                                while (cnt && adv >= v->iov_len) {
                                    adv -= v->iov_len;
                                    v++;
                                    cnt--;
                                }
                                if (cnt && adv) {
                                    v->iov_base = (char*)v->iov_base + adv;
                                    v->iov_len -= adv;
                                }
                            }
                        }
                        file_pos += 32 + (uint64_t)cs;
                        unique++;
                    }
                    if (is_mine) k++;
                }
            } catch (const AppError& e) {
                if (!e.msg.empty()) fail(e.msg);
            }
            t_write += now_s() - tw0;
        }
        pthread_barrier_wait(&sh->bar); // every record is written
        if (getenv("SQUISH_TIMING"))
            fprintf(stderr, "[squishrs timing] rank %d of %d: ctx_create %.3f s, buffers %.3f, file reads %.3f, to the device %.3f, hash %.3f, dedup + compress %.3f, "
                            "from the device %.3f, record writes %.3f, at barriers %.3f\n", rank, world, t_ctx, t_alloc, t_read, t_h2d, t_hash, t_rest, t_d2h, t_write, t_bar);
        if (rank == 0 && !sh->failed) {
            try { // writer.rs:177-184 (chunk count), 292-329 (manifest)
                std::string m;
                put_u32(m, (uint32_t)files.size());
                for (size_t f = 0; f < files.size(); f++) {
                    std::string const rel = strip_prefix(files[f], input);
                    put_u32(m, (uint32_t)rel.size());
                    m += rel;
                    put_u64(m, sizes[f]);
                    uint32_t const nc = first_chunk[f + 1] - first_chunk[f];
                    put_u32(m, nc);
                    m.append((const char*)all_hashes + (size_t)first_chunk[f] * 16, (size_t)nc * 16);
                }
                if (pwrite(out_fd, &unique, 8, 19) != 8) throw AppError{std::string("I/O error: ") + strerror(errno)};
                size_t done = 0;
                while (done < m.size()) {
                    ssize_t const w = pwrite(out_fd, m.data() + done, m.size() - done, (off_t)(file_pos + done));
                    if (w <= 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
                    done += (size_t)w;
                }
                if (ftruncate(out_fd, (off_t)(file_pos + m.size())) != 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
            } catch (const AppError& e) {
                fail(e.msg);
            }
        }
        if (out_fd >= 0) close(out_fd);
        if (ctx) {
            for (void* p : {(void*)d_data, (void*)d_hashes, (void*)d_all, (void*)d_isnew, (void*)d_comp, (void*)d_idx}) sq_dev_free(ctx, p);
            host_free(h_in);
            host_free(h_comp);
            sq_ctx_destroy(ctx);
        }
        return sh->failed ? 1 : 0;
    };
    fflush(stdout);
    fflush(stderr);
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        pid_t const pid = fork(); // (before this process has touched a GPU: it never does)
        if (pid < 0) throw AppError{std::string("I/O error: fork: ") + strerror(errno)};
        if (pid == 0) _exit(rank_main(r));
        kids.push_back(pid);
    }
    bool bad = false;
    for (size_t left = kids.size(); left;) { // a rank that dies (not: fails) would leave the others at the barrier: they are ended then
        int st = 0;
        pid_t const k = waitpid(-1, &st, 0);
        if (k < 0) break;
        auto it = std::find(kids.begin(), kids.end(), k);
        if (it == kids.end()) continue;
        *it = -1;
        left--;
        if (!WIFEXITED(st)) {
            bad = true;
            for (pid_t o : kids)
                if (o > 0) kill(o, SIGKILL);
        } else if (WEXITSTATUS(st) != 0)
            bad = true;
    }
    std::string const msg = sh->msg;
    bool const failed = sh->failed != 0;
    pthread_barrier_destroy(&sh->bar);
    munmap(shm, sh_bytes);
    if (failed) throw AppError{msg.empty() ? std::string("Unknown error: a rank failed") : msg};
    if (bad) throw AppError{"Unknown error: a rank ended abnormally"};
    struct stat st;
    if (stat(output.c_str(), &st) != 0) throw AppError{std::string("I/O error: ") + strerror(errno)};
    return (uint64_t)st.st_size;
}

// ---- list (archive/reader.rs:46-118,155-219; cmd/mod.rs:60-158) --------------------------------
struct Summary {
    uint64_t archive_size = 0, total_orig = 0, unique_chunks = 0, timestamp = 0;
    std::string version;
    std::vector<std::pair<std::string, uint64_t>> files;
    bool bad_utf8 = false;
};
static void rd(FILE* f, void* p, size_t n) {
    if (fread(p, 1, n, f) != n) throw AppError{"Error reading from squish: failed to fill whole buffer"};
}
// String::from_utf8's rule (reader.rs:178,350): well-formed UTF-8, no overlong forms, no surrogates, nothing above U+10FFFF
static bool valid_utf8(const std::string& s) {
    size_t i = 0, n = s.size();
    auto cont = [&](size_t k) { return k < n && ((uint8_t)s[k] & 0xC0) == 0x80; };
    while (i < n) {
        uint8_t const c = (uint8_t)s[i];
        if (c < 0x80) {
            i += 1;
        } else if (c >= 0xC2 && c <= 0xDF) {
            if (!cont(i + 1)) return false;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (!cont(i + 1) || !cont(i + 2)) return false;
            uint8_t const d = (uint8_t)s[i + 1];
            if ((c == 0xE0 && d < 0xA0) || (c == 0xED && d > 0x9F)) return false;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (!cont(i + 1) || !cont(i + 2) || !cont(i + 3)) return false;
            uint8_t const d = (uint8_t)s[i + 1];
            if ((c == 0xF0 && d < 0x90) || (c == 0xF4 && d > 0x8F)) return false;
            i += 4;
        } else
            return false;
    }
    return true;
}
static Summary read_summary(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw AppError{"Specified file does not exist: `" + path + "`"};
    Summary s;
    struct stat st;
    stat(path.c_str(), &st);
    s.archive_size = (uint64_t)st.st_size;
    char hdr[11];
    if (fread(hdr, 1, 11, f) != 11) throw AppError{"I/O error: failed to fill whole buffer"};
    if (memcmp(hdr, "squish", 6) != 0) throw AppError{"Archive format error: Invalid archive header: prefix mismatch"};
    s.version.assign(hdr + 6, 5);
    { // header.rs:119-163: major.minor must match
        std::string v = s.version;
        size_t d1 = v.find('.'), d2 = v.find('.', d1 == std::string::npos ? 0 : d1 + 1);
        if (d1 == std::string::npos) throw AppError{"Archive format error: Invalid version format in archive header"};
        std::string maj = v.substr(0, d1), mnr = v.substr(d1 + 1, d2 == std::string::npos ? std::string::npos : d2 - d1 - 1);
        if (maj != "1" || mnr != "2")
            throw AppError{"Archive format error: Incompatible version: archive " + maj + "." + mnr + " vs current 1.2"};
    }
    rd(f, &s.timestamp, 8);
    rd(f, &s.unique_chunks, 8);
    for (uint64_t i = 0; i < s.unique_chunks; i++) {
        uint8_t h[16];
        uint64_t osz, csz;
        rd(f, h, 16);
        rd(f, &osz, 8);
        rd(f, &csz, 8);
        if (fseek(f, (long)csz, SEEK_CUR)) throw AppError{"Error reading from squish: seek"};
    }
    uint32_t nfiles;
    rd(f, &nfiles, 4);
    for (uint32_t i = 0; i < nfiles; i++) {
        uint32_t plen, cc;
        uint64_t osz;
        rd(f, &plen, 4);
        std::string p(plen, '\0');
        rd(f, &p[0], plen);
        if (!valid_utf8(p)) { // get_summary's String::from_utf8 (reader.rs:178) ends the pass here: `list` reports it,
            s.bad_utf8 = true; // `unpack` (which only takes the header fields from this summary) at its own pass over the table
            break;
        }
        rd(f, &osz, 8);
        rd(f, &cc, 4);
        fseek(f, (long)cc * 16, SEEK_CUR);
        s.total_orig += osz;
        s.files.emplace_back(p, osz);
    }
    fclose(f);
    return s;
}
static std::string date_of(uint64_t ts) { // header.rs:87-95 "%H:%M %d/%m/%Y" local time
    time_t t = (time_t)ts;
    struct tm tmv;
    localtime_r(&t, &tmv);
    char buf[64];
    strftime(buf, sizeof(buf), "%H:%M %d/%m/%Y", &tmv);
    return buf;
}
static void list_cmd(const std::string& squish, bool simple) {
    Summary s = read_summary(squish);
    if (s.bad_utf8) throw AppError{"Illegal UTF8 detected"};
    double ratio = s.total_orig ? (double)s.archive_size / (double)s.total_orig * 100.0 : 0.0;
    if (simple) { // lib.rs:67-80
        printf("squish_size(bytes): %" PRIu64 ", original_size(bytes): %" PRIu64
               ", compression ratio: %.2f%%, number_of_files: %zu, chunks_count: %" PRIu64 "\n",
               s.archive_size, s.total_orig, ratio, s.files.size(), s.unique_chunks);
        printf("%10s  File Path\n", "Size (Bytes)");
        printf("----------  --------------------\n");
        for (auto& f : s.files) printf("%10" PRIu64 "  %s\n", f.second, f.first.c_str());
        return;
    }
    // cmd/mod.rs:60-158: two prettytable tables, FORMAT_NO_LINESEP_WITH_TITLE (borders, a separator line under the
    // title row only, cells padded with one space)
    auto table = [](const std::vector<std::string>& title, bool span, const std::vector<std::pair<std::string, std::string>>& rows) {
        size_t w0 = span ? 0 : title[0].size(), w1 = span ? 0 : title[1].size();
        for (auto& r : rows) {
            w0 = std::max(w0, r.first.size());
            w1 = std::max(w1, r.second.size());
        }
        if (span && title[0].size() > w0 + w1 + 3) w1 = title[0].size() - w0 - 3; // the spanning title widens the last column
        std::string const line = "+" + std::string(w0 + 2, '-') + "+" + std::string(w1 + 2, '-') + "+\n";
        std::string o = line;
        if (span)
            o += "| " + title[0] + std::string(w0 + w1 + 3 - title[0].size(), ' ') + " |\n";
        else
            o += "| " + title[0] + std::string(w0 - title[0].size(), ' ') + " | " + title[1] + std::string(w1 - title[1].size(), ' ') + " |\n";
        o += line;
        for (auto& r : rows)
            o += "| " + r.first + std::string(w0 - r.first.size(), ' ') + " | " + r.second + std::string(w1 - r.second.size(), ' ') + " |\n";
        o += line;
        return o;
    };
    char pct[32];
    snprintf(pct, sizeof pct, "%.1f%%", ratio);
    printf("\nSquash breakdown:\n%s",
           table({"Squash Summary"}, true,
                 {{"Creation Date (UTC)", date_of(s.timestamp)},
                  {"Squish Version", s.version},
                  {"Compressed size", format_bytes(s.archive_size)},
                  {"Original size", format_bytes(s.total_orig)},
                  {"Compression Ratio", pct},
                  {"Number of files", with_commas(s.files.size())},
                  {"Number of chunks", with_commas(s.unique_chunks)}})
               .c_str());
    std::map<std::string, size_t> dirs;
    for (auto& f : s.files) dirs[f.first.substr(0, f.first.find('/'))]++;
    std::vector<std::pair<std::string, size_t>> v(dirs.begin(), dirs.end());
    std::stable_sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.second > b.second; });
    std::vector<std::pair<std::string, std::string>> rows;
    for (auto& d : v) rows.push_back({d.first, with_commas(d.second)});
    printf("\n\nTop-level directory breakdown:\n%s\n", table({"Directory", "File Count"}, false, rows).c_str());
}

// ---- unpack (archive/reader.rs:232-413; lib.rs:86-106) ------------------------------------------
// read_chunks: every chunk record is decoded into a map keyed by its hash (a later record with the same
// hash replaces the earlier one, like HashMap::insert); rebuild_files: each file is the concatenation of
// its chunks.  Decoding runs on up to `threads` host threads (the reference's Rayon pool, -j); frames are
// decoded by zstd_frame_decoder.hpp with the record's original_size as the capacity
// (zstd::bulk::decompress(&compressed_data, orig_size), reader.rs:302-303).
static std::string os_error(int e) { return std::string(strerror(e)) + " (os error " + std::to_string(e) + ")"; }
// Frame_Content_Size of a zstd frame header (RFC 8878 3.1.1.1), if the header carries one.
static bool frame_content_size(const uint8_t* p, size_t n, uint64_t* fcs) {
    if (n < 6 || p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD) return false;
    unsigned const fhd = p[4], flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
    size_t const pos = 5 + (single ? 0 : 1) + (did == 0 ? 0 : did == 1 ? 1 : did == 2 ? 2 : 4);
    size_t const fsz = flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8;
    if (fsz == 0 || pos + fsz > n) return false;
    uint64_t v = 0;
    for (size_t k = 0; k < fsz; k++) v |= (uint64_t)p[pos + k] << (8 * k);
    *fcs = fsz == 2 ? v + 256 : v;
    return true;
}
static void unpack_cmd(const std::string& squish, const std::string& outdir, long threads, int device) {
    Summary hdr = read_summary(squish); // ArchiveReader::new: header / version checks, table offsets
    FILE* f = fopen(squish.c_str(), "rb");
    if (!f) throw AppError{"Specified file does not exist: `" + squish + "`"};
    if (fseek(f, 27, SEEK_SET)) throw AppError{"Error reading from squish: seek"};
    struct Rec {
        std::string hash;
        uint64_t osz;
        const uint8_t* fptr = nullptr; // the record's frame: a stretch of the mapped archive, or `frame` (no mapping)
        size_t flen = 0;
        std::vector<uint8_t> frame, data;
        const uint8_t* dptr = nullptr; // the decoded bytes: `data`, or a stretch of a GPU batch's output buffer (outs)
        size_t dlen = 0;
        std::string err;
    };
    std::vector<Rec> recs((size_t)hdr.unique_chunks);
    double const t_read0 = now_s();
    std::vector<std::unique_ptr<uint8_t[]>> outs; // GPU decode: every batch's output buffer stays alive until the files are written
    // The records (read_chunks, reader.rs:259-314).  The archive is mapped and a record's frame is a view of the mapping (the
    // reference reads each into a Vec; here 8.6 GB of frames took 2.1 s to copy out of the page cache before anything was
    // decoded); without a mapping every frame is read into a buffer of its own as before.
    struct Mapping {
        const uint8_t* p = nullptr;
        size_t n = 0;
        ~Mapping() { if (p) munmap((void*)p, n); }
    } map;
    if (hdr.archive_size > 0 && !getenv("SQUISH_NO_MMAP")) {
        void* m = mmap(nullptr, (size_t)hdr.archive_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fileno(f), 0);
        if (m != MAP_FAILED) map.p = (const uint8_t*)m, map.n = (size_t)hdr.archive_size;
    }
    if (map.p) {
        uint64_t pos = 27;
        for (auto& r : recs) {
            if (pos + 32 > map.n) throw AppError{"Error reading from squish: failed to fill whole buffer"};
            uint64_t csz;
            r.hash.assign((const char*)map.p + pos, 16);
            memcpy(&r.osz, map.p + pos + 16, 8);
            memcpy(&csz, map.p + pos + 24, 8);
            if (csz > hdr.archive_size || pos + 32 + csz > map.n) throw AppError{"Error reading from squish: failed to fill whole buffer"};
            r.fptr = map.p + pos + 32;
            r.flen = (size_t)csz;
            pos += 32 + csz;
        }
        if (fseek(f, (long)pos, SEEK_SET)) throw AppError{"Error reading from squish: seek"};
    } else {
        for (auto& r : recs) {
            uint64_t csz;
            r.hash.resize(16);
            rd(f, &r.hash[0], 16);
            rd(f, &r.osz, 8);
            rd(f, &csz, 8);
            if (csz > hdr.archive_size) throw AppError{"Error reading from squish: failed to fill whole buffer"};
            r.frame.resize((size_t)csz);
            rd(f, r.frame.data(), (size_t)csz);
            r.fptr = r.frame.data();
            r.flen = r.frame.size();
        }
    }
    if (getenv("SQUISH_TIMING"))
        fprintf(stderr, "[squishrs timing] unpack: %zu records %s in %.3f s\n", recs.size(), map.p ? "mapped" : "read", now_s() - t_read0);
    // Where the frames are decoded: on the GPU (sq_decompress_batch: one wavefront per record, batches of records whose
    // outputs fit 4 GiB) when a gfx950 device is there and the archive is large, on host threads (zstd_frame_decoder.hpp,
    // the same statement of RFC 8878) otherwise -- `unpack` and `list` must work on a machine without a GPU, as the
    // reference's do, and a wave decodes ONE record at ~9 MiB/s: the GPU wins by decoding thousands at once (16.7 GiB/s
    // measured on 2,048 records of 2 MiB, against ~1.9 GiB/s for 16 host threads), not on a few hundred chunks, where
    // its fixed costs (context, 0.25 s for a 2 MiB record) exceed the host's whole run.  Default: GPU from 1 GiB of
    // recorded original size on.  SQUISH_UNPACK=host|gpu pins the choice (gpu: fail when there is no device);
    // SQUISH_TIMING=1 says which ran.
    bool on_gpu = false;
    {
        const char* mode = getenv("SQUISH_UNPACK");
        bool const want_host = mode && !strcmp(mode, "host"), want_gpu = mode && !strcmp(mode, "gpu");
        uint64_t total_orig = 0;
        for (auto& r : recs) total_orig += r.osz;
        bool const large = total_orig >= (1ull << 30);
        sq_ctx* ctx = nullptr;
        if (!want_host && !recs.empty() && (large || want_gpu)) {
            sq_config cfg;
            memset(&cfg, 0, sizeof(cfg));
            cfg.abi_version = SQ_ABI_VERSION;
            cfg.device = device;
            cfg.dedup_capacity = 1024;
            int const rc = sq_ctx_create(&cfg, &ctx);
            if (rc != SQ_OK) {
                ctx = nullptr;
                if (want_gpu) throw AppError{std::string("sq_ctx_create: ") + sq_strerror(rc)};
            }
        }
        double const t0 = now_s();
        if (ctx) {
            on_gpu = true;
            // A record's output needs no more room than its frame says it holds: the recorded original size is the
            // capacity zstd::bulk::decompress gets (reader.rs:302-303) and the pack path records 2 MiB for EVERY chunk
            // (writer.rs:255), so 32,768 chunks of 256 KiB used to travel back as 16 batches of 4 GiB, an eighth of
            // it data.  With min(recorded size, Frame_Content_Size) the batches' buffers hold the data back to back
            // (to 16 bytes) and are kept as they are: the files are rebuilt from them without another copy.
            uint64_t const out_budget = 4ull << 30;
            size_t i0 = 0;
            std::unique_ptr<uint8_t[]> in;
            size_t in_cap = 0;
            double t_alloc = 0, t_fill = 0, t_call = 0; // (t_fill includes t_alloc)
            int n_batches = 0;
            std::vector<uint64_t> off, ooff, want(recs.size());
            std::vector<uint32_t> len, ocap, olen, stat;
            for (size_t i = 0; i < recs.size(); i++) {
                uint64_t fcs = 0;
                want[i] = recs[i].osz;
                if (frame_content_size(recs[i].fptr, recs[i].flen, &fcs) && fcs < want[i]) want[i] = fcs;
            }
            while (i0 < recs.size()) {
                size_t i1 = i0;
                uint64_t in_bytes = 0, out_bytes = 0;
                while (i1 < recs.size() && (i1 == i0 || (out_bytes + want[i1] <= out_budget && in_bytes + recs[i1].flen <= out_budget))) {
                    if (want[i1] > 0xFFFFFFFFull || recs[i1].flen > 0xFFFFFFFFull) break;
                    in_bytes += (recs[i1].flen + 15) & ~(uint64_t)15;
                    out_bytes += (want[i1] + 15) & ~(uint64_t)15;
                    i1++;
                }
                if (i1 == i0) { // a record no batch can take (> 4 GiB): the host decoder's business
                    try {
                        recs[i0].data = zfd::decompress(recs[i0].fptr, recs[i0].flen, (size_t)recs[i0].osz);
                    } catch (const zfd::DecodeError& e) {
                        recs[i0].err = e.msg;
                    }
                    i0++;
                    continue;
                }
                size_t const n = i1 - i0;
                double const ta = now_s();
                if ((size_t)in_bytes + 16 > in_cap) {
                    in.reset();
                    in.reset(new uint8_t[(size_t)in_bytes + 16]); // (not zero-filled: every byte the decoder reads is written below)
                    in_cap = (size_t)in_bytes + 16;
                }
                outs.emplace_back(new uint8_t[(size_t)out_bytes + 16]);
                uint8_t* const out = outs.back().get();
                t_alloc += now_s() - ta;
                off.resize(n), ooff.resize(n), len.resize(n), ocap.resize(n), olen.assign(n, 0), stat.assign(n, 0);
                uint64_t a = 0, b = 0;
                for (size_t k = 0; k < n; k++) {
                    Rec& r = recs[i0 + k];
                    off[k] = a;
                    len[k] = (uint32_t)r.flen;
                    a += (r.flen + 15) & ~(uint64_t)15;
                    ooff[k] = b;
                    ocap[k] = (uint32_t)want[i0 + k];
                    b += (want[i0 + k] + 15) & ~(uint64_t)15;
                }
                memset(in.get() + a, 0, 16);
                {   // the frames into the batch buffer, on up to -j threads (8.6 GB out of the page cache: 0.9 s on one)
                    uint8_t* const dst = in.get();
                    std::atomic<size_t> nextk{0};
                    auto copy = [&]() {
                        for (;;) {
                            size_t const k0 = nextk.fetch_add(64);
                            if (k0 >= n) return;
                            for (size_t k = k0; k < std::min(n, k0 + 64); k++) {
                                const Rec& r = recs[i0 + k];
                                if (r.flen) memcpy(dst + off[k], r.fptr, r.flen);
                                size_t const pad = (size_t)(((r.flen + 15) & ~(uint64_t)15) - r.flen);
                                if (pad) memset(dst + off[k] + r.flen, 0, pad);
                            }
                        }
                    };
                    unsigned const hw = std::thread::hardware_concurrency();
                    size_t const nt = std::min<size_t>((size_t)std::max(1L, std::min<long>(threads, hw ? (long)hw : 1L)), (n + 63) / 64);
                    std::vector<std::thread> pool;
                    for (size_t t = 1; t < nt; t++) pool.emplace_back(copy);
                    copy();
                    for (auto& t : pool) t.join();
                }
                double const tb = now_s();
                t_fill += tb - ta;
                int const rc = sq_decompress_batch(ctx, in.get(), off.data(), len.data(), (uint32_t)n, out, ooff.data(), ocap.data(),
                                                   olen.data(), stat.data());
                t_call += now_s() - tb;
                n_batches++;
                if (rc != SQ_OK && rc != SQ_E_DECOMPRESS) {
                    sq_ctx_destroy(ctx);
                    throw AppError{std::string("sq_decompress_batch: ") + sq_strerror(rc)};
                }
                for (size_t k = 0; k < n; k++) {
                    Rec& r = recs[i0 + k];
                    if (stat[k])
                        r.err = sq_decompress_strerror(stat[k]);
                    else
                        r.dptr = out + ooff[k], r.dlen = olen[k]; // (no copy: the batch's buffer is kept)
                    std::vector<uint8_t>().swap(r.frame);
                }
                i0 = i1;
            }
            double const td = now_s();
            sq_ctx_destroy(ctx);
            if (getenv("SQUISH_TIMING"))
                fprintf(stderr, "[squishrs timing] unpack: %d GPU batches: buffers %.3f s, frames into the batch buffer %.3f s, sq_decompress_batch %.3f s, ctx_destroy %.3f s\n",
                        n_batches, t_alloc, t_fill - t_alloc, t_call, now_s() - td);
        } else {
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (;;) {
                    size_t const i = next.fetch_add(1);
                    if (i >= recs.size()) return;
                    try {
                        recs[i].data = zfd::decompress(recs[i].fptr, recs[i].flen, (size_t)recs[i].osz);
                    } catch (const zfd::DecodeError& e) {
                        recs[i].err = e.msg;
                    }
                    std::vector<uint8_t>().swap(recs[i].frame);
                }
            };
            unsigned hw = std::thread::hardware_concurrency();
            size_t nt = (size_t)std::max(1L, std::min<long>(threads, hw ? (long)hw : 1L));
            nt = std::min(nt, std::max<size_t>(1, recs.size()));
            std::vector<std::thread> pool;
            for (size_t t = 1; t < nt; t++) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        }
        if (getenv("SQUISH_TIMING"))
            fprintf(stderr, "[squishrs timing] unpack: %zu records decoded on the %s in %.3f s\n", recs.size(), on_gpu ? "GPU" : "host", now_s() - t0);
    }
    struct View {
        const uint8_t* p;
        size_t n;
    };
    std::unordered_map<std::string, View> chunk_map;
    for (auto& r : recs) {
        if (!r.err.empty()) throw AppError{"Error reading from squish: " + r.err};
        chunk_map[r.hash] = r.dptr ? View{r.dptr, r.dlen} : View{r.data.data(), r.data.size()};
    }
    // file table: ALL entries are read first (rebuild_files, reader.rs:337-375: a damaged table is an error before any
    // file is created), then the files are rebuilt in parallel (reader.rs:378-407, entries.par_iter() on the -j pool)
    struct Entry {
        std::string rel;
        std::vector<std::string> hashes;
        std::string err;
    };
    uint32_t nfiles;
    rd(f, &nfiles, 4);
    std::vector<Entry> entries;
    for (uint32_t i = 0; i < nfiles; i++) {
        uint32_t plen, cc;
        uint64_t osz;
        rd(f, &plen, 4);
        if (plen > hdr.archive_size) throw AppError{"Error reading from squish: failed to fill whole buffer"};
        Entry e;
        e.rel.assign(plen, '\0');
        rd(f, &e.rel[0], plen);
        if (!valid_utf8(e.rel)) throw AppError{"Illegal UTF8 detected"}; // reader.rs:350
        rd(f, &osz, 8);
        rd(f, &cc, 4);
        if ((uint64_t)cc * 16 > hdr.archive_size) throw AppError{"Error reading from squish: failed to fill whole buffer"};
        e.hashes.assign(cc, std::string(16, '\0'));
        for (auto& h : e.hashes) rd(f, &h[0], 16);
        entries.push_back(std::move(e));
    }
    fclose(f);
    namespace fs = std::filesystem;
    std::atomic<size_t> next_file{0};
    std::atomic<bool> failed{false};
    auto rebuild = [&]() {
        for (;;) {
            size_t const i = next_file.fetch_add(1);
            if (i >= entries.size() || failed.load()) return; // (try_for_each: nobody starts a new file after a failure)
            Entry& e = entries[i];
            fs::path const full = fs::path(outdir) / fs::path(e.rel); // Path::join: an absolute `rel` replaces outdir
            if (full.has_parent_path()) {
                std::error_code ec;
                fs::create_directories(full.parent_path(), ec);
                if (ec && !fs::is_directory(full.parent_path())) { // (another thread may have been creating the same directory)
                    e.err = "Error creating directory `" + full.parent_path().string() + "`: " + os_error(ec.value());
                    failed.store(true);
                    return;
                }
            }
            FILE* o = fopen(full.c_str(), "wb");
            if (!o) {
                e.err = "Error creating file `" + full.string() + "`: " + os_error(errno);
                failed.store(true);
                return;
            }
            for (auto& h : e.hashes) {
                auto it = chunk_map.find(h);
                if (it == chunk_map.end()) {
                    e.err = "Missing Chunk for File: `" + e.rel + "`";
                    break;
                }
                View const d = it->second;
                if (d.n && fwrite(d.p, 1, d.n, o) != d.n) {
                    e.err = "Error creating directory `" + e.rel + "`: " + os_error(errno); // (sic) reader.rs:392-394
                    break;
                }
            }
            if (fclose(o) && e.err.empty()) e.err = "Error creating directory `" + e.rel + "`: " + os_error(errno);
            if (!e.err.empty()) {
                failed.store(true);
                return;
            }
        }
    };
    {
        unsigned const hw = std::thread::hardware_concurrency();
        size_t nt = (size_t)std::max(1L, std::min<long>(threads, hw ? (long)hw : 1L));
        nt = std::min(nt, std::max<size_t>(1, entries.size()));
        double const t0 = now_s();
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; t++) pool.emplace_back(rebuild);
        rebuild();
        for (auto& t : pool) t.join();
        if (getenv("SQUISH_TIMING"))
            fprintf(stderr, "[squishrs timing] unpack: %zu files rebuilt by %zu threads in %.3f s\n", entries.size(), nt, now_s() - t0);
    }
    for (auto& e : entries) // the first failure in table order
        if (!e.err.empty()) throw AppError{e.err};
}

// clap's behaviour for `squishrs` (cmd/mod.rs:11-58): -h/--help prints the help on stdout and exits 0; an unknown
// flag, a missing value or a missing positional is an error on stderr with exit code 2; -j/--max-threads is global.
static const char* kHelp =
    "Compact, compress, and deduplicate files into a single archive\n\n"
    "Usage: squishrs [OPTIONS] <COMMAND>\n\nCommands:\n"
    "  pack    Pack a directory\n  list    List files in an archive\n  unpack  Extract archive contents\n"
    "  help    Print this message or the help of the given subcommand(s)\n\n"
    "Options:\n  -j, --max-threads <MAX_THREADS>  Maximum number of threads to use [default: 25]\n"
    "      --device <ORDINAL>           GPU ordinal (this implementation) [default: 0]\n"
    "  -h, --help                       Print help\n";
static int help_for(const std::string& cmd) {
    if (cmd == "pack")
        printf("Compress and deduplicate a directory into a .squish archive file\n\nUsage: squishrs pack [OPTIONS] <INPUT>\n\n"
               "Arguments:\n  <INPUT>  \n\nOptions:\n  -o, --output <OUTPUT>            \n"
               "  -j, --max-threads <MAX_THREADS>  Maximum number of threads to use [default: 25]\n"
               "      --gpus <N>                   Pack on GPUs 0..N-1, one process each (this implementation)\n"
               "      --devices <A,B,...>          ... on these GPU ordinals\n  -h, --help                       Print help\n");
    else if (cmd == "list")
        printf("Display the contents of a .squish archive\n\nUsage: squishrs list [OPTIONS] <SQUISH>\n\n"
               "Arguments:\n  <SQUISH>  \n\nOptions:\n      --simple                     \n"
               "  -j, --max-threads <MAX_THREADS>  Maximum number of threads to use [default: 25]\n  -h, --help                       Print help\n");
    else if (cmd == "unpack")
        printf("Unpacks all files from a .squish archive into a target directory\n\nUsage: squishrs unpack [OPTIONS] <SQUISH>\n\n"
               "Arguments:\n  <SQUISH>  \n\nOptions:\n  -o, --output <OUTPUT>            \n"
               "  -j, --max-threads <MAX_THREADS>  Maximum number of threads to use [default: 25]\n  -h, --help                       Print help\n");
    else
        fputs(kHelp, stdout);
    return 0;
}
static int cli_error(const std::string& what, const std::string& usage_line) {
    fprintf(stderr, "error: %s\n\nUsage: %s\n\nFor more information, try '--help'.\n", what.c_str(), usage_line.c_str());
    return 2;
}

int main(int argc, char** argv) {
    g_tty = isatty(1);
    std::vector<std::string> pos;
    std::string output;
    bool simple = false, have_output = false, want_help = false;
    int device = 0;
    std::vector<int> devices; // pack --gpus N / --devices a,b,...: one process per entry (pack_multi)
    long max_threads = 25; // cmd/mod.rs:16 (host worker count; the GPU path does not use Rayon)
    auto usage_of = [&]() -> std::string {
        if (pos.empty()) return "squishrs [OPTIONS] <COMMAND>";
        if (pos[0] == "pack") return "squishrs pack [OPTIONS] <INPUT>";
        if (pos[0] == "list") return "squishrs list [OPTIONS] <SQUISH>";
        if (pos[0] == "unpack") return "squishrs unpack [OPTIONS] <SQUISH>";
        return "squishrs [OPTIONS] <COMMAND>";
    };
    bool only_positionals = false; // after "--"
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto value = [&](const char* name) -> const char* {
            if (i + 1 < argc) return argv[++i];
            exit(cli_error(std::string("a value is required for '") + name + "' but none was supplied", usage_of()));
        };
        if (only_positionals || a.empty() || a[0] != '-' || a == "-")
            pos.push_back(a);
        else if (a == "--")
            only_positionals = true;
        else if (a == "-j" || a == "--max-threads")
            max_threads = atol(value("--max-threads <MAX_THREADS>"));
        else if (a.rfind("--max-threads=", 0) == 0)
            max_threads = atol(a.c_str() + 14);
        else if (a.rfind("-j", 0) == 0 && a.size() > 2)
            max_threads = atol(a.c_str() + 2);
        else if ((a == "-o" || a == "--output") && !pos.empty() && pos[0] != "list") {
            output = value("--output <OUTPUT>");
            have_output = true;
        } else if (a.rfind("--output=", 0) == 0 && !pos.empty() && pos[0] != "list") {
            output = a.substr(9);
            have_output = true;
        } else if (a == "--simple" && !pos.empty() && pos[0] == "list")
            simple = true;
        else if (a == "--device")
            device = atoi(value("--device <ORDINAL>"));
        else if (a == "--gpus" && !pos.empty() && pos[0] == "pack") {
            int const n = atoi(value("--gpus <N>"));
            devices.clear();
            for (int d = 0; d < n; d++) devices.push_back(d);
        } else if (a == "--devices" && !pos.empty() && pos[0] == "pack") { // (explicit ordinals; one may repeat: ranks that share a GPU)
            devices.clear();
            std::string const v = value("--devices <A,B,...>");
            for (size_t i = 0; i < v.size();) {
                size_t j = v.find(',', i);
                if (j == std::string::npos) j = v.size();
                devices.push_back(atoi(v.substr(i, j - i).c_str()));
                i = j + 1;
            }
        }
        else if (a == "-h" || a == "--help")
            want_help = true;
        else if ((a == "-V" || a == "--version") && !pos.empty()) {
            printf("squishrs-%s %s\n", pos[0].c_str(), kVersion);
            return 0;
        } else
            return cli_error("unexpected argument '" + a + "' found", usage_of());
    }
    if (want_help) return help_for(pos.empty() ? "" : pos[0]);
    if (pos.empty()) { // clap: a missing subcommand prints the help on stderr, exit code 2
        fputs(kHelp, stderr);
        return 2;
    }
    if (pos[0] == "help") return help_for(pos.size() > 1 ? pos[1] : "");
    if (pos[0] != "pack" && pos[0] != "list" && pos[0] != "unpack")
        return cli_error("unrecognized subcommand '" + pos[0] + "'", "squishrs [OPTIONS] <COMMAND>");
    if (pos.size() < 2)
        return cli_error(std::string("the following required arguments were not provided:\n  <") + (pos[0] == "pack" ? "INPUT" : "SQUISH") + ">", usage_of());
    if (pos.size() > 2) return cli_error("unexpected argument '" + pos[2] + "' found", usage_of());
    try {
        if (pos[0] == "pack") {
            std::string input = pos[1];
            if (!have_output) output = input + ".squish"; // lib.rs:31 (untrimmed input)
            uint64_t batch = 4ull << 30; // 2,048 chunks of 2 MiB: one per compressor wave (an eighth of the input at most, see pack())
            if (const char* e = getenv("SQUISH_BATCH_BYTES")) batch = strtoull(e, nullptr, 10);
            uint32_t super = 4096; // chunks per super-batch of the N-GPU path (all ranks together)
            if (const char* e = getenv("SQUISH_SUPER_BATCH_CHUNKS")) super = (uint32_t)std::max(1, atoi(e));
            if (devices.size() > 64 || (devices.empty() && false)) return cli_error("invalid value for '--gpus <N>'", usage_of());
            uint64_t sz = devices.size() >= 1 ? pack_multi(input, output, devices, super, max_threads) : pack(input, output, device, batch, max_threads);
            std::string shown = output.rfind("./", 0) == 0 ? output.substr(2) : output;
            printf("%s\nCompressed to %s\n%s: %s\n", green("Packing complete!").c_str(), shown.c_str(),
                   blue("Final archive size").c_str(), format_bytes(sz).c_str()); // lib.rs:49-55
        } else if (pos[0] == "list") {
            list_cmd(pos[1], simple);
        } else if (pos[0] == "unpack") { // lib.rs:86-106
            std::string squish = pos[1];
            if (!have_output) {
                const std::string suf = ".squish";
                output = (squish.size() >= suf.size() && squish.compare(squish.size() - suf.size(), suf.size(), suf) == 0)
                             ? squish.substr(0, squish.size() - suf.size())
                             : squish;
            }
            unpack_cmd(squish, output, max_threads, device);
            printf("%s\n%s was unsquished into /%s\n", green("Unpacking complete!").c_str(), squish.c_str(), output.c_str());
        }
    } catch (const AppError& e) {
        fprintf(stderr, "%s: %s\n", red("Error").c_str(), e.msg.c_str()); // main.rs:6-9
        return 1;
    }
    return 0;
}
