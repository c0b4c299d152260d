// corpus_gen.hip -- synthetic-workload generator kernels for bench.py and the GPU tests (gfx950).
// NOT part of the product (libsquish_gpu.so): measurement tooling, built to tools/libsq_corpus.so by
// __graft_entry__.build().
//
// sqgen_random_chunks: the device twin of tools/corpus.py:random_bytes -- chunk i is the byte stream of 64
// interleaved xoshiro256** generators (one per lane) seeded through a splitmix64 chain from
// (seed_base ^ (first + i)); round r of a chunk is the 512 bytes out[r*64 + lane].  One kernel launch for the
// whole workload (the torch-op version this replaces issued ~10^5 tiny elementwise kernels, which is slow and
// was the trigger of a profiler crash under `rocprofv3 --pmc`, see DESIGN.md "Incidents").
// sqgen_gather_chunks: dst chunk i = src chunk index[i] (exact duplicates for BASELINE configs[2] / [4]).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 splitmix_out(u64 x) { // output of the splitmix64 step whose state after the add is x
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ u64 rotl64(u64 x, int k) { return (x << k) | (x >> (64 - k)); }

// one wave per (chunk, slice of rounds)
// index == nullptr: chunk i carries stream (first + i); otherwise stream index[i] (equal entries = exact duplicates)
__global__ void __launch_bounds__(256) k_random_chunks(u64* __restrict__ out, u64 seed_base, u64 first, u64 n, u64 words_per_chunk,
                                                       const long long* __restrict__ index) {
    const u32 lane = threadIdx.x & 63;
    const u64 chunk = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (chunk >= n) return;
    const u64 G = 0x9E3779B97F4A7C15ull;
    const u64 x0 = seed_base ^ (index ? (u64)index[chunk] : first + chunk);
    u64 s0 = splitmix_out(x0 + (4 * lane + 1) * G), s1 = splitmix_out(x0 + (4 * lane + 2) * G),
        s2 = splitmix_out(x0 + (4 * lane + 3) * G), s3 = splitmix_out(x0 + (4 * lane + 4) * G);
    u64* const o = out + chunk * words_per_chunk;
    const u64 rounds = words_per_chunk / 64;
    for (u64 r = 0; r < rounds; r++) {
        o[r * 64 + lane] = rotl64(s1 * 5, 7) * 9;
        const u64 t = s1 << 17;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = rotl64(s3, 45);
    }
}

__global__ void __launch_bounds__(256) k_gather_chunks(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                       const long long* __restrict__ index, u64 n, u64 vec_per_chunk) {
    const u64 chunk = blockIdx.x;
    if (chunk >= n) return;
    const uint4* s = src + (u64)index[chunk] * vec_per_chunk;
    uint4* d = dst + chunk * vec_per_chunk;
    for (u64 i = threadIdx.x; i < vec_per_chunk; i += blockDim.x) d[i] = s[i];
}

extern "C" {

// d_out: n * L bytes (L a multiple of 512, d_out 8-byte aligned).  Returns 0 or a hipError_t.
int sqgen_random_chunks(void* d_out, uint64_t seed_base, uint64_t first, uint64_t n, uint64_t L, void* stream) {
    if (L % 512 != 0 || ((uintptr_t)d_out & 7)) return -1;
    if (n == 0) return 0;
    const u32 wpb = 4;
    hipLaunchKernelGGL(k_random_chunks, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(64 * wpb), 0, (hipStream_t)stream,
                       (u64*)d_out, seed_base, first, n, L / 8, (const long long*)nullptr);
    return (int)hipGetLastError();
}

// chunk i = stream d_index[i] (int64, device): equal entries give exact duplicates (BASELINE configs[2] / [4]).
int sqgen_random_chunks_indexed(void* d_out, uint64_t seed_base, const void* d_index, uint64_t n, uint64_t L, void* stream) {
    if (L % 512 != 0 || ((uintptr_t)d_out & 7) || !d_index) return -1;
    if (n == 0) return 0;
    const u32 wpb = 4;
    hipLaunchKernelGGL(k_random_chunks, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(64 * wpb), 0, (hipStream_t)stream,
                       (u64*)d_out, seed_base, (u64)0, n, L / 8, (const long long*)d_index);
    return (int)hipGetLastError();
}

// dst chunk i (L bytes, L a multiple of 16, both buffers 16-byte aligned) = src chunk d_index[i] (int64, device).
int sqgen_gather_chunks(void* d_dst, const void* d_src, const void* d_index, uint64_t n, uint64_t L, void* stream) {
    if (L % 16 != 0 || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_src & 15)) return -1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gather_chunks, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, (uint4*)d_dst,
                       (const uint4*)d_src, (const long long*)d_index, n, L / 16);
    return (int)hipGetLastError();
}
}
