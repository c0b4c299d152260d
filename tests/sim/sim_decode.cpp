// tests/sim/sim_decode.cpp -- TEST INFRASTRUCTURE.  Builds the product's GPU zstd decoder source
// (squishrs_amd/csrc/sq_zstd_decode.h) with -DSQ_SIM, where every wave primitive has a scalar body, so its control
// logic can be checked on a machine without a GPU against frames of the real libzstd (tests/golden/).  Never linked
// into libsquish_gpu.so; not a fallback for anything.
#define SQ_SIM 1
#include "../../squishrs_amd/csrc/sq_zstd_decode.h"
#include <stdlib.h>

extern "C" unsigned sim_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_len, int poison) {
    static thread_local sqdz::DecLds lds;
    memset(&lds, poison ? 0x5A : 0, sizeof(lds));
    sqdz::dec_lds_init(&lds);
    u8* lit = (u8*)malloc(SQDZ_LITBUF_BYTES);
    memset(lit, poison ? 0xA5 : 0, SQDZ_LITBUF_BYTES);
    u32 len = 0;
    u32 const st = sqdz::decode_frames_wave(src, (u32)n, dst, (u32)cap, lit, &lds, &len);
    free(lit);
    *out_len = st ? 0 : len;
    return st;
}
extern "C" unsigned sim_decode_lds_bytes(void) { return (unsigned)sizeof(sqdz::DecLds); }
