/*
 * zstd12_entropy.h -- CPU ORACLE (test infrastructure, see sq_oracle.h).
 * Entropy stage of the zstd level-12 block compressor: FSE, Huffman, literals section,
 * sequences section.  Restates the published libzstd 1.5.7 encoder (the C library behind
 * zstd-sys 2.0.15+zstd.1.5.7, Cargo.lock:1637-1640) that
 * /root/reference/src/util/chunk.rs:89-90 calls as zstd::bulk::compress(chunk, 12).
 * libzstd function names are quoted in comments so a reader can follow the correspondence.
 */
#ifndef ZSTD12_ENTROPY_H
#define ZSTD12_ENTROPY_H
#include <stddef.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;

#define ERR ((size_t)-1)
static inline int is_err(size_t c) { return c > (size_t)-120; }

static inline u32 highbit32(u32 v) { return 31 - (u32)__builtin_clz(v); }

/* ---- FSE compression table (FSE_CTable) ---- */
#define FSE_MAXLOG 9
#define FSE_MAXSYM 52
typedef struct {
    int deltaFindState;
    u32 deltaNbBits;
} fse_tt;
typedef struct {
    u16 tableLog;
    u16 maxSym;
    u16 stateTable[1 << FSE_MAXLOG];
    fse_tt tt[FSE_MAXSYM + 1];
} fse_ct;

/* ---- Huffman compression table (HUF_CElt[]) ---- */
typedef struct {
    u8 nbBits[256];
    u16 val[256];
    u32 maxSym;   /* HUF_CTableHeader.maxSymbolValue */
    u32 tableLog; /* HUF_CTableHeader.tableLog */
} huf_ct;

enum { REP_NONE = 0, REP_CHECK = 1, REP_VALID = 2 };
enum { SET_BASIC = 0, SET_RLE = 1, SET_COMPRESSED = 2, SET_REPEAT = 3 };

/* ZSTD_entropyCTables_t */
typedef struct {
    huf_ct huf;
    u32 huf_repeat;
    fse_ct ll, of, ml;
    u32 ll_repeat, of_repeat, ml_repeat;
} entropy_t;

/* SeqDef */
typedef struct {
    u32 offBase;
    u16 litLength;
    u16 mlBase;
} seqdef;

typedef struct {
    seqdef* seqs;
    u32 nseq;
    u8* lits;
    u32 nlits;
    u32 longLengthType; /* 0 none, 1 literalLength, 2 matchLength */
    u32 longLengthPos;
    u8 *llCode, *ofCode, *mlCode; /* capacity >= max nseq */
} seqstore;

extern const u8 zs_LL_bits[36];
extern const u8 zs_ML_bits[53];
u32 zs_LLcode(u32 litLength); /* ZSTD_LLcode */
u32 zs_MLcode(u32 mlBase);    /* ZSTD_MLcode */

void entropy_reset(entropy_t* e); /* ZSTD_reset_compressedBlockState (tables part) */

/* ZSTD_entropyCompressSeqStore: returns compressed block body size, 0 = "emit raw block",
 * ERR on internal error. */
size_t entropy_compress_seqstore(const seqstore* ss, const entropy_t* prev, entropy_t* next,
                                 u32 strategy, u8* dst, size_t dstCap, size_t blockSize);

#endif
