"""Build-container helpers: ctypes bindings to the REAL libzstd 1.5.7 (the C library that
zstd-sys 2.0.15+zstd.1.5.7 wraps, Cargo.lock:1637-1640) and to python-xxhash, used ONLY to pin
the CPU oracle and to generate tests/golden/ fixtures.  Never imported by the product.

libzstd 1.5.7 is found in the image's pillow wheel; if it is absent (or another version) every
entry point raises LookupError and callers skip."""
import ctypes
import glob
import os

_CANDIDATES = [
    "/usr/local/lib/python3.10/dist-packages/pillow.libs/libzstd-*.so*",
    "/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so*",
    "/usr/lib/python3/dist-packages/pillow.libs/libzstd-*.so*",
]


class ZSTD_Sequence(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint), ("litLength", ctypes.c_uint),
                ("matchLength", ctypes.c_uint), ("rep", ctypes.c_uint)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    for pat in _CANDIDATES:
        for p in sorted(glob.glob(pat)):
            try:
                z = ctypes.CDLL(p)
                z.ZSTD_versionNumber.restype = ctypes.c_uint
                if z.ZSTD_versionNumber() != 10507:
                    continue
            except OSError:
                continue
            z.ZSTD_compress.restype = ctypes.c_size_t
            z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_int]
            z.ZSTD_compressBound.restype = ctypes.c_size_t
            z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            z.ZSTD_decompress.restype = ctypes.c_size_t
            z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                          ctypes.c_size_t]
            z.ZSTD_isError.restype = ctypes.c_uint
            z.ZSTD_isError.argtypes = [ctypes.c_size_t]
            z.ZSTD_createCCtx.restype = ctypes.c_void_p
            z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
            z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
            z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            z.ZSTD_generateSequences.restype = ctypes.c_size_t
            z.ZSTD_generateSequences.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                 ctypes.c_void_p, ctypes.c_size_t]
            z.ZSTD_sequenceBound.restype = ctypes.c_size_t
            z.ZSTD_sequenceBound.argtypes = [ctypes.c_size_t]
            z.ZSTD_compress2.restype = ctypes.c_size_t
            z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_void_p, ctypes.c_size_t]
            _lib = z
            return z
    raise LookupError("libzstd 1.5.7 not found in this image")


def available():
    try:
        lib()
        return True
    except LookupError:
        return False


def compress(data: bytes, level: int = 12) -> bytes:
    """== zstd::bulk::compress(data, level) (src/util/chunk.rs:89-90)."""
    z = lib()
    cap = z.ZSTD_compressBound(len(data))
    dst = ctypes.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, data, len(data), level)
    if z.ZSTD_isError(n):
        raise RuntimeError("ZSTD_compress failed")
    return dst.raw[:n]


def decompress(frame: bytes, cap: int) -> bytes:
    z = lib()
    dst = ctypes.create_string_buffer(max(cap, 1))
    n = z.ZSTD_decompress(dst, cap, frame, len(frame))
    if z.ZSTD_isError(n):
        raise RuntimeError("ZSTD_decompress failed")
    return dst.raw[:n]


def generate_sequences(data: bytes, level: int = 12):
    """Per-block match-finder output: list of (offset, litLength, matchLength, rep); block
    delimiters are (0, lastLits, 0, 0)."""
    z = lib()
    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 100, level)  # ZSTD_c_compressionLevel
    cap = z.ZSTD_sequenceBound(len(data))
    arr = (ZSTD_Sequence * cap)()
    n = z.ZSTD_generateSequences(cctx, arr, cap, data, len(data))
    z.ZSTD_freeCCtx(cctx)
    if z.ZSTD_isError(n):
        raise RuntimeError("ZSTD_generateSequences failed")
    return [(arr[i].offset, arr[i].litLength, arr[i].matchLength, arr[i].rep) for i in range(n)]


def xxh3_128_le(data: bytes) -> bytes:
    """== hash_chunk (src/util/chunk.rs:46-49): xxh3_128(data).to_le_bytes()."""
    import xxhash
    return xxhash.xxh3_128_intdigest(data).to_bytes(16, "little")


def parse_frame(frame: bytes):
    """Tiny RFC 8878 frame walker for debugging: returns (content_size, [(type, size, body)])."""
    assert frame[:4] == b"\x28\xb5\x2f\xfd"
    fhd = frame[4]
    fcs_code = fhd >> 6
    single = (fhd >> 5) & 1
    pos = 5
    if not single:
        pos += 1
    fcs_len = [1 if single else 0, 2, 4, 8][fcs_code]
    fcs = int.from_bytes(frame[pos:pos + fcs_len], "little") if fcs_len else None
    if fcs_code == 1:
        fcs += 256
    pos += fcs_len
    blocks = []
    while True:
        h = int.from_bytes(frame[pos:pos + 3], "little")
        pos += 3
        last, typ, size = h & 1, (h >> 1) & 3, h >> 3
        body_len = size if typ in (0, 2) else 1
        blocks.append((("raw", "rle", "cmp")[typ], size, frame[pos:pos + body_len]))
        pos += body_len
        if last:
            break
    assert pos == len(frame), (pos, len(frame))
    return fcs, blocks
