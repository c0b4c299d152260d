"""ctypes binding of libsquish_gpu.so (the C ABI in include/squish_gpu.h).

The library is the product: there is no Python or CPU implementation behind it.  If it is missing
or no gfx950 device is usable, every call raises -- nothing falls back to the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SQUISH_GPU_LIB selects another build of the same library (e.g. the -DSQ_PROF phase-timer build)
LIB_PATH = os.environ.get("SQUISH_GPU_LIB") or os.path.join(_HERE, "libsquish_gpu.so")

SQ_OK = 0
SQ_E_INVAL, SQ_E_NOMEM, SQ_E_CAP, SQ_E_DEVICE, SQ_E_COMPRESS, SQ_E_NODEVICE, SQ_E_DECOMPRESS = -1, -2, -3, -4, -5, -6, -7
SQ_ABI_VERSION = 1
SQ_CHUNK_SIZE = 2048 * 1024
SQ_CFG_NEVER_FUSE_HASH, SQ_CFG_ALWAYS_FUSE_HASH = 1, 2

EXPORTS = [
    "sq_abi_version", "sq_strerror", "sq_compress_bound", "sq_ctx_create", "sq_ctx_destroy",
    "sq_ctx_set_stream", "sq_ctx_synchronize", "sq_hash_chunk", "sq_compress_chunk",
    "sq_pack_batch", "sq_unique_count", "sq_hash_batch", "sq_dedup_global", "sq_compress_batch",
    "sq_calibrate_read_bw", "sq_last_timings", "sq_dedup_reset", "sq_compress_batch_slots", "sq_slots_bound",
    "sq_host_alloc", "sq_host_free", "sq_debug_stage2_list", "sq_pack_submit", "sq_pack_wait", "sq_calibrate_gather", "sq_calibrate_gather2",
    "sq_shard_submit_hash", "sq_shard_wait_hash", "sq_shard_submit_rest", "sq_shard_submit_rest_idx",
    "sq_decompress_batch", "sq_decompress_strerror", "sq_debug_stage2_kernel", "sq_ctx_trim", "sq_dev_alloc", "sq_dev_free", "sq_dev_copy",
]


class SqConfig(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_uint32), ("device", ctypes.c_int32),
                ("max_chunk", ctypes.c_uint32), ("max_batch_chunks", ctypes.c_uint32),
                ("max_batch_bytes", ctypes.c_uint64), ("dedup_capacity", ctypes.c_uint64),
                ("compress_waves", ctypes.c_uint32), ("flags", ctypes.c_uint32)]


class SqTimings(ctypes.Structure):
    _fields_ = [("hash_ms", ctypes.c_float), ("dedup_ms", ctypes.c_float),
                ("compress_ms", ctypes.c_float), ("compact_ms", ctypes.c_float), ("fast_ms", ctypes.c_float),
                ("fast_decided", ctypes.c_uint32), ("fused", ctypes.c_uint32)]


class SquishError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsquish_gpu: {msg} ({code})")
        self.code = code


_lib = None


def load():
    """Load libsquish_gpu.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SquishError(SQ_E_NODEVICE, f"{LIB_PATH} not built; run __graft_entry__.build()")
    # One HIP runtime per process: torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).
    # Importing torch first makes the loader resolve our DT_NEEDED libamdhip64.so.7 to that copy;
    # the other order would give the process two runtimes and torch would see no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p = ctypes.c_void_p, ctypes.c_void_p
    L.sq_abi_version.restype = ctypes.c_uint32
    L.sq_strerror.restype = ctypes.c_char_p
    L.sq_strerror.argtypes = [ctypes.c_int]
    L.sq_compress_bound.restype = ctypes.c_size_t
    L.sq_compress_bound.argtypes = [ctypes.c_size_t]
    L.sq_ctx_create.argtypes = [ctypes.POINTER(SqConfig), ctypes.POINTER(vp)]
    L.sq_ctx_destroy.argtypes = [vp]
    L.sq_ctx_destroy.restype = None
    L.sq_ctx_set_stream.argtypes = [vp, vp]
    L.sq_ctx_synchronize.argtypes = [vp]
    L.sq_ctx_trim.argtypes = [vp]
    L.sq_hash_chunk.argtypes = [u8p, ctypes.c_size_t, u8p]
    L.sq_compress_chunk.argtypes = [u8p, ctypes.c_size_t, u8p, ctypes.c_size_t,
                                    ctypes.POINTER(ctypes.c_size_t)]
    L.sq_pack_batch.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p, u8p, u8p, ctypes.c_uint64, vp]
    L.sq_unique_count.restype = ctypes.c_uint64
    L.sq_unique_count.argtypes = [vp]
    L.sq_hash_batch.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p]
    L.sq_dedup_global.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, u8p]
    L.sq_compress_batch.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p, u8p, ctypes.c_uint64, vp]
    L.sq_compress_batch_slots.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p, u8p, ctypes.c_uint64, vp, vp]
    L.sq_slots_bound.restype = ctypes.c_uint64
    L.sq_slots_bound.argtypes = [vp, ctypes.c_uint32]
    L.sq_calibrate_read_bw.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)]
    L.sq_calibrate_gather.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    L.sq_calibrate_gather2.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    L.sq_last_timings.argtypes = [vp, ctypes.POINTER(SqTimings)]
    L.sq_dedup_reset.argtypes = [vp]
    L.sq_pack_submit.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p, u8p, u8p, ctypes.c_uint64, ctypes.c_uint32]
    L.sq_pack_wait.argtypes = [vp, vp, vp]
    L.sq_shard_submit_hash.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p]
    L.sq_shard_wait_hash.argtypes = [vp]
    L.sq_shard_submit_rest.argtypes = [vp, u8p, ctypes.c_uint64, ctypes.c_uint64, u8p, u8p, ctypes.c_uint64, ctypes.c_uint32]
    L.sq_shard_submit_rest_idx.argtypes = [vp, u8p, ctypes.c_uint64, vp, u8p, u8p, ctypes.c_uint64, ctypes.c_uint32]
    L.sq_debug_stage2_list.argtypes = [vp, vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    L.sq_debug_stage2_kernel.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.sq_decompress_batch.argtypes = [vp, u8p, vp, vp, ctypes.c_uint32, u8p, vp, vp, vp, vp]
    L.sq_decompress_strerror.restype = ctypes.c_char_p
    L.sq_decompress_strerror.argtypes = [ctypes.c_uint32]
    _lib = L
    return L


def check(code):
    if code != SQ_OK:
        raise SquishError(code, load().sq_strerror(code).decode())
