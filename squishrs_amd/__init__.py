"""squishrs_amd -- MI355X (gfx950) implementation of squishRS's pack-side hot path:
per-chunk XXH3-128 content hash, global dedup set, zstd level-12 frame compressor, behind the C
ABI of include/squish_gpu.h (libsquish_gpu.so, hand-written HIP in squishrs_amd/csrc/)."""
from ._lib import SquishError, LIB_PATH  # noqa: F401
from .chunk import (CHUNK_SIZE, COMPRESSION_LEVEL, ChunkStore, InsertReturn, compress_bound,  # noqa: F401
                    compress_chunk, hash_chunk)
