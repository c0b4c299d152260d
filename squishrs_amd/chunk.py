"""Host-side mirror of the reference's chunk API (src/util/chunk.rs) over the C ABI.

Same names and argument meaning as the reference: `hash_chunk`, `ChunkStore.insert/len/is_empty`,
`CHUNK_SIZE`, `InsertReturn{hash, compressed_data}`; errors surface as SquishError whose text for
compressor failures is the reference's "Compression error" (src/util/errors.rs:25-26).
Batch forms (`insert_batch`, device-resident `pack_device`) are additive.

torch is used only for device memory, streams and (in parallel.py) torch.distributed."""
import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import SqConfig, SqTimings, SquishError, check

CHUNK_SIZE = 2048 * 1024   # src/util/chunk.rs:11
COMPRESSION_LEVEL = 12     # src/util/chunk.rs:12


@dataclass
class InsertReturn:        # src/util/chunk.rs:14-17
    hash: bytes
    compressed_data: Optional[bytes]


def compress_bound(n: int) -> int:
    return int(_lib.load().sq_compress_bound(n))


def hash_chunk(chunk: bytes) -> bytes:
    """== hash_chunk (src/util/chunk.rs:46-49), computed by the HIP kernel."""
    L = _lib.load()
    out = ctypes.create_string_buffer(16)
    buf = (ctypes.c_char * max(len(chunk), 1)).from_buffer_copy(chunk or b"\0")
    check(L.sq_hash_chunk(ctypes.addressof(buf), len(chunk), ctypes.addressof(out)))
    return out.raw


def compress_chunk(chunk: bytes) -> bytes:
    """== zstd::bulk::compress(chunk, 12) (src/util/chunk.rs:89-90), computed by the HIP kernel."""
    L = _lib.load()
    cap = compress_bound(len(chunk)) + 64
    dst = ctypes.create_string_buffer(cap)
    n = ctypes.c_size_t(0)
    buf = (ctypes.c_char * max(len(chunk), 1)).from_buffer_copy(chunk or b"\0")
    check(L.sq_compress_chunk(ctypes.addressof(buf), len(chunk), ctypes.addressof(dst), cap,
                              ctypes.byref(n)))
    return dst.raw[:n.value]


class ChunkStore:
    """== ChunkStore (src/util/chunk.rs:19-24,52-56,80-136); the dedup set lives in HBM."""

    def __init__(self, device: int = 0, max_chunk: int = CHUNK_SIZE, dedup_capacity: int = 1 << 22,
                 compress_waves: int = 0, fuse_hash: Optional[bool] = None):
        """fuse_hash: None = the library chooses per batch (from the previous batch's duplicate ratio) whether the
        first compress stage also computes the hashes; True / False force one form (same results)."""
        L = _lib.load()
        flags = 0 if fuse_hash is None else (_lib.SQ_CFG_ALWAYS_FUSE_HASH if fuse_hash else _lib.SQ_CFG_NEVER_FUSE_HASH)
        cfg = SqConfig(_lib.SQ_ABI_VERSION, device, max_chunk, 0, 0, dedup_capacity, compress_waves, flags)
        h = ctypes.c_void_p()
        check(L.sq_ctx_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self._L = L
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.sq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- reference API ---------------------------------------------------------------------
    def insert(self, chunk: bytes) -> InsertReturn:
        return self.insert_batch([chunk])[0]

    def len(self) -> int:
        return int(self._L.sq_unique_count(self._h))

    def __len__(self):
        return self.len()

    def is_empty(self) -> bool:
        return self.len() == 0

    def reset(self):
        """ChunkStore::new on the same device buffers (empties the dedup set)."""
        check(self._L.sq_dedup_reset(self._h))

    # -- batch forms -----------------------------------------------------------------------
    def insert_batch(self, chunks: Sequence[bytes]) -> List[InsertReturn]:
        """n x insert in index order through sq_pack_batch with host buffers."""
        n = len(chunks)
        if n == 0:
            return []
        lens = np.array([len(c) for c in chunks], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        offs[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
        data = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        hashes = np.zeros(n * 16, dtype=np.uint8)
        is_new = np.zeros(n, dtype=np.uint8)
        cap = int(sum(compress_bound(int(x)) + 64 for x in lens))
        comp = np.zeros(cap, dtype=np.uint8)
        comp_off = np.zeros(n + 1, dtype=np.uint64)
        check(self._L.sq_pack_batch(self._h, data.ctypes.data, offs.ctypes.data, lens.ctypes.data, n,
                                    hashes.ctypes.data, is_new.ctypes.data, comp.ctypes.data, cap,
                                    comp_off.ctypes.data))
        out = []
        for i in range(n):
            h = hashes[16 * i:16 * i + 16].tobytes()
            if is_new[i]:
                out.append(InsertReturn(h, comp[int(comp_off[i]):int(comp_off[i + 1])].tobytes()))
            else:
                out.append(InsertReturn(h, None))
        return out

    # -- device-resident stage API (torch tensors on this store's device) ---------------------
    def set_stream(self, stream_ptr: int):
        check(self._L.sq_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def synchronize(self):
        check(self._L.sq_ctx_synchronize(self._h))

    def hash_device(self, data, offs: np.ndarray, lens: np.ndarray, hashes):
        check(self._L.sq_hash_batch(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data,
                                    len(lens), hashes.data_ptr()))

    def dedup_device(self, all_hashes, n_total: int, my_first: int, my_n: int, is_new):
        check(self._L.sq_dedup_global(self._h, all_hashes.data_ptr(), n_total, my_first, my_n,
                                      is_new.data_ptr() if my_n else None))

    def compress_device(self, data, offs: np.ndarray, lens: np.ndarray, select, comp) -> np.ndarray:
        comp_off = np.zeros(len(lens) + 1, dtype=np.uint64)
        check(self._L.sq_compress_batch(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data,
                                        len(lens), select.data_ptr() if select is not None else None,
                                        comp.data_ptr(), comp.numel(), comp_off.ctypes.data))
        return comp_off

    def compress_device_slots(self, data, offs: np.ndarray, lens: np.ndarray, select, comp):
        """sq_compress_batch_slots: frames stay in their slots of `comp`; returns (offsets[n], lengths[n])."""
        comp_off = np.zeros(len(lens), dtype=np.uint64)
        comp_len = np.zeros(len(lens), dtype=np.uint32)
        check(self._L.sq_compress_batch_slots(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data,
                                              len(lens), select.data_ptr() if select is not None else None,
                                              comp.data_ptr(), comp.numel(), comp_off.ctypes.data,
                                              comp_len.ctypes.data))
        return comp_off, comp_len

    def slots_bound(self, lens: np.ndarray) -> int:
        return int(self._L.sq_slots_bound(lens.ctypes.data, len(lens)))

    def pack_device(self, data, offs: np.ndarray, lens: np.ndarray, hashes, is_new, comp) -> np.ndarray:
        """sq_pack_batch with every buffer resident in HBM (nothing crosses PCIe but offsets)."""
        comp_off = np.zeros(len(lens) + 1, dtype=np.uint64)
        check(self._L.sq_pack_batch(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data,
                                    len(lens), hashes.data_ptr(), is_new.data_ptr(), comp.data_ptr(),
                                    comp.numel(), comp_off.ctypes.data))
        return comp_off

    def pack_submit(self, data, offs: np.ndarray, lens: np.ndarray, hashes, is_new, comp, reset: bool = False):
        """sq_pack_submit: enqueue hash -> dedup -> compress (slots layout) of one device-resident batch and return;
        at most two batches in flight, each with its own hashes / is_new / comp buffers."""
        check(self._L.sq_pack_submit(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data, len(lens),
                                     hashes.data_ptr(), is_new.data_ptr(), comp.data_ptr(), comp.numel(),
                                     1 if reset else 0))

    def pack_wait(self, n: int):
        """sq_pack_wait: (slot offsets[n], frame lengths[n]) of the oldest batch in flight."""
        comp_off = np.zeros(n, dtype=np.uint64)
        comp_len = np.zeros(n, dtype=np.uint32)
        check(self._L.sq_pack_wait(self._h, comp_off.ctypes.data, comp_len.ctypes.data))
        return comp_off, comp_len

    def shard_submit_hash(self, data, offs: np.ndarray, lens: np.ndarray, hashes):
        """sq_shard_submit_hash: enqueue the hash of this rank's batch on the next free lane."""
        check(self._L.sq_shard_submit_hash(self._h, data.data_ptr(), offs.ctypes.data, lens.ctypes.data, len(lens),
                                           hashes.data_ptr()))

    def shard_wait_hash(self):
        """sq_shard_wait_hash: the hashes of the batch submitted last are in place (the exchange may start)."""
        check(self._L.sq_shard_wait_hash(self._h))

    def shard_submit_rest(self, all_hashes, n_total: int, my_first: int, is_new, comp, reset: bool = False):
        """sq_shard_submit_rest: global dedup over every rank's hashes + compression of this rank's new chunks, enqueued
        behind the hash; collected by pack_wait."""
        check(self._L.sq_shard_submit_rest(self._h, all_hashes.data_ptr(), n_total, my_first, is_new.data_ptr(),
                                           comp.data_ptr(), comp.numel(), 1 if reset else 0))

    def shard_submit_rest_idx(self, all_hashes, n_total: int, my_idx, is_new, comp, reset: bool = False):
        """sq_shard_submit_rest_idx: as shard_submit_rest for any assignment of chunks to ranks -- all_hashes in GLOBAL
        chunk-index order, my_idx (device int32/uint32 tensor) = the global index of each of this rank's chunks."""
        check(self._L.sq_shard_submit_rest_idx(self._h, all_hashes.data_ptr(), n_total, my_idx.data_ptr(), is_new.data_ptr(),
                                               comp.data_ptr(), comp.numel(), 1 if reset else 0))

    def decompress_device(self, frames, offs: np.ndarray, lens: np.ndarray, out, out_offs: np.ndarray, out_caps: np.ndarray):
        """sq_decompress_batch on device tensors: record i = frames[offs[i] : offs[i] + lens[i]] is decoded into
        out[out_offs[i] : out_offs[i] + out_caps[i]] (== zstd::bulk::decompress(&data, orig_size), reader.rs:302-303).
        Returns (out_len[n], status[n]); raises only for errors other than "some record did not decode"."""
        n = len(lens)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        out_offs = np.ascontiguousarray(out_offs, dtype=np.uint64)
        out_caps = np.ascontiguousarray(out_caps, dtype=np.uint32)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint32)
        rc = self._L.sq_decompress_batch(self._h, frames.data_ptr(), offs.ctypes.data, lens.ctypes.data, n, out.data_ptr(),
                                         out_offs.ctypes.data, out_caps.ctypes.data, out_len.ctypes.data, status.ctypes.data)
        if rc != _lib.SQ_OK and rc != _lib.SQ_E_DECOMPRESS:
            check(rc)
        return out_len, status

    def decompress_batch(self, frames: Sequence[bytes], caps: Sequence[int]) -> List[Optional[bytes]]:
        """Host convenience form: decodes every record on the GPU; entry i is the decoded bytes, or None when the record
        did not decode (self.last_decode_errors[i] then holds libzstd's message for it)."""
        n = len(frames)
        if n == 0:
            return []
        lens = np.array([len(f) for f in frames], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        offs[1:] = np.cumsum((lens.astype(np.uint64) + 15) & ~np.uint64(15))[:-1]
        total = int(offs[-1] + lens[-1])
        buf = bytearray(total + 16)
        for f, o in zip(frames, offs):
            buf[int(o):int(o) + len(f)] = f
        ocap = np.array(caps, dtype=np.uint32)
        ooff = np.zeros(n, dtype=np.uint64)
        ooff[1:] = np.cumsum((ocap.astype(np.uint64) + 15) & ~np.uint64(15))[:-1]
        out = ctypes.create_string_buffer(int(ooff[-1] + ocap[-1]) + 16)
        src = (ctypes.c_char * len(buf)).from_buffer(buf)
        out_len = np.zeros(n, dtype=np.uint32)
        status = np.zeros(n, dtype=np.uint32)
        rc = self._L.sq_decompress_batch(self._h, ctypes.addressof(src), offs.ctypes.data, lens.ctypes.data, n, ctypes.addressof(out),
                                         ooff.ctypes.data, ocap.ctypes.data, out_len.ctypes.data, status.ctypes.data)
        if rc != _lib.SQ_OK and rc != _lib.SQ_E_DECOMPRESS:
            check(rc)
        self.last_decode_errors = [None if s == 0 else self._L.sq_decompress_strerror(int(s)).decode() for s in status]
        raw = out.raw
        return [raw[int(o):int(o) + int(l)] if s == 0 else None for o, l, s in zip(ooff, out_len, status)]

    def trim(self):
        """give the context's cached device buffers (workspace pools, staging) back to the device (sq_ctx_trim)"""
        check(self._L.sq_ctx_trim(self._h))

    def stage2_kernel(self):
        """(loaded, used_last): is the second build of the compressor kernel (squish_s2_w3.co) available, and did the most
        recent compress stage run it (include/squish_gpu.h sq_debug_stage2_kernel)"""
        a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
        check(self._L.sq_debug_stage2_kernel(self._h, ctypes.byref(a), ctypes.byref(b)))
        return bool(a.value), bool(b.value)

    def stage2_list(self, cap: int = 1 << 20) -> np.ndarray:
        """Indices of the chunks the last compress launch's first stage left to the compressor proper."""
        out = np.zeros(cap, dtype=np.uint32)
        k = ctypes.c_uint32(0)
        check(self._L.sq_debug_stage2_list(self._h, out.ctypes.data, cap, ctypes.byref(k)))
        return out[:k.value].copy()

    def read_bw_ms(self, buf) -> float:
        ms = ctypes.c_float(0)
        check(self._L.sq_calibrate_read_bw(self._h, buf.data_ptr(), buf.numel(), ctypes.byref(ms)))
        return ms.value

    def gather_rate(self, buf, per_lane: int = 512) -> float:
        """sq_calibrate_gather: random 16-byte loads per second over `buf` (device tensor)."""
        ms = ctypes.c_float(0)
        req = ctypes.c_uint64(0)
        check(self._L.sq_calibrate_gather(self._h, buf.data_ptr(), buf.numel(), per_lane, ctypes.byref(ms), ctypes.byref(req)))
        return req.value / (ms.value / 1e3)

    def gather_rate2(self, buf, nbytes: int, req_bytes: int = 16, depth: int = 8, waves_per_cu: int = 32, per_lane: int = 512) -> float:
        """sq_calibrate_gather2: random requests of req_bytes per second over the first nbytes of `buf`."""
        ms = ctypes.c_float(0)
        req = ctypes.c_uint64(0)
        check(self._L.sq_calibrate_gather2(self._h, buf.data_ptr(), nbytes, per_lane, req_bytes, depth, waves_per_cu, ctypes.byref(ms), ctypes.byref(req)))
        return req.value / (ms.value / 1e3)

    def timings(self) -> SqTimings:
        t = SqTimings()
        check(self._L.sq_last_timings(self._h, ctypes.byref(t)))
        return t
