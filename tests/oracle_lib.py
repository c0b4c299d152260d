"""ctypes access to the CPU oracle (oracle/_build/libsq_oracle.so).  TEST INFRASTRUCTURE: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_build", "libsq_oracle.so")


class Seq(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint), ("litLength", ctypes.c_uint),
                ("matchLength", ctypes.c_uint), ("rep", ctypes.c_uint)]


class SqoFile(ctypes.Structure):
    _fields_ = [("rel_path", ctypes.c_char_p), ("data", ctypes.c_void_p), ("size", ctypes.c_uint64)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.sqo_compress_chunk.restype = ctypes.c_size_t
        L.sqo_compress_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.sqo_compress_bound.restype = ctypes.c_size_t
        L.sqo_compress_bound.argtypes = [ctypes.c_size_t]
        L.sqo_hash_chunk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        L.sqo_decompress_frame.restype = ctypes.c_size_t
        L.sqo_decompress_frame.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        L.sqo_store_new.restype = ctypes.c_void_p
        L.sqo_store_free.argtypes = [ctypes.c_void_p]
        L.sqo_store_insert.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.sqo_store_len.restype = ctypes.c_uint64
        L.sqo_store_len.argtypes = [ctypes.c_void_p]
        L.sqo_pack_archive.restype = ctypes.c_size_t
        L.sqo_pack_archive.argtypes = [ctypes.POINTER(SqoFile), ctypes.c_uint32, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_void_p)]
        L.sqo_free.argtypes = [ctypes.c_void_p]

    def hash_chunk(self, data: bytes) -> bytes:
        out = ctypes.create_string_buffer(16)
        self.lib.sqo_hash_chunk(data, len(data), out)
        return out.raw

    def compress_bound(self, n: int) -> int:
        return int(self.lib.sqo_compress_bound(n))

    def compress_chunk(self, data: bytes) -> bytes:
        cap = self.compress_bound(len(data)) + 64
        dst = ctypes.create_string_buffer(cap)
        n = self.lib.sqo_compress_chunk(data, len(data), dst, cap)
        assert n > 0, "oracle compress failed"
        return dst.raw[:n]

    def decompress(self, frame: bytes, cap: int) -> bytes:
        dst = ctypes.create_string_buffer(max(cap, 1))
        n = self.lib.sqo_decompress_frame(frame, len(frame), dst, cap)
        assert n != (1 << 64) - 1, "oracle decoder rejected the frame"
        return dst.raw[:n]

    def pack_archive(self, files, timestamp: int = 0) -> bytes:
        """files: list of (rel_path:str, data:bytes) in walk order."""
        arr = (SqoFile * max(len(files), 1))()
        keep = []
        for i, (p, d) in enumerate(files):
            b = ctypes.create_string_buffer(d, max(len(d), 1))
            keep.append(b)
            arr[i].rel_path = p.encode()
            arr[i].data = ctypes.addressof(b)
            arr[i].size = len(d)
        out = ctypes.c_void_p()
        n = self.lib.sqo_pack_archive(arr, len(files), timestamp, ctypes.byref(out))
        assert out.value, "oracle pack failed"
        res = ctypes.string_at(out.value, n)
        self.lib.sqo_free(out)
        return res


class OracleStore:
    """ChunkStore restatement (chunk.rs:80-100) for parity tests."""

    def __init__(self, o: Oracle):
        self.o = o
        self.h = o.lib.sqo_store_new()

    def insert(self, chunk: bytes):
        cap = self.o.compress_bound(len(chunk)) + 64
        dst = ctypes.create_string_buffer(cap)
        hsh = ctypes.create_string_buffer(16)
        n = ctypes.c_size_t(0)
        r = self.o.lib.sqo_store_insert(self.h, chunk, len(chunk), hsh, dst, cap, ctypes.byref(n))
        assert r >= 0
        return hsh.raw, (dst.raw[:n.value] if r == 1 else None)

    def len(self):
        return int(self.o.lib.sqo_store_len(self.h))

    def __del__(self):
        try:
            self.o.lib.sqo_store_free(self.h)
        except Exception:  # noqa: BLE001
            pass


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _cached = Oracle(ctypes.CDLL(SO))
    return _cached
