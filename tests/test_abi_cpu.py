"""CPU suite: the C-ABI library loads and exports every symbol include/squish_gpu.h declares; no
compute calls without a GPU, and the product refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "squish_gpu.h")).read()
    return sorted(set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", h)))


def test_header_symbols_exported():
    from squishrs_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 16
    for s in decl:
        assert hasattr(L, s), f"{s} declared in squish_gpu.h but not exported"
    assert sorted(_lib.EXPORTS) == decl


def test_abi_basics_no_gpu_needed():
    from squishrs_amd import _lib
    L = _lib.load()
    assert L.sq_abi_version() == 1
    assert L.sq_strerror(-5) == b"Compression error"      # errors.rs:25-26
    assert L.sq_compress_bound(0) == 64 and L.sq_compress_bound(262144) == 262144 + 1024
    assert L.sq_compress_bound(2097152) == 2097152 + 8192


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from squishrs_amd import ChunkStore, SquishError, hash_chunk
    with pytest.raises(SquishError):
        ChunkStore()
    with pytest.raises(SquishError):
        hash_chunk(b"hello squish")


def test_product_does_not_touch_oracle():
    """The shipped package and library must not import, link or read anything under oracle/."""
    pkg = os.path.join(ROOT, "squishrs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "sqo_" not in txt and "sq_oracle" not in txt, f


def test_hand_issued_lds_reads_are_not_touched_before_their_wait():
    """Stage 1 issues the 18 filter reads of a round as inline asm with one hand-placed wait (sq_zstd_fast.h); the
    compiler must not copy / spill their destination registers in between.  tools/check_async_reads.py compiles the
    library to ISA (hipcc cross-compiles without a GPU) and scans every such group."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_reads.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_queue_loops_keep_the_shape_that_has_never_hung():
    """DESIGN.md section 9 (round 3's hang from an inlined per-chunk body): tools/check_queue_loops.py compiles the library
    to ISA and asserts that k_compress_batch reaches its per-chunk body through one s_swappc_b64 to compress_one, that the
    queue's atomic add of both persistent kernels sits in the lane-election region only with EXEC restored behind it, and
    that the only out-of-line callees inside stage 1's inlined body are stack-less leaves that do not touch the registers
    holding the kernel's spilled scalars."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_queue_loops.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_second_build_of_the_compressor_kernel_carries_the_symbol_the_library_asks_for():
    """sq_api.hip loads squish_s2_w3.co (the compressor kernel at 3 waves per SIMD) and looks k_compress_batch up by its mangled
    name (SQZ_K_COMPRESS_BATCH_SYMBOL): the name in the header must be the name in the build."""
    import re
    hdr = open(os.path.join(ROOT, "squishrs_amd", "csrc", "sq_zstd_kernel.h")).read()
    sym = re.search(r'#define SQZ_K_COMPRESS_BATCH_SYMBOL "(\w+)"', hdr).group(1)
    co = os.path.join(ROOT, "squishrs_amd", "squish_s2_w3.co")
    assert os.path.exists(co), "build() did not produce squish_s2_w3.co"
    assert sym.encode() in open(co, "rb").read()
    assert sym.encode() in open(os.path.join(ROOT, "squishrs_amd", "libsquish_gpu.so"), "rb").read()
