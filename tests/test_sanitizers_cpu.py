"""CPU suite: the HOST driver under sanitizers ("-fsanitize=address,undefined,thread in CPU-only
test builds"; the reference relies on Rust's ownership rules and `clippy -D warnings`,
/root/reference/.github/workflows/ci.yml:30-31).  `make -C squishrs_amd/host asan tsan` builds squishrs_main.cpp +
zstd_frame_decoder.hpp with AddressSanitizer + UBSan and with ThreadSanitizer; `list` and `unpack` (the reader and the
RFC 8878 frame decoder, /root/reference/src/archive/reader.rs:46-118,232-413 -- no GPU involved) then run over an
oracle-made archive and over a sweep of truncated / corrupted copies of it: every run must end with exit code 0 or with
the reference's "Error: ..." + exit code 1, and no sanitizer may report anything."""
import os
import subprocess

import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "squishrs_amd", "host")
# (the HIP runtime the driver links keeps allocations until exit: leak checking is about the driver, not about it)
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
           TSAN_OPTIONS="report_signal_unsafe=0")
REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: ThreadSanitizer", "SUMMARY: ")


@pytest.fixture(scope="module")
def builds():
    from squishrs_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    subprocess.check_call(["make", "-s", "-C", HOST, "asan", "tsan"])
    return os.path.join(HOST, "squishrs_asan"), os.path.join(HOST, "squishrs_tsan")


@pytest.fixture(scope="module")
def archive(tmp_path_factory, oracle):
    files = [
        ("a/text.txt", corpus.make("text", 1, 300000)),
        ("a/b/records.bin", corpus.make("records", 2, 70000)),
        ("a/b/tiny.txt", b"hello squish"),
        ("empty.bin", b""),
        ("big/mixed.bin", corpus.make("mixed", 3, 2 * 2097152 + 4321)),      # three chunks, a btopt tail
        ("big/dup.bin", corpus.make("mixed", 3, 2 * 2097152 + 4321)),        # whole-file duplicate
        ("r/random.bin", corpus.random_bytes(9, 9, 262144)),
        ("z/zeros.bin", bytes(400000)),
        ("z/runs.bin", corpus.make("runs", 5, 150000)),
    ]
    d = tmp_path_factory.mktemp("san")
    p = d / "good.squish"
    p.write_bytes(oracle.pack_archive(files, timestamp=1700000000))
    return p, dict(files)


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, env=ENV, timeout=600)
    for m in REPORTS:
        assert m not in r.stderr, f"{os.path.basename(exe)} {' '.join(args[:1])}: sanitizer report\n{r.stderr[-3000:]}"
    return r


def _tree(root):
    out = {}
    for d, _, fs in os.walk(root):
        for f in fs:
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = open(p, "rb").read()
    return out


def test_list_and_unpack_clean_under_asan_ubsan(builds, archive, tmp_path):
    asan, _ = builds
    arch, files = archive
    r = _run(asan, "list", str(arch), "--simple")
    assert r.returncode == 0 and f"number_of_files: {len(files)}" in r.stdout
    assert _run(asan, "list", str(arch)).returncode == 0
    for j in ("1", "8"):
        out = tmp_path / ("out" + j)
        r = _run(asan, "-j", j, "unpack", str(arch), "-o", str(out))
        assert r.returncode == 0, r.stderr
        assert _tree(out) == files


def test_unpack_with_8_threads_clean_under_tsan(builds, archive, tmp_path):
    _, tsan = builds
    arch, files = archive
    out = tmp_path / "out"
    r = _run(tsan, "-j", "8", "unpack", str(arch), "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert _tree(out) == files
    assert _run(tsan, "list", str(arch), "--simple").returncode == 0


def test_damaged_archives_fail_cleanly_under_asan_ubsan(builds, archive, tmp_path):
    """Truncations at 24 places and single-byte damage at 40 places (preamble, chunk-record headers, frame headers,
    Huffman / FSE table descriptions, bit streams, the manifest): the reader either still succeeds (damage in a
    payload that decodes to other bytes is not detectable: the format carries no checksum) or reports "Error" with
    exit code 1 (/root/reference/tests/cli_tests.rs:82-99) -- never a crash, never a sanitizer report."""
    asan, _ = builds
    arch, _files = archive
    good = arch.read_bytes()
    n = len(good)
    cases = []
    for k in range(24):
        cut = [0, 5, 11, 19, 26, 27, 40, 59, 60][k] if k < 9 else (n * (k - 8)) // 16 - (k % 3)
        cases.append(good[:max(0, min(n - 1, cut))])
    import numpy as np
    rng = np.random.default_rng(99)
    spots = [0, 6, 12, 20, 27, 43, 44, 51, 52, 59, 60, 61, 62, 63, 64, 70, 90, 200] + [int(x) for x in rng.integers(27, n, size=18)] + [n - 1, n - 17, n - 40, n - 100]
    for sp in spots:
        b = bytearray(good)
        b[sp] ^= 0xFF if sp % 2 else 0x5A
        cases.append(bytes(b))
    bad = tmp_path / "bad.squish"
    codes = {0: 0, 1: 0}
    for i, blob in enumerate(cases):
        bad.write_bytes(blob)
        for cmd in (("list", str(bad), "--simple"), ("unpack", str(bad), "-o", str(tmp_path / f"o{i}"))):
            r = _run(asan, *cmd)
            assert r.returncode in (0, 1), (i, cmd[0], r.returncode, r.stderr[-500:])
            if r.returncode == 1:
                assert "Error" in r.stderr, (i, r.stderr[-300:])
            codes[r.returncode] += 1
    assert codes[1] >= 30       # most of the damage is detected
