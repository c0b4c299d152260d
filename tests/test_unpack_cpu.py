"""CPU suite: `squishrs unpack` (the C++ host driver's reader + zstd frame decoder,
/root/reference/src/archive/reader.rs:232-413, lib.rs:86-106) -- needs no GPU.

* archives written by the oracle's .squish restatement (every strategy of the level-12 encoder, dedup,
  empty files, nested paths) unpack to the original tree;
* frames produced by the REAL libzstd 1.5.7 at other levels / with checksum / without content size /
  concatenated / skippable (tests/golden/zstd_decode_frames.json, tools/gen_decode_golden.py) decode to
  their content when wrapped into an archive record;
* every golden frame of the pack path with literal bytes (tests/golden/zstd12_frames.json "literals");
* the reference's own round-trip fixture (tests/roundtrip.rs) and CLI error behaviour (tests/cli_tests.rs)."""
import hashlib
import json
import os
import struct
import subprocess

import pytest

import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "squishrs_amd", "host", "squishrs")


@pytest.fixture(scope="module", autouse=True)
def _cli_built():
    from squishrs_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "squishrs_amd", "host")])


def _run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, cwd=cwd)


def _archive(records, files, ts=0):
    """records: [(hash16, orig_size, frame)], files: [(rel, size, [hash16])] -> .squish v1.2 bytes (App. A)"""
    b = bytearray(b"squish1.2.0") + struct.pack("<QQ", ts, len(records))
    for h, osz, fr in records:
        b += h + struct.pack("<QQ", osz, len(fr)) + fr
    b += struct.pack("<I", len(files))
    for rel, size, hashes in files:
        p = rel.encode()
        b += struct.pack("<I", len(p)) + p + struct.pack("<QI", size, len(hashes)) + b"".join(hashes)
    return bytes(b)


def _tree(root):
    out = {}
    for d, _, fs in os.walk(root):
        for f in fs:
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = open(p, "rb").read()
    return out


def test_unpack_oracle_archive(tmp_path, oracle):
    files = [
        ("a/text.txt", corpus.make("text", 1, 300000)),        # btlazy2 + lazy2 sized chunks
        ("a/b/records.bin", corpus.make("records", 2, 70000)),
        ("a/b/c/runs.bin", corpus.make("runs", 3, 50000)),
        ("skewed.bin", corpus.make("skewed", 4, 20000)),
        ("tiny.txt", b"hello squish"),
        ("small.bin", corpus.make("text", 5, 9000)),           # btopt strategy (<= 16 KiB)
        ("empty.dat", b""),                                    # zero chunks (writer.rs:241-245)
        ("dup/text_again.txt", corpus.make("text", 1, 300000)),  # exact duplicate: no second record
        ("big/mixed.bin", corpus.make("mixed", 6, (2 << 20) + 12345)),  # a 2 MiB chunk (row-hash lazy2) + a tail chunk
        ("rand.bin", corpus.make("random", 7, 262144)),        # raw blocks
        ("planted.bin", corpus.make("planted4", 8, 262144)),
    ]
    arch = oracle.pack_archive(files, timestamp=1700000000)
    a = tmp_path / "t.squish"
    a.write_bytes(arch)
    out = tmp_path / "out"
    r = _run("unpack", str(a), "--output", str(out))
    assert r.returncode == 0, r.stderr
    assert "Unpacking complete!" in r.stdout and f"{a} was unsquished into /{out}" in r.stdout  # lib.rs:100-105
    got = _tree(out)
    assert sorted(got) == sorted(p for p, _ in files)
    for p, d in files:
        assert got[p] == d, p
    # default output name: the archive path without ".squish" (lib.rs:88-93); -j accepted
    r = _run("-j", "2", "unpack", "t.squish", cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert _tree(tmp_path / "t") == got
    # list agrees with what was unpacked
    r = _run("list", str(a), "--simple")
    assert r.returncode == 0 and f"number_of_files: {len(files)}" in r.stdout


def _expected(case):
    if case["kind"] == "concat":
        return b"".join(corpus.make(k, s, n) for k, s, n in case["parts"])
    return corpus.make(case["kind"], case["seed"], case["nbytes"])


def test_unpack_decodes_real_libzstd_frames(tmp_path):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd_decode_frames.json")))
    assert len(fx["cases"]) >= 60
    records, files, want = [], [], {}
    for i, c in enumerate(fx["cases"]):
        data = _expected(c)
        assert hashlib.sha256(data).hexdigest() == c["sha256"], c["name"]
        h = hashlib.md5(c["name"].encode()).digest()  # any 16 distinct bytes serve as the record key
        records.append((h, 2 << 20, bytes.fromhex(c["frame"])))
        files.append((f"f/{i:03d}.bin", len(data), [h]))
        want[f"f/{i:03d}.bin"] = data
    a = tmp_path / "z.squish"
    a.write_bytes(_archive(records, files))
    out = tmp_path / "o"
    r = _run("unpack", str(a), "-o", str(out))
    assert r.returncode == 0, r.stderr
    got = _tree(out)
    for p, d in want.items():
        assert got[p] == d, p


def test_unpack_literal_goldens_of_the_pack_path(tmp_path, oracle):
    # frames whose bytes came from real libzstd 1.5.7 at level 12 (known answers of SURVEY App. B)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "zstd12_frames.json")))
    records, files, want = [], [], {}
    for i, (name, c) in enumerate(sorted(fx["literals"].items())):
        if not isinstance(c, dict) or "frame" not in c:
            continue
        fr = bytes.fromhex(c["frame"])
        data = bytes.fromhex(c["hex"])
        h = hashlib.md5(name.encode()).digest()
        records.append((h, 2 << 20, fr))
        files.append((f"{i}.bin", len(data), [h, h]))  # the same chunk twice in one file
        want[f"{i}.bin"] = data + data
    assert len(records) >= 8
    # ... and the generated cases small enough to carry their frame bytes (all three strategies)
    n = 0
    for j, g in enumerate(fx["generated"]):
        if not g.get("frame"):
            continue
        data = corpus.make(g["kind"], g["seed"], g["len"])
        h = hashlib.md5(b"g%d" % j).digest()
        records.append((h, 2 << 20, bytes.fromhex(g["frame"])))
        files.append((f"g/{j}.bin", len(data), [h]))
        want[f"g/{j}.bin"] = data
        n += 1
    assert n >= 80
    a = tmp_path / "l.squish"
    a.write_bytes(_archive(records, files))
    r = _run("unpack", str(a), "-o", str(tmp_path / "o"))
    assert r.returncode == 0, r.stderr
    assert _tree(tmp_path / "o") == want


def test_unpack_reference_roundtrip_fixture(tmp_path):
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "roundtrip_hello_squish.json")))
    a = tmp_path / "r.squish"
    a.write_bytes(bytes.fromhex(fx["archive"]))
    r = _run("unpack", str(a), "--output", str(tmp_path / "restored"))   # tests/roundtrip.rs
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "restored" / "file.txt").read_bytes() == b"hello squish"


def test_unpack_errors(tmp_path, oracle):
    r = _run("unpack", "nonexistent.squish", "--output", str(tmp_path / "o"), cwd=tmp_path)   # cli_tests.rs:101-120
    assert r.returncode != 0 and "Error" in r.stderr and "Specified file does not exist" in r.stderr
    bad = tmp_path / "garbage.squish"
    bad.write_bytes(b"this is not an archive at all")
    r = _run("unpack", str(bad), "-o", str(tmp_path / "o"))
    assert r.returncode != 0 and "Archive format error" in r.stderr
    # a file that names a chunk no record holds (reader.rs:396-400)
    h0, h1 = b"\x01" * 16, b"\x02" * 16
    fr = oracle.compress_chunk(b"some bytes " * 50)
    a = tmp_path / "m.squish"
    a.write_bytes(_archive([(h0, 2 << 20, fr)], [("x/y.txt", 550, [h0, h1])]))
    r = _run("unpack", str(a), "-o", str(tmp_path / "o2"))
    assert r.returncode != 0 and "Missing Chunk for File: `x/y.txt`" in r.stderr
    # damaged frame, truncated record, capacity too small (zstd::bulk::decompress(.., orig_size))
    dmg = bytearray(fr)
    dmg[len(dmg) // 2] ^= 0x5A
    dmg[6] ^= 0xFF
    for rec, what in (((h0, 2 << 20, bytes(dmg)), "damaged"), ((h0, 2 << 20, fr[:-3]), "truncated"), ((h0, 100, fr), "capacity")):
        a.write_bytes(_archive([rec], [("y.txt", 550, [h0])]))
        r = _run("unpack", str(a), "-o", str(tmp_path / ("o_" + what)))
        assert r.returncode != 0 and "Error reading from squish" in r.stderr, what
    a.write_bytes(_archive([(h0, 2 << 20, fr)], [("y.txt", 550, [h0])])[:-5])
    r = _run("unpack", str(a), "-o", str(tmp_path / "o3"))
    assert r.returncode != 0 and "Error" in r.stderr


def test_cli_clap_behaviour_and_list_tables(tmp_path, oracle):
    """f3: what clap does for /root/reference/src/cmd/mod.rs:11-58 -- --help on stdout with exit code 0, an unknown
    flag / a missing argument / an unknown subcommand on stderr with exit code 2, -j global -- and the two
    prettytable tables of the non-simple listing (cmd/mod.rs:97-165: labels of src/cmd/tests.rs:38-58)."""
    r = _run("--help")
    assert r.returncode == 0 and "Usage: squishrs [OPTIONS] <COMMAND>" in r.stdout and "pack" in r.stdout and r.stderr == ""
    r = _run("pack", "--help")
    assert r.returncode == 0 and "Usage: squishrs pack [OPTIONS] <INPUT>" in r.stdout
    r = _run("pack", "--bogus", "x")
    assert r.returncode == 2 and "error: unexpected argument '--bogus' found" in r.stderr and r.stdout == ""
    r = _run("list")
    assert r.returncode == 2 and "required arguments were not provided" in r.stderr and "<SQUISH>" in r.stderr
    r = _run("frobnicate", "x")
    assert r.returncode == 2 and "unrecognized subcommand 'frobnicate'" in r.stderr
    r = _run()
    assert r.returncode == 2 and "Usage: squishrs [OPTIONS] <COMMAND>" in r.stderr
    r = _run("list", "x.squish", "-o", "y")                 # -o belongs to pack / unpack only
    assert r.returncode == 2 and "unexpected argument '-o'" in r.stderr
    # listing of an archive of the oracle: 1,234 files in two top-level directories
    files = [(f"big/f{i:04d}.txt", b"x" * (i % 7)) for i in range(1200)] + [(f"small/g{i}.bin", corpus.text_like(i, 3000)) for i in range(34)]
    arch = tmp_path / "a.squish"
    arch.write_bytes(oracle.pack_archive(files, timestamp=1_700_000_000))
    r = _run("-j", "3", "list", str(arch))                  # the global -j in front of the subcommand
    assert r.returncode == 0, r.stderr
    out = r.stdout
    for label in ("Squash breakdown:", "Squash Summary", "Creation Date (UTC)", "Squish Version", "Compressed size", "Original size",
                  "Compression Ratio", "Number of files", "Number of chunks", "Top-level directory breakdown:", "Directory", "File Count"):
        assert label in out, label
    assert "| Number of files     | 1,234" in out and "14/11/2023 |" in out and "| 1.2.0" in out
    lines = out.splitlines()
    i = lines.index("Top-level directory breakdown:")
    assert lines[i + 1].startswith("+-") and lines[i + 2].startswith("| Directory") and lines[i + 3].startswith("+-")
    assert lines[i + 4].startswith("| big ") and "| 1,200 " in lines[i + 4]      # sorted by count, descending
    assert lines[i + 5].startswith("| small ") and lines[i + 6].startswith("+-")
    r = _run("list", str(arch), "--simple")
    assert r.returncode == 0 and "number_of_files: 1234" in r.stdout


def _archive_raw(records, files, ts=0):
    """as _archive, but the paths are bytes (any bytes)"""
    b = bytearray(b"squish1.2.0") + struct.pack("<QQ", ts, len(records))
    for h, osz, fr in records:
        b += h + struct.pack("<QQ", osz, len(fr)) + fr
    b += struct.pack("<I", len(files))
    for p, size, hashes in files:
        b += struct.pack("<I", len(p)) + p + struct.pack("<QI", size, len(hashes)) + b"".join(hashes)
    return bytes(b)


def test_illegal_utf8_paths_and_table_read_before_any_file(tmp_path, oracle):
    """reader.rs:178,350: a path that is not UTF-8 is AppError::IllegalUTF8 for `list` and `unpack`; rebuild_files reads the WHOLE
    file table before it creates a file (reader.rs:337-375), so a table damaged at its end leaves the output directory empty."""
    h0 = b"\x07" * 16
    data = b"squish " * 40
    fr = oracle.compress_chunk(data)
    for bad in (b"a\xffb.txt", b"\xc0\xaf.txt", b"\xed\xa0\x80.txt", b"\xf4\x90\x80\x80", b"trunc\xe2\x82"):
        a = tmp_path / "u.squish"
        a.write_bytes(_archive_raw([(h0, 2 << 20, fr)], [(b"ok.txt", len(data), [h0]), (bad, len(data), [h0])]))
        out = tmp_path / ("o_" + bad.hex())
        r = _run("unpack", str(a), "-o", str(out))
        assert r.returncode == 1 and "Illegal UTF8 detected" in r.stderr, (bad, r.stderr)
        assert not out.exists() or _tree(out) == {}, "a file was created before the table had been read"
        r = _run("list", str(a), "--simple")
        assert r.returncode == 1 and "Illegal UTF8 detected" in r.stderr, (bad, r.stderr)
    # well-formed multi-byte paths pass
    a = tmp_path / "v.squish"
    name = "déjà/文件-\U0001F600.txt"
    a.write_bytes(_archive([(h0, 2 << 20, fr)], [(name, len(data), [h0])]))
    r = _run("unpack", str(a), "-o", str(tmp_path / "ov"))
    assert r.returncode == 0, r.stderr
    assert _tree(tmp_path / "ov") == {name: data}
    # table cut short after its first (good) entry: nothing is written
    whole = _archive([(h0, 2 << 20, fr)], [("first.txt", len(data), [h0]), ("second.txt", len(data), [h0])])
    a.write_bytes(whole[:-7])
    r = _run("unpack", str(a), "-o", str(tmp_path / "ot"))
    assert r.returncode == 1 and "Error reading from squish" in r.stderr
    assert not (tmp_path / "ot").exists() or _tree(tmp_path / "ot") == {}


def test_unpack_rebuilds_many_files_in_parallel(tmp_path, oracle):
    """rebuild_files' par_iter (reader.rs:378-407): 600 files in 40 directories over 30 records, -j 8 and -j 1 give the same tree; a
    missing chunk in one of them is reported whatever the thread count."""
    import random
    rng = random.Random(5)
    recs, blobs = [], []
    for i in range(30):
        d = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 300, 5000]))) + (b"abc%d " % i) * rng.randrange(1, 400)
        h = oracle.hash_chunk(d)
        recs.append((h, 2 << 20, oracle.compress_chunk(d)))
        blobs.append((h, d))
    files, want = [], {}
    for i in range(600):
        picks = [rng.choice(blobs) for _ in range(rng.randrange(0, 4))]
        rel = "d%02d/sub/f%03d.bin" % (i % 40, i)
        files.append((rel, sum(len(d) for _, d in picks), [h for h, _ in picks]))
        want[rel] = b"".join(d for _, d in picks)
    a = tmp_path / "many.squish"
    a.write_bytes(_archive(recs, files))
    for j in ("8", "1"):
        out = tmp_path / ("o" + j)
        r = _run("-j", j, "unpack", str(a), "-o", str(out))
        assert r.returncode == 0, r.stderr
        assert _tree(out) == want
    files[333] = (files[333][0], 1, [b"\xee" * 16])
    a.write_bytes(_archive(recs, files))
    for j in ("8", "1"):
        r = _run("-j", j, "unpack", str(a), "-o", str(tmp_path / ("m" + j)))
        assert r.returncode == 1 and "Missing Chunk for File: `%s`" % files[333][0] in r.stderr, r.stderr
