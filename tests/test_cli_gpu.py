"""GPU: the C++ `squishrs` host driver (pack through sq_pack_batch) against the oracle's .squish
restatement -- whole archive byte-identical except the 8 timestamp bytes --
plus the reference's CLI behaviours from tests/cli_tests.rs, including the pack -> unpack round trips."""
import json
import os
import subprocess

import pytest

import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "squishrs_amd", "host", "squishrs")


def _manifest_order(arch: bytes):
    count = int.from_bytes(arch[19:27], "little")
    pos = 27
    for _ in range(count):
        csz = int.from_bytes(arch[pos + 24:pos + 32], "little")
        pos += 32 + csz
    n = int.from_bytes(arch[pos:pos + 4], "little")
    pos += 4
    names = []
    for _ in range(n):
        pl = int.from_bytes(arch[pos:pos + 4], "little")
        names.append(arch[pos + 4:pos + 4 + pl].decode())
        pos += 4 + pl + 8
        cc = int.from_bytes(arch[pos:pos + 4], "little")
        pos += 4 + 16 * cc
    assert pos == len(arch)
    return names


def _run(*args, cwd=None):
    return subprocess.run([CLI, *args], cwd=cwd, capture_output=True, text=True, timeout=600)


def test_pack_tree_byte_identical_to_oracle(tmp_path, oracle):
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    tree = tmp_path / "in"
    files = {
        "a.txt": b"hello squish",
        "empty.bin": b"",
        "sub/text.txt": corpus.text_like(1, 300000),
        "sub/dup.txt": corpus.text_like(1, 300000),           # exact duplicate -> one record
        "sub/deep/random.bin": corpus.random_bytes(5, 1, 70000),
        "big/mixed.bin": corpus.mixed(4, 2 * 2097152 + 12345),  # 3 chunks, tail of 12,345 B (btopt)
        "big/zeros.bin": bytes(2097152 + 100),
    }
    for rel, data in files.items():
        p = tree / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    out = tmp_path / "out.squish"
    r = _run("pack", str(tree), "-o", str(out))
    assert r.returncode == 0, r.stderr
    assert "Packing complete!" in r.stdout and "Final archive size" in r.stdout
    arch = out.read_bytes()
    order = _manifest_order(arch)
    assert sorted(order) == sorted(files)
    exp = oracle.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)
    assert arch[:11] == b"squish1.2.0"
    assert arch[19:] == exp[19:], "archive differs from the oracle restatement"
    assert len(arch) == os.path.getsize(out)
    # list --simple
    r = _run("list", str(out), "--simple")
    assert r.returncode == 0 and f"number_of_files: {len(files)}" in r.stdout


def test_config0_1000_files_of_64k_random(tmp_path, oracle):
    """BASELINE configs[0]'s shape through the product's own driver: 1,000 files x 65,536 random bytes in one directory
    -> `squishrs pack` -> the archive equals the oracle's .squish restatement of the same files in manifest order
    (every byte but the timestamp: preamble, 1,000 chunk records of 32 + 65,546 bytes, manifest), `list --simple`
    reports the 1,000 files (/root/reference/tests/cli_tests.rs:55-80 checks that line), unpack restores them."""
    tree = tmp_path / "c0"
    tree.mkdir()
    files = {}
    for i in range(1000):
        files["f%06d.bin" % i] = corpus.random_bytes(0, i, 65536)
        (tree / ("f%06d.bin" % i)).write_bytes(files["f%06d.bin" % i])
    out = tmp_path / "c0.squish"
    r = _run("pack", str(tree), "-o", str(out))
    assert r.returncode == 0, r.stderr
    arch = out.read_bytes()
    order = _manifest_order(arch)
    assert sorted(order) == sorted(files)
    exp = oracle.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)
    assert int.from_bytes(arch[19:27], "little") == 1000
    assert len(arch) == 27 + 1000 * (32 + 65546) + 4 + sum(4 + len(rel) + 8 + 4 + 16 for rel in order)
    assert arch[:11] == b"squish1.2.0" and arch[19:] == exp[19:], "archive differs from the oracle restatement"
    r = _run("list", str(out), "--simple")
    assert r.returncode == 0 and "number_of_files: 1000" in r.stdout
    back = tmp_path / "back"
    r = _run("unpack", str(out), "-o", str(back))
    assert r.returncode == 0, r.stderr
    for rel in ("f000000.bin", "f000499.bin", "f000999.bin"):
        assert (back / rel).read_bytes() == files[rel]
    assert sum(len(fs) for _, _, fs in os.walk(back)) == 1000


def test_roundtrip_fixture_of_reference(tmp_path):
    # tests/roundtrip.rs: one file "file.txt" = "hello squish" -> the 124-byte archive of App. A
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "roundtrip_hello_squish.json")))
    d = tmp_path / "in"
    d.mkdir()
    (d / "file.txt").write_bytes(b"hello squish")
    out = tmp_path / "o.squish"
    assert _run("pack", str(d), "-o", str(out)).returncode == 0
    got = out.read_bytes()
    exp = bytes.fromhex(fx["archive"])
    assert len(got) == 124 and got[:11] == exp[:11] and got[19:] == exp[19:]


def test_empty_dir_and_errors(tmp_path):
    d = tmp_path / "empty"
    d.mkdir()
    out = tmp_path / "e.squish"
    assert _run("pack", str(d), "-o", str(out)).returncode == 0
    r = _run("list", str(out), "--simple")
    assert r.returncode == 0 and "number_of_files: 0" in r.stdout       # cli_tests.rs:55-80
    bad = tmp_path / "garbage.squish"
    bad.write_bytes(b"this is not an archive at all")
    r = _run("list", str(bad))
    assert r.returncode != 0 and "Error" in r.stderr                    # cli_tests.rs:82-99
    r = _run("list", str(tmp_path / "missing.squish"))
    assert r.returncode != 0 and "Error" in r.stderr                    # cli_tests.rs:101-120


def test_pack_unpack_roundtrip(tmp_path):
    # tests/cli_tests.rs:12-53 (two small files) and :122-159 (nested directories), plus every chunk kind
    # of the pack path: what the GPU packed, the host reader restores byte for byte
    tree = tmp_path / "input"
    files = {
        "file1.txt": b"hello",
        "file2.bin": bytes([0, 1, 2, 3, 4]),
        "dir1/dir2/nested.txt": b"nested file content",
        "dir1/text.txt": corpus.text_like(3, 400000),
        "dir1/again.txt": corpus.text_like(3, 400000),
        "r/random256k.bin": corpus.random_bytes(9, 1, 262144),
        "r/planted.bin": corpus.planted(5, 262144, 4),
        "big/mixed.bin": corpus.mixed(8, 2097152 + 54321),
        "small/records.bin": corpus.records(2, 12000),
        "none.bin": b"",
    }
    for rel, data in files.items():
        p = tree / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    arch = tmp_path / "archive.squish"
    r = _run("pack", str(tree), "--output", str(arch))
    assert r.returncode == 0, r.stderr
    out = tmp_path / "output"
    r = _run("unpack", str(arch), "--output", str(out))
    assert r.returncode == 0, r.stderr
    assert "Unpacking complete!" in r.stdout
    for rel, data in files.items():
        assert (out / rel).read_bytes() == data, rel
    n = sum(len(fs) for _, _, fs in os.walk(out))
    assert n == len(files)
    # the same archive with the records decoded on the GPU (sq_decompress_batch; the default takes the GPU only from 1 GiB on)
    # and, pinned, on the host: identical trees, and SQUISH_TIMING says where the frames were decoded
    for mode, word in (("gpu", "on the GPU"), ("host", "on the host")):
        o2 = tmp_path / ("output_" + mode)
        r = subprocess.run([CLI, "unpack", str(arch), "--output", str(o2)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SQUISH_UNPACK=mode, SQUISH_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert word in r.stderr, r.stderr
        for rel, data in files.items():
            assert (o2 / rel).read_bytes() == data, (mode, rel)
    # the driver's three stages (reader / GPU / writer threads) rotate three batches: with batches of a few
    # chunks the rotation wraps many times, and the archive must not change by a byte (timestamp aside)
    small = tmp_path / "small_batches.squish"
    r = subprocess.run([CLI, "pack", str(tree), "-o", str(small)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, SQUISH_BATCH_BYTES="700000"))
    assert r.returncode == 0, r.stderr
    a, b = arch.read_bytes(), small.read_bytes()
    assert len(a) == len(b) and a[:11] == b[:11] and a[19:] == b[19:]


def test_sharded_pack_launcher_one_gpu_equals_driver_archive(tmp_path, oracle):
    """squishrs_amd.sharded_pack (the multi-GPU launcher, here with ONE rank and the real HIP stages, small
    super-batches so the dedup set persists across them): its archive equals the C++ driver's and the oracle's byte
    for byte except the timestamp; -j 1 and -j 16 readers of the driver give the same bytes."""
    import sys
    tree = tmp_path / "in"
    files = {
        "a/one.txt": corpus.text_like(11, 123456),
        "a/two.bin": corpus.mixed(12, 2097152 + 777),
        "b/one_again.txt": corpus.text_like(11, 123456),
        "b/rnd.bin": corpus.random_bytes(5, 9, 262144),
        "c/empty": b"",
        "c/zeros.bin": bytes(300000),
        "c/two_again.bin": corpus.mixed(12, 2097152 + 777),
    }
    for rel, data in files.items():
        p = tree / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    drv = tmp_path / "driver.squish"
    r = _run("-j", "16", "pack", str(tree), "-o", str(drv))
    assert r.returncode == 0, r.stderr
    drv1 = tmp_path / "driver_j1.squish"
    r = _run("-j", "1", "pack", str(tree), "-o", str(drv1))
    assert r.returncode == 0, r.stderr
    a, a1 = drv.read_bytes(), drv1.read_bytes()
    assert a[19:] == a1[19:]
    shd = tmp_path / "sharded.squish"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "squishrs_amd.sharded_pack", str(tree), "-o", str(shd), "--batch-chunks", "3"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr
    b = shd.read_bytes()
    assert b[:11] == a[:11] and b[19:] == a[19:], "sharded launcher's archive differs from the driver's"
    order = _manifest_order(a)
    assert a[19:] == oracle.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)[19:]


def test_sharded_pack_three_ranks_on_one_gpu_rehearsal(tmp_path, oracle):
    """Three ranks of the sharded launcher on the ONE GPU of this box (every rank drives the real HIP stages on
    device 0; gloo carries the hash all-gather and the frames, since RCCL refuses two ranks per device): cross-rank
    duplicates, a rank with no chunk in some super-batches, the single ordered writer.  The archive must equal the
    oracle's.  (The RCCL transport itself is first exercised by the driver's multi-GPU run.)"""
    import socket
    import sys
    tree = tmp_path / "in"
    files = {f"d{i % 3}/f{i:02d}.bin": (corpus.text_like(20 + i % 4, 90000 + 1000 * (i % 4)) if i % 2 else corpus.random_bytes(6, i % 5, 262144))
             for i in range(14)}
    files["d0/big.bin"] = corpus.mixed(33, 2097152 + 4242)
    files["d2/big_copy.bin"] = files["d0/big.bin"]
    for rel, data in files.items():
        p = tree / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "three.squish"
    env = dict(os.environ, PYTHONPATH=ROOT, SQUISH_DEVICE="0", SQUISH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "squishrs_amd.sharded_pack", str(tree), "-o", str(out), "--batch-chunks", "5"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a = out.read_bytes()
    order = _manifest_order(a)
    assert sorted(order) == sorted(files)
    assert a[19:] == oracle.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)[19:]


def test_bench_config4_three_ranks_on_one_gpu_rehearsal():
    """bench.py --gpus 3 (BASELINE configs[4] shape: 30 % of all chunks exact copies of a chunk on another rank) with
    all three ranks on this box's one GPU and gloo for the hash all-gather: every rank asserts its is_new flags
    against the first-occurrence model and unique == 0.7 N n inside bench.py; here: it ran and says so."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SQUISH_DEVICE="0", SQUISH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # block-cyclic stripes of 16 chunks (the default), then contiguous ranges (--stripe 0)
    for extra, chunks, unique in ((), "1600", 3360), (("--stripe", "0"), "1500", 3150):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
                            "--chunks", chunks, *extra], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 3 and line["config"]["unique_chunks"] == unique and line["scaling"] == "weak"
        assert line["config"]["parity"]["frames_compared_with_oracle"] > 0 if "frames_compared_with_oracle" in line["config"]["parity"] else line["config"]["parity"]["frames_compared_with_oracle_bytes"] > 0
        assert ("block-cyclic" in line["config"]["workload"]) == (not extra)
        # every rank's own clock, host phase times and kernel times travel in the line (the first run on real links must be diagnosable)
        pr = line["per_rank"]["ranks"]
        assert len(pr) == 3 and all(r["ms_per_step"] > 0 and r["kernels_ms"]["stage1"] > 0 and r["host_ms_per_step"]["all_gather_of_hashes"] >= 0 for r in pr)
        assert max(r["ms_per_step"] for r in pr) <= line["ms_per_step"] * 1.0001


def test_pack_on_three_ranks_in_the_product_binary_equals_the_one_gpu_archive(tmp_path, oracle):
    """`squishrs pack --devices 0,0,0`: the N-GPU path of the C++ driver (pack_multi: one process per rank forked before any GPU
    call, hashes exchanged through shared host memory, sq_shard_submit_hash / sq_shard_submit_rest_idx, every rank writes its own
    records) rehearsed with three ranks on ONE GPU, with super-batches small enough that stripes, cross-rank duplicates and
    duplicates across super-batches all occur.  The archive must equal the one-GPU archive (and the oracle's) except the timestamp.
    Reference: the Rayon fan-out + shared DashMap + ordered writer of /root/reference/src/archive/writer.rs:153-195."""
    tree = tmp_path / "in"
    files = {}
    for i in range(40):
        kind = ["random", "text", "mixed", "records"][i % 4]
        files[f"d{i % 5}/f{i:02d}.bin"] = corpus.make(kind, 700 + (i % 23), 100000 + 37000 * (i % 7))     # i % 23: duplicates of whole files
    files["big/a.bin"] = corpus.mixed(9, 3 * 2097152 + 777)
    files["big/b.bin"] = corpus.mixed(9, 3 * 2097152 + 777)[:2 * 2097152] + corpus.random_bytes(3, 5, 2097152 + 5)  # shares two chunks with a.bin
    files["empty"] = b""
    for rel, data in files.items():
        p = tree / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)
    one = tmp_path / "one.squish"
    r = _run("pack", str(tree), "-o", str(one))
    assert r.returncode == 0, r.stderr
    for devs, super_chunks in (("0,0,0", "48"), ("0,0", "4096")):
        multi = tmp_path / f"multi_{devs.count(',') + 1}.squish"
        r = subprocess.run([CLI, "pack", "--devices", devs, str(tree), "-o", str(multi)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, SQUISH_SUPER_BATCH_CHUNKS=super_chunks))
        assert r.returncode == 0, r.stdout + r.stderr
        a, b = one.read_bytes(), multi.read_bytes()
        assert a[:11] == b[:11] and a[19:] == b[19:], f"--devices {devs}: archive differs from the one-GPU archive"
    order = _manifest_order(one.read_bytes())
    exp = oracle.pack_archive([(rel, files[rel]) for rel in order], timestamp=0)
    assert one.read_bytes()[19:] == exp[19:]
    r = _run("pack", "--gpus", "0", str(tree), "-o", str(tmp_path / "x.squish"))     # --gpus 0: no rank, the one-GPU path
    assert r.returncode == 0
