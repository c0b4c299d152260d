"""End-to-end sharded `pack`: N processes (one per GPU, torch.distributed) produce ONE .squish archive that is
byte-identical to the single-GPU / `-j 1` archive.

What it mirrors in the reference: ArchiveWriter::pack fans the files out, collects per-file results IN ORDER and a
single writer thread emits the chunk records (/root/reference/src/archive/writer.rs:153-195,
/root/reference/src/fsutil/writer.rs:17-39); the preamble / count patch / manifest are
/root/reference/src/archive/writer.rs:66-107,292-329 and src/util/header.rs:35-63,192-233.

Here the canonical chunk sequence (files in walk order, 2 MiB chunks, writer.rs:229-271) is cut into super-batches
of `batch_chunks`; inside a super-batch the chunks go to the ranks in block-cyclic stripes (parallel.stripe_assign:
stripe t of 16 chunks belongs to rank t % N -- with contiguous ranges rank 0 would hold first occurrences only and
compress everything it has while the others skip their duplicates; stripes spread that work).  Two super-batches are
in flight per rank (ShardedPacker.submit / wait): while batch k compresses, batch k+1 is read, copied to the device,
hashed and its hashes exchanged.  Per super-batch each rank: reads its own chunks -> hash -> ONE all-gather of the 16-byte
hashes (RCCL over xGMI with the "nccl" backend; "gloo" in the CPU tests) -> identical global first-occurrence dedup on
every rank -> compresses its own new chunks -> a second, tiny all-gather of the record sizes (8 bytes per chunk) whose
exclusive scan gives every record its place in the file -> every rank pwrite()s ITS OWN records there.  No frame crosses
a link and there is no single writer to funnel through; the order of the records is the global chunk-index order by
construction.  Rank 0 adds the preamble, patches the unique count and appends the manifest.

Launcher:  torchrun --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m squishrs_amd.sharded_pack <dir> -o out.squish
`stages` is the object that runs the device stages: the product passes a ChunkStore (HIP); the CPU tests inject an
oracle-backed stand-in to exercise the host logic under gloo."""
import os
import struct
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .parallel import ShardedPacker, stripe_assign

CHUNK_SIZE = 2048 * 1024          # src/util/chunk.rs:11
MAGIC = b"squish1.2.0"            # src/util/header.rs:10-14 (magic + version)


def walk_files(root: str) -> List[Tuple[str, str]]:
    """(relative path, absolute path) of every file under root in the order of the reference's walk
    (/root/reference/src/fsutil/directory.rs:39-73 and squishrs_main.cpp walk_dir: iterative DFS with stack.pop(),
    OS read_dir order, is_dir() follows symlinks) -- the same order the single-GPU driver packs in."""
    root = root.rstrip("/\\") or root
    stack, files = [root], []
    while stack:
        d = stack.pop()
        dirs = []
        with os.scandir(d) as it:
            for e in it:
                p = d + "/" + e.name
                if os.path.isdir(p):
                    dirs.append(p)
                else:
                    files.append(p)
        stack.extend(dirs)
    out = []
    for p in files:
        rel = p[len(root):].lstrip("/")
        out.append((rel, p))
    return out


def chunk_plan(sizes: Sequence[int]):
    """The canonical chunk sequence: for file f of `sizes[f]` bytes, ceil(size / 2 MiB) chunks (an empty file has
    none, writer.rs:241-245).  Returns arrays (file index, offset in file, length) per global chunk index."""
    fi, fo, ln = [], [], []
    for f, s in enumerate(sizes):
        o = 0
        while o < s:
            l_ = min(CHUNK_SIZE, s - o)
            fi.append(f)
            fo.append(o)
            ln.append(l_)
            o += l_
    return np.array(fi, dtype=np.int64), np.array(fo, dtype=np.int64), np.array(ln, dtype=np.uint32)


WRITE_RUN_BYTES = 64 << 20         # a run of records is flushed at this size (one pwritev call stays far below Linux's
                                   # 0x7ffff000-byte cap per write, and no second multi-GiB host copy is ever built)
IOV_MAX = 1024                     # iovecs per pwritev (POSIX IOV_MAX on Linux)


def pwritev_all(fd: int, bufs: Sequence, at: int, _pwritev=None) -> int:
    """Writes every byte of `bufs` (bytes / memoryviews) at file offset `at`: loops over short writes, raises on a write
    of 0 bytes.  Returns the number of bytes written.  (_pwritev: test hook, defaults to os.pwritev.)"""
    pw = _pwritev or os.pwritev
    views = [memoryview(b).cast("B") for b in bufs if len(b)]
    total = 0
    i = 0
    while i < len(views):
        batch = views[i:i + IOV_MAX]
        want = sum(len(v) for v in batch)
        n = pw(fd, batch, at)
        if n <= 0:
            raise OSError("pwritev wrote %d of %d bytes at offset %d" % (n, want, at))
        total += n
        at += n
        # drop what was written; a short write leaves the tail of one view
        while n and i < len(views):
            if n >= len(views[i]):
                n -= len(views[i])
                i += 1
            else:
                views[i] = views[i][n:]
                n = 0
    return total


class _Staging:
    """Host staging buffer of the reader: page-locked when the device is a GPU (one H2D DMA per super-batch), reused
    between super-batches (two in flight -> two buffers, handed out in turn)."""

    def __init__(self, pinned: bool):
        self.pinned = pinned
        self.bufs = [None, None]
        self.turn = 0

    def get(self, nbytes: int) -> torch.Tensor:
        k = self.turn
        self.turn ^= 1
        b = self.bufs[k]
        if b is None or b.numel() < nbytes:
            cap = max(nbytes, 1 << 20)
            try:
                b = torch.empty(cap, dtype=torch.uint8, pin_memory=self.pinned)
            except RuntimeError:        # page-locking refused (ulimit): ordinary memory still works, the copy is just slower
                b = torch.empty(cap, dtype=torch.uint8)
            self.bufs[k] = b
        return b[:nbytes]


def _read_chunks_idx(files, fi, fo, ln, idx, staging: Optional[_Staging] = None, threads: int = 8):
    """Bytes of the chunks with global indices idx (ascending) back to back at 256-byte aligned starts (the staging
    layout), read with up to `threads` concurrent pread()s straight into the staging buffer (the reference reads files on
    all Rayon workers, writer.rs:153-165).  Returns (uint8 tensor, offsets)."""
    count = len(idx)
    offs = np.zeros(count, dtype=np.uint64)
    pos = 0
    for i, g in enumerate(idx):
        offs[i] = pos
        pos += (int(ln[g]) + 255) & ~255
    buf = staging.get(pos + 256) if staging is not None else torch.empty(pos + 256, dtype=torch.uint8)
    view = memoryview(buf.numpy())
    fds: Dict[int, int] = {}
    for g in idx:
        f = int(fi[g])
        if f not in fds:
            fds[f] = os.open(files[f][1], os.O_RDONLY)

    def one(i):
        g = idx[i]
        at, want, fo_ = int(offs[i]), int(ln[g]), int(fo[g])
        fd = fds[int(fi[g])]
        got = 0
        while got < want:
            n = os.preadv(fd, [view[at + got:at + want]], fo_ + got)
            if n <= 0:
                raise OSError("file changed while packing: %s" % files[int(fi[g])][1])
            got += n
        pad = (-want) & 255
        if pad:
            view[at + want:at + want + pad] = bytes(pad)

    try:
        if count:
            if threads > 1 and count > 1:
                with ThreadPoolExecutor(max_workers=min(threads, count)) as ex:
                    list(ex.map(one, range(count)))
            else:
                for i in range(count):
                    one(i)
    finally:
        for fd in fds.values():
            os.close(fd)
    view[pos:pos + 256] = bytes(256)
    return buf, offs


def pack_sharded(files: List[Tuple[str, str]], out_path: Optional[str], stages, device="cpu",
                 group: Optional[dist.ProcessGroup] = None, timestamp: Optional[int] = None,
                 batch_chunks: int = 4096, contents: Optional[List[bytes]] = None, stripe: int = 16,
                 read_threads: int = 8, stats: Optional[dict] = None, _pwritev=None) -> Optional[int]:
    """Packs `files` into out_path.  Every rank writes its own chunk records into the one file (rank 0 also the
    preamble, the count and the manifest).  Returns the archive size on rank 0, None elsewhere.
    contents: optional in-memory file bodies (tests) instead of reading files[i][1].
    stats (optional dict): filled with this rank's busy seconds per host stage -- read, h2d, submit (hash + exchange of
    the hashes + enqueue of the rest), wait (the GPU stage as the host sees it), d2h, exchange2 (record sizes), write --
    and the counts `chunks` / `compressed` / `super_batches`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    packer = ShardedPacker(stages, group)
    comm_dev = "cpu" if (dist.is_initialized() and dist.get_backend(group) == "gloo") else device
    on_gpu = str(device).startswith("cuda")
    sizes = [len(c) for c in contents] if contents is not None else [os.path.getsize(p) for _, p in files]
    fi, fo, ln = chunk_plan(sizes)
    n_chunks = len(ln)
    all_hashes = np.zeros((n_chunks, 16), dtype=np.uint8)   # every rank ends up with every hash (the all-gather)
    ts = int(time.time()) if timestamp is None else int(timestamp)
    if rank == 0:
        with open(out_path, "wb") as f0:
            f0.write(MAGIC + struct.pack("<Q", ts) + struct.pack("<Q", 0))   # count patched at the end (offset 19)
    if world > 1:
        dist.barrier(group=group)                            # the file exists before anybody else opens it
    fd = os.open(out_path, os.O_WRONLY)
    file_pos = len(MAGIC) + 16                               # where the next super-batch's records start (same on every rank)
    unique = 0
    staging = _Staging(pinned=on_gpu)
    d2h = _Staging(pinned=on_gpu)                            # frames of a super-batch come back in ONE copy into this
    tm = {k: 0.0 for k in ("read", "h2d", "submit", "wait", "d2h", "exchange2", "write")}
    cnts = {"chunks": 0, "compressed": 0, "super_batches": 0}

    def read(idx):
        """this rank's chunks (global indices idx, ascending) back to back at 256-byte aligned starts"""
        if contents is None:
            return _read_chunks_idx(files, fi, fo, ln, idx, staging, read_threads)
        offs = np.zeros(len(idx), dtype=np.uint64)
        parts, pos = [], 0
        for i, g in enumerate(idx):
            offs[i] = pos
            b = contents[int(fi[g])][int(fo[g]):int(fo[g]) + int(ln[g])]
            pad = (-len(b)) & 255
            parts.append(b + bytes(pad))
            pos += len(b) + pad
        return torch.frombuffer(bytearray(b"".join(parts) + bytes(256)), dtype=torch.uint8), offs

    inflight = []    # per super-batch in flight: what finish() needs

    def submit(b0):
        nb = min(batch_chunks, n_chunks - b0)
        owner, counts, perm = stripe_assign(nb, world, stripe)
        mine = np.nonzero(owner == rank)[0]                  # this rank's chunks of the super-batch (global order kept)
        cnt = len(mine)
        t0 = time.perf_counter()
        buf, offs = read(b0 + mine)
        t1 = time.perf_counter()
        lens = ln[b0 + mine].copy()
        data = buf.to(device)                                # (blocking: the lane's stream, not torch's, runs the hash)
        hashes = torch.zeros(max(cnt, 1) * 16, dtype=torch.uint8, device=device)
        is_new = torch.zeros(max(cnt, 1), dtype=torch.uint8, device=device)
        cap = int(sum(((int(x) + (int(x) >> 8) + 64 + 32 + 15) // 16) * 16 for x in lens)) + 64    # >= sq_slots_bound
        comp = torch.empty(cap, dtype=torch.uint8, device=device)
        perm_t = torch.from_numpy(perm).to(device)
        idx_t = torch.from_numpy(mine.astype(np.int32)).to(device)
        t2 = time.perf_counter()
        packer.submit(data, offs, lens, hashes, is_new, comp, reset=False, perm=perm_t, my_idx=idx_t)
        t3 = time.perf_counter()
        tm["read"] += t1 - t0
        tm["h2d"] += t2 - t1
        tm["submit"] += t3 - t2
        inflight.append((b0, nb, mine, counts, perm, lens, is_new, comp, data, hashes))

    def finish():
        nonlocal file_pos, unique
        b0, nb, mine, counts, perm, lens, is_new, comp, _data, _hashes = inflight.pop(0)
        cnt = len(mine)
        t0 = time.perf_counter()
        co, cl = packer.wait()
        t1 = time.perf_counter()
        ordered = packer.last_all_hashes                     # this super-batch's hashes in global order
        all_hashes[b0:b0 + nb] = ordered[:nb * 16].cpu().numpy().reshape(nb, 16)
        flags = is_new[:cnt].cpu().numpy().astype(bool)
        # ONE copy brings the super-batch's frames back: the slot area up to the end of the last new frame (raw frames
        # fill their slots; on compressible data the slack travels too, where the GPU stage is the slow one by far)
        co = np.asarray(co, dtype=np.int64)
        cl = np.asarray(cl, dtype=np.int64)
        newj = np.nonzero(flags)[0]
        used = int((co[newj] + cl[newj]).max()) if len(newj) else 0
        host = None
        if used:
            host = d2h.get(used)
            host.copy_(comp[:used])                          # (blocking; page-locked destination on a GPU)
        frames = memoryview(host.numpy()) if host is not None else None
        t2 = time.perf_counter()
        # record sizes of ALL chunks of the super-batch in global order (the second, small exchange: 8 bytes per chunk):
        # an exclusive scan of them places every record; nobody's frames cross a link
        rec = np.where(flags, 32 + cl[:cnt], 0).astype(np.int64)
        n_max = int(counts.max()) if nb else 0
        gathered = packer.gather_ragged(torch.from_numpy(rec).to(comm_dev), max(n_max, 1)).cpu().numpy()
        rank_major = np.concatenate([gathered[r, :int(counts[r])] for r in range(world)]) if nb else np.zeros(0, dtype=np.int64)
        sizes_g = rank_major[perm]                           # per chunk in global order
        starts = file_pos + np.concatenate([[0], np.cumsum(sizes_g)[:-1]]) if nb else np.zeros(0, dtype=np.int64)
        t3 = time.perf_counter()
        # this rank's records: consecutive chunks of a stripe are consecutive in the file -> one pwritev per run, a run
        # being flushed when the next record is not adjacent or WRITE_RUN_BYTES are waiting (nothing is joined or copied:
        # the iovecs point into the host copy of the slots)
        run, run_at, run_bytes = [], None, 0
        for j in newj:
            g = mine[j]
            at = int(starts[g])
            if run_at is not None and (run_at + run_bytes != at or run_bytes >= WRITE_RUN_BYTES):
                pwritev_all(fd, run, run_at, _pwritev)
                run, run_at, run_bytes = [], None, 0
            if run_at is None:
                run_at = at
            # chunk record (fsutil/writer.rs:23-33): hash, original_size (= the 2 MiB buffer size whatever the
            # chunk length, writer.rs:255), compressed_size, frame
            run.append(all_hashes[b0 + g].tobytes() + struct.pack("<QQ", CHUNK_SIZE, int(cl[j])))
            run.append(frames[int(co[j]):int(co[j]) + int(cl[j])])
            run_bytes += 32 + int(cl[j])
        if run:
            pwritev_all(fd, run, run_at, _pwritev)
        t4 = time.perf_counter()
        file_pos += int(sizes_g.sum())
        unique += int(np.count_nonzero(sizes_g))
        tm["wait"] += t1 - t0
        tm["d2h"] += t2 - t1
        tm["exchange2"] += t3 - t2
        tm["write"] += t4 - t3
        cnts["chunks"] += cnt
        cnts["compressed"] += len(newj)
        cnts["super_batches"] += 1

    for b0 in range(0, n_chunks, batch_chunks):
        if len(inflight) == 2:
            finish()
        submit(b0)
    while inflight:
        finish()
    os.close(fd)
    if stats is not None:
        stats.update(tm)
        stats.update(cnts)
    if world > 1:
        dist.barrier(group=group)                            # every rank's records are in the file
    if rank != 0:
        return None
    # ---- manifest (writer.rs:292-329): file_count, then per file path_len, path, orig_size, chunk_count, hashes
    with open(out_path, "r+b") as fh:
        fh.seek(file_pos)
        fh.write(struct.pack("<I", len(files)))
        g = 0
        for f, (rel, _) in enumerate(files):
            p = rel.encode()
            nch = (sizes[f] + CHUNK_SIZE - 1) // CHUNK_SIZE
            fh.write(struct.pack("<I", len(p)) + p + struct.pack("<QI", sizes[f], nch))
            fh.write(all_hashes[g:g + nch].tobytes())
            g += nch
        size = fh.tell()
        fh.seek(19)
        fh.write(struct.pack("<Q", unique))     # header.rs:224-233 patch_u64
    return size


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m squishrs_amd.sharded_pack",
                                 description="multi-GPU squishrs pack: one process per GPU (launch with torchrun)")
    ap.add_argument("input")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("--batch-chunks", type=int, default=4096)
    ap.add_argument("-j", "--max-threads", type=int, default=8, help="reader threads per rank")
    a = ap.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "sharded_pack needs GPUs (the product has no CPU path)"
    # SQUISH_DEVICE / SQUISH_BACKEND: rehearse several ranks on ONE GPU (all on that device, gloo for the exchange)
    dev_ord = int(os.environ.get("SQUISH_DEVICE", local_rank))
    torch.cuda.set_device(dev_ord)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("SQUISH_BACKEND", "nccl"))
    from .chunk import ChunkStore
    files = walk_files(a.input)
    sizes = [os.path.getsize(p) for _, p in files]
    n_chunks = sum((s + CHUNK_SIZE - 1) // CHUNK_SIZE for s in sizes)
    out = a.output or (os.path.basename(os.path.normpath(a.input)) + ".squish")   # lib.rs:49-55 default name
    st = ChunkStore(device=dev_ord, dedup_capacity=n_chunks + 1024)
    t0 = time.time()
    stats = {}
    size = pack_sharded(files, out, st, device="cuda", batch_chunks=a.batch_chunks, read_threads=a.max_threads, stats=stats)
    if world > 1:
        dist.barrier()
    if os.environ.get("SQUISH_TIMING"):   # per-rank host stage times (busy seconds; stages of different super-batches overlap)
        print("[sharded_pack rank %d] " % rank + " ".join("%s=%.3f" % (k, v) if isinstance(v, float) else "%s=%d" % (k, v)
                                                          for k, v in stats.items()), file=sys.stderr)
    if rank == 0:
        total = sum(sizes)
        print(f"Packed {len(files)} files ({total} bytes) into {out} ({size} bytes) on {world} GPU(s) in {time.time() - t0:.2f} s")
    st.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
