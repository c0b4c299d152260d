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
from typing import List, Optional, Sequence, Tuple

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


def _read_chunks_idx(files, fi, fo, ln, idx):
    """Bytes of the chunks with global indices idx (ascending) back to back at 256-byte aligned starts (the staging layout)."""
    count = len(idx)
    offs = np.zeros(count, dtype=np.uint64)
    pos = 0
    for i, g in enumerate(idx):
        offs[i] = pos
        pos += (int(ln[g]) + 255) & ~255
    buf = np.zeros(pos + 256, dtype=np.uint8)
    cur, fh = -1, None
    for i, g in enumerate(idx):
        if int(fi[g]) != cur:
            if fh:
                fh.close()
            cur = int(fi[g])
            fh = open(files[cur][1], "rb")
        fh.seek(int(fo[g]))
        b = fh.read(int(ln[g]))
        assert len(b) == int(ln[g]), "file changed while packing"
        buf[int(offs[i]):int(offs[i]) + len(b)] = np.frombuffer(b, dtype=np.uint8)
    if fh:
        fh.close()
    return buf, offs


def pack_sharded(files: List[Tuple[str, str]], out_path: Optional[str], stages, device="cpu",
                 group: Optional[dist.ProcessGroup] = None, timestamp: Optional[int] = None,
                 batch_chunks: int = 4096, contents: Optional[List[bytes]] = None, stripe: int = 16) -> Optional[int]:
    """Packs `files` into out_path.  Every rank writes its own chunk records into the one file (rank 0 also the
    preamble, the count and the manifest).  Returns the archive size on rank 0, None elsewhere.
    contents: optional in-memory file bodies (tests) instead of reading files[i][1]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    packer = ShardedPacker(stages, group)
    comm_dev = "cpu" if (dist.is_initialized() and dist.get_backend(group) == "gloo") else device
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

    def read(idx):
        """this rank's chunks (global indices idx, ascending) back to back at 256-byte aligned starts"""
        if contents is None:
            return _read_chunks_idx(files, fi, fo, ln, idx)
        offs = np.zeros(len(idx), dtype=np.uint64)
        parts, pos = [], 0
        for i, g in enumerate(idx):
            offs[i] = pos
            b = contents[int(fi[g])][int(fo[g]):int(fo[g]) + int(ln[g])]
            pad = (-len(b)) & 255
            parts.append(b + bytes(pad))
            pos += len(b) + pad
        return np.frombuffer(b"".join(parts) + bytes(256), dtype=np.uint8).copy(), offs

    inflight = []    # per super-batch in flight: what finish() needs

    def submit(b0):
        nb = min(batch_chunks, n_chunks - b0)
        owner, counts, perm = stripe_assign(nb, world, stripe)
        mine = np.nonzero(owner == rank)[0]                  # this rank's chunks of the super-batch (global order kept)
        cnt = len(mine)
        buf, offs = read(b0 + mine)
        lens = ln[b0 + mine].copy()
        data = torch.from_numpy(buf).to(device)
        hashes = torch.zeros(max(cnt, 1) * 16, dtype=torch.uint8, device=device)
        is_new = torch.zeros(max(cnt, 1), dtype=torch.uint8, device=device)
        cap = int(sum(((int(x) + (int(x) >> 8) + 64 + 32 + 15) // 16) * 16 for x in lens)) + 64    # >= sq_slots_bound
        comp = torch.zeros(cap, dtype=torch.uint8, device=device)
        perm_t = torch.from_numpy(perm).to(device)
        idx_t = torch.from_numpy(mine.astype(np.int32)).to(device)
        packer.submit(data, offs, lens, hashes, is_new, comp, reset=False, perm=perm_t, my_idx=idx_t)
        inflight.append((b0, nb, mine, counts, perm, lens, is_new, comp, data, hashes))

    def finish():
        nonlocal file_pos, unique
        b0, nb, mine, counts, perm, lens, is_new, comp, _data, _hashes = inflight.pop(0)
        cnt = len(mine)
        co, cl = packer.wait()
        ordered = packer.last_all_hashes                     # this super-batch's hashes in global order
        all_hashes[b0:b0 + nb] = ordered[:nb * 16].cpu().numpy().reshape(nb, 16)
        flags = is_new[:cnt].cpu().numpy().astype(bool)
        # record sizes of ALL chunks of the super-batch in global order (the second, small exchange: 8 bytes per chunk):
        # an exclusive scan of them places every record; nobody's frames cross a link
        rec = np.where(flags, 32 + cl.astype(np.int64), 0).astype(np.int64)
        n_max = int(counts.max()) if nb else 0
        gathered = packer.gather_ragged(torch.from_numpy(rec).to(comm_dev), max(n_max, 1)).cpu().numpy()
        rank_major = np.concatenate([gathered[r, :int(counts[r])] for r in range(world)]) if nb else np.zeros(0, dtype=np.int64)
        sizes_g = rank_major[perm]                           # per chunk in global order
        starts = file_pos + np.concatenate([[0], np.cumsum(sizes_g)[:-1]]) if nb else np.zeros(0, dtype=np.int64)
        # this rank's records: consecutive chunks of a stripe are consecutive in the file -> one pwrite per run
        run, run_at = [], None
        for j, g in enumerate(mine):
            if not flags[j]:
                continue
            if run_at is not None and run_at + sum(len(x) for x in run) != int(starts[g]):
                os.pwrite(fd, b"".join(run), run_at)
                run, run_at = [], None
            if run_at is None:
                run_at = int(starts[g])
            # chunk record (fsutil/writer.rs:23-33): hash, original_size (= the 2 MiB buffer size whatever the
            # chunk length, writer.rs:255), compressed_size, frame
            run.append(all_hashes[b0 + g].tobytes() + struct.pack("<QQ", CHUNK_SIZE, int(cl[j])))
            run.append(comp[int(co[j]):int(co[j]) + int(cl[j])].cpu().numpy().tobytes())
        if run:
            os.pwrite(fd, b"".join(run), run_at)
        file_pos += int(sizes_g.sum())
        unique += int(np.count_nonzero(sizes_g))

    for b0 in range(0, n_chunks, batch_chunks):
        if len(inflight) == 2:
            finish()
        submit(b0)
    while inflight:
        finish()
    os.close(fd)
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
    size = pack_sharded(files, out, st, device="cuda", batch_chunks=a.batch_chunks)
    if world > 1:
        dist.barrier()
    if rank == 0:
        total = sum(sizes)
        print(f"Packed {len(files)} files ({total} bytes) into {out} ({size} bytes) on {world} GPU(s) in {time.time() - t0:.2f} s")
    st.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
