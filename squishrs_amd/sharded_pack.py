"""End-to-end sharded `pack`: N processes (one per GPU, torch.distributed) produce ONE .squish archive that is
byte-identical to the single-GPU / `-j 1` archive.

What it mirrors in the reference: ArchiveWriter::pack fans the files out, collects per-file results IN ORDER and a
single writer thread emits the chunk records (/root/reference/src/archive/writer.rs:153-195,
/root/reference/src/fsutil/writer.rs:17-39); the preamble / count patch / manifest are
/root/reference/src/archive/writer.rs:66-107,292-329 and src/util/header.rs:35-63,192-233.

Here the canonical chunk sequence (files in walk order, 2 MiB chunks, writer.rs:229-271) is cut into super-batches
of `batch_chunks`; every super-batch is split into contiguous index ranges, one per rank.  Per super-batch each
rank: reads its own chunks -> hash -> ONE all-gather of hashes (RCCL over xGMI with the "nccl" backend; "gloo" in
the CPU tests) -> identical global first-occurrence dedup on every rank -> compresses its own new chunks -> sends
(flags, frame lengths, frames) to rank 0.  Rank 0 is the single writer: it emits the records of rank 0, 1, ... of
each super-batch in that order = global chunk-index order, patches the unique count and appends the manifest.

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

from .parallel import ShardedPacker, shard_range

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


def _read_chunks(files, fi, fo, ln, first, count):
    """Bytes of chunks [first, first+count) back to back at 256-byte aligned starts (the staging layout)."""
    offs = np.zeros(count, dtype=np.uint64)
    pos = 0
    for i in range(count):
        offs[i] = pos
        pos += (int(ln[first + i]) + 255) & ~255
    buf = np.zeros(pos + 256, dtype=np.uint8)
    cur, fh = -1, None
    for i in range(count):
        g = first + i
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
                 batch_chunks: int = 4096, contents: Optional[List[bytes]] = None) -> Optional[int]:
    """Packs `files` into out_path (written by rank 0 only).  Returns the archive size on rank 0, None elsewhere.
    contents: optional in-memory file bodies (tests) instead of reading files[i][1]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    packer = ShardedPacker(stages, group)
    # frames travel to the writer as device tensors with RCCL, as host tensors with gloo (CPU tests, rehearsals)
    comm_dev = "cpu" if (dist.is_initialized() and dist.get_backend(group) == "gloo") else device
    sizes = [len(c) for c in contents] if contents is not None else [os.path.getsize(p) for _, p in files]
    fi, fo, ln = chunk_plan(sizes)
    n_chunks = len(ln)
    all_hashes = np.zeros((n_chunks, 16), dtype=np.uint8)   # every rank ends up with every hash (the all-gather)
    fh = None
    unique = 0
    if rank == 0:
        fh = open(out_path, "wb")
        ts = int(time.time()) if timestamp is None else int(timestamp)
        fh.write(MAGIC + struct.pack("<Q", ts) + struct.pack("<Q", 0))   # count patched at the end (offset 19)

    def read(first, count):
        if contents is None:
            return _read_chunks(files, fi, fo, ln, first, count)
        offs = np.zeros(count, dtype=np.uint64)
        parts, pos = [], 0
        for i in range(count):
            g = first + i
            offs[i] = pos
            b = contents[int(fi[g])][int(fo[g]):int(fo[g]) + int(ln[g])]
            pad = (-len(b)) & 255
            parts.append(b + bytes(pad))
            pos += len(b) + pad
        return np.frombuffer(b"".join(parts) + bytes(256), dtype=np.uint8).copy(), offs

    for b0 in range(0, n_chunks, batch_chunks):
        nb = min(batch_chunks, n_chunks - b0)
        first, cnt = shard_range(nb, rank, world)
        buf, offs = read(b0 + first, cnt)
        lens = ln[b0 + first:b0 + first + cnt].copy()
        data = torch.from_numpy(buf).to(device)
        hashes = torch.zeros(max(cnt, 1) * 16, dtype=torch.uint8, device=device)
        is_new = torch.zeros(max(cnt, 1), dtype=torch.uint8, device=device)
        cap = int(sum(int(x) + (int(x) >> 8) + 64 + 48 for x in lens)) + 64
        comp = torch.zeros(cap, dtype=torch.uint8, device=device)
        # hash -> all-gather -> global dedup -> compress own new chunks (frames packed back to back, index order)
        packer_n = cnt
        stages.hash_device(data, offs, lens, hashes) if cnt else None
        gathered, n_total, my_first = packer.gather_hashes(hashes, packer_n)
        assert n_total == nb and my_first == first
        stages.dedup_device(gathered, n_total, my_first, cnt, is_new)
        comp_off = stages.compress_device(data, offs, lens, is_new, comp) if cnt else np.zeros(1, dtype=np.uint64)
        all_hashes[b0:b0 + nb] = gathered[:nb * 16].cpu().numpy().reshape(nb, 16)
        flags = is_new[:cnt].cpu().numpy().astype(np.uint8)
        clen = np.diff(comp_off.astype(np.int64)).astype(np.int64) if cnt else np.zeros(0, dtype=np.int64)
        blob = comp[:int(comp_off[-1])]
        # ---- deliver to the single writer in global index order
        if rank == 0:
            for r in range(world):
                r_first, r_cnt = shard_range(nb, r, world)
                if r == 0:
                    r_flags, r_clen, r_blob = flags, clen, blob.cpu().numpy()
                else:
                    meta = torch.zeros(2 * r_cnt + 1, dtype=torch.int64, device=comm_dev)
                    dist.recv(meta, src=r, group=group)
                    m = meta.cpu().numpy()
                    r_flags, r_clen = m[:r_cnt].astype(np.uint8), m[r_cnt:2 * r_cnt]
                    r_blob_t = torch.zeros(max(int(m[2 * r_cnt]), 1), dtype=torch.uint8, device=comm_dev)
                    dist.recv(r_blob_t, src=r, group=group)
                    r_blob = r_blob_t.cpu().numpy()
                pos = 0
                for j in range(r_cnt):
                    if r_flags[j]:
                        g = b0 + r_first + j
                        # chunk record (fsutil/writer.rs:23-33): hash, original_size (= the 2 MiB buffer size
                        # whatever the chunk length, writer.rs:255), compressed_size, frame
                        fh.write(all_hashes[g].tobytes() + struct.pack("<QQ", CHUNK_SIZE, int(r_clen[j])))
                        fh.write(r_blob[pos:pos + int(r_clen[j])].tobytes())
                        pos += int(r_clen[j])
                        unique += 1
        else:
            meta = torch.from_numpy(np.concatenate([flags.astype(np.int64), clen, [int(comp_off[-1])]]).astype(np.int64)).to(comm_dev)
            dist.send(meta, dst=0, group=group)
            dist.send((blob if blob.numel() else torch.zeros(1, dtype=torch.uint8, device=device)).to(comm_dev), dst=0, group=group)
    if rank != 0:
        return None
    # ---- manifest (writer.rs:292-329): file_count, then per file path_len, path, orig_size, chunk_count, hashes
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
    fh.close()
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
