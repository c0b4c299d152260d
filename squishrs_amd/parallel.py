"""Multi-GPU pack path: one process per GPU, chunks sharded in contiguous
global index ranges, ONE collective -- an all-gather of the per-rank 16-byte hashes (RCCL over xGMI
when the backend is "nccl"; "gloo" in the CPU tests) -- after which every rank resolves "first
occurrence in global index order" identically and compresses only its own new chunks.

The reference has no distributed code; the dedup rule reproduced is ChunkStore::insert's
exactly-once-per-hash (src/util/chunk.rs:80-100) in the canonical `-j 1` order (App. E).

`stages` is the object that runs the three device stages; the product passes a ChunkStore (HIP).
Tests inject a CPU stand-in to exercise the collective/host logic under gloo."""
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous split of [0, n_total) -- global index = first + local index."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def stripe_assign(n_total: int, world: int, stripe: int):
    """Block-cyclic assignment of the chunks of a super-batch to ranks: stripe t = chunks [t * stripe, (t + 1) * stripe) of
    the global order belongs to rank t % world.  With contiguous ranges the first rank holds first occurrences only and
    compresses all of its chunks while later ranks skip their duplicates of earlier ranks' chunks; stripes spread the
    first occurrences -- the compression work -- evenly.  Returns (owner[g], counts[r], perm[g]): perm[g] = position of
    chunk g in the rank-major concatenation an all-gather of the ranks' arrays produces."""
    g = np.arange(n_total, dtype=np.int64)
    owner = (g // max(1, stripe)) % world
    counts = np.bincount(owner, minlength=world).astype(np.int64)
    starts = np.zeros(world, dtype=np.int64)
    starts[1:] = np.cumsum(counts)[:-1]
    local = np.zeros(n_total, dtype=np.int64)
    for r in range(world):
        idx = np.nonzero(owner == r)[0]
        local[idx] = np.arange(len(idx))
    return owner, counts, starts[owner] + local


class ShardedPacker:
    def __init__(self, stages, group: Optional[dist.ProcessGroup] = None):
        self.stages = stages
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # Stream ordering.  The device stages run on the ctx's stream; the collective (and the torch.cat of its
        # result) run on torch's current stream.  With more than one rank the ctx is therefore moved onto torch's
        # current stream HERE, so hash -> all-gather -> dedup are ordered by the stream itself; a single rank
        # has no collective between the stages and keeps whatever stream the ctx has.
        if self.world > 1 and torch.cuda.is_available() and hasattr(stages, "set_stream"):
            stages.set_stream(torch.cuda.current_stream().cuda_stream)
        self._comm = None       # stream of the exchanges of the pipelined form (submit / wait)
        self._inflight = []     # per batch in flight: what must stay alive until it is collected
        # host-side seconds this rank spent per phase of submit() / wait() since reset_stats(): hash_wait = launching the
        # hash and waiting for it, exchange = the all-gather of the hashes (+ the permutation into global order), rest =
        # enqueueing the global dedup + compression, wait = blocking in wait() for a batch's frames; `batches` counts them
        self.stats = {"hash_wait": 0.0, "exchange": 0.0, "rest": 0.0, "wait": 0.0, "batches": 0}

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0 if k == "batches" else 0.0

    def gather_hashes(self, hashes: torch.Tensor, n_local: int, uniform: bool = False):
        """All-gather of the ranks' hash arrays (16 B x n_local, padded to the max count).
        Returns (all_hashes in global index order, n_total, my_first).  uniform=True: the caller vouches that every
        rank holds n_local chunks (bench.py), which saves the exchange of the counts and its host round trip."""
        if self.world == 1:
            return hashes[: n_local * 16], n_local, 0
        if hashes.is_cuda and dist.get_backend(self.group) == "gloo":
            # (rehearsal of the multi-rank path on one GPU: gloo moves host tensors; RCCL takes device tensors)
            torch.cuda.current_stream().synchronize()
            g, n_total, my_first = self.gather_hashes(hashes.cpu(), n_local, uniform)
            return g.to(hashes.device), n_total, my_first
        dev = hashes.device
        if uniform:
            recv = torch.empty(self.world * n_local * 16, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(recv, hashes[: n_local * 16].contiguous(), group=self.group)
            return recv, self.world * n_local, self.rank * n_local
        cnt = torch.tensor([n_local], dtype=torch.int64, device=dev)
        counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, cnt, group=self.group)
        counts_h = counts.cpu().numpy()
        n_max = int(counts_h.max())
        n_total = int(counts_h.sum())
        my_first = int(counts_h[: self.rank].sum())
        send = hashes[: n_max * 16]
        if n_local < n_max:
            send = torch.zeros(n_max * 16, dtype=torch.uint8, device=dev)
            send[: n_local * 16] = hashes[: n_local * 16]
        recv = torch.empty(self.world * n_max * 16, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=self.group)
        if (counts_h == n_max).all():
            return recv, n_total, my_first
        parts = [recv[r * n_max * 16: r * n_max * 16 + int(counts_h[r]) * 16] for r in range(self.world)]
        return torch.cat(parts), n_total, my_first

    def gather_ragged(self, t: torch.Tensor, n_max: int):
        """All-gather of a 1-D tensor of at most n_max entries per rank (padded): returns the (world, n_max) result."""
        if self.world == 1:
            return t.reshape(1, -1)
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            torch.cuda.current_stream().synchronize()
            return self.gather_ragged(t.cpu(), n_max).to(t.device)
        send = torch.zeros(n_max, dtype=t.dtype, device=t.device)
        send[: t.numel()] = t
        recv = torch.empty(self.world * n_max, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.view(self.world, n_max)

    def pack(self, data, offs: np.ndarray, lens: np.ndarray, hashes, is_new, comp, layout: str = "packed",
             uniform: bool = False):
        """hash -> all-gather -> global dedup -> compress own new chunks.
        layout "packed": frames back to back in `comp`, returns comp_off[n + 1];
        layout "slots" : frames left in their slots of `comp` (no packing pass), returns
                         (comp_off[n], comp_len[n])."""
        n = len(lens)
        self.stages.hash_device(data, offs, lens, hashes)
        all_hashes, n_total, my_first = self.gather_hashes(hashes, n, uniform)
        self.stages.dedup_device(all_hashes, n_total, my_first, n, is_new)
        if layout == "slots":
            return self.stages.compress_device_slots(data, offs, lens, is_new, comp)
        return self.stages.compress_device(data, offs, lens, is_new, comp)

    # ---- pipelined form: two batches in flight, the exchange of batch k+1 under the compression of batch k ----------
    def submit(self, data, offs: np.ndarray, lens: np.ndarray, hashes, is_new, comp, reset: bool = False,
               uniform: bool = False, perm=None, my_idx=None):
        """Enqueue one batch (slots layout): hash -> [host waits for the hashes] -> all-gather -> global dedup +
        compression of this rank's new chunks enqueued behind it.  Returns at once; collect with wait().  At most two
        batches in flight, each with its own hashes / is_new / comp buffers.  The collective runs on a stream of its
        own: torch's current stream may be one of the context's lanes (see __init__) and would put the exchange
        behind the other batch's kernels."""
        import time
        n = len(lens)
        st = self.stages
        t0 = time.perf_counter()
        st.shard_submit_hash(data, offs, lens, hashes)
        st.shard_wait_hash()
        t1 = time.perf_counter()

        def exchange():
            g, n_total, my_first = self.gather_hashes(hashes, n, uniform)
            if perm is not None:
                # any assignment of chunks to ranks (stripe_assign): the rank-major gather result is put into GLOBAL
                # chunk-index order (perm: tensor on the hashes' device) -- on the exchange's stream, like the gather
                g = g.view(-1, 16)[perm].reshape(-1).contiguous()
            return g, n_total, my_first

        if self.world > 1 and torch.cuda.is_available() and hashes.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            with torch.cuda.stream(self._comm):
                all_hashes, n_total, my_first = exchange()
            self._comm.synchronize()
        else:
            all_hashes, n_total, my_first = exchange()
            if hashes.is_cuda:
                torch.cuda.current_stream().synchronize()   # (the permutation ran on torch's stream, the rest runs on the lane's)
        t2 = time.perf_counter()
        if perm is not None:   # ... and this rank names its chunks by their global index (my_idx)
            st.shard_submit_rest_idx(all_hashes, n_total, my_idx, is_new, comp, reset)
        else:
            st.shard_submit_rest(all_hashes, n_total, my_first, is_new, comp, reset)
        t3 = time.perf_counter()
        self.stats["hash_wait"] += t1 - t0
        self.stats["exchange"] += t2 - t1
        self.stats["rest"] += t3 - t2
        self.stats["batches"] += 1
        self._inflight.append((all_hashes, n, my_idx))

    def wait(self):
        """(slot offsets[n], frame lengths[n]) of the oldest batch in flight."""
        import time
        self.last_all_hashes, n, _idx = self._inflight.pop(0)   # (the batch's hashes in global order: the caller's manifest)
        t0 = time.perf_counter()
        r = self.stages.pack_wait(n)
        self.stats["wait"] += time.perf_counter() - t0
        return r
