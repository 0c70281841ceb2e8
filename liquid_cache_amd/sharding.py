"""Row-range sharding of cache entries across the GPUs of one node, and the (only) collectives of the scan path.

Every 8192-row batch is independent (predicates never look across rows), so the unit of placement is the row
range (file, row group, batch) — the high 32 bits and the low 16 bits of the DataFusion layer's `ParquetArrayID`
(reference: src/datafusion/src/cache/id.rs:15-21).  ALL columns of a row range live on the same rank, so
multi-column conjunctions chain their selections on one device with no inter-GPU traffic; sharding by raw EntryID
hash would split the columns of one batch across devices and force a mask exchange per conjunct.

Exchange steps (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  * COUNT(*)-style consumers: one 8-byte all-reduce of the per-rank hit counts;
  * mask consumers: all-gather of the per-rank mask segments (disjoint row ranges, word-aligned per entry, so the
    gathered buffers concatenate without bit shifting).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def row_range_key(entry_id: int) -> Tuple[int, int, int]:
    """(file, row group, batch) of a ParquetArrayID-packed EntryID — the column id is NOT part of the key."""
    e = int(entry_id)
    return (e >> 48) & 0xFFFF, (e >> 32) & 0xFFFF, e & 0xFFFF


def assign_row_ranges(entry_ids: Sequence[int], world_size: int,
                      weights: Dict[int, int] | None = None) -> List[List[int]]:
    """Contiguous row-range shards, balanced by weight (staged bytes or rows; default 1 per entry).

    Returns one list of entry ids per rank; entries sharing a row range always land on the same rank and ranks own
    contiguous runs of row ranges in (file, row group, batch) order."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    groups: Dict[Tuple[int, int, int], List[int]] = {}
    for e in entry_ids:
        groups.setdefault(row_range_key(e), []).append(int(e))
    keys = sorted(groups)
    w = [sum((weights or {}).get(e, 1) for e in groups[k]) for k in keys]
    total = sum(w)
    shards: List[List[int]] = [[] for _ in range(world_size)]
    acc, rank = 0, 0
    for k, wk in zip(keys, w):
        # move to the next rank once this rank holds its share (never leave a later rank without candidates)
        while rank < world_size - 1 and acc >= total * (rank + 1) / world_size:
            rank += 1
        shards[rank].extend(sorted(groups[k]))
        acc += wk
    return shards


def contiguous_batch_range(total_batches: int, rank: int, world_size: int):
    """[first, last) batch of `rank` when a table of `total_batches` equally weighted batches is split into contiguous,
    batch-aligned row ranges (what assign_row_ranges gives for equal weights) — bench.py's strong-scaling split."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    return total_batches * rank // world_size, total_batches * (rank + 1) // world_size


def rank_of_entry(entry_id: int, shards: Sequence[Sequence[int]]) -> int:
    key = row_range_key(entry_id)
    for r, s in enumerate(shards):
        if any(row_range_key(e) == key for e in s):
            return r
    raise KeyError(entry_id)


def all_reduce_count(local_count, group=None):
    """Sum of per-rank hit counts (torch tensor, in place); the exchange step of COUNT(*) queries."""
    import torch.distributed as dist
    dist.all_reduce(local_count, op=dist.ReduceOp.SUM, group=group)
    return local_count


def all_gather_mask_segments(local_words, group=None):
    """All-gather per-rank mask segment buffers (1-D int64 tensors of possibly different length).

    Returns the list of per-rank tensors in rank order; concatenating them gives the hit mask of the whole table in
    shard order (per-entry segments are 64-bit aligned)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local_words.numel()], dtype=torch.int64, device=local_words.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(max(m, 1), dtype=local_words.dtype, device=local_words.device)
    padded[: local_words.numel()] = local_words
    out = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[: int(s.item())] for o, s in zip(out, sizes)]


class PipelinedCountAllReduce:
    """COUNT(*) exchange of a repeated scan, overlapped with the next scan.

    Two count buffers alternate: step i reduces into buffer i & 1 and starts its all-reduce asynchronously (RCCL runs it
    on its own stream, after the kernels already queued on the current stream); step i + 2 waits for that all-reduce
    (a stream-side wait) before it overwrites the buffer.  `drain()` waits for everything outstanding."""

    def __init__(self, make_buffer, world_size: int, group=None):
        self.buffers = [make_buffer(), make_buffer()]
        self.pending = [None, None]
        self.world = world_size
        self.group = group
        self.steps = 0

    def acquire(self):
        """Buffer for this step's local count (its previous all-reduce, if any, has completed in stream order)."""
        b = self.steps & 1
        self.steps += 1
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
        self._current = b
        return self.buffers[b]

    def submit(self):
        """Start the all-reduce of the buffer handed out by the last acquire()."""
        if self.world > 1:
            import torch.distributed as dist
            self.pending[self._current] = dist.all_reduce(self.buffers[self._current], op=dist.ReduceOp.SUM,
                                                          group=self.group, async_op=True)

    def drain(self):
        for b in range(2):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def last(self):
        """The (globally reduced, after drain()) count of the most recent step."""
        return self.buffers[(self.steps - 1) & 1]


class Communicator:
    """The exchange steps behind the C ABI (lc_comm_*): RCCL over xGMI for device contexts, a shared-memory file for
    host-only contexts (CPU tests).  Rank 0 creates the unique id (`Communicator.unique_id(cache)`), the host distributes
    its 128 bytes (here: whatever the caller has — a file, torch.distributed's store, MPI) and every rank calls
    `Communicator(cache, rank, world, id)`."""

    def __init__(self, cache, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _native as N
        self._lib, self._cache, self._C, self._N = cache._lib, cache, C, N
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        N.check(self._lib.lc_comm_init(cache.handle, rank, world, buf, C.byref(h)), cache.handle)
        self._h = h
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id(cache) -> bytes:
        import ctypes as C
        from . import _native as N
        buf = (C.c_uint8 * 128)()
        N.check(cache._lib.lc_comm_unique_id(cache.handle, buf), cache.handle)
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allreduce_count(self, total_ptr: int, stream: int = 0):
        """In-place sum over the ranks of the u64 at `total_ptr` (device pointer; host pointer for host-only contexts)."""
        self._N.check(self._lib.lc_comm_allreduce_count(self._h, total_ptr, stream or None), self._cache.handle)

    def allgather_mask(self, local_ptr: int, local_words: int, all_ptr: int, words_per_rank, stream: int = 0):
        C = self._C
        wpr = (C.c_uint64 * self.world)(*[int(w) for w in words_per_rank])
        self._N.check(self._lib.lc_comm_allgather_mask(self._h, local_ptr or None, int(local_words), all_ptr, wpr,
                                                        stream or None), self._cache.handle)


class PipelinedAbiCountAllReduce:
    """PipelinedCountAllReduce on the C ABI (lc_comm_allreduce_count, RCCL directly, no torch.distributed collective):
    the all-reduce of step i runs on a side HIP stream behind an event recorded after step i's scan, so it overlaps the
    scan of step i + 1; step i + 2 makes the scan stream wait for it before the buffer is overwritten."""

    def __init__(self, make_buffer, comm: "Communicator", torch):
        self.torch = torch
        self.comm = comm
        self.buffers = [make_buffer(), make_buffer()]
        self.side = torch.cuda.Stream()
        self.done = [None, None]
        self.steps = 0

    def acquire(self):
        b = self.steps & 1
        self.steps += 1
        if self.done[b] is not None:
            self.torch.cuda.current_stream().wait_event(self.done[b])
            self.done[b] = None
        self._current = b
        return self.buffers[b]

    def submit(self):
        if self.comm.world > 1:
            torch = self.torch
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.side.wait_event(ready)
            self.comm.allreduce_count(self.buffers[self._current].data_ptr(), self.side.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.done[self._current] = ev

    def drain(self):
        for b in range(2):
            if self.done[b] is not None:
                self.done[b].synchronize()
                self.done[b] = None

    def last(self):
        return self.buffers[(self.steps - 1) & 1]
