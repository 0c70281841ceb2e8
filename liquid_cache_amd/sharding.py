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
