"""world_size-2 `gloo` tests of the multi-GPU path (runs on CPU).

The scan itself shards with no data-path collective; what is exercised here is everything around it: row-range
shard assignment, the COUNT(*) all-reduce and the mask-segment all-gather.  Per-shard hit masks come from the CPU
oracle (test infrastructure) — on the GPU box the same collectives carry the masks produced by lc_scan_eval.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_table(n_batches=6, rows=700, seed=3):
    """Two columns over the same row ranges: an int column and a string column."""
    rng = np.random.default_rng(seed)
    ints = [rng.integers(0, 1000, size=rows if b < n_batches - 1 else rows // 3, dtype=np.int32) for b in range(n_batches)]
    words = ["google", "yandex", "mail", "maps", "goo", "gle"]
    strs = [["http://%s.%s/%d" % (words[rng.integers(6)], words[rng.integers(6)], rng.integers(50))
             for _ in range(len(ints[b]))] for b in range(n_batches)]
    return ints, strs


def _worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import liquid_oracle as lo
    import liquid_cache_amd as lc
    from liquid_cache_amd import sharding as sh
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ints, strs = _make_table()
    nb = len(ints)
    ids = [lc.ParquetArrayID.new(0, b // 3, col, b % 3) for b in range(nb) for col in (1, 13)]
    shards = sh.assign_row_ranges(ids, world)
    mine = shards[rank]
    # conjunction `c1 >= 500 AND URL LIKE '%google%'` evaluated per owned row range; both columns are local
    words, count = [], 0
    for b in range(nb):
        e_int, e_str = lc.ParquetArrayID.new(0, b // 3, 1, b % 3), lc.ParquetArrayID.new(0, b // 3, 13, b % 3)
        if int(e_int) not in mine:
            assert int(e_str) not in mine
            continue
        assert int(e_str) in mine
        li = lo.encode_primitive(lo.PHYS["int32"], ints[b])
        ls, st = lo.encode_byte_view(strs[b], fingerprints=True)
        m1 = lo.eval_predicate(li, lo.GE, 500).filter_mask()
        m2 = lo.eval_predicate(ls, lo.LIKE, b"%google%", m1, symtab=st).filter_mask()
        final = lo.and_then(m1, m2)
        count += int(final.sum())
        seg = np.zeros(((len(final) + 63) // 64) * 8, np.uint8)
        packed = np.packbits(final, bitorder="little")
        seg[: len(packed)] = packed
        words.append(seg.view(np.int64))
    local = torch.from_numpy(np.concatenate(words)) if words else torch.zeros(0, dtype=torch.int64)
    total = sh.all_reduce_count(torch.tensor([count], dtype=torch.int64))
    gathered = sh.all_gather_mask_segments(local)
    if rank == 0:
        np.save(out_path, np.concatenate([g.numpy() for g in gathered]))
        with open(out_path + ".count", "w") as f:
            f.write(str(int(total.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scan_matches_single_process(tmp_path, world):
    out = str(tmp_path / "mask.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ints, strs = _make_table()
    want_segments, want_count = [], 0
    for b in range(len(ints)):
        m = (ints[b] >= 500) & np.array(["google" in s for s in strs[b]])
        want_count += int(m.sum())
        seg = np.zeros(((len(m) + 63) // 64) * 8, np.uint8)
        p = np.packbits(m, bitorder="little")
        seg[: len(p)] = p
        want_segments.append(seg.view(np.int64))
    got = np.load(out)
    assert int(open(out + ".count").read()) == want_count
    # shards are contiguous row ranges in order, so rank-order concatenation == table order
    assert got.tolist() == np.concatenate(want_segments).tolist()


def _reducer_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    from liquid_cache_amd.sharding import PipelinedCountAllReduce
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    red = PipelinedCountAllReduce(lambda: torch.zeros((), dtype=torch.int64), world)
    seen = []
    for step in range(7):
        buf = red.acquire()
        buf.fill_(1000 * step + rank + 1)     # this rank's COUNT(*) of step `step`
        red.submit()
        if step >= 2:                          # the buffer of step-2 was waited for by acquire(): it holds the global sum
            pass
    red.drain()
    seen.append(int(red.last().item()))
    # the other buffer holds the reduced value of the step before the last one
    seen.append(int(red.buffers[(red.steps - 2) & 1].item()))
    if rank == 0:
        with open(out_path, "w") as f:
            f.write(",".join(map(str, seen)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_count_all_reduce(tmp_path, world):
    """bench.py's exchange step: asynchronous all-reduce of step i overlapping step i+1 (two alternating buffers)."""
    out = str(tmp_path / "red.txt")
    mp.spawn(_reducer_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    last, before = map(int, open(out).read().split(","))
    ranks = sum(r + 1 for r in range(world))
    assert last == 1000 * 6 * world + ranks and before == 1000 * 5 * world + ranks
