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


def test_strong_scaling_split_covers_the_table_once():
    """bench.py --rows-total: contiguous batch-aligned ranges per rank, together exactly the table; the same split
    assign_row_ranges produces for equally weighted batches."""
    import sys
    sys.path.insert(0, ROOT)
    import liquid_cache_amd as lc
    from liquid_cache_amd import sharding as sh
    for total in (1, 7, 8, 12_207, 73_247):
        for world in (1, 2, 3, 8):
            ranges = [sh.contiguous_batch_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    ids = [lc.ParquetArrayID.new(0, b // 54, 10, b % 54) for b in range(1000)]
    shards = sh.assign_row_ranges(ids, 8)
    assert [len(s) for s in shards] == [b - a for a, b in (sh.contiguous_batch_range(1000, r, 8) for r in range(8))]


def _q6_worker(rank, world, port, out_path):
    """Config 4 sharded: lineitem-shaped columns l_shipdate / l_discount / l_quantity, row-range shards, the five Q6
    conjuncts chained per owned batch (oracle masks stand in for lc_scan_eval), COUNT(*) all-reduced and the final
    masks all-gathered."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import liquid_oracle as lo
    import liquid_cache_amd as lc
    from liquid_cache_amd import sharding as sh
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ship, disc, qty = _make_lineitem()
    nb = len(ship)
    ids = [lc.ParquetArrayID.new(0, b // 4, col, b % 4) for b in range(nb) for col in (10, 6, 4)]
    mine = set(sh.assign_row_ranges(ids, world)[rank])
    count, words = 0, []
    for b in range(nb):
        owned = [int(lc.ParquetArrayID.new(0, b // 4, col, b % 4)) in mine for col in (10, 6, 4)]
        assert all(owned) or not any(owned)       # the three columns of a row range live on one rank
        if not owned[0]:
            continue
        l_ship = lo.encode_primitive(lo.PHYS["date32"], ship[b])
        l_disc = lo.encode_decimal([int(x) for x in disc[b]], precision=15, scale=2)
        l_qty = lo.encode_decimal([int(x) for x in qty[b]], precision=15, scale=2)
        sel = np.ones(len(ship[b]), bool)
        for liquid, op, lit in ((l_ship, lo.GE, 8766), (l_ship, lo.LT, 9131), (l_disc, lo.GE, 5), (l_disc, lo.LE, 7),
                                (l_qty, lo.LT, 2400)):
            r = lo.eval_predicate(liquid, op, lit, sel)          # BooleanArray of popcount(sel) rows ...
            sel = lo.and_then(sel, r.filter_mask())              # ... re-expanded: boolean_buffer_and_then
        count += int(sel.sum())
        seg = np.zeros(((len(sel) + 63) // 64) * 8, np.uint8)
        packed = np.packbits(sel, bitorder="little")
        seg[: len(packed)] = packed
        words.append(seg.view(np.int64))
    total = sh.all_reduce_count(torch.tensor([count], dtype=torch.int64))
    gathered = sh.all_gather_mask_segments(torch.from_numpy(np.concatenate(words)) if words else torch.zeros(0, dtype=torch.int64))
    if rank == 0:
        np.save(out_path, np.concatenate([g.numpy() for g in gathered]))
        with open(out_path + ".count", "w") as f:
            f.write(str(int(total.item())))
    dist.destroy_process_group()


def _make_lineitem(n_batches=9, rows=900, seed=11):
    rng = np.random.default_rng(seed)
    lens = [rows if b < n_batches - 1 else rows // 2 for b in range(n_batches)]
    ship = [rng.integers(8036, 10561, size=n, dtype=np.int32) for n in lens]       # 1992-01-02 .. 1998-12-01
    disc = [rng.integers(0, 11, size=n, dtype=np.int64) for n in lens]
    qty = [rng.integers(1, 51, size=n, dtype=np.int64) * 100 for n in lens]
    return ship, disc, qty


@pytest.mark.parametrize("world", [2, 3])
def test_tpch_q6_sharded_by_row_range(tmp_path, world):
    out = str(tmp_path / "q6.npy")
    mp.spawn(_q6_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ship, disc, qty = _make_lineitem()
    want_count, segs = 0, []
    for b in range(len(ship)):
        m = (ship[b] >= 8766) & (ship[b] < 9131) & (disc[b] >= 5) & (disc[b] <= 7) & (qty[b] < 2400)
        want_count += int(m.sum())
        seg = np.zeros(((len(m) + 63) // 64) * 8, np.uint8)
        p = np.packbits(m, bitorder="little")
        seg[: len(p)] = p
        segs.append(seg.view(np.int64))
    assert int(open(out + ".count").read()) == want_count > 0
    assert np.load(out).tolist() == np.concatenate(segs).tolist()


def test_bench_q6_shards_hold_the_same_bytes_for_every_world_size():
    """bench.py --workload tpch_q6 (config 4): the synthetic lineitem batches are keyed by the GLOBAL batch index and the
    ranks own contiguous batch ranges, so the per-batch COUNT(*) the ranks check against — and their sum, the number
    every world size must report — do not depend on the number of ranks."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from liquid_cache_amd import _native as N
    from liquid_cache_amd import sharding as sh
    L = N.load()
    total_rows, bs = 200_000, 8192
    total_batches = (total_rows + bs - 1) // bs
    bufs = [np.zeros(bs, np.int64) for _ in range(3)]

    def counts_of(world):
        out = {}
        for rank in range(world):
            b0, b1 = sh.contiguous_batch_range(total_batches, rank, world)
            rows = min(total_rows, b1 * bs) - b0 * bs
            for b in range((rows + bs - 1) // bs):
                n = min(bs, rows - b * bs)
                sh_, di, qt = bench.q6_synth_batch(L, 42, b + b0, n, bufs)
                assert (b + b0) not in out
                out[b + b0] = (n, bench.q6_expected_count(sh_, di, qt))
        return out

    one = counts_of(1)
    assert sum(n for n, _ in one.values()) == total_rows and sum(c for _, c in one.values()) > 0
    for world in (2, 3, 8):
        assert counts_of(world) == one


# ------------------------------------------------------------------------------------------------------------------
# the exchange steps behind the C ABI (lc_comm_*), shared-memory backend of host-only contexts: what a Rust host binds
# ------------------------------------------------------------------------------------------------------------------
def _abi_worker(rank, world, id_path, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import liquid_cache_amd as lc
    from liquid_cache_amd import sharding as sh
    cache = lc.LiquidCacheBuilder.new().with_host_only().build()
    if rank == 0:
        uid = sh.Communicator.unique_id(cache)
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)  # the host's own distribution channel: here a file
    else:
        import time
        for _ in range(2000):
            if os.path.exists(id_path):
                break
            time.sleep(0.005)
        uid = open(id_path, "rb").read()
    comm = sh.Communicator(cache, rank, world, uid)
    rng = np.random.default_rng(100 + rank)
    results = []
    for rnd in range(5):
        # COUNT(*): per-rank partial counts -> the global count on every rank
        part = np.array([int(rng.integers(0, 1 << 40)) + rank], np.uint64)
        mine = int(part[0])
        comm.allreduce_count(part.ctypes.data)
        # mask: per-rank segments of different lengths (incl. an empty one) -> the concatenation on every rank
        wpr = [((r * 7 + rnd * 3) % 5) * 11 for r in range(world)]
        local = rng.integers(0, 1 << 62, size=wpr[rank], dtype=np.int64).view(np.uint64)
        out = np.zeros(max(sum(wpr), 1), np.uint64)
        comm.allgather_mask(local.ctypes.data if local.size else 0, local.size, out.ctypes.data, wpr)
        results.append((mine, int(part[0]), local.tolist(), out[: sum(wpr)].tolist(), wpr))
    comm.close()
    cache.close()
    np.save(out_path % rank, np.array([repr(results)]))


@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_exchange_steps_over_shared_memory(product_lib, tmp_path, world):
    id_path = str(tmp_path / "comm_id")
    out = str(tmp_path / "res_%d.npy")
    mp.spawn(_abi_worker, args=(world, id_path, out), nprocs=world, join=True)
    res = [eval(str(np.load(out % r)[0])) for r in range(world)]  # noqa: S307 (our own repr)
    for rnd in range(5):
        total = sum(res[r][rnd][0] for r in range(world)) % (1 << 64)
        concat = []
        for r in range(world):
            concat += res[r][rnd][2]
        for r in range(world):
            assert res[r][rnd][1] == total, (rnd, r)
            assert res[r][rnd][3] == concat, (rnd, r)
            assert res[r][rnd][4] == res[0][rnd][4]


# ------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N (default for N > 1): STRONG scaling of the URL workload — one table split by contiguous row ranges —
# end to end on CPU: bench.py's own shard arithmetic and generator keys, the oracle in place of lc_scan_eval, the exchange
# steps through lc_comm_* (shared-memory backend)
# ------------------------------------------------------------------------------------------------------------------
_URL_TOTAL_ROWS, _URL_BS = 8192 * 5 + 1234, 8192


def _url_batch(seed, gb, rows):
    from liquid_cache_amd import _native as N
    offs = np.zeros(_URL_BS + 1, np.int32)
    data = np.zeros(_URL_BS * 512, np.uint8)
    n = N.load_bench().lc_synth_url_batch(seed, gb, rows, min(300, rows), 20000, offs.ctypes.data, data.ctypes.data, data.size)
    raw = bytes(data[:n])
    return [raw[offs[i]: offs[i + 1]] for i in range(rows)]


def _url_strong_worker(rank, world, id_path, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle import liquid_oracle as lo
    import liquid_cache_amd as lc
    from liquid_cache_amd import sharding as sh
    # the shard bench.py main() gives this rank: --rows <table>, world > 1 -> rows_total = rows, batch-aligned contiguous range
    args = bench.parse_args(["--gpus", str(world), "--rows", str(_URL_TOTAL_ROWS)])
    assert args.scaling == "auto"
    total_batches = (_URL_TOTAL_ROWS + _URL_BS - 1) // _URL_BS
    b0, b1 = sh.contiguous_batch_range(total_batches, rank, world)
    args.batch0 = b0
    assert bench.url_seed(args, rank) == args.seed          # rank-free: the union of the shards is the one-GPU table
    rows_mine = max(0, min(_URL_TOTAL_ROWS, b1 * _URL_BS) - b0 * _URL_BS)
    count, segs = 0, []
    for gb in range(b0, b1):
        rows = min(_URL_BS, rows_mine - (gb - b0) * _URL_BS)
        vals = _url_batch(bench.url_seed(args, rank), gb, rows)
        liquid, st = lo.encode_byte_view(vals, fingerprints=True)
        m = lo.eval_predicate(liquid, lo.LIKE, b"%google%", None, symtab=st).filter_mask()
        count += int(m.sum())
        seg = np.zeros(((len(m) + 63) // 64) * 8, np.uint8)
        p = np.packbits(m, bitorder="little")
        seg[: len(p)] = p
        segs.append(seg.view(np.uint64))
    cache = lc.LiquidCacheBuilder.new().with_host_only().build()
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(sh.Communicator.unique_id(cache))
        os.replace(id_path + ".tmp", id_path)
    else:
        import time
        for _ in range(4000):
            if os.path.exists(id_path):
                break
            time.sleep(0.005)
    comm = sh.Communicator(cache, rank, world, open(id_path, "rb").read())
    part = np.array([count], np.uint64)
    comm.allreduce_count(part.ctypes.data)
    local = np.concatenate(segs) if segs else np.zeros(0, np.uint64)
    wpr = []
    for r in range(world):
        a, b = sh.contiguous_batch_range(total_batches, r, world)
        wpr.append(sum((min(_URL_BS, _URL_TOTAL_ROWS - gb * _URL_BS) + 63) // 64 for gb in range(a, b)))
    assert wpr[rank] == local.size
    out = np.zeros(max(sum(wpr), 1), np.uint64)
    comm.allgather_mask(local.ctypes.data if local.size else 0, local.size, out.ctypes.data, wpr)
    comm.close()
    cache.close()
    if rank == world - 1:
        np.save(out_path, out[: sum(wpr)])
        with open(out_path + ".count", "w") as f:
            f.write(str(int(part[0])))


@pytest.mark.parametrize("world", [2, 3])
def test_bench_url_strong_split_through_lc_comm(product_lib, tmp_path, world):
    out = str(tmp_path / "url_mask.npy")
    mp.spawn(_url_strong_worker, args=(world, str(tmp_path / "comm_id"), out), nprocs=world, join=True)
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seed = bench.parse_args([]).seed
    want_count, segs = 0, []
    for gb in range((_URL_TOTAL_ROWS + _URL_BS - 1) // _URL_BS):
        rows = min(_URL_BS, _URL_TOTAL_ROWS - gb * _URL_BS)
        m = np.array([b"google" in v for v in _url_batch(seed, gb, rows)])
        want_count += int(m.sum())
        seg = np.zeros(((len(m) + 63) // 64) * 8, np.uint8)
        p = np.packbits(m, bitorder="little")
        seg[: len(p)] = p
        segs.append(seg.view(np.uint64))
    assert int(open(out + ".count").read()) == want_count > 0
    assert np.load(out).tolist() == np.concatenate(segs).tolist()
