#!/usr/bin/env python3
"""bench.py — ClickBench "Q21" hot-cache filter scan on MI355X (liquid_cache_amd).

One "step" = one pass of the decode + predicate-pushdown hot path over the whole staged column chunk:
`URL LIKE '%google%'` (benchmark/clickbench/queries/q20.sql / q21.sql of the reference share this scan) over a
synthetic 100 M-row ClickBench-shaped URL column that is fully transcoded (dictionary + FSST + fingerprints) and
resident in HBM before the timed region starts.  Reported: filtered rows/s (value), algorithmic GB/s, the
roofline object of the dominant kernel (live HIP-event timing on the launch stream) and a CPU baseline (the C
oracle restating the reference's algorithm) on a bounded sample of the same data.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU, row-range sharding (every rank stages and scans its own 8192-row batches; weak
scaling: per-GPU rows fixed).  The only exchange step of the COUNT(*)-style query is the sum of per-rank hit
counts: one 8-byte all-reduce over RCCL per step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=99_997_497, help="rows per GPU (ClickBench hits = 99,997,497)")
    p.add_argument("--batch-size", type=int, default=8192)
    p.add_argument("--uniques", type=int, default=2200, help="distinct URLs per batch (nano_hits: ~2,150-2,250)")
    p.add_argument("--row-group-batches", type=int, default=54, help="batches sharing one FSST symbol table")
    p.add_argument("--needle", default="google")
    p.add_argument("--needle-ppm", type=int, default=159,
                   help="distinct URLs per million that contain the needle (ClickBench hits: 15,911 of 99,997,497 rows match)")
    p.add_argument("--workload", default="url_like", choices=["url_like", "int64_gt"])
    p.add_argument("--int-bits", type=int, default=62, help="int64_gt: FoR bit width of every batch (WatchID ~62)")
    p.add_argument("--cpu-batches", type=int, default=0, help="batches in the CPU-baseline sample (0 = auto)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-fingerprints", action="store_true",
                   help="url_like: stage the column without the SubstringSearch hint (no fingerprints, no signature index: "
                        "every dictionary value is walked)")
    p.add_argument("--no-q21", action="store_true", help="skip the secondary q21.sql pushdown pipeline measurement")
    p.add_argument("--seed", type=int, default=42)
    return p.parse_args(argv)


def stage_url_column(cache, lc, N, args, rank, n_batches, threads):
    """Generate + transcode + stage the URL column through the public API; returns entry ids."""
    L = N.load()
    import pyarrow as pa
    rows_total = args.rows
    bs = args.batch_size
    ids = [lc.ParquetArrayID.new(rank, b // args.row_group_batches, 13, b % args.row_group_batches)
           for b in range(n_batches)]

    def do_row_group(rg):
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 512, np.uint8)
        first = rg * args.row_group_batches
        for b in range(first, min(first + args.row_group_batches, n_batches)):
            rows = min(bs, rows_total - b * bs)
            n = L.lc_synth_url_batch(args.seed + rank * 1_000_003, b, rows, min(args.uniques, rows), args.needle_ppm,
                                     offs.ctypes.data, data.ctypes.data, data.size)
            arr = pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1]), pa.py_buffer(data[:n]))
            cache.insert(ids[b], arr, None if args.no_fingerprints else lc.CacheExpression.SUBSTRING_SEARCH)
        return rg

    n_rg = (n_batches + args.row_group_batches - 1) // args.row_group_batches
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(do_row_group, range(n_rg)))
    return ids


def stage_phrase_column(cache, lc, N, args, rank, n_batches, threads):
    """SearchPhrase-shaped column (no SubstringSearch hint: its predicate is `<> ''`), same row ranges as the URLs."""
    L = N.load()
    import pyarrow as pa
    bs = args.batch_size
    ids = [lc.ParquetArrayID.new(rank, b // args.row_group_batches, 39, b % args.row_group_batches)
           for b in range(n_batches)]

    def do_row_group(rg):
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 64, np.uint8)
        first = rg * args.row_group_batches
        for b in range(first, min(first + args.row_group_batches, n_batches)):
            rows = min(bs, args.rows - b * bs)
            n = L.lc_synth_phrase_batch(args.seed + rank * 1_000_003, b, rows, 600, 870, offs.ctypes.data,
                                        data.ctypes.data, data.size)
            arr = pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1]), pa.py_buffer(data[:max(n, 1)]))
            cache.insert(ids[b], arr)
        return rg

    n_rg = (n_batches + args.row_group_batches - 1) // args.row_group_batches
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(do_row_group, range(n_rg)))
    return ids


def q21_pipeline(cache, lc, N, args, rank, n_batches, threads, url_scan, like_expr, torch, stream):
    """The other reading of "Q21" (SURVEY §8d (ii)): q21.sql = SELECT "SearchPhrase", MIN("URL"), COUNT(*) ... WHERE
    "URL" LIKE '%google%' AND "SearchPhrase" <> '' GROUP BY ...: pushed-down part = `SearchPhrase <> ''` first (NotEq
    sorts before LIKE, row_filter.rs:499-515), URL LIKE on the narrowed selection, then get().with_selection() of both
    columns for the surviving rows; everything stays on the device."""
    import pyarrow as pa
    sp_ids = stage_phrase_column(cache, lc, N, args, rank, n_batches, threads)
    sp_scan = cache.scan(sp_ids)
    words = int(url_scan.mask_words)
    assert int(sp_scan.mask_words) == words
    ne_expr = lc.LiquidExpr.try_new("!=", b"", pa.string(), None)
    m1 = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    m2 = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(url_scan.entries, 1), dtype=torch.int32, device="cuda")
    c1 = torch.zeros(max(url_scan.entries, 1), dtype=torch.int32, device="cuda")
    cap = 1 << 20
    row_offs = torch.zeros(url_scan.entries + 1, dtype=torch.int64, device="cuda")
    row_offs2 = torch.zeros(url_scan.entries + 1, dtype=torch.int64, device="cuda")
    refs = [torch.zeros(cap, dtype=torch.int64, device="cuda") for _ in range(2)]
    voffs = [torch.zeros(cap + 1, dtype=torch.int64, device="cuda") for _ in range(2)]
    data = [torch.zeros(cap * 64, dtype=torch.uint8, device="cuda") for _ in range(2)]
    out = {}

    def run():
        sp_scan.eval(ne_expr, m1.data_ptr(), 0, c1.data_ptr(), stream)
        url_scan.eval(like_expr, m2.data_ptr(), m1.data_ptr(), counts.data_ptr(), stream)
        # both projections in stream order, no host round trip; sizes are read once after the timed loop
        url_scan.gather_bytes_async(row_offs.data_ptr(), refs[0].data_ptr(), voffs[0].data_ptr(), cap, data[0].data_ptr(),
                                    data[0].numel(), m2.data_ptr(), 0, stream)
        sp_scan.gather_bytes_async(row_offs2.data_ptr(), refs[1].data_ptr(), voffs[1].data_ptr(), cap, data[1].data_ptr(),
                                   data[1].numel(), m2.data_ptr(), 0, stream)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    iters = max(5, args.steps)
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    n_ne = int(c1.sum(dtype=torch.int64).item())
    k_out = int(row_offs[-1].item())
    assert k_out == int(row_offs2[-1].item()) == int(counts.sum(dtype=torch.int64).item()) and k_out <= cap
    out.update(rows_out=k_out, url_bytes=int(voffs[0][k_out].item()), phrase_bytes=int(voffs[1][k_out].item()))
    assert out["url_bytes"] <= data[0].numel() and out["phrase_bytes"] <= data[1].numel()
    res = {"query": "q21.sql pushdown: SearchPhrase <> '' -> URL LIKE '%%%s%%' -> get(URL), get(SearchPhrase)" % args.needle,
           "ms": ms, "rows_per_s": url_scan.rows / (ms * 1e-3), "rows_after_searchphrase": n_ne,
           "rows_out": out["rows_out"], "url_bytes_out": out["url_bytes"], "phrase_bytes_out": out["phrase_bytes"]}
    sp_scan.close()
    return res


def measure_get_with_selection(scan, lc, args, base, words, counts, torch, stream):
    """get-with-selection over the same Int64 column: the selected rows' decoded values compacted in row
    order.  Extra measurement next to the headline (not part of `value`); rows chosen by a second predicate."""
    import pyarrow as pa
    res = {}
    for sel_name, sel_frac in (("10pct", 0.1), ("0.1pct", 0.001)):
        sel_lit = base + int((1 << args.int_bits) * (1.0 - sel_frac))
        sel_mask = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
        scan.eval(lc.LiquidExpr.try_new(">", sel_lit, pa.int64()), sel_mask.data_ptr(), 0, counts.data_ptr(), stream)
        k_sel = int(counts.sum(dtype=torch.int64).item())
        vals = torch.zeros(max(k_sel, 1) + 8, dtype=torch.int64, device="cuda")
        offs = torch.zeros(scan.entries + 1, dtype=torch.int64, device="cuda")
        for _ in range(2):
            scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel_mask.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = max(5, args.steps)
        e0.record()
        for _ in range(iters):
            scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel_mask.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / iters
        # algorithmic bytes: n*W/8 packed + n/8 selection read, k*sizeof(T) written
        g_bytes = scan.rows * args.int_bits // 8 + scan.rows // 8 + k_sel * 8
        res[sel_name] = {"kernels": "k_sel_entry_counts + k_scan_{tile_sums,tiles,apply} + k_fixed_gather<u64>",
                         "selected_rows": k_sel, "ms": g_ms, "algorithmic_bytes": int(g_bytes),
                         "achieved_gbs": g_bytes / (g_ms * 1e-3) / 1e9,
                         "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "rows_per_s": scan.rows / (g_ms * 1e-3)}
    return res


def stage_int_column(cache, lc, N, args, rank, n_batches, threads):
    L = N.load()
    import pyarrow as pa
    bs = args.batch_size
    ids = [lc.ParquetArrayID.new(rank, b // args.row_group_batches, 0, b % args.row_group_batches)
           for b in range(n_batches)]

    def do_chunk(c):
        buf = np.zeros(bs, np.int64)
        for b in range(c, n_batches, threads):
            rows = min(bs, args.rows - b * bs)
            L.lc_synth_int64_batch(args.seed + rank * 1_000_003, b, rows, args.int_bits, 4_000_000_000_000_000_000 >> (64 - args.int_bits) if args.int_bits < 63 else 0, buf.ctypes.data)
            cache.insert(ids[b], pa.array(buf[:rows]))
        return c

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(do_chunk, range(threads)))
    return ids


def usable_cores() -> int:
    """Cores this process may really use: CPU affinity capped by the cgroup CPU quota (a container on a 256-thread host
    is often limited to a handful of cores, and os.cpu_count() does not see that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline_url(cache, lc, N, args, rank, n_sample, pattern, threads):
    """Oracle (CPU restatement of the reference algorithm) on the first n_sample batches, single thread."""
    from oracle import liquid_oracle as lo
    import pyarrow as pa
    L = N.load()
    bs = args.batch_size
    blobs, symtabs = [None] * n_sample, {}
    rows_total = sum(min(bs, args.rows - b * bs) for b in range(n_sample))

    def prep(c):  # regenerate + transcode the sample's batches (same bytes the GPU scanned), untimed
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 512, np.uint8)
        for b in range(c, n_sample, threads):
            rows = min(bs, args.rows - b * bs)
            n = L.lc_synth_url_batch(args.seed + rank * 1_000_003, b, rows, min(args.uniques, rows), args.needle_ppm,
                                     offs.ctypes.data, data.ctypes.data, data.size)
            arr = pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1]), pa.py_buffer(data[:n]))
            eid = lc.ParquetArrayID.new(rank, b // args.row_group_batches, 13, b % args.row_group_batches)
            path = lc.ParquetArrayID.column_access_path(eid)
            blobs[b] = (cache.transcode(arr, None if args.no_fingerprints else lc.CacheExpression.SUBSTRING_SEARCH, path), path)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(prep, range(threads)))
    for _, path in blobs:
        if path not in symtabs:
            symtabs[path] = lo.symtab_load(cache.symbol_table(path))
    # native loops over the batches (oracle/lo_bench.c): single thread = the reported baseline; one batch per OpenMP task
    # over all host cores = the figure SURVEY §8d asks to report beside it
    bl = [blob for blob, _ in blobs]
    sts = [symtabs[path] for _, path in blobs]
    lo.bench_eval_batches(bl[:8], sts[:8], lo.LIKE, pattern, 1)  # warm
    t0 = time.perf_counter()
    hits = lo.bench_eval_batches(bl, sts, lo.LIKE, pattern, 1)
    dt = time.perf_counter() - t0
    cores = usable_cores()
    t1 = time.perf_counter()
    hits_mt = lo.bench_eval_batches(bl, sts, lo.LIKE, pattern, cores)
    dt_mt = time.perf_counter() - t1
    cpu_baseline_url.all_cores = {"value": rows_total / dt_mt, "unit": "rows/s", "cores": cores, "kind": "port",
                                  "sample": "same batches, one batch per OpenMP task, %.2f s" % dt_mt,
                                  "hits_match": hits_mt == hits}
    return rows_total / dt, rows_total, hits, dt


def cpu_baseline_int(cache, lc, N, args, rank, n_sample, literal):
    from oracle import liquid_oracle as lo
    import pyarrow as pa
    L = N.load()
    bs = args.batch_size
    buf = np.zeros(bs, np.int64)
    blobs = []
    rows_total = 0
    base = 4_000_000_000_000_000_000 >> (64 - args.int_bits) if args.int_bits < 63 else 0
    for b in range(n_sample):
        rows = min(bs, args.rows - b * bs)
        L.lc_synth_int64_batch(args.seed + rank * 1_000_003, b, rows, args.int_bits, base, buf.ctypes.data)
        blobs.append(cache.transcode(pa.array(buf[:rows])))
        rows_total += rows
    t0 = time.perf_counter()
    hits = lo.bench_eval_batches(blobs, None, lo.GT, literal, 1)
    dt = time.perf_counter() - t0
    cores = usable_cores()
    t1 = time.perf_counter()
    hits_mt = lo.bench_eval_batches(blobs, None, lo.GT, literal, cores)
    dt_mt = time.perf_counter() - t1
    cpu_baseline_url.all_cores = {"value": rows_total / dt_mt, "unit": "rows/s", "cores": cores, "kind": "port",
                                  "sample": "same batches, one batch per OpenMP task, %.2f s" % dt_mt,
                                  "hits_match": hits_mt == hits}
    return rows_total / dt, rows_total, hits, dt


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel, from the rocprofv3 --pmc passes of the newest profiled round
    (profiles/<round>/hbm_traffic.json, produced by scripts/profile_round.sh; counters cannot be read from inside the
    process).  None when this workload has not been profiled."""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "*", "hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if workload in d:
            return int(d[workload]["traffic_bytes"]), os.path.relpath(f, root)
    return None, None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: liquid_cache_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N

    cache = lc.LiquidCacheBuilder.new().with_device(local_rank).with_batch_size(args.batch_size).build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, min(world, 8))))

    t_stage = time.perf_counter()
    if args.workload == "url_like":
        ids = stage_url_column(cache, lc, N, args, rank, n_batches, threads)
        pattern = ("%" + args.needle + "%").encode()
        import pyarrow as pa
        expr = lc.LiquidExpr.try_new("like", pattern, pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
        workload = "clickbench_q21_url_like_%s%s" % (args.needle, "_no_fingerprints" if args.no_fingerprints else "")
        dtype = "u8"
    else:
        ids = stage_int_column(cache, lc, N, args, rank, n_batches, threads)
        import pyarrow as pa
        base = 4_000_000_000_000_000_000 >> (64 - args.int_bits) if args.int_bits < 63 else 0
        literal = base + (1 << (args.int_bits - 1)) if args.int_bits < 64 else 0
        expr = lc.LiquidExpr.try_new(">", literal, pa.int64())
        workload = "clickbench_int64_gt_w%d" % args.int_bits
        dtype = "int64"
    t_stage = time.perf_counter() - t_stage

    scan = cache.scan(ids)
    words = int(scan.mask_words)
    mask = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
    # COUNT(*) partials: two buffers so that the all-reduce of step i (RCCL's own stream) overlaps the scan of step i+1;
    # int32 accumulators while the global count fits (one reduce kernel; an int64 sum of int32 counts costs torch an
    # extra cast kernel per step)
    from liquid_cache_amd.sharding import PipelinedCountAllReduce
    acc_dtype = torch.int32 if scan.rows * world < 2**31 else torch.int64
    reducer = PipelinedCountAllReduce(lambda: torch.zeros((), dtype=acc_dtype, device="cuda"), world)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        total = reducer.acquire()
        scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
        torch.sum(counts, dim=(0,), dtype=acc_dtype, out=total)  # COUNT(*) of this shard
        reducer.submit()  # the query's only exchange step: COUNT(*) partials -> global count (4-8 bytes)

    drain = reducer.drain

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if os.environ.get("LC_DUMP_COUNTS"):  # kernel-instrumentation aid (see LC_DEBUG_FLAGS in lc_kernels.hip)
        import numpy as _np
        _np.save(os.environ["LC_DUMP_COUNTS"], counts.cpu().numpy())
    ms_per_step = elapsed / args.steps * 1e3
    rows_all = scan.rows * world
    hits = int(reducer.last().item())

    # roofline of the dominant kernel: HIP events on the launch stream, same launches as the timed region
    alg_bytes = scan.algorithmic_bytes(expr, with_selection=False)
    kernel_ms = scan.eval_timed(expr, mask.data_ptr(), max(5, args.steps), 0, counts.data_ptr(), stream)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    gather = None
    if args.workload == "int64_gt" and rank == 0:
        try:  # secondary measurement: it must never cost the headline line
            gather = measure_get_with_selection(scan, lc, args, base, words, counts, torch, stream)
        except Exception as e:  # noqa: BLE001
            gather = {"error": "%s: %s" % (type(e).__name__, e)}

    q21 = None
    if args.workload == "url_like" and rank == 0 and world == 1 and not args.no_q21:
        try:  # secondary measurement: it must never cost the headline line
            q21 = q21_pipeline(cache, lc, N, args, rank, n_batches, threads, scan, expr, torch, stream)
        except Exception as e:  # noqa: BLE001
            q21 = {"error": "%s: %s" % (type(e).__name__, e)}

    out = None
    if rank == 0:
        traffic, traffic_src = measured_traffic(workload)
        out = {
            "metric": "filtered rows/s (+ GB/s scanned), ClickBench Q21 hot cache",
            "value": rows_all / elapsed * args.steps,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": int(scan.rows), "batch_rows": args.batch_size,
                       "batches_per_gpu": int(scan.entries), "distinct_per_batch": args.uniques,
                       "parallelism": "row-range shards x%d, 8-byte count all-reduce per step (overlapped with the next scan)" % world,
                       "predicate": ("URL LIKE '%%%s%%'" % args.needle) if args.workload == "url_like" else "col > literal",
                       "hits": hits, "stage_seconds": round(t_stage, 2)},
            "gb_per_s_scanned": alg_bytes * world / (elapsed / args.steps) / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # what the kernel really pulls through HBM (the signature index and the skipped row phase
                         # make it far less than the algorithmic bytes of the reference algorithm for LIKE)
                         "traffic_gbs": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                         "kernel": "k_str_pred" if args.workload == "url_like" else "k_fixed_pred<u64>",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        if gather is not None:
            out["get_with_selection"] = gather
        if q21 is not None:
            out["q21_pipeline"] = q21
        if world == 1 and not args.no_cpu_baseline:
            n_sample = args.cpu_batches or n_batches  # ~5 s (LIKE) / ~1 s (int) of single-thread CPU work at 100 M rows
            if args.workload == "url_like":
                v, rows_s, hits_s, dt = cpu_baseline_url(cache, lc, N, args, rank, n_sample, pattern, threads)
            else:
                v, rows_s, hits_s, dt = cpu_baseline_int(cache, lc, N, args, rank, n_sample, literal)
            out["cpu_baseline"] = {"value": v, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": "first %d batches (%d rows) of the same column, %.1f s" % (n_sample, rows_s, dt)}
            if getattr(cpu_baseline_url, "all_cores", None):
                out["cpu_baseline_all_cores"] = cpu_baseline_url.all_cores
        print(json.dumps(out), flush=True)
    scan.close()
    cache.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
