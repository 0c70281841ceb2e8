#!/usr/bin/env python3
"""bench.py — ClickBench "Q21" hot-cache filter scan on MI355X (liquid_cache_amd).

One "step" = one pass of the decode + predicate-pushdown hot path over the whole staged column chunk, COUNT(*)
included: `URL LIKE '%google%'` (benchmark/clickbench/queries/q20.sql / q21.sql of the reference share this scan) over a
synthetic 100 M-row ClickBench-shaped URL column that is fully transcoded (dictionary + FSST + fingerprints) and
resident in HBM before the timed region starts.  Reported on ONE JSON line: filtered rows/s (value), the roofline
object of the dominant kernel (live HIP-event timing on the launch stream; `achieved` counts the bytes the kernel itself
has to move, `reference_algorithm_equivalent_gbs` the reference algorithm's bytes of SURVEY §8d; hot and L3-cold), a CPU baseline (the C oracle
restating the reference's algorithm on the same bytes, whose hit count must equal the GPU's), and — at N=1 — the
secondary workloads of BASELINE.json's other configs (Int64 `>`, narrow integer / date / decimal columns, the TPC-H Q6
chain, get-with-selection, the q21.sql pushdown pipeline, LIKE without the signature index / without fingerprints).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU, row-range sharding (every rank stages and scans its own 8192-row batches; weak scaling
by default: per-GPU rows fixed; `--rows-total` fixes the table instead = strong scaling).  Exchange step per scan:
`--exchange count` (default) one 8-byte all-reduce of the per-rank COUNT(*) over RCCL; `--exchange mask` the all-gather
of the per-rank hit-mask segments into the single Arrow BooleanArray north_star names (12.5 MB for 100 M rows).
"""
from __future__ import annotations

import argparse
import copy
import datetime
import decimal
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor as _ThreadPoolExecutor

_THREAD_DEVICE = [None]  # the rank's device (set in main): HIP's current device is per thread and defaults to device 0


def _bind_thread():
    """Worker threads of a rank work on the rank's device (torch.cuda.set_device is per thread)."""
    if _THREAD_DEVICE[0] is not None:
        import torch
        torch.cuda.set_device(_THREAD_DEVICE[0])


def ThreadPoolExecutor(max_workers=None):  # noqa: N802 — the stdlib's, with every worker bound to the rank's device
    return _ThreadPoolExecutor(max_workers=max_workers, initializer=_bind_thread)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FLUSH_BYTES = 1 << 30  # scratch overwritten between cold launches (Infinity Cache: 256 MiB)
MIN_TRAFFIC_ROUND = "r5"  # oldest profiles/<round>/hbm_traffic.json whose kernels are the ones this tree launches (round 5
                          # changed the headline kernel's outputs — no mask for COUNT(*) — and the narrow-integer kernels)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=99_997_497, help="rows per GPU (ClickBench hits = 99,997,497)")
    p.add_argument("--rows-total", type=int, default=0,
                   help="strong scaling: rows of the whole table, split evenly over the ranks (overrides --rows)")
    p.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                   help="--gpus N > 1: 'strong' (default) splits ONE --rows table (ClickBench hits: 99,997,497 rows) by contiguous "
                        "row ranges, the metric's '1/2/4/8 GPU' line; 'weak' stages --rows rows on every rank")
    p.add_argument("--batch-size", type=int, default=8192)
    p.add_argument("--uniques", type=int, default=2200, help="distinct URLs per batch (nano_hits: ~2,150-2,250)")
    p.add_argument("--row-group-batches", type=int, default=54, help="batches sharing one FSST symbol table")
    p.add_argument("--needle", default="google")
    p.add_argument("--needle-ppm", type=int, default=159,
                   help="distinct URLs per million that contain the needle (ClickBench hits: 15,911 of 99,997,497 rows match)")
    p.add_argument("--workload", default="url_like", choices=["url_like", "int64_gt", "tpch_q6"],
                   help="tpch_q6: BASELINE.json config 4 (l_shipdate range + l_discount range + l_quantity, row-range "
                        "sharded; --rows-total defaults to SF100's 600,037,902 lineitem rows)")
    p.add_argument("--int-bits", type=int, default=62, help="int64_gt: FoR bit width of every batch (WatchID ~62)")
    p.add_argument("--int-kind", default="int64", choices=["int64", "int16", "date32", "decimal"],
                   help="int64_gt: Arrow type the integers are staged as (int16 / date32 / decimal128(15,2) for the narrow-"
                        "column kernels)")
    p.add_argument("--int-base", type=int, default=None, help="int64_gt: smallest value (default: a large id-like base)")
    p.add_argument("--exchange", default="count", choices=["count", "mask"],
                   help="multi-GPU exchange step per scan: COUNT(*) all-reduce or all-gather of the hit-mask segments")
    p.add_argument("--comm", default="abi", choices=["torch", "abi"],
                   help="who runs the exchange step: the library's own C ABI (lc_comm_*: RCCL directly, what a Rust host would "
                        "bind; the unique id travels over torch's store; default since round 6 — checked by a known-answer "
                        "all-reduce at start-up, torch.distributed takes over if that fails or does not return) or "
                        "torch.distributed (RCCL through PyTorch)")
    p.add_argument("--cpu-batches", type=int, default=0, help="batches in the CPU-baseline sample (0 = whole column)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-fingerprints", action="store_true",
                   help="url_like: stage the column without the SubstringSearch hint (no fingerprints, no signature index: "
                        "every dictionary value is walked)")
    p.add_argument("--no-signatures", action="store_true",
                   help="url_like: stage without the bigram signature index (reference layout only: the reference's "
                        "fingerprint prefilter decides the candidates)")
    p.add_argument("--no-row-lists", action="store_true", help="url_like: stage without the inverted row lists")
    p.add_argument("--no-like-pipeline", action="store_true",
                   help="url_like: evaluate LIKE with the one-wave-per-entry kernel only (A/B of the scan-level pipeline)")
    p.add_argument("--no-needle-classes", action="store_true", help="skip the timing of the LIKE needle classes")
    p.add_argument("--like-path", type=int, default=0, help="LC_OPT_LIKE_PATH (A/B aid): 0 auto, 1 k_str_pred, 3 k_like_lean for every needle")
    p.add_argument("--rotate", type=int, default=0,
                   help="resident tables the timed loop rotates through (one per scan) so that a scan never finds its data in "
                        "the 256 MiB Infinity Cache; 0 = as many as make the cycle READ >= 2.2 x 256 MiB (1..32)")
    p.add_argument("--scans-per-step", type=int, default=0,
                   help="scans (one pass of the predicate over one resident table each) that make up ONE timed step; 0 = 256 "
                        "(tpch_q6: 16), so that --steps 20 times tens of milliseconds instead of a third of one")
    p.add_argument("--full-line", action="store_true",
                   help="print the whole record (every secondary, ~40 KB) as the last stdout line instead of the compact one "
                        "(A/B scripts); the whole record is written to --detail-path either way")
    p.add_argument("--detail-path", default=os.path.join(ROOT, "bench_detail.json"),
                   help="where the whole record (secondaries, needle classes, sweep) goes; the stdout line stays <= 4 KB")
    p.add_argument("--stage-on-device", action="store_true",
                   help="url_like: transcode the URL batches on the device (lc_insert_arrow_batch_device) instead of the host")
    p.add_argument("--no-q21", action="store_true", help="skip the secondary q21.sql pushdown pipeline measurement")
    p.add_argument("--no-secondary", action="store_true", help="skip every secondary workload (profiling runs)")
    p.add_argument("--no-cold", action="store_true",
                   help="skip the L3-cold timing (rocprofv3 --stats runs: the kernel average then only holds hot launches)")
    p.add_argument("--secondary-rows", type=int, default=0, help="rows of the secondary workloads (0 = --rows)")
    p.add_argument("--secondary-set", default="all",
                   help="comma list of the secondary groups to run: q21,rowgroup,int,micro,q6,sweep,staging,like,concurrent (kernel A/B runs)")
    p.add_argument("--sweep-rows", type=int, default=0,
                   help="rows of the ClickBench pushdown sweep (config 5); 0 = the whole table (--rows)")
    p.add_argument("--seed", type=int, default=42)
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------- staging
def url_seed(args, rank):
    """Generator seed of the URL column: rank-free when the ranks hold row ranges of ONE table (strong scaling)."""
    return args.seed + (0 if getattr(args, "batch0", None) is not None else rank * 1_000_003)


def stage_url_column(cache, lc, N, args, rank, n_batches, threads, file_id=None):
    """Generate + transcode + stage the URL column through the public API; returns entry ids.  Under strong scaling
    (args.batch0 = this rank's first GLOBAL batch) the batches are generated by their global index from a rank-free seed and
    carry their global ids: the union of the shards is the table a single GPU stages."""
    L = N.load()
    import pyarrow as pa
    rows_total = args.rows
    bs = args.batch_size
    strong = getattr(args, "batch0", None) is not None
    b0 = args.batch0 if strong else 0
    fid = (0 if strong else rank) if file_id is None else file_id
    rgb = args.row_group_batches
    ids = [lc.ParquetArrayID.new(fid, (b + b0) // rgb, 13, (b + b0) % rgb) for b in range(n_batches)]
    seed = url_seed(args, rank)

    def do_row_group(rg):
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 512, np.uint8)
        arrs, bids = [], []
        for gb in range(max(rg * rgb, b0), min((rg + 1) * rgb, b0 + n_batches)):
            b = gb - b0
            rows = min(bs, rows_total - b * bs)
            n = N.load_bench().lc_synth_url_batch(seed, gb, rows, min(args.uniques, rows), args.needle_ppm,
                                     offs.ctypes.data, data.ctypes.data, data.size)
            arrs.append(pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1].copy()), pa.py_buffer(data[:n].copy())))
            bids.append(ids[b])
        # the batches of one row group go in together: one upload and one signature-builder launch per row group
        hint = None if args.no_fingerprints else lc.CacheExpression.SUBSTRING_SEARCH
        if args.stage_on_device:
            cache.insert_device(bids, arrs, hint)  # dictionary + FSST + index on the device (byte-identical entries)
        else:
            cache.insert_batch(bids, arrs, hint)
        return rg

    if n_batches > 0:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(do_row_group, range(b0 // rgb, (b0 + n_batches - 1) // rgb + 1)))
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
            n = N.load_bench().lc_synth_phrase_batch(args.seed + rank * 1_000_003, b, rows, 600, 870, offs.ctypes.data,
                                        data.ctypes.data, data.size)
            arr = pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1]), pa.py_buffer(data[:max(n, 1)]))
            cache.insert(ids[b], arr)
        return rg

    n_rg = (n_batches + args.row_group_batches - 1) // args.row_group_batches
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(do_row_group, range(n_rg)))
    return ids


def int_base(bits: int) -> int:
    return 4_000_000_000_000_000_000 >> (64 - bits) if bits < 63 else 0


def _dec_array(pa, unscaled: np.ndarray):
    buf = np.zeros((len(unscaled), 2), np.int64)
    buf[:, 0] = unscaled
    return pa.Array.from_buffers(pa.decimal128(15, 2), len(unscaled), [None, pa.py_buffer(buf)])


def stage_int_column(cache, lc, N, args, rank, rows_total, threads, bits=None, base=None, col=0, kind="int64",
                     on_batch=None):
    """Uniform integers in [base, base + 2^bits) per batch, staged as Int64 / Int16 / Date32 / Decimal128(15,2).
    `on_batch(b, values)` sees the generated int64 values (expected results are accumulated while staging)."""
    L = N.load()
    import pyarrow as pa
    bs = args.batch_size
    bits = args.int_bits if bits is None else bits
    base = int_base(bits) if base is None else base
    n_batches = (rows_total + bs - 1) // bs
    strong = getattr(args, "batch0", None) is not None  # row ranges of ONE table: global batch index, rank-free seed
    b0 = args.batch0 if strong else 0
    ids = [lc.ParquetArrayID.new(0 if strong else rank, (b + b0) // args.row_group_batches, col, (b + b0) % args.row_group_batches)
           for b in range(n_batches)]
    seed = args.seed + (0 if strong else rank * 1_000_003) + col * 7919

    def do_chunk(c):
        buf = np.zeros(bs, np.int64)
        for b in range(c, n_batches, threads):
            rows = min(bs, rows_total - b * bs)
            N.load_bench().lc_synth_int64_batch(seed, b + b0, rows, bits, base, buf.ctypes.data)
            v = buf[:rows]
            if on_batch is not None:
                on_batch(b, v)
            if kind == "int64":
                arr = pa.array(v)
            elif kind == "int16":
                arr = pa.array(v.astype(np.int16))
            elif kind == "date32":
                arr = pa.array(v.astype(np.int32), type=pa.date32())
            elif kind == "uint32":
                arr = pa.array(v.astype(np.uint32))
            elif kind == "int32":
                arr = pa.array(v.astype(np.int32))
            elif kind == "float64":
                arr = pa.array(v.astype(np.float64) / 100.0)  # two decimals: ALP exponent 2, packed like the integers
            else:
                arr = _dec_array(pa, v)
            cache.insert(ids[b], arr)
        return c

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(do_chunk, range(threads)))
    return ids


# ---------------------------------------------------------------------------------------------------------- helpers
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


def measured_traffic(key, kernel=None):
    """HBM bytes per launch of the dominant kernel, from the rocprofv3 --pmc passes of the newest profiled round
    (profiles/<round>/hbm_traffic.json, produced by scripts/profile_round.sh + scripts/pmc_summary.py: FETCH_SIZE scaled
    by the factor calibrated on known byte counts, plus WRITE_SIZE; counters cannot be read from inside the process).
    `kernel`: the figure is only taken from a profile of THAT kernel (a round that replaced the kernel must not inherit the
    old one's bytes).  None when this workload has not been profiled."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "hbm_traffic.json")), reverse=True):
        if os.path.basename(os.path.dirname(f)) < MIN_TRAFFIC_ROUND:
            continue  # (profiles of rounds whose kernels have since been replaced: unmeasured, not inherited)
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        e = d.get(key)
        if not isinstance(e, dict) or "traffic_bytes" not in e:
            continue
        fam = (kernel or "").split("<")[0].split(" ")[0]
        kernels = {k: v for k, v in e.items() if isinstance(v, dict)}
        summed = not any(v.get("traffic_bytes") == e["traffic_bytes"] for v in kernels.values())  # (a multi-kernel workload)
        if fam and not any(k.startswith(fam) and (summed or v.get("traffic_bytes") == e["traffic_bytes"]) for k, v in kernels.items()):
            return None, None  # the newest profile of this workload is of another kernel: unmeasured, not inherited
        return int(e["traffic_bytes"]), os.path.relpath(f, ROOT)
    return None, None


def roofline(kernel, kernel_ms, alg_bytes, kernel_bytes, cold_ms=None, traffic=None, traffic_src=None):
    """`achieved` = bytes the kernel itself has to move (lc_scan_traffic_model) / kernel time: a true fraction of the
    HBM peak.  The time is the L3-COLD one when it was measured (Infinity Cache flushed before every launch: a hot-cache
    query over a 100-column table never finds a 100 MB column in the 256 MiB memory-side cache); the back-to-back figure
    is kept as `*_hot`.  `reference_algorithm_equivalent_gbs` = the reference algorithm's bytes / the same
    time: what the scan is worth to the query.  It is NOT a bandwidth — it exceeds `achieved` (and the HBM peak) wherever
    the kernel's index structures spare it bytes."""
    t_ms = cold_ms if cold_ms is not None else kernel_ms
    ach = kernel_bytes / (t_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "timing": "l3_cold" if cold_ms is not None else "back_to_back",
           "traffic": traffic, "traffic_source": traffic_src,
           "traffic_gbs": (traffic / (t_ms * 1e-3) / 1e9) if traffic else None,
           # the same fraction by MEASURED HBM bytes (PMC): what the memory system actually delivered for this kernel
           "frac_by_traffic": (traffic / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "kernel": kernel, "kernel_ms": t_ms, "kernel_bytes_per_launch": int(kernel_bytes),
           "algorithmic_bytes_per_launch": int(alg_bytes),
           "reference_algorithm_equivalent_gbs": alg_bytes / (t_ms * 1e-3) / 1e9}
    if cold_ms is not None:
        out.update({"kernel_ms_hot": kernel_ms, "achieved_hot": kernel_bytes / (kernel_ms * 1e-3) / 1e9,
                    "frac_hot": kernel_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    # (names of earlier rounds, same numbers)
                    "kernel_ms_l3_cold": cold_ms, "frac_l3_cold": ach / HBM_PEAK_GBS})
    return out


def _fracs(node, path=""):
    """(name, frac) of every roofline-style object below `node` (dicts that carry both "frac" and a kernel / kernels)."""
    found = []
    if isinstance(node, dict):
        if isinstance(node.get("frac"), (int, float)) and ("kernel" in node or "kernels" in node or "achieved_gbs" in node):
            found.append((path, float(node["frac"])))
        for k, v in node.items():
            if isinstance(v, dict):
                found += _fracs(v, (path + "." if path else "") + str(k))
    return found


def compact_line(out, detail_path):
    """The record the driver parses: <= 4 KB.  Mirrors the reference's per-iteration record
    (benchmark/src/inprocess_runner.rs:278-352: query, iteration, time, cache/IO counters): what was run, how long it took,
    the roofline object of the dominant kernel, the CPU baseline and a handful of scalars that summarise the secondaries;
    everything else is in `detail` (bench_detail.json)."""
    g = lambda d, *ks: next((d[k] for k in ks if isinstance(d, dict) and k in d), None)  # noqa: E731
    cfg, rf = out.get("config", {}), out.get("roofline", {})
    keep_cfg = ("workload", "rows_per_gpu", "rows_all_gpus", "batches_per_gpu", "predicate", "hits", "cpu_hits",
                "hits_match_cpu_oracle", "hits_match_numpy", "needles_checked", "needle_masks_all_match_cpu_oracle",
                "rotating_columns", "rotating_columns_checked_against_oracle", "scans_per_step", "us_per_scan",
                "cycle_read_bytes", "index_bytes", "index_build_ms", "index_build_ms_steady", "first_evaluation_us", "next_scan_first_evaluation_us", "stage_seconds",
                "exchange_by", "granularity", "clickbench_protocol_us", "index_builds_wait_ms")
    c = {k: cfg[k] for k in keep_cfg if k in cfg}
    for k in ("parallelism", "evaluation_path", "step"):
        if k in cfg:
            c[k] = str(cfg[k])[:200]
    own = g(rf, "kernel_bytes_per_launch", "algorithmic_bytes")
    r = {"bound": rf.get("bound", "hbm"), "kernel": str(rf.get("kernel"))[:80], "kernel_ms": rf.get("kernel_ms"),
         "timing": rf.get("timing"), "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"),
         "frac": rf.get("frac"), "kernel_ms_hot": rf.get("kernel_ms_hot"), "frac_hot": rf.get("frac_hot"),
         "own_bytes": own, "algorithmic_bytes": g(rf, "algorithmic_bytes_per_launch", "algorithmic_bytes"),
         "traffic": rf.get("traffic"), "traffic_over_own": (rf["traffic"] / own) if rf.get("traffic") and own else None,
         "frac_by_traffic": rf.get("frac_by_traffic"), "traffic_source": rf.get("traffic_source")}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = c
    line["gb_per_s_scanned"] = out.get("gb_per_s_scanned")
    line["roofline"] = r
    if "cpu_baseline" in out:
        line["cpu_baseline"] = out["cpu_baseline"]
        ac = out.get("cpu_baseline_all_cores")
        if isinstance(ac, dict):
            line["cpu_baseline_all_cores"] = {k: ac.get(k) for k in ("value", "unit", "cores", "kind")}
    if "scaling_model" in out:
        line["scaling_model"] = {k: v for k, v in out["scaling_model"].items() if k != "note"}
    sec = out.get("secondary")
    if isinstance(sec, dict):
        fr = _fracs(sec)
        sm = {}
        if fr:
            worst = min(fr, key=lambda t: t[1])
            sm["secondary_objects"] = len(fr)
            sm["worst_secondary_frac"] = round(worst[1], 4)
            sm["worst_secondary"] = worst[0][:60]
        for name, keys in (("int64_gt_w62_frac", ("int64_gt_w62", "frac")), ("date32_gt_w12_frac", ("date32_gt_w12", "frac")),
                           ("decimal_gt_w4_frac", ("decimal_gt_w4", "frac")), ("int64_gt_w17_frac", ("int64_gt_w17", "frac")),
                           ("tpch_q6_chain_frac", ("tpch_q6_pushdown", "full_size", "frac")),
                           ("q21_pipeline_ms", ("q21_pipeline", "ms")),
                           ("q21_pipeline_partitioned_ms", ("q21_pipeline", "ms_partitioned_lists")),
                           ("byte_view_gather_after_like_ms", ("micro", "byte_view_gather_after_like", "kernel_ms")),
                           ("byte_view_gather_after_like_slotted_ms", ("micro", "byte_view_gather_after_like", "slotted_call_ms")),
                           ("url_like_no_signatures_ms", ("url_like_no_signatures", "kernel_ms")),
                           ("url_like_no_fingerprints_ms", ("url_like_no_fingerprints", "kernel_ms")),
                           ("clickbench_sweep_ms", ("clickbench_pushdown_sweep", "ms_all_queries")),
                           ("like_stream_cached_ms_per_query", ("mixed_table_like_stream", "indexes_cached_between_queries", "ms_per_query_mean")),
                           ("concurrent_table_scans_rows_per_s", ("concurrent_table_scans", "rows_per_s")),
                           ("like_stream_rebuilt_ms_per_query", ("mixed_table_like_stream", "budget_of_3_indexes_lru_thrash", "ms_per_query_mean")),
                           ("rowgroup_rows_per_s", ("rowgroup_granularity", "rows_per_s")),
                           ("rowgroup_many_rows_per_s", ("rowgroup_granularity", "many_rows_per_s")),
                           ("eval_predicate_call_us", ("rowgroup_granularity", "eval_predicate_call_us"))):
            node = sec
            for k in keys:
                node = node.get(k) if isinstance(node, dict) else None
            if isinstance(node, (int, float)):
                sm[name] = round(float(node), 5)
        sm["secondary_errors"] = sum(1 for v in sec.values() if isinstance(v, dict) and "error" in v)
        line["summary"] = sm
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    s = json.dumps(line)
    if len(s) > 4000:  # never let a long string push the line past what the driver keeps
        for k in ("evaluation_path", "parallelism", "step"):
            if k in c:
                c[k] = c[k][:80]
        line.pop("cpu_baseline_all_cores", None)
        s = json.dumps(line)
    return s


def emit(out, args):
    """Whole record -> --detail-path (and gpurun_out/, which travels back from a GPU box); compact record -> stdout."""
    paths = [args.detail_path]
    gout = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(gout):
        paths.append(os.path.join(gout, os.path.basename(args.detail_path)))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f)
        except OSError as e:
            print("warning: could not write %s: %s" % (p, e), file=sys.stderr)
    print(json.dumps(out) if args.full_line else compact_line(out, args.detail_path), flush=True)


def add_read_probe(r, cache, N):
    """The MEASURED ceiling next to the nominal one: time of a kernel that only reads the same number of bytes once
    (lc_probe_stream_read: non-temporal 16-byte loads, eight in flight per lane, best of five grid sizes), hot and L3-cold.  `frac_of_read_probe` = probe time /
    kernel time: 1.0 means the scan kernel moves its bytes as fast as this device streams that many bytes at all."""
    import ctypes as C
    try:
        B = N.load_bench()
        best = None
        for grid in (256, 512, 1024, 2048, 8192):
            hot, cold = C.c_double(), C.c_double()
            if B.lc_probe_stream_read(cache._ctx, max(int(r["kernel_bytes_per_launch"]), 65536), 10, grid, C.byref(hot), C.byref(cold)) != 0:
                return r
            if best is None or cold.value < best[1]:
                best = (hot.value, cold.value, grid)
        probe = {"bytes": int(r["kernel_bytes_per_launch"]), "hot_us": best[0], "cold_us": best[1], "grid": best[2],
                 "cold_gbs": r["kernel_bytes_per_launch"] / best[1] / 1e3, "hot_gbs": r["kernel_bytes_per_launch"] / best[0] / 1e3}
        r["read_probe"] = probe
        if r.get("timing") in ("l3_cold", "timed_region_hip_events_l3_cold_rotation"):
            r["frac_of_read_probe"] = best[1] / (r["kernel_ms"] * 1e3)
            if r.get("kernel_ms_hot"):
                r["frac_of_read_probe_hot"] = best[0] / (r["kernel_ms_hot"] * 1e3)
        else:
            r["frac_of_read_probe"] = best[0] / (r["kernel_ms"] * 1e3)
    except Exception as e:  # noqa: BLE001
        r["read_probe"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return r


def time_pred(scan, expr, torch, stream, iters, kernel, words=None, with_cold=True, probe=None, tkey=None):
    """HIP-event kernel time (hot, and with the Infinity Cache flushed before every launch) + byte model of one predicate.
    `tkey`: the workload's key in profiles/<round>/hbm_traffic.json (PMC traffic of the same kernel, when profiled)."""
    words = int(scan.mask_words) if words is None else words
    mask = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
    scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = scan.eval_timed(expr, mask.data_ptr(), iters, 0, counts.data_ptr(), stream)
    cold = scan.eval_timed_cold(expr, mask.data_ptr(), max(3, iters // 4), FLUSH_BYTES, 0, counts.data_ptr(), stream) \
        if with_cold else None
    alg, own = scan.traffic_model(expr, False)
    if kernel is None:  # the kernel the library says it ran (lc_scan_explain: "k_like_scanall (...)", "k_like_flat: ...")
        kernel = scan.explain(expr).split(" ")[0].rstrip(":")
    traffic, traffic_src = measured_traffic(tkey, kernel) if tkey else (None, None)
    r = roofline(kernel, ms, alg, own, cold, traffic, traffic_src)
    if probe is not None:
        add_read_probe(r, *probe)
    r["hits"] = int(counts.sum(dtype=torch.int64).item())
    r["rows_per_s"] = scan.rows / (r["kernel_ms"] * 1e-3)
    return r, mask, counts


# -------------------------------------------------------------------------------------------------- secondary workloads
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
           "ms_mask_form": ms, "rows_after_searchphrase": n_ne,
           "rows_out": out["rows_out"], "url_bytes_out": out["url_bytes"], "phrase_bytes_out": out["phrase_bytes"]}
    # Round 5, the same pipeline with SPARSE results: the LIKE appends its hit rows as a list (no 12.5 MB mask of zeros), both
    # projections are ONE launch each over that list (Arrow BinaryView records + data buffer).  Same conjunction, evaluated in
    # the reference's order (NotEq first: row_filter.rs:499-515); the rows out are the same set (checked below).
    hcap = 1 << 20
    hits = torch.zeros(hcap, dtype=torch.int64, device="cuda")
    n_hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    views = [torch.zeros((hcap, 2), dtype=torch.int64, device="cuda") for _ in range(2)]
    nbytes = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(2)]

    def run_hits():
        sp_scan.eval(ne_expr, m1.data_ptr(), 0, 0, stream)
        url_scan.eval_hits(like_expr, hits.data_ptr(), hcap, n_hits.data_ptr(), m1.data_ptr(), 0, 0, 0, stream)
        url_scan.gather_bytes_hits(hits.data_ptr(), n_hits.data_ptr(), hcap, views[0].data_ptr(), data[0].data_ptr(),
                                   min(data[0].numel(), (1 << 31) - 1), nbytes[0].data_ptr(), 0, stream)
        sp_scan.gather_bytes_hits(hits.data_ptr(), n_hits.data_ptr(), hcap, views[1].data_ptr(), data[1].data_ptr(),
                                  min(data[1].numel(), (1 << 31) - 1), nbytes[1].data_ptr(), 0, stream)

    for _ in range(2):
        run_hits()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        run_hits()
    torch.cuda.synchronize()
    ms_h = (time.perf_counter() - t0) / iters * 1e3
    k_h = int(n_hits.item())
    # the list holds exactly the rows of the mask form; the gathered lengths add up to the mask form's bytes
    hv = hits[:k_h].cpu().numpy().view(np.uint64)
    want_rows = refs[0][:k_out].cpu().numpy().view(np.uint64)
    assert k_h == k_out and np.array_equal(np.sort(hv), np.sort(want_rows)), "hit list differs from the mask form's rows"
    for c in range(2):
        lens = views[c][:k_h, 0].cpu().numpy().view(np.int32).reshape(-1, 2)[:, 0]
        assert int(lens.astype(np.int64).sum()) == (out["url_bytes"], out["phrase_bytes"])[c], "gathered bytes differ"
    res.update(ms_reference_order=ms_h,
               kernels_reference_order="k_str_pred (SearchPhrase <> '') + k_like_flat (hit list) + 2 x k_str_gather_hits",
               hit_list_equals_mask_form=True)
    # ... and with the conjuncts in the order a sparse pipeline wants them (a conjunction commutes): the SELECTIVE one first —
    # the LIKE leaves 16,635 rows as a hit list — and `SearchPhrase <> ''` evaluated on those rows only (lc_scan_filter_hits:
    # per record, on the row's own value), instead of mapping 100 M keys to find 13 M rows of which 2,153 survive.  The four
    # device counters are zeroed by ONE memset (LC_HITS_COUNTERS_ZEROED).
    hits2 = torch.zeros(hcap, dtype=torch.int64, device="cuda")
    ctr = torch.zeros(4, dtype=torch.int64, device="cuda")
    c_ptr = [ctr.data_ptr() + 8 * i for i in range(4)]

    gcap = min(hcap, data[0].numel() // 256)  # rows the projections are sized for (slotted form: 128 bytes per row + long values)

    # round 6: the PARTITIONED list form (LC_HITS_PARTITIONED: 16 partitions, a counter per 128-byte line) — the producers claim
    # list space on 16 addresses instead of one; [list 1 counters | list 2 counters | the two byte counters]
    P, CS = N.HITS_PARTITIONS, N.HITS_COUNTER_STRIDE
    pctr = torch.zeros(2 * P * CS + 2, dtype=torch.int64, device="cuda")
    pc_ptr = [pctr.data_ptr(), pctr.data_ptr() + 8 * P * CS, pctr.data_ptr() + 16 * P * CS, pctr.data_ptr() + 16 * P * CS + 8]

    def run_sparse(slotted=True, partitioned=False):
        if partitioned:
            N.check(cache._lib.lc_device_memset(cache.handle, pctr.data_ptr(), 0, pctr.numel() * 8, stream), cache.handle)
            cp = pc_ptr
        else:
            N.check(cache._lib.lc_device_memset(cache.handle, ctr.data_ptr(), 0, 32, stream), cache.handle)
            cp = c_ptr
        url_scan.eval_hits(like_expr, hits.data_ptr(), hcap, cp[0], 0, 0, 0, 0, stream, counters_zeroed=True, partitioned=partitioned)
        sp_scan.filter_hits(ne_expr, hits.data_ptr(), cp[0], hcap, hits2.data_ptr(), gcap if partitioned else hcap, cp[1], stream,
                            counters_zeroed=True, partitioned=partitioned)
        url_scan.gather_bytes_hits(hits2.data_ptr(), cp[1], gcap, views[0].data_ptr(), data[0].data_ptr(),
                                   min(data[0].numel(), (1 << 31) - 1), cp[2], 0, stream, counters_zeroed=True, slotted=slotted,
                                   partitioned=partitioned)
        sp_scan.gather_bytes_hits(hits2.data_ptr(), cp[1], gcap, views[1].data_ptr(), data[1].data_ptr(),
                                  min(data[1].numel(), (1 << 31) - 1), cp[3], 0, stream, counters_zeroed=True, slotted=slotted,
                                  partitioned=partitioned)

    def time_sparse(slotted, partitioned=False):
        for _ in range(2):
            run_sparse(slotted, partitioned)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_sparse(slotted, partitioned)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    ms_part = None
    try:
        ms_part = time_sparse(True, True)
        pv = pctr.cpu().numpy()
        n1 = int(pv[0: P * CS: CS].sum())
        n2p = pv[P * CS: 2 * P * CS: CS]
        stride2 = gcap // P
        h2 = hits2.cpu().numpy().view(np.uint64)
        got_p = np.concatenate([h2[p * stride2: p * stride2 + int(n2p[p])] for p in range(P)])
        assert int(n2p.max()) <= stride2, "partitioned pipeline: a partition overflowed"
        assert np.array_equal(np.sort(got_p), np.sort(want_rows)), "partitioned sparse pipeline: rows differ from the mask form's"
        for c in range(2):
            lens = views[c][: len(got_p), 0].cpu().numpy().view(np.int32).reshape(-1, 2)[:, 0]
            assert int(lens.astype(np.int64).sum()) == (out["url_bytes"], out["phrase_bytes"])[c], "partitioned pipeline: gathered bytes differ"
    except AssertionError:
        raise
    except Exception as e:  # noqa: BLE001
        res["partitioned_error"] = "%s: %s" % (type(e).__name__, e)
    ms_dense = time_sparse(False)
    ms_s = time_sparse(True)  # (the buffers checked below are the slotted run's)
    cv = ctr.cpu().numpy()
    hv2 = hits2[: int(cv[1])].cpu().numpy().view(np.uint64)
    assert int(cv[1]) == k_out and np.array_equal(np.sort(hv2), np.sort(want_rows)), "sparse pipeline: rows differ from the mask form's"
    for c in range(2):
        lens = views[c][: int(cv[1]), 0].cpu().numpy().view(np.int32).reshape(-1, 2)[:, 0]
        assert int(lens.astype(np.int64).sum()) == (out["url_bytes"], out["phrase_bytes"])[c], "sparse pipeline: gathered bytes differ"
    if ms_part is not None:
        assert n1 == int(cv[0]), "partitioned pipeline: the LIKE's hit count %d differs from the contiguous form's %d" % (n1, int(cv[0]))
        res["ms_partitioned_lists"] = ms_part
        res["partitioned_pipeline_equals_mask_form"] = True
    res.update(ms=ms_s, ms_dense_data_buffers=ms_dense, rows_per_s=url_scan.rows / (ms_s * 1e-3), rows_after_like=int(cv[0]),
               kernels="k_like_flat (hit list) + k_pred_hits (SearchPhrase <> '' on the listed rows) + 2 x k_str_gather_hits "
                       "(LC_GATHER_SLOTTED: a row's bytes in its 128-byte slot; ms_dense_data_buffers: the dense form)",
               sparse_pipeline_equals_mask_form=True)
    # the step after the path: GROUP BY "SearchPhrase" with MIN("URL") and COUNT(*) as per-entry partials on the device
    # (lc_scan_group_partials) instead of handing the selected strings to a host-side partial aggregate
    try:
        pcap = max(k_out, 1) + 64
        partials = torch.zeros((pcap, 4), dtype=torch.int32, device="cuda")
        n_part = torch.zeros(1, dtype=torch.int64, device="cuda")

        def run_partials():
            sp_scan.eval(ne_expr, m1.data_ptr(), 0, c1.data_ptr(), stream)
            url_scan.eval(like_expr, m2.data_ptr(), m1.data_ptr(), counts.data_ptr(), stream)
            sp_scan.group_partials(url_scan, partials.data_ptr(), pcap, n_part.data_ptr(), m2.data_ptr(), False, stream)

        for _ in range(2):
            run_partials()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run_partials()
        torch.cuda.synchronize()
        res["ms_with_group_partials"] = (time.perf_counter() - t0) / iters * 1e3
        npart = int(n_part.item())
        p = partials[:npart].cpu().numpy().view(np.uint32)
        res["group_partials"] = npart
        res["group_partials_rows_covered"] = int(p[:, 2].sum())  # == rows_out: every selected row is in exactly one partial
        assert res["group_partials_rows_covered"] == k_out and npart <= k_out
    except Exception as e:  # noqa: BLE001
        res["group_partials_error"] = "%s: %s" % (type(e).__name__, e)
    sp_scan.close()
    return res


def measure_get_with_selection(scan, lc, bits, base, counts, torch, stream, iters):
    """get-with-selection over an Int64 column: the selected rows' decoded values compacted in row order.
    `necessary_bytes`: selection words + the packed blocks that hold a selected row (whole 128*W-byte block when more than
    16 of its rows are selected, else two 8-byte words per selected row) + the values written."""
    import pyarrow as pa
    res = {}
    words = int(scan.mask_words)
    for sel_name, sel_frac in (("10pct", 0.1), ("0.1pct", 0.001)):
        sel_lit = base + int((1 << bits) * (1.0 - sel_frac))
        sel_mask = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
        scan.eval(lc.LiquidExpr.try_new(">", sel_lit, pa.int64()), sel_mask.data_ptr(), 0, counts.data_ptr(), stream)
        k_sel = int(counts.sum(dtype=torch.int64).item())
        vals = torch.zeros(max(k_sel, 1) + 8, dtype=torch.int64, device="cuda")
        offs = torch.zeros(scan.entries + 1, dtype=torch.int64, device="cuda")
        for _ in range(2):
            scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel_mask.data_ptr(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            scan.gather_fixed(vals.data_ptr(), vals.numel() * 8, offs.data_ptr(), sel_mask.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / iters
        # per 1024-row block (16 mask words; every batch but the last is 8 full blocks, entry segments are word aligned)
        m = sel_mask.cpu().numpy().view(np.uint64)
        pc = np.unpackbits(m.view(np.uint8)).reshape(-1, 64).sum(axis=1)
        pad = (-len(pc)) % 16
        per_block = np.concatenate([pc, np.zeros(pad, pc.dtype)]).reshape(-1, 16).sum(axis=1)
        dense = per_block > 16
        necessary = scan.rows // 8 + int(dense.sum()) * 128 * bits + int(per_block[~dense].sum()) * 16 + k_sel * 8
        alg = scan.rows * bits // 8 + scan.rows // 8 + k_sel * 8   # n*W/8 + n/8 read, k*sizeof(T) written
        g_traffic = measured_traffic("gather_10pct", "k_fixed_gather")[0] if sel_name == "10pct" and scan.rows == 99_997_497 else None
        res[sel_name] = {"kernels": "k_sel_entry_counts + k_scan_{tile_sums,tiles,apply} + k_fixed_gather<u64>",
                         "traffic": g_traffic,
                         "selected_rows": k_sel, "ms": g_ms, "necessary_bytes": int(necessary),
                         "achieved_gbs": necessary / (g_ms * 1e-3) / 1e9,
                         "frac": necessary / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes": int(alg), "reference_algorithm_equivalent_gbs": alg / (g_ms * 1e-3) / 1e9,
                         "rows_per_s": scan.rows / (g_ms * 1e-3)}
    return res


def secondary_int_columns(cache, lc, N, args, rows, threads, torch, stream, iters):
    """`col > literal` (50 % selective) over integer-like columns of the widths the other BASELINE configs use."""
    import pyarrow as pa
    out = {}
    specs = [("int64_gt_w62", "int64", 62, None, pa.int64(), "k_fixed_pred<u64> (LDS staged)", 50),
             ("date32_gt_w12", "date32", 12, 8036, pa.date32(), "k_fixed_pred_reg<u32>", 51),
             ("int16_gt_w12", "int16", 12, 0, pa.int16(), "k_fixed_pred_reg<u16>", 52),
             ("decimal_gt_w4", "decimal", 4, 0, pa.decimal128(15, 2), "k_fixed_pred_reg<u64>", 53),
             ("int64_gt_w17", "int64", 17, 1000, pa.int64(), "k_fixed_pred_reg<u64>", 54)]
    for name, kind, bits, base, dtype, kernel, col in specs:
        try:
            base_v = int_base(bits) if base is None else base
            ids = stage_int_column(cache, lc, N, args, 1, rows, threads, bits=bits, base=base_v, col=col, kind=kind)
            scan = cache.scan(ids)
            lit = base_v + (1 << (bits - 1))
            if kind == "decimal":
                lit_v = decimal.Decimal(lit) / 100
            elif kind == "date32":
                lit_v = datetime.date(1970, 1, 1) + datetime.timedelta(days=lit)
            else:
                lit_v = lit
            r, _, counts = time_pred(scan, lc.LiquidExpr.try_new(">", lit_v, dtype), torch, stream, iters, kernel, probe=(cache, N),
                                     tkey=name)
            r["rows"] = int(scan.rows)
            if name == "int64_gt_w62":
                r["get_with_selection"] = measure_get_with_selection(scan, lc, bits, base_v, counts, torch, stream, iters)
            out[name] = r
            scan.close()
            cache.evict(ids)
        except Exception as e:  # noqa: BLE001 - a secondary measurement must never cost the headline line
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def secondary_micro(cache, lc, N, args, rows, threads, torch, stream, iters, url_scan=None):
    """The reference's own micro-benchmark shapes and the hot-path kernels earlier rounds never timed, each
    with a roofline object (own bytes / HIP-event time, hot and L3-cold):
      * u32 bit widths {1,3,7,11,19,27} (core/bench/bitpacking.rs:12-13), `>` at 50 %, and W=11 at selectivity
        {0.01,0.1,0.3,0.7,0.9} (bitpacking.rs:62-75);
      * Int32 `= 500` over uniform [0,1024) at batch 16384 (datafusion/bench/filter_pushdown.rs:23-31,115-119);
      * config 2 at selectivity {0.01 %, 1 %, 50 %} (Int64 W=62 / W=17, Date32 W=12);
      * ALP float predicate (float_array.rs:294-316), string Eq / ordering through the prefix keys
        (byte_view_array/comparisons.rs:21-151, 351-405), date-part extraction over decoded values."""
    import ctypes as C
    import pyarrow as pa
    out = {}

    def run(name, kind, bits, base, dtype, kernel, col, lits, a=None, op=">"):
        try:
            ids = stage_int_column(cache, lc, N, a or args, 1, rows, threads, bits=bits, base=base, col=col, kind=kind)
            scan = cache.scan(ids)
            for tag, lit in lits:
                if kind == "date32":
                    lit_v = datetime.date(1970, 1, 1) + datetime.timedelta(days=int(lit))
                elif kind == "float64":
                    lit_v = float(lit) / 100.0
                else:
                    lit_v = int(lit)
                r, _, _ = time_pred(scan, lc.LiquidExpr.try_new(op, lit_v, dtype), torch, stream, iters, kernel)
                r["rows"] = int(scan.rows)
                r["selectivity"] = r["hits"] / max(int(scan.rows), 1)
                out[name + tag] = r
            scan.close()
            cache.evict(ids)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    frac = lambda bits, base, s: base + int((1 << bits) * (1.0 - s))  # noqa: E731  (`>` literal for selectivity s)
    for w in (1, 3, 7, 11, 19, 27):
        lits = [("", (1 << (w - 1)) - (1 if w == 1 else 0))]
        if w == 11:
            lits += [("_sel%g" % sv, frac(w, 0, sv)) for sv in (0.01, 0.1, 0.3, 0.7, 0.9)]
        run("uint32_gt_w%d" % w, "uint32", w, 0, pa.uint32(), "k_fixed_pred_reg<u32>", 60 + w, lits)
    a16 = copy.copy(args)
    a16.batch_size = 16384
    run("int32_eq_500_batch16384", "int32", 10, 0, pa.int32(), "k_fixed_pred_reg<u32>", 95, [("", 500)], a=a16, op="=")
    sels = (("_sel0.01pct", 1e-4), ("_sel1pct", 1e-2), ("_sel50pct", 0.5))
    b62 = int_base(62)
    run("int64_gt_w62", "int64", 62, b62, pa.int64(), "k_fixed_pred<u64> (LDS staged)", 96, [(t, frac(62, b62, sv)) for t, sv in sels])
    run("int64_gt_w17", "int64", 17, 1000, pa.int64(), "k_fixed_pred_reg<u64>", 97, [(t, frac(17, 1000, sv)) for t, sv in sels])
    run("date32_gt_w12", "date32", 12, 8036, pa.date32(), "k_fixed_pred_reg<u32>", 98, [(t, frac(12, 8036, sv)) for t, sv in sels])
    run("float64_alp_gt_w17", "float64", 17, 0, pa.float64(), "k_fixed_pred_reg<u64> (ALP, packed-domain range)", 99,
        [("", frac(17, 0, 0.5))])
    if url_scan is not None:
        try:  # string Eq / ordering: prefix keys decide almost every dictionary value, the row lists / keys give the rows
            offs = np.zeros(args.batch_size + 1, np.int32)
            data = np.zeros(args.batch_size * 512, np.uint8)
            nb = N.load_bench().lc_synth_url_batch(url_seed(args, 0), 0, args.batch_size, min(args.uniques, args.batch_size),
                                                   args.needle_ppm, offs.ctypes.data, data.ctypes.data, data.size)
            v0 = bytes(data[:nb][offs[0]: offs[1]])
            for tag, op, lit in (("string_eq_existing_value", "=", v0), ("string_lt", "<", b"http://m"), ("string_ge", ">=", b"http://m")):
                r, _, _ = time_pred(url_scan, lc.LiquidExpr.try_new(op, lit, pa.string()), torch, stream, max(3, iters // 2), None)
                r["rows"] = int(url_scan.rows)
                r["predicate"] = "URL %s %r" % (op, lit[:40])
                out[tag] = r
        except Exception as e:  # noqa: BLE001
            out["string_eq_ordering"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:  # get().with_selection() of a byte-view column, device resident (k_sel_entry_counts + scans + k_str_sel_rows +
            # k_str_decode_sel): 1 % of the rows, and the rows a selective LIKE leaves
            words = int(url_scan.mask_words)
            like = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
            m_like = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
            url_scan.eval(like, m_like.data_ptr(), 0, 0, stream)
            g = torch.Generator(device="cuda")
            g.manual_seed(7)
            # ~1.5 % of the rows: the AND of six random words has one bit in 64 set
            m_rand = torch.randint(-(1 << 62), 1 << 62, (6, max(words, 1)), dtype=torch.int64, device="cuda", generator=g)
            m_1pct = m_rand[0] & m_rand[1] & m_rand[2] & m_rand[3] & m_rand[4] & m_rand[5]
            cap = 1 << 21
            row_offs = torch.zeros(url_scan.entries + 1, dtype=torch.int64, device="cuda")
            refs = torch.zeros(cap, dtype=torch.int64, device="cuda")
            voffs = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
            data = torch.zeros(cap * 128 + (64 << 20), dtype=torch.uint8, device="cuda")  # (slotted form: cap slots + long values)
            for tag, m in (("byte_view_gather_1.5pct", m_1pct), ("byte_view_gather_after_like", m_like)):
                def run():
                    url_scan.gather_bytes_async(row_offs.data_ptr(), refs.data_ptr(), voffs.data_ptr(), cap, data.data_ptr(),
                                                data.numel(), m.data_ptr(), 0, stream)
                run()
                torch.cuda.synchronize()
                k = int(row_offs[-1].item())
                nbytes = int(voffs[min(k, cap)].item())
                assert 0 < k <= cap and nbytes <= data.numel()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n_it = max(3, iters // 2)
                e0.record()
                for _ in range(n_it):
                    run()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n_it
                # what the gather has to move: the selection words twice (counts, rows), per selected row its key (2), the
                # offset pair (~8), ~44 compressed bytes, reference + length + value offset out (20) and the decoded bytes
                need = 2 * words * 8 + k * (2 + 8 + 20) + int(nbytes * 0.58) + nbytes
                out[tag + "_mask_form"] = {"bound": "hbm", "kernel": "k_sel_entry_counts + k_scan_* + k_str_sel_rows + k_scan_* + k_str_decode_sel",
                            "kernel_ms": ms, "rows": int(url_scan.rows), "selected_rows": k, "bytes_out": nbytes,
                            "kernel_bytes_per_launch": int(need), "achieved": need / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": need / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "timing": "back_to_back", "traffic": None,
                            "rows_out_per_s": k / (ms * 1e-3)}
                # round 5: the same rows as a HIT LIST (what lc_scan_eval_hits leaves), gathered by ONE launch into BinaryView
                # records + a data buffer
                hits_t = torch.zeros(cap, dtype=torch.int64, device="cuda")
                n_h = torch.zeros(1, dtype=torch.int64, device="cuda")
                n_b = torch.zeros(1, dtype=torch.int64, device="cuda")
                views_t = torch.zeros((cap, 2), dtype=torch.int64, device="cuda")
                url_scan.mask_to_hits(m.data_ptr(), hits_t.data_ptr(), cap, n_h.data_ptr(), 0, stream)

                def run_h():
                    url_scan.gather_bytes_hits(hits_t.data_ptr(), n_h.data_ptr(), cap, views_t.data_ptr(), data.data_ptr(),
                                               min(data.numel(), (1 << 31) - 1), n_b.data_ptr(), 0, stream)
                run_h()
                torch.cuda.synchronize()
                assert int(n_h.item()) == k
                lens_h = views_t[:k, 0].cpu().numpy().view(np.int32).reshape(-1, 2)[:, 0].astype(np.int64)
                assert int(lens_h.sum()) == nbytes, "hit-list gather: lengths differ from the mask form's"
                # timed by a C loop (lc_bench_gather_bytes_hits_timed: events around back-to-back calls of the public entry point; a
                # Python loop measures its own pace).  Two forms: the stand-alone call, which zeroes its byte counter with a one-wave
                # kernel in front of the gather, and the gather kernel as a pipeline runs it (counters zeroed once up front; the
                # calls append behind each other in the data buffer)
                B = N.load_bench()
                ms_c = C.c_float()
                n_c = max(n_it, 20)
                cap_b = min(data.numel(), (1 << 31) - 1)
                n_app = max(1, min(n_c, cap_b // max(nbytes + 64, 1) - 1))

                def timed_c(flags, n):
                    N.check(B.lc_bench_gather_bytes_hits_timed(cache._ctx, url_scan._h, C.c_void_p(hits_t.data_ptr()),
                                                               C.c_void_p(n_h.data_ptr()), cap, C.c_void_p(views_t.data_ptr()),
                                                               C.c_void_p(data.data_ptr()), cap_b, C.c_void_p(n_b.data_ptr()), flags,
                                                               C.c_void_p(stream or None), n, C.byref(ms_c)), cache._ctx)
                    return float(ms_c.value)
                ms_call = timed_c(0, n_c)
                ms_h = timed_c(1, n_app)
                assert int(n_b.item()) == n_app * nbytes, "appended gathers: byte total differs"
                # LC_GATHER_SLOTTED: record i's bytes at i * 128 (longer values behind the slots): nothing to claim for the common
                # value, no barrier, no store stage — the same Arrow array over a sparse buffer
                lens_dense = views_t[:k, 0].clone()
                ms_slot = timed_c(2, n_c)
                same = bool(((views_t[:k, 0] & 0xFFFFFFFF) == (lens_dense & 0xFFFFFFFF)).all().item())
                assert same, "slotted gather: lengths differ from the dense form's"
                long_bytes = int(n_b.item())
                # what it has to move: per row its record (8), key (2), offset pair (~8), prefix key (8), ~0.58 compressed
                # bytes per decoded byte, the view (16) and the decoded bytes
                need_h = k * (8 + 2 + 8 + 8 + 16) + int(nbytes * 0.58) + nbytes
                out[tag] = {"bound": "hbm", "kernel": "k_str_gather_hits (one launch over the hit list)", "kernel_ms": ms_h,
                            "rows": int(url_scan.rows), "selected_rows": k, "bytes_out": nbytes,
                            "kernel_bytes_per_launch": int(need_h), "achieved": need_h / (ms_h * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": need_h / (ms_h * 1e-3) / 1e9 / HBM_PEAK_GBS, "timing": "back_to_back",
                            "traffic": None, "rows_out_per_s": k / (ms_h * 1e-3), "mask_form_ms": ms,
                            "call_ms_with_counter_reset": ms_call, "appended_calls_timed": n_app,
                            "slotted_call_ms": ms_slot, "slotted_bytes_behind_the_slots": long_bytes,
                            "slotted_rows_out_per_s": k / (ms_slot * 1e-3)}
        except Exception as e:  # noqa: BLE001
            out["byte_view_gather"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:  # date-part extraction over decoded Date32 values, in place (k_date_component / lossy reconstruction)
        ids = stage_int_column(cache, lc, N, args, 1, min(rows, 33_554_432), threads, bits=12, base=8036, col=94, kind="date32")
        scan = cache.scan(ids)
        n = int(scan.rows)
        vals = torch.randint(8036, 8036 + 4096, (n,), dtype=torch.int32, device="cuda")
        scan.date_part(vals.data_ptr(), n, lc.Date32Field.YEAR, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            scan.date_part(vals.data_ptr(), n, lc.Date32Field.YEAR, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out["date_part_year_in_place"] = {"bound": "hbm", "kernel": "k_date_lossy<i32>", "kernel_ms": ms, "rows": n,
                                          "kernel_bytes_per_launch": 8 * n, "achieved": 8 * n / (ms * 1e-3) / 1e9,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "timing": "back_to_back", "traffic": None}
        scan.close()
        cache.evict(ids)
    except Exception as e:  # noqa: BLE001
        out["date_part_year_in_place"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:  # boolean_buffer_and_then through the host API: a per-call latency (8192-bit masks), not a bandwidth
        left = np.random.default_rng(5).random(8192) < 0.5
        right = np.random.default_rng(6).random(int(left.sum())) < 0.5
        lc.boolean_buffer_and_then(cache, left, right)
        t0 = time.perf_counter()
        for _ in range(200):
            lc.boolean_buffer_and_then(cache, left, right)
        out["mask_and_then_host_call"] = {"unit": "us per call (8192-bit left, host buffers in and out)",
                                          "value": (time.perf_counter() - t0) / 200 * 1e6}
    except Exception as e:  # noqa: BLE001
        out["mask_and_then_host_call"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def secondary_rowgroup(cache, lc, N, args, ids, expr, whole_scan_hits, rows):
    """The number a drop-in gets when it calls at the REFERENCE's granularity: one evaluation per row group (the ~54 batches
    of one ColumnAccessPath; liquid_stream.rs:358-430, liquid_cache_reader.rs:264-294) from T host threads on their own
    streams, the row-group scans created once and kept — and per-entry lc_eval_predicate calls.  Driver: C++ over the public
    ABI (lc_bench_rowgroup_run), so that no Python / GIL time is in the numbers."""
    import ctypes as C
    B = N.load_bench()
    rgb = args.row_group_batches
    ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
    begins = list(range(0, len(ids), rgb)) + [len(ids)]
    gb = np.ascontiguousarray(np.asarray(begins, dtype=np.uint64))
    pred = expr.as_predicate()
    out = {"row_groups": len(begins) - 1, "entries_per_row_group": rgb, "rows": int(rows),
           "driver": "lc_bench_rowgroup_run: one lc_scan_eval_count per unit and pass, T threads x own stream, scans kept"}
    runs = {}
    best = None
    for label, threads, gps, with_mask in (("t1", 1, 1, 0), ("t4", 4, 1, 0), ("t8", 8, 1, 0), ("t16", 16, 1, 0),
                                           ("t8_mask", 8, 1, 1), ("t8_x8", 8, 8, 0), ("t1_x226", 1, len(begins) - 1, 0)):
        st = N.RowGroupStats()
        rc = B.lc_bench_rowgroup_run(cache._ctx, len(begins) - 1, gb.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p), threads, 5,
                                     with_mask, gps, C.byref(st))
        if rc != 0:
            runs[label] = {"error": "lc_bench_rowgroup_run rc %d" % rc}
            continue
        assert int(st.hits) == whole_scan_hits, "row-group calls: COUNT(*) %d != the whole scan's %d" % (st.hits, whole_scan_hits)
        r = {"threads": threads, "row_groups_per_call": gps, "mask_written": bool(with_mask), "calls_per_pass": int(st.units),
             "pass_us": st.wall_s / st.passes * 1e6, "rows_per_s": rows * st.passes / st.wall_s,
             "call_us_host_side": st.call_us_mean, "first_pass_ms": st.first_pass_s * 1e3,
             "us_per_call_wall": st.wall_s / st.passes / max(int(st.units), 1) * 1e6 * threads}
        runs[label] = r
        if gps == 1 and not with_mask and (best is None or r["rows_per_s"] > best["rows_per_s"]):
            best = r
    out["runs"] = runs
    if best:
        out["rows_per_s"] = best["rows_per_s"]
        out["best_threads"] = best["threads"]
        out["hits_equal_whole_scan"] = True
    # MANY row groups per call (round 6): a thread's partition of the row groups in ONE call per pass with per-row-group counts —
    # mode 0: lc_eval_predicate_row_groups (ids in, host counts out, the scan from the context's scan cache);
    # mode 1: lc_scan_eval_count_groups on a kept scan + one stream wait
    B.lc_bench_rowgroup_many.restype = C.c_int32
    B.lc_bench_rowgroup_many.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.POINTER(N.RowGroupStats)]
    many = {}
    best_many = None
    for label, threads, mode in (("ids_t1", 1, 0), ("ids_t4", 4, 0), ("ids_t16", 16, 0), ("scan_t1", 1, 1), ("scan_t4", 4, 1),
                                 ("scan_t16", 16, 1)):
        st = N.RowGroupStats()
        rc = B.lc_bench_rowgroup_many(cache._ctx, len(begins) - 1, gb.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p), threads, 20, mode,
                                      C.byref(st))
        if rc != 0:
            many[label] = {"error": "lc_bench_rowgroup_many rc %d" % rc}
            continue
        assert int(st.hits) == whole_scan_hits, "many-row-group calls: COUNT(*) %d != the whole scan's %d" % (st.hits, whole_scan_hits)
        r = {"threads": threads, "call": "lc_eval_predicate_row_groups" if mode == 0 else "lc_scan_eval_count_groups + stream wait",
             "row_groups_per_call": (len(begins) - 1) // threads, "pass_us": st.wall_s / st.passes * 1e6,
             "rows_per_s": rows * st.passes / st.wall_s, "call_us": st.call_us_mean, "first_pass_ms": st.first_pass_s * 1e3}
        many[label] = r
        if best_many is None or r["rows_per_s"] > best_many["rows_per_s"]:
            best_many = dict(r, label=label)
    out["many_row_groups_per_call"] = many
    if best_many:
        out["many_rows_per_s"] = best_many["rows_per_s"]
        out["many_best"] = best_many["label"]
    # the per-entry drop-in call (host buffers out): what `impl LiquidArray for GpuLiquidArray` pays per batch
    n_e = min(len(ids), 2048)
    for threads in (1, 8):
        us, hits = C.c_double(), C.c_uint64()
        rc = B.lc_bench_entry_calls(cache._ctx, n_e, ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p),
                                    threads, 2, args.batch_size, C.byref(us), C.byref(hits))
        if rc == 0:
            out["eval_predicate_call_us" if threads == 1 else "eval_predicate_call_us_8_callers"] = us.value
            out["eval_predicate_hits_first_%d_entries" % n_e] = int(hits.value)
        else:
            out["eval_predicate_call_error"] = rc
    return out


def secondary_concurrent_tables(cache, N, args, tables, expr, want_hits, rows_per_table):
    """Throughput when several independent whole-table scans are in flight at once (what a server running queries of several
    sessions does): one lc_scan_eval_count per TABLE and pass, T host threads on their own streams (lc_bench_rowgroup_run with a
    table as the unit; the scans and their indexes are created in an untimed first pass).  The headline `value` stays the
    ONE-stream figure — a scan is latency bound (one wave per group, ~1 round of the device), so a second stream fills what
    the first leaves idle; this says by how much."""
    import ctypes as C
    B = N.load_bench()
    ids_np = np.ascontiguousarray(np.concatenate([np.asarray([int(e) for e in t], dtype=np.uint64) for t in tables]))
    begins = np.ascontiguousarray(np.cumsum([0] + [len(t) for t in tables]).astype(np.uint64))
    pred = expr.as_predicate()
    out = {"tables": len(tables), "rows_per_table": int(rows_per_table),
           "driver": "lc_bench_rowgroup_run: unit = a whole table, one lc_scan_eval_count per unit and pass, T threads x own stream"}
    runs = {}
    for threads in (1, 2, 4, 8):
        if threads > len(tables):
            break
        st = N.RowGroupStats()
        rc = B.lc_bench_rowgroup_run(cache._ctx, len(tables), begins.ctypes.data_as(C.POINTER(C.c_uint64)),
                                     ids_np.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(C.byref(pred), C.c_void_p), threads, 16, 0, 1,
                                     C.byref(st))
        if rc != 0:
            runs["t%d" % threads] = {"error": "lc_bench_rowgroup_run rc %d" % rc}
            continue
        assert int(st.hits) == int(want_hits), "concurrent table scans: COUNT(*) %d != %d" % (st.hits, want_hits)
        n_scans = int(st.units) * int(st.passes)
        runs["t%d" % threads] = {"threads": threads, "us_per_scan": st.wall_s / n_scans * 1e6,
                                 "rows_per_s": rows_per_table * n_scans / st.wall_s}
    out["runs"] = runs
    ok = [r for r in runs.values() if "rows_per_s" in r]
    if ok:
        best = max(ok, key=lambda r: r["rows_per_s"])
        out["rows_per_s"] = best["rows_per_s"]
        out["best_threads"] = best["threads"]
        out["us_per_scan"] = best["us_per_scan"]
    return out


def make_abi_communicator(cache, N, Communicator, rank, world, torch, dist, dry_run):
    """The library's own communicator (lc_comm_*: RCCL through the C ABI), its 128-byte id broadcast over the launcher's process
    group — and PROVEN before it is used: a known-answer all-reduce (rank + 1 -> world (world + 1) / 2) runs in a helper thread
    with a time limit, and the ranks agree (over torch.distributed) whether everyone passed.  One failure, wrong sum or rank that
    did not return, and every rank falls back to torch.distributed: a scaling run is never lost to this path.  dry_run (ranks
    sharing one GPU under LC_BENCH_TEST_BACKEND): the shared-memory test backend, which RCCL's refusal of duplicate devices
    makes necessary."""
    import threading
    note = None
    comm = None
    ok = 1
    try:
        if dry_run:
            cache.set_option(N.OPT_COMM_SHARED_MEMORY, 1)
        uid = [Communicator.unique_id(cache) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        result = {}

        def probe():
            try:
                _bind_thread()  # (a new thread starts on device 0)
                # (the communicator's creation is a collective as well: inside the time limit, like the all-reduce)
                result["comm"] = Communicator(cache, rank, world, uid[0])
                comm = result["comm"]
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    t = torch.full((), rank + 1, dtype=torch.int64, device="cuda")
                    side.synchronize()
                    comm.allreduce_count(t.data_ptr(), side.cuda_stream)
                    side.synchronize()
                    result["sum"] = int(t.item())
            except Exception as e:  # noqa: BLE001
                result["error"] = "%s: %s" % (type(e).__name__, e)

        th = threading.Thread(target=probe, daemon=True)
        th.start()
        th.join(60.0)
        comm = result.get("comm")
        if th.is_alive():
            ok, note = 0, "its creation / known-answer all-reduce did not return within 60 s"
        elif result.get("sum") != world * (world + 1) // 2:
            ok, note = 0, "its known-answer all-reduce gave %s" % (result.get("error") or result.get("sum"))
    except Exception as e:  # noqa: BLE001 — never lose the scaling run to this path
        ok, note = 0, "%s: %s" % (type(e).__name__, e)
    agreed = torch.tensor([ok], dtype=torch.int64, device="cuda")
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if int(agreed.item()) == 1:
        return comm, "lc_comm (%s through the C ABI), known-answer all-reduce passed on every rank" % (
            "shared-memory test backend" if dry_run else "RCCL")
    return None, "torch.distributed (lc_comm not used: %s)" % (note or "another rank failed its check")


def clickbench_protocol(lc, N, args, rank, n_batches, threads, expr, want_hits, torch, stream, device):
    """ClickBench's own protocol on a FRESH context (nothing cached, no index anywhere): the table is staged, then q20
    (`SELECT COUNT(*) FROM hits WHERE URL LIKE '%google%'`) runs 5 times, and every run is what a host in the reference's call
    shape does per query (liquid_cache_reader.rs:264-339: a reader per query names its entries) — lc_scan_create over the id
    list, one COUNT(*) evaluation, the count read back, lc_scan_destroy — host wall clock, synchronised.  Run 1 is answered by
    the entry-level index while the scan-level one is built off the path; the paper's figure is the mean of the last 3."""
    fresh = (lc.LiquidCacheBuilder.new().with_device(device).with_batch_size(args.batch_size)
             .with_index_options(signatures=not args.no_signatures, row_lists=not args.no_row_lists).build())
    try:
        ids = stage_url_column(fresh, lc, N, args, rank, n_batches, threads)
        ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
        total = torch.zeros((), dtype=torch.int64, device="cuda")
        runs, paths = [], []
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        for _ in range(5):
            cur.synchronize()
            t0 = time.perf_counter()
            sc = fresh.scan(ids_np)
            sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
            # (the count is read back on the QUERY's stream: a device-wide synchronise would wait for the builder's stream too)
            got = int(total.item())
            k = int(sc.info().last_like_kernel)
            sc.close()
            runs.append((time.perf_counter() - t0) * 1e6)
            paths.append(N.LIKE_KERNEL_NAMES.get(k, "?"))
            assert got == want_hits, "clickbench protocol: COUNT(*) %d != %d" % (got, want_hits)
        # ... and with the scan-level index in place (five back-to-back runs take ~6 ms, the build ~5: ClickBench's runs are
        # seconds apart)
        sc = fresh.scan(ids_np)
        t_w = time.perf_counter()
        sc.index_wait()
        waited_ms = (time.perf_counter() - t_w) * 1e3
        sc.close()
        steady = []
        for _ in range(4):
            cur.synchronize()
            t0 = time.perf_counter()
            sc = fresh.scan(ids_np)
            sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
            got = int(total.item())
            k = int(sc.info().last_like_kernel)
            sc.close()
            steady.append((time.perf_counter() - t0) * 1e6)
            paths.append(N.LIKE_KERNEL_NAMES.get(k, "?"))
            assert got == want_hits, "clickbench protocol: COUNT(*) %d != %d" % (got, want_hits)
        return {"runs_us": [round(x, 1) for x in runs], "run1_us": round(runs[0], 1), "mean_last3_us": round(float(np.mean(runs[2:])), 1),
                "index_in_place_runs_us": [round(x, 1) for x in steady], "index_in_place_mean_last3_us": round(float(np.mean(steady[1:])), 1),
                "waited_for_the_builder_ms": round(waited_ms, 2),
                "answered_by": paths, "query": "q20: scan create + COUNT(*) WHERE URL LIKE '%google%' + count read back + scan destroy",
                "rows": int(args.rows)}
    finally:
        fresh.close()


def secondary_like_stream(cache, lc, N, args, cols, expr, want_hits, torch, stream, index_bytes, live_indexes):
    """Index residency under a mixed query stream (round 4's review: nothing exercised eviction): LIKE COUNT(*) queries round
    robin over SIX resident string tables, a scan per query as DataFusion builds a reader per query — created, evaluated once,
    destroyed.  What a query costs depends on where its table's scan-level index is: cached from the last query on that table
    (adopted: records + one launch), gone (rebuilt: ~10 ms), or not allowed (budget held by live scans: the entry-level index
    of k_like_lean serves).  Every COUNT(*) is checked."""
    out = {"tables": len(cols), "index_bytes_per_table": int(index_bytes),
           "query": "scan create + URL LIKE COUNT(*) (d_mask_out = NULL) + scan destroy, host wall clock, synchronised"}
    total = torch.zeros((), dtype=torch.int64, device="cuda")
    cols = [np.ascontiguousarray(np.asarray([int(e) for e in ids_c], dtype=np.uint64)) for ids_c in cols]

    def query(c, keep=False):
        t0 = time.perf_counter()
        sc = cache.scan(cols[c])
        sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
        got = int(total.item())  # (waits for the query's stream only: a builder at work on another table is not waited for)
        path = N.LIKE_KERNEL_NAMES.get(int(sc.info().last_like_kernel), "?")  # (what answered; does not wait for a build)
        if not keep:
            sc.close()
        dt = time.perf_counter() - t0
        assert got == want_hits[c], "LIKE stream: table %d COUNT(*) %d != %d" % (c, got, want_hits[c])
        return dt, path, sc

    def run(label, budget_indexes, cache_n, rounds, held=()):
        cache.set_option(N.OPT_LIKE_INDEX_CACHE, cache_n)
        cache.set_option(N.OPT_LIKE_INDEX_BUDGET_BYTES, int((budget_indexes + live_indexes) * index_bytes * 1.02) if budget_indexes else 0)
        live = [query(c, keep=True)[2] for c in held]
        for c in range(len(cols)):  # one untimed round: whatever can be cached is
            if c not in held:
                query(c)
        times, paths = [], {}
        for _ in range(rounds):
            for c in range(len(cols)):
                if c in held:
                    continue
                dt, path, _ = query(c)
                times.append(dt)
                paths[path] = paths.get(path, 0) + 1
        for sc in live:
            sc.close()
        out[label] = {"queries": len(times), "ms_per_query_mean": float(np.mean(times)) * 1e3,
                      "ms_per_query_max": float(np.max(times)) * 1e3, "paths": paths}

    try:
        run("indexes_cached_between_queries", 0, 8, 5)                    # every query finds its table's scan (and index) kept
        run("budget_of_3_indexes_lru_thrash", 3, 8, 4)                    # six tables through three index slots: a table without
        #                                                                   one is answered by k_like_lean, builds run off the path
        run("budget_held_by_3_live_scans", 3, 8, 5, held=(0, 1, 2))       # the other three tables: k_like_lean (entry-level index)
        cache.set_option(N.OPT_SCAN_CACHE, 0)
        run("scan_cache_off_indexes_cached", 0, 8, 3)                     # round 5's shape: a scan really created per query
        cache.set_option(N.OPT_SCAN_CACHE, 32)
    finally:
        cache.set_option(N.OPT_SCAN_CACHE, 32)
        cache.set_option(N.OPT_LIKE_INDEX_CACHE, 4)
        cache.set_option(N.OPT_LIKE_INDEX_BUDGET_BYTES, 0)
    return out


def q6_literals():
    epoch = datetime.date(1970, 1, 1)
    return ((datetime.date(1992, 1, 2) - epoch).days, (datetime.date(1994, 1, 1) - epoch).days,
            (datetime.date(1995, 1, 1) - epoch).days)


def q6_synth_batch(L, seed, global_batch, n, bufs):
    from liquid_cache_amd import _native as N
    """Synthetic lineitem columns of one batch, keyed by the GLOBAL batch index.  ship: 2^12 days from 1992-01-02 (W=12;
    the SF100 column spans 2,526 days); discount 0..15 hundredths (W=4, TPC-H has 0..10); quantity 1..64, x100 as the
    unscaled Decimal(15,2) (W=13, TPC-H has 1..50)."""
    ship, disc, qty = bufs
    N.load_bench().lc_synth_int64_batch(seed + 101, global_batch, n, 12, q6_literals()[0], ship.ctypes.data)
    N.load_bench().lc_synth_int64_batch(seed + 102, global_batch, n, 4, 0, disc.ctypes.data)
    N.load_bench().lc_synth_int64_batch(seed + 103, global_batch, n, 6, 1, qty.ctypes.data)
    return ship[:n], disc[:n], qty[:n] * 100


def q6_expected_count(sh, di, qt):
    _, d1, d2 = q6_literals()
    return int(((sh >= d1) & (sh < d2) & (di >= 5) & (di <= 7) & (qt < 2400)).sum())


def q6_expected_qty_sum(sh, di, qt):
    _, d1, d2 = q6_literals()
    return int(qt[(sh >= d1) & (sh < d2) & (di >= 5) & (di <= 7) & (qt < 2400)].sum())


def q6_expected_qty_disc_product(sh, di, qt):
    _, d1, d2 = q6_literals()
    m = (sh >= d1) & (sh < d2) & (di >= 5) & (di <= 7) & (qt < 2400)
    return int((qt[m].astype(np.int64) * di[m].astype(np.int64)).sum())


def stage_q6_columns(cache, lc, N, args, rows, threads, batch0=0):
    """l_shipdate (Date32), l_discount and l_quantity (Decimal128(15,2)) for `rows` rows starting at global batch
    `batch0` (the synthetic generator is keyed by the GLOBAL batch index, so a shard holds the same bytes whatever the
    number of ranks).  Returns (ids per column, expected COUNT(*) per batch from numpy)."""
    import pyarrow as pa
    bs = args.batch_size
    n_batches = (rows + bs - 1) // bs
    L = N.load()
    ids = {c: [lc.ParquetArrayID.new(2, (b + batch0) // args.row_group_batches, c, (b + batch0) % args.row_group_batches)
               for b in range(n_batches)] for c in (10, 6, 4)}
    expected = np.zeros(n_batches, np.int64)
    expected_qty = np.zeros(n_batches, np.int64)
    expected_qd = np.zeros(n_batches, np.int64)

    def stage(chunk):
        bufs = [np.zeros(bs, np.int64) for _ in range(3)]
        for b in range(chunk, n_batches, threads):
            n = min(bs, rows - b * bs)
            sh, di, qt = q6_synth_batch(L, args.seed, b + batch0, n, bufs)
            expected[b] = q6_expected_count(sh, di, qt)
            expected_qty[b] = q6_expected_qty_sum(sh, di, qt)
            expected_qd[b] = q6_expected_qty_disc_product(sh, di, qt)
            cache.insert(ids[10][b], pa.array(sh.astype(np.int32), type=pa.date32()))
            cache.insert(ids[6][b], _dec_array(pa, di))
            cache.insert(ids[4][b], _dec_array(pa, qt))

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(stage, range(threads)))
    stage_q6_columns.last_expected_qty_sum = int(expected_qty.sum())
    stage_q6_columns.last_expected_qty_disc = int(expected_qd.sum())
    return ids, expected


def q6_conjuncts(lc, s_ship, s_disc, s_qty):
    import pyarrow as pa
    DEC = pa.decimal128(15, 2)
    E = lc.LiquidExpr.try_new
    conj = [(s_ship, E(">=", datetime.date(1994, 1, 1), pa.date32())), (s_ship, E("<", datetime.date(1995, 1, 1), pa.date32())),
            (s_disc, E(">=", decimal.Decimal("0.05"), DEC)), (s_disc, E("<=", decimal.Decimal("0.07"), DEC)),
            (s_qty, E("<", decimal.Decimal("24.00"), DEC))]
    fused = [(s_ship, [conj[0][1], conj[1][1]]), (s_disc, [conj[2][1], conj[3][1]]), (s_qty, [conj[4][1]])]
    return conj, fused


Q6_WIDTHS = {"ship": 12, "disc": 4, "qty": 13}


def q6_one_launch_bytes(rows):
    # the fused chain reads every column once and writes ONE mask (the intermediate masks stay on the chip)
    return rows * sum(Q6_WIDTHS.values()) // 8 + rows // 8


def q6_algorithmic_bytes(rows, passes):
    # n*W/8 + selection n/8 (all but the first pass) + n/8 out
    return sum(rows * Q6_WIDTHS[c] // 8 + (rows // 8 if has_sel else 0) + rows // 8 for c, has_sel in passes)


def secondary_tpch_q6(cache, lc, N, args, rows, threads, torch, stream, iters):
    """TPC-H Q6-shaped pushdown: l_shipdate >= d1 AND l_shipdate < d2 AND l_discount >= 0.05 AND
    l_discount <= 0.07 AND l_quantity < 24, every mask the selection of the next predicate; `fused` evaluates the two
    range pairs in one pass each (lc_scan_eval_and).  COUNT(*) per batch is checked against numpy, exactly."""
    ids, expected = stage_q6_columns(cache, lc, N, args, rows, threads)
    ids_ship, ids_disc, ids_qty = ids[10], ids[6], ids[4]
    s_ship, s_disc, s_qty = cache.scan(ids_ship), cache.scan(ids_disc), cache.scan(ids_qty)
    words = int(s_ship.mask_words)
    masks = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(2)]
    counts = torch.zeros(s_ship.entries, dtype=torch.int32, device="cuda")
    conj, fused = q6_conjuncts(lc, s_ship, s_disc, s_qty)

    def run_chain():
        sel = 0
        for i, (scan, expr) in enumerate(conj):
            out = masks[i & 1]
            scan.eval(expr, out.data_ptr(), sel, counts.data_ptr(), stream)
            sel = out.data_ptr()

    def run_fused():
        sel = 0
        for i, (scan, exprs) in enumerate(fused):
            out = masks[i & 1]
            ok = scan.eval_and(exprs, out.data_ptr(), sel, counts.data_ptr(), stream)
            assert ok
            sel = out.data_ptr()

    from liquid_cache_amd.pushdown import CompiledFilter
    compiled = CompiledFilter.from_conjunction(fused)

    def run_one_launch():
        # the whole conjunction as ONE kernel (k_fixed_chain through lc_scan_eval_filter): every wave takes its entry
        # through the three columns, the intermediate masks never leave the chip
        final = compiled.run(masks[0].data_ptr(), masks[1].data_ptr(), counts.data_ptr(), 0, 0, stream)
        assert final == masks[(len(fused) - 1) & 1].data_ptr() or final == masks[0].data_ptr()

    res = {"rows": int(rows), "conjuncts": 5, "count": int(expected.sum())}
    for tag, fn, passes in (("chained_5_passes", run_chain, [("ship", 0), ("ship", 1), ("disc", 1), ("disc", 1), ("qty", 1)]),
                            ("fused_3_passes", run_fused, [("ship", 0), ("disc", 1), ("qty", 1)]),
                            ("one_launch_3_columns", run_one_launch, [("ship", 0), ("disc", 0), ("qty", 0)])):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        got = counts.cpu().numpy().astype(np.int64)
        assert got.tolist() == expected.tolist(), "device COUNT(*) per batch differs from numpy (%s)" % tag
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        alg = q6_one_launch_bytes(rows) if tag == "one_launch_3_columns" else q6_algorithmic_bytes(rows, passes)
        res[tag] = {"ms": ms, "rows_per_s": rows / (ms * 1e-3), "algorithmic_bytes": int(alg),
                    "achieved_gbs": alg / (ms * 1e-3) / 1e9, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the step after the path: SUM / MIN / MAX(l_quantity) of the rows the chain selected, on the device (lc_scan_aggregate)
    agg_out = torch.zeros(6, dtype=torch.int64, device="cuda")
    final_mask = masks[(len(fused) - 1) & 1]
    run_fused()
    s_qty.aggregate(agg_out.data_ptr(), final_mask.data_ptr(), stream)
    torch.cuda.synchronize()
    a = [int(x) for x in agg_out.cpu().numpy().astype(np.int64)]
    assert a[0] == int(expected.sum()) and a[1] == stage_q6_columns.last_expected_qty_sum and a[2] == 0, \
        "device SUM(l_quantity) under the Q6 mask differs from numpy"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        s_qty.aggregate(agg_out.data_ptr(), final_mask.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    res["aggregate_sum_min_max_quantity"] = {"ms": e0.elapsed_time(e1) / iters, "selected_rows": a[0],
                                             "sum_unscaled": a[1], "matches_numpy": True}
    # Q6's aggregate is a SUM over a product of two decimal columns (sum(l_extendedprice * l_discount)); with the columns
    # staged here: SUM(l_quantity * l_discount) under the mask, exact, on the device (lc_scan_sum_product)
    s_qty.sum_product(s_disc, agg_out.data_ptr(), final_mask.data_ptr(), stream)
    torch.cuda.synchronize()
    a = [int(x) for x in agg_out.cpu().numpy().astype(np.int64)]
    assert a[0] == int(expected.sum()) and a[1] == stage_q6_columns.last_expected_qty_disc and a[2] == 0, \
        "device SUM(l_quantity * l_discount) under the Q6 mask differs from numpy"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        s_qty.sum_product(s_disc, agg_out.data_ptr(), final_mask.data_ptr(), stream)
    e1.record()
    torch.cuda.synchronize()
    res["aggregate_sum_quantity_times_discount"] = {"ms": e0.elapsed_time(e1) / iters, "selected_rows": a[0],
                                                    "sum_unscaled": a[1], "matches_numpy": True}
    # the chain is worth the 5-pass algorithmic bytes to the query whichever way it is run
    alg5 = res["chained_5_passes"]["algorithmic_bytes"]
    res["fused_3_passes"]["effective_gbs_vs_5_pass_bytes"] = alg5 / (res["fused_3_passes"]["ms"] * 1e-3) / 1e9
    for s in (s_ship, s_disc, s_qty):
        s.close()
    cache.evict(ids_ship + ids_disc + ids_qty)
    return res


def secondary_transcode_rate(cache, lc, N, args, rows, threads):
    """Arrow -> Liquid staging rate: the host transcoder (`cache.insert`, one thread per batch stripe —
    what the reference's background transcode threads do) against the on-device one (`lc_insert_arrow_device`: the raw
    values cross PCIe once, min / max and FastLanes packing are kernels), same Int64 W=62 and Date32 W=12 batches; the
    entries of both paths are byte-identical (tests), so only the time is reported."""
    import pyarrow as pa
    bs = args.batch_size
    n_batches = max(1, min(rows, 16_777_216) // bs)
    L = N.load()
    res = {"rows": n_batches * bs, "batches": n_batches}
    for tag, bits, base, to_arrow in (("int64_w62", 62, int_base(62), lambda v: pa.array(v)),
                                      ("date32_w12", 12, 8036, lambda v: pa.array(v.astype(np.int32), type=pa.date32()))):
        buf = np.zeros(bs, np.int64)
        arrays = []
        for b in range(n_batches):
            N.load_bench().lc_synth_int64_batch(args.seed + 7, b, bs, bits, base, buf.ctypes.data)
            arrays.append(to_arrow(buf.copy()))
        ids_h = [lc.ParquetArrayID.new(8, b // args.row_group_batches, 1, b % args.row_group_batches) for b in range(n_batches)]
        ids_d = [lc.ParquetArrayID.new(8, b // args.row_group_batches, 2, b % args.row_group_batches) for b in range(n_batches)]

        def host_stripe(c):
            for b in range(c, n_batches, threads):
                cache.insert(ids_h[b], arrays[b])

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(host_stripe, range(threads)))
        t_host = time.perf_counter() - t0
        chunk = 256  # arrays per device call
        t0 = time.perf_counter()
        for c in range(0, n_batches, chunk):
            cache.insert_device(ids_d[c:c + chunk], arrays[c:c + chunk])
        t_dev = time.perf_counter() - t0
        res[tag] = {"host_threads": threads, "host_rows_per_s": n_batches * bs / t_host, "host_seconds": t_host,
                    "device_rows_per_s": n_batches * bs / t_dev, "device_seconds": t_dev,
                    "device_calls": (n_batches + chunk - 1) // chunk}
        cache.evict(ids_h + ids_d)
    # URL-shaped Utf8 batches: dictionary + FSST + prefix keys + fingerprints + compact offsets + the acceleration index.
    # Symbol tables are trained beforehand (once per column chunk, on the host, for both paths: transcode.rs:16-33), so the
    # timed part is the per-batch encoding — host threads over row groups against one device call per row group.
    try:
        nb = max(1, min(n_batches, 512))
        rgb = args.row_group_batches
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 512, np.uint8)
        arrays = []
        for b in range(nb):
            n = N.load_bench().lc_synth_url_batch(args.seed + 11, b, bs, min(args.uniques, bs), args.needle_ppm,
                                                  offs.ctypes.data, data.ctypes.data, data.size)
            arrays.append(pa.StringArray.from_buffers(bs, pa.py_buffer(offs.copy()), pa.py_buffer(data[:n].copy())))
        hint = lc.CacheExpression.SUBSTRING_SEARCH
        ids_h = [lc.ParquetArrayID.new(9, b // rgb, 1, b % rgb) for b in range(nb)]
        ids_d = [lc.ParquetArrayID.new(9, b // rgb, 2, b % rgb) for b in range(nb)]
        paths = [1_000_000 + b // rgb for b in range(nb)]
        groups = [list(range(g, min(g + rgb, nb))) for g in range(0, nb, rgb)]
        warm = [lc.ParquetArrayID.new(9, g[0] // rgb, 3, 0) for g in groups]
        for g, w in zip(groups, warm):
            cache.insert(w, arrays[g[0]], hint, path_id=paths[g[0]])  # trains the row group's symbol table

        def host_group(g):
            cache.insert_batch([ids_h[b] for b in g], [arrays[b] for b in g], hint, path_ids=[paths[b] for b in g])

        def dev_group(g):
            cache.insert_device([ids_d[b] for b in g], [arrays[b] for b in g], hint, path_ids=[paths[b] for b in g])

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(host_group, groups))
        t_host = time.perf_counter() - t0
        dev_group(groups[0])  # (first launch of the two kernels)
        cache.evict([ids_d[b] for b in groups[0]])
        t0 = time.perf_counter()
        for g in groups:
            dev_group(g)
        t_dev1 = time.perf_counter() - t0
        cache.evict(ids_d)
        dev_threads = min(threads, 4)
        t_dev = None
        for _ in range(2):  # (the first round lets every thread's call allocate its pinned / device staging buffers)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=dev_threads) as ex:
                list(ex.map(dev_group, groups))
            t_dev = time.perf_counter() - t0
        same = all(cache.entry_bytes(ids_h[b]) == cache.entry_bytes(ids_d[b]) for b in range(0, nb, max(1, nb // 16)))
        raw_bytes = int(sum(a.nbytes for a in arrays))
        res["url_utf8"] = {"rows": nb * bs, "arrow_bytes": raw_bytes, "host_threads": threads,
                           "host_rows_per_s": nb * bs / t_host, "host_seconds": t_host,
                           "device_rows_per_s_one_thread": nb * bs / t_dev1, "device_seconds_one_thread": t_dev1,
                           "device_threads": dev_threads, "device_rows_per_s": nb * bs / t_dev, "device_seconds": t_dev,
                           "device_calls": len(groups), "sampled_entries_byte_identical": bool(same)}
        cache.evict(ids_h + ids_d + warm)
    except Exception as e:  # noqa: BLE001
        res["url_utf8"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def secondary_like_variants(lc, N, args, rank, n_batches, threads, torch, stream, iters, pattern):
    """The LIKE scan in the reference-algorithm regimes: the reference's fingerprint prefilter only (no bigram signature
    index staged), and a column staged without the SubstringSearch hint (no fingerprints: every dictionary value walked)."""
    import copy
    import pyarrow as pa
    out = {}
    for name, sig, nofp in (("url_like_no_signatures", False, False),
                            ("url_like_no_fingerprints", True, True)):
        try:
            cache2 = (lc.LiquidCacheBuilder.new().with_device(torch.cuda.current_device())
                      .with_index_options(signatures=sig).build())
            a2 = copy.copy(args)
            a2.no_fingerprints = nofp
            ids = stage_url_column(cache2, lc, N, a2, rank, n_batches, threads)
            scan = cache2.scan(ids)
            hint = lc.CacheExpression.SUBSTRING_SEARCH
            expr = lc.LiquidExpr.try_new("like", pattern, pa.string(), hint)
            r, _, _ = time_pred(scan, expr, torch, stream, max(3, iters // 2), None, with_cold=True, probe=(cache2, N),
                                tkey=name)
            out[name] = r
            scan.close()
            cache2.close()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def secondary_clickbench_sweep(cache, lc, args, rows, threads, torch, stream, iters):
    """BASELINE.json config 5: the pushed-down predicates of all 43 ClickBench queries (24 have a WHERE clause) over a
    synthetic hits-shaped table, each run as the reference's row filter would (conjunct order of row_filter.rs:499-515,
    every mask the selection of the next conjunct, adjacent ranges on one column fused, IN list as a Kleene OR) — device
    resident, no host round trip inside a query.  Per query: milliseconds per evaluation and surviving rows."""
    from liquid_cache_amd import clickbench as cb
    from liquid_cache_amd.pushdown import LiquidRowFilter, PushdownExecutor
    t0 = time.perf_counter()
    columns, ids, _ = cb.stage_hits(cache, rows, seed=args.seed, batch_size=args.batch_size,
                                    row_group_batches=args.row_group_batches, threads=threads)
    t_stage = time.perf_counter() - t0
    ex = PushdownExecutor(columns)
    any_scan = next(iter(columns.values())).scan
    words = int(any_scan.mask_words)
    masks = [torch.zeros(max(words, 1), dtype=torch.int64, device="cuda") for _ in range(2)]
    counts = torch.zeros(max(any_scan.entries, 1), dtype=torch.int32, device="cuda")
    out = {"rows": int(rows), "columns": len(columns), "stage_seconds": round(t_stage, 1), "queries": {}}
    total_ms = 0.0
    for q in range(cb.N_QUERIES):
        conj = cb.QUERIES.get(q)
        if not conj:
            continue
        rf = LiquidRowFilter(conj)
        cf = ex.compile(rf)  # the whole filter is one call into the library per evaluation (lc_scan_eval_filter)
        a, b = masks[0].data_ptr(), masks[1].data_ptr()
        for _ in range(2):
            cf.run(a, b, counts.data_ptr(), 0, 0, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            cf.run(a, b, counts.data_ptr(), 0, 0, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        total_ms += ms
        # bytes of the query's passes by the library's own byte model (lc_scan_traffic_model per conjunct; passes behind the
        # first are priced WITH their selection words but as if no entry were skipped — the kernels do skip entries whose
        # selection is empty, so for multi-conjunct queries this is an upper bound and `frac` with it)
        steps = ex.plan(rf)
        own = alg = 0
        for k, stp in enumerate(steps):
            for sc, ex_ in zip(stp.scans if stp.kind == "or" else stp.scans * len(stp.exprs), stp.exprs):
                a_b, o_b = sc.traffic_model(ex_, k > 0)
                own += int(o_b)
                alg += int(a_b)
        out["queries"]["q%d" % q] = {"passes": len(steps), "conjuncts": sum(len(s.exprs) for s in steps),
                                     "ms": ms, "rows_per_s": rows / (ms * 1e-3),
                                     "rows_out": int(counts.sum(dtype=torch.int64).item()),
                                     "own_bytes_model": own, "algorithmic_bytes": alg,
                                     "achieved_gbs": own / (ms * 1e-3) / 1e9,
                                     "frac": own / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "filter": " AND ".join(s.text for s in steps)[:160]}
    out["total_ms_all_queries"] = total_ms
    for c in columns.values():
        c.scan.close()
    cache.evict([e for v in ids.values() for e in v])
    return out


# ---------------------------------------------------------------------------------------------------------- CPU baseline
# Full-size parity beyond the headline needle: the same column scanned for these needles by the GPU and by the CPU oracle
# (all cores, untimed).  `mail` is the needle that exposed the speculative-walk false positives of round 2 (3,112 rows in the
# 4 of 226 row groups whose symbol table holds "mail" under the code of a frequently escaped byte).
EXTRA_PARITY_NEEDLES = ("mail", "file", "ru/")
# The needle classes of the LIKE path (secondary.like_needle_classes): every one is timed (hot and L3-cold) and its hit
# mask compared bit for bit with the CPU oracle's over the whole column.  (label, operator, needle)
NEEDLE_CLASSES = (
    ("1_byte", "like", "q"),                                   # no bigram: the reference's fingerprint filter decides
    ("selective_6", "like", "google"),                         # the headline
    ("selective_4", "like", "file"),
    ("mid_3", "like", "ru/"),
    ("non_selective_4", "like", "mail"),                       # ~19 % of the rows
    ("selective_16", "like", "yandex.ru/search"),              # > 15 bytes: the LDS automaton image grows to 17 KB
    ("selective_24", "like", "market.yandex.ru/catalog"),
    ("long_34", "like", "images.yandex.ru/search/catalog/it"),  # > 31 bytes: automaton table walked in global memory
    ("not_like_6", "not_like", "google"),
    ("not_like_absent", "not_like", "zzzzqqq"),
)


def cpu_baseline_url(cache, lc, N, args, rank, n_sample, pattern, threads, extra_patterns=()):
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
            n = N.load_bench().lc_synth_url_batch(args.seed + rank * 1_000_003, b, rows, min(args.uniques, rows), args.needle_ppm,
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
    all_cores = {"value": rows_total / dt_mt, "unit": "rows/s", "cores": cores, "kind": "port",
                 "sample": "same batches, one batch per OpenMP task, %.2f s" % dt_mt, "hits": int(hits_mt)}
    # checker leg (untimed): the oracle's hit MASK and per-batch counts for the headline pattern and the extra needles, in
    # scan layout, for a bit-for-bit comparison with the GPU's over the whole sample
    seg = np.zeros(n_sample + 1, np.uint64)
    for b in range(n_sample):
        seg[b + 1] = seg[b] + (min(bs, args.rows - b * bs) + 63) // 64
    checks = {}
    for op, p in (("like", pattern),) + tuple(extra_patterns):
        t1 = time.perf_counter()
        r = lo.bench_eval_batches_masks(bl, sts, lo.OP_NAMES[op], p, seg, cores)
        checks[(op, p.decode())] = r + (time.perf_counter() - t1,)
    all_cores["_checks"] = checks
    return rows_total / dt, rows_total, int(hits), dt, all_cores


def oracle_url_sample_counts(cache, lc, N, args, rank, file_id, batches, pattern, threads):
    """Checker leg: the oracle's per-batch COUNT(*) of `pattern` for the given batches of a staged URL column
    (regenerated from its seed and transcoded with the column's own symbol tables: the bytes the GPU holds)."""
    from oracle import liquid_oracle as lo
    import pyarrow as pa
    bs = args.batch_size
    blobs = [None] * len(batches)
    seed = url_seed(args, rank)

    def prep(c):
        offs = np.zeros(bs + 1, np.int32)
        data = np.zeros(bs * 512, np.uint8)
        for k in range(c, len(batches), threads):
            b = batches[k]
            rows = min(bs, args.rows - b * bs)
            n = N.load_bench().lc_synth_url_batch(seed, b, rows, min(args.uniques, rows), args.needle_ppm,
                                                  offs.ctypes.data, data.ctypes.data, data.size)
            arr = pa.StringArray.from_buffers(rows, pa.py_buffer(offs[: rows + 1]), pa.py_buffer(data[:n]))
            eid = lc.ParquetArrayID.new(file_id, b // args.row_group_batches, 13, b % args.row_group_batches)
            path = lc.ParquetArrayID.column_access_path(eid)
            blobs[k] = (cache.transcode(arr, lc.CacheExpression.SUBSTRING_SEARCH, path), path)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(prep, range(threads)))
    symtabs = {}
    for _, path in blobs:
        if path not in symtabs:
            symtabs[path] = lo.symtab_load(cache.symbol_table(path))
    seg = np.zeros(len(batches) + 1, np.uint64)
    for k, b in enumerate(batches):
        seg[k + 1] = seg[k] + (min(bs, args.rows - b * bs) + 63) // 64
    _, _, cpu_counts, = lo.bench_eval_batches_masks([bl for bl, _ in blobs], [symtabs[pth] for _, pth in blobs], lo.LIKE,
                                                    pattern, seg, usable_cores())
    return np.asarray(cpu_counts).astype(np.int64)


def cpu_baseline_int(cache, lc, N, args, rank, n_sample, literal, base):
    from oracle import liquid_oracle as lo
    import pyarrow as pa
    L = N.load()
    bs = args.batch_size
    buf = np.zeros(bs, np.int64)
    blobs = []
    rows_total = 0
    for b in range(n_sample):
        rows = min(bs, args.rows - b * bs)
        N.load_bench().lc_synth_int64_batch(args.seed + rank * 1_000_003, b, rows, args.int_bits, base, buf.ctypes.data)
        v = buf[:rows]
        if args.int_kind == "int16":
            arr = pa.array(v.astype(np.int16))
        elif args.int_kind == "date32":
            arr = pa.array(v.astype(np.int32), type=pa.date32())
        elif args.int_kind == "decimal":
            arr = _dec_array(pa, v)
        else:
            arr = pa.array(v)
        blobs.append(cache.transcode(arr))
        rows_total += rows
    t0 = time.perf_counter()
    hits = lo.bench_eval_batches(blobs, None, lo.GT, literal, 1)
    dt = time.perf_counter() - t0
    cores = usable_cores()
    t1 = time.perf_counter()
    hits_mt = lo.bench_eval_batches(blobs, None, lo.GT, literal, cores)
    dt_mt = time.perf_counter() - t1
    all_cores = {"value": rows_total / dt_mt, "unit": "rows/s", "cores": cores, "kind": "port",
                 "sample": "same batches, one batch per OpenMP task, %.2f s" % dt_mt, "hits": int(hits_mt)}
    return rows_total / dt, rows_total, int(hits), dt, all_cores


def cpu_baseline_q6(cache, lc, args, ids, expected, n_sample):
    """The five Q6 conjuncts chained per batch by the CPU oracle (eval_predicate over the selection of the previous
    conjunct, boolean_buffer_and_then in between) on the Liquid bytes of the first n_sample batches — the bytes the GPU
    scanned, read back from HBM.  Single thread; COUNT(*) per batch checked against numpy."""
    from oracle import liquid_oracle as lo
    _, d1, d2 = q6_literals()
    blobs = [(cache.entry_bytes(ids[10][b]), cache.entry_bytes(ids[6][b]), cache.entry_bytes(ids[4][b])) for b in range(n_sample)]
    rows = 0
    t0 = time.perf_counter()
    for b, (l_ship, l_disc, l_qty) in enumerate(blobs):
        n = lo.array_info(l_ship).len
        sel = np.ones(n, bool)
        for liquid, op, lit in ((l_ship, lo.GE, d1), (l_ship, lo.LT, d2), (l_disc, lo.GE, 5), (l_disc, lo.LE, 7), (l_qty, lo.LT, 2400)):
            sel = lo.and_then(sel, lo.eval_predicate(liquid, op, lit, sel).filter_mask())
        assert int(sel.sum()) == int(expected[b]), "CPU oracle COUNT(*) of batch %d differs from numpy" % b
        rows += n
    dt = time.perf_counter() - t0
    return {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port", "hits": int(expected[:n_sample].sum()),
            "sample": "first %d batches (%d rows) of the same three columns, 5 chained conjuncts, %.1f s" % (n_sample, rows, dt)}


# ---------------------------------------------------------------------------------------------------------- main
def run_tpch_q6(cache, lc, N, args, rank, world, batch0, threads, scaling, torch, dist):
    """BASELINE.json config 4 as a bench workload: the Q6-shaped chain over this rank's contiguous row range of a
    --rows-total table (strong scaling; the shards hold the same bytes whatever the world size, so COUNT(*) is the same
    number for every N and equals numpy's).  A step = the three fused passes (ship-date pair, discount pair, quantity),
    every mask the selection of the next, COUNT(*) produced by the last kernel and summed over ranks by an 8-byte
    all-reduce that overlaps the next step."""
    from liquid_cache_amd.sharding import PipelinedCountAllReduce
    t_stage = time.perf_counter()
    ids, expected = stage_q6_columns(cache, lc, N, args, args.rows, threads, batch0)
    t_stage = time.perf_counter() - t_stage
    s_ship, s_disc, s_qty = cache.scan(ids[10]), cache.scan(ids[6]), cache.scan(ids[4])
    words = int(s_ship.mask_words)
    masks = [torch.zeros(max(words, 1), dtype=torch.int64, device="cuda") for _ in range(2)]
    counts = torch.zeros(max(s_ship.entries, 1), dtype=torch.int32, device="cuda")
    _, fused = q6_conjuncts(lc, s_ship, s_disc, s_qty)
    reducer = PipelinedCountAllReduce(lambda: torch.zeros((), dtype=torch.int64, device="cuda"), world)
    stream = torch.cuda.current_stream().cuda_stream

    from liquid_cache_amd.pushdown import CompiledFilter
    compiled = CompiledFilter.from_conjunction(fused)

    def chain(total_ptr, counts_ptr):
        # one call, one kernel: lc_scan_eval_filter runs the three columns as k_fixed_chain, COUNT(*) from the same kernel
        compiled.run(masks[0].data_ptr(), masks[1].data_ptr(), counts_ptr, 0, total_ptr, stream)

    sps = args.scans_per_step if args.scans_per_step > 0 else 16  # chains per timed step (a chain over 600 M rows: ~0.55 ms)

    def step():
        for _ in range(sps):
            total = reducer.acquire()
            chain(total.data_ptr(), 0)
            reducer.submit()

    for _ in range(args.warmup):
        step()
    reducer.drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    reducer.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    hits = int(reducer.last().item())
    # per-batch counts of the last pass == numpy's per batch; their sum over ranks == the fused COUNT(*)
    scratch = torch.zeros((), dtype=torch.int64, device="cuda")
    chain(scratch.data_ptr(), counts.data_ptr())
    torch.cuda.synchronize()
    got = counts.cpu().numpy().astype(np.int64)[: len(expected)]
    assert got.tolist() == expected.tolist(), "device COUNT(*) per batch differs from numpy"
    sums = torch.tensor([int(expected.sum()), int(s_ship.rows)], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    assert int(sums[0].item()) == hits, "fused COUNT(*) %d != numpy %d" % (hits, int(sums[0].item()))
    rows_all = int(sums[1].item())
    # kernel time of one chain on this rank (HIP events on the launch stream)
    iters = max(5, args.steps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        chain(scratch.data_ptr(), 0)
    e1.record()
    torch.cuda.synchronize()
    chain_ms = e0.elapsed_time(e1) / iters
    if rank != 0:
        return
    rows = int(s_ship.rows)
    alg3 = q6_one_launch_bytes(rows)
    alg5 = q6_algorithmic_bytes(rows, [("ship", 0), ("ship", 1), ("disc", 1), ("disc", 1), ("qty", 1)])
    out = {
        "metric": "filtered rows/s (+ GB/s scanned), TPC-H Q6-shaped pushdown chain (BASELINE.json config 4)",
        "value": rows_all * sps / elapsed * args.steps, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "timed_region_s": elapsed,
        "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "date32+decimal128", "data": "synthetic",
        "config": {"workload": "tpch_q6_shipdate_discount_quantity_chain", "rows_per_gpu": rows, "rows_all_gpus": rows_all,
                   "batch_rows": args.batch_size, "batches_per_gpu": int(s_ship.entries), "first_global_batch": batch0,
                   "parallelism": "contiguous row-range shards x%d (assign_row_ranges over equal batches), 8-byte "
                                  "COUNT(*) all-reduce per step" % world,
                   "predicate": "l_shipdate >= 1994-01-01 AND l_shipdate < 1995-01-01 AND l_discount BETWEEN 0.05 AND "
                                "0.07 AND l_quantity < 24", "hits": hits, "hits_match_numpy": True,
                   "stage_seconds": round(t_stage, 2), "scans_per_step": sps,
                   "us_per_scan": elapsed / args.steps / sps * 1e6,
                   "step": "%d chains, each one launch over the rank's row range of the three columns" % sps},
        "gb_per_s_scanned": alg5 * world / (elapsed / args.steps / sps) / 1e9,
        "roofline": {"bound": "hbm", "kernel": "k_fixed_chain (one launch over 3 columns: u32 W=12, u64 W=4, u64 W=13)",
                     "achieved": alg3 / (chain_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg3 / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": measured_traffic("tpch_q6", "k_fixed_chain")[0] if rows_all == 600_037_902 and world == 1 else None,
                     "kernel_ms": chain_ms,
                     "algorithmic_bytes": int(alg3), "effective_gbs_vs_5_pass_bytes": alg5 / (chain_ms * 1e-3) / 1e9},
    }
    if world == 1 and not args.no_secondary:
        out["roofline"]["kernel_bytes_per_launch"] = int(alg3)
        out["roofline"]["timing"] = "back_to_back"  # (2.25 GB per pass: far beyond the Infinity Cache, hot = cold)
        add_read_probe(out["roofline"], cache, N)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_q6(cache, lc, args, ids, expected, min(len(expected), args.cpu_batches or 1500))
    emit(out, args)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    scaling = "weak"
    if args.workload == "tpch_q6" and not args.rows_total:
        args.rows_total = 600_037_902  # TPC-H SF100 lineitem
    if world > 1 and not args.rows_total and args.scaling != "weak":
        # the metric's "1/2/4/8 GPU" line: ONE table of --rows rows (ClickBench hits: 99,997,497) split by contiguous row
        # ranges (ParquetArrayID keeps file / row group / batch of a row range together, datafusion/src/cache/id.rs:15-21)
        args.rows_total = args.rows
    if args.scaling == "strong" and not args.rows_total:
        args.rows_total = args.rows
    batch0 = 0
    args.batch0 = None
    if args.rows_total:
        # strong scaling: contiguous, batch-aligned row ranges per rank (liquid_cache_amd.sharding.assign_row_ranges
        # over equally weighted batches gives exactly this split)
        from liquid_cache_amd.sharding import contiguous_batch_range
        total_batches = (args.rows_total + args.batch_size - 1) // args.batch_size
        b0, b1 = contiguous_batch_range(total_batches, rank, world)
        args.rows = max(0, min(args.rows_total, b1 * args.batch_size) - b0 * args.batch_size)
        batch0 = b0
        scaling = "strong"
        if world > 1:
            args.batch0 = b0  # generators and ids take GLOBAL batch indices: the shards' union is the one-GPU table

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: liquid_cache_amd has no CPU fallback")
    # LC_BENCH_TEST_BACKEND=gloo: dry run of the multi-rank code path on a box with ONE GPU (every rank on device 0, the
    # collectives through gloo) — a test aid for the launcher / sharding / reporting logic, never a measurement
    test_backend = os.environ.get("LC_BENCH_TEST_BACKEND")
    if test_backend:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    _THREAD_DEVICE[0] = local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_backend:
            dist.init_process_group(backend=test_backend)
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    import pyarrow as pa

    cache = (lc.LiquidCacheBuilder.new().with_device(local_rank).with_batch_size(args.batch_size)
             .with_index_options(signatures=not args.no_signatures, row_lists=not args.no_row_lists,
                                 like_pipeline_min_entries=-1 if args.no_like_pipeline else None,
                                 like_path=args.like_path or None).build())
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, min(world, 8))))

    if args.workload == "tpch_q6":
        return run_tpch_q6(cache, lc, N, args, rank, world, batch0, threads, scaling, torch, dist)

    t_stage = time.perf_counter()
    if args.workload == "url_like":
        ids = stage_url_column(cache, lc, N, args, rank, n_batches, threads)
        pattern = ("%" + args.needle + "%").encode()
        expr = lc.LiquidExpr.try_new("like", pattern, pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
        workload = "clickbench_q21_url_like_%s%s" % (args.needle, "_no_fingerprints" if args.no_fingerprints else "")
        dtype, kernel = "u8", "k_str_pred"
    else:
        base = int_base(args.int_bits) if args.int_base is None else args.int_base
        ids = stage_int_column(cache, lc, N, args, rank, args.rows, threads, base=base, kind=args.int_kind)
        literal = base + (1 << (args.int_bits - 1)) if args.int_bits < 64 else 0
        if args.int_kind == "decimal":
            expr = lc.LiquidExpr.try_new(">", decimal.Decimal(literal) / 100, pa.decimal128(15, 2))
        elif args.int_kind == "date32":
            expr = lc.LiquidExpr.try_new(">", datetime.date(1970, 1, 1) + datetime.timedelta(days=literal), pa.date32())
        else:
            expr = lc.LiquidExpr.try_new(">", literal, pa.int16() if args.int_kind == "int16" else pa.int64())
        workload = "clickbench_%s_gt_w%d" % (args.int_kind, args.int_bits)
        lanes = {"int64": "u64", "decimal": "u64", "date32": "u32", "int16": "u16"}[args.int_kind]
        dtype, kernel = args.int_kind, ("k_fixed_pred_reg<%s>" % lanes if args.int_bits <= 32 else "k_fixed_pred<%s>" % lanes)
    t_stage = time.perf_counter() - t_stage

    scan = cache.scan(ids)
    words = int(scan.mask_words)
    mask = torch.zeros(max(words, 1), dtype=torch.int64, device="cuda")
    counts = torch.zeros(max(scan.entries, 1), dtype=torch.int32, device="cuda")
    # what the FIRST evaluation of this predicate on a fresh scan costs (host wall clock, device drained before and after):
    # the scan's records, the scan-level index of k_like_flat, the folded automata and the plan's trial launch — a query's
    # first run pays it, the steady state of the timed loop does not
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    scan.eval(expr, mask.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().synchronize()  # (the query's stream: the scan-level index is being built on the builder's meanwhile)
    first_eval_us = (time.perf_counter() - t_first) * 1e6
    # (the rotation below is sized by the bytes of the kernel the TIMED loop runs: with the index built off the query path the
    # first evaluation was answered by k_like_lean, whose 26.5 MB would size the cycle 20 % too short for k_like_flat's 20.8 MB)
    scan.index_wait()
    scan.eval(expr, mask.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.current_stream().synchronize()
    # The timed loop ROTATES through several resident tables of the same shape (other seeds), one per scan: a hot-cache
    # query never finds its column in the 256 MiB memory-side Infinity Cache, and back-to-back passes over ONE 40-160 MB
    # column would (round 2: 27.9 us hot vs 35.9 us cold).  The cycle is sized so that the bytes it READS (the kernel's own
    # bytes minus the mask words it writes, when it writes them) exceed 2.2 x 256 MiB: when a table comes round again,
    # more than twice the cache's capacity of other lines has passed through it (round 4's cap of 8 tables stopped
    # guaranteeing that once the kernel's own bytes fell to 33 MB).
    sparse_count = world == 1 or args.exchange == "count"  # COUNT(*) consumers take no mask (d_mask_out = NULL)
    _, own0 = scan.traffic_model(expr, False, no_mask=sparse_count)
    _, own0_mask = scan.traffic_model(expr, False)
    # (kernels that cannot skip the mask words are charged for them either way)
    read0 = max(int(own0) - (int(words) * 8 if int(own0) == int(own0_mask) else 0), 1)
    n_rot = args.rotate if args.rotate > 0 else max(1, min(32, -(-int(2.2 * (256 << 20)) // read0)))
    if args.rotate <= 0 and args.workload == "url_like":
        # every resident URL table takes ~1 GB of Liquid bytes + ~2 GB of scan-level index: leave room for the secondaries
        free_b, _tot = torch.cuda.mem_get_info()
        n_rot = max(1, min(n_rot, int(free_b * 0.5) // (3 << 30)))
    t_rot = time.perf_counter()
    scans = [scan]
    rot_ids = [ids]
    for r in range(1, n_rot):
        a2 = copy.copy(args)
        a2.seed = args.seed + 7919 * r
        if args.workload == "url_like":
            ids_r = stage_url_column(cache, lc, N, a2, rank, n_batches, threads,
                                     file_id=1000 + 16 * r + (0 if args.batch0 is not None else rank))
        else:
            ids_r = stage_int_column(cache, lc, N, a2, rank, args.rows, threads, base=base, kind=args.int_kind, col=200 + r)
        scans.append(cache.scan(ids_r))
        rot_ids.append(ids_r)
    t_stage += time.perf_counter() - t_rot
    # COUNT(*) partials: written by the predicate kernel itself (lc_scan_eval_count); two buffers so that the all-reduce
    # of step i (RCCL's own stream) overlaps the scan of step i+1
    from liquid_cache_amd.sharding import (Communicator, PipelinedAbiCountAllReduce, PipelinedCountAllReduce,
                                           all_gather_mask_segments)
    comm = None
    comm_used = "torch.distributed"
    if args.comm == "abi" and world > 1:
        comm, comm_used = make_abi_communicator(cache, N, Communicator, rank, world, torch, dist, bool(test_backend))
    make_total = lambda: torch.zeros((), dtype=torch.int64, device="cuda")  # noqa: E731
    reducer = PipelinedAbiCountAllReduce(make_total, comm, torch) if comm else PipelinedCountAllReduce(make_total, world)
    stream = torch.cuda.current_stream().cuda_stream
    gathered = [None]
    step_no = [0]
    words_per_rank = None
    if comm and args.exchange == "mask":
        wl = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
        dist.all_gather(wl, torch.tensor([words], dtype=torch.int64, device="cuda"))
        words_per_rank = [int(x.item()) for x in wl]
        gathered[0] = torch.zeros(max(sum(words_per_rank), 1), dtype=torch.int64, device="cuda")

    # ONE STEP = `sps` scans: each scan is one pass of the predicate over ONE resident table (--rows rows per rank), COUNT(*)
    # included, followed by its exchange step; consecutive scans take consecutive tables of the rotation.  (Round 4 timed 20
    # single scans = 0.33 ms, where one scheduler hiccup moves the result by tens of percent.)
    sps = args.scans_per_step if args.scans_per_step > 0 else 256
    want_mask = args.exchange == "mask" and world > 1
    mask_ptr = mask.data_ptr() if (want_mask or not sparse_count) else 0

    def one_scan():
        total = reducer.acquire()
        sc = scans[step_no[0] % n_rot]
        step_no[0] += 1
        # COUNT(*) of this shard from the predicate kernel itself; the hit mask only when the exchange step consumes it
        sc.eval_count(expr, mask_ptr, total.data_ptr(), 0, 0, stream)
        reducer.submit()  # exchange step of COUNT(*) queries: the partial counts -> global count (8 bytes)
        if want_mask:
            # exchange step of mask consumers: the per-rank segments -> one BooleanArray (row-range shards concatenate)
            if comm:
                comm.allgather_mask(mask.data_ptr(), words, gathered[0].data_ptr(), words_per_rank, stream)
            else:
                gathered[0] = all_gather_mask_segments(mask)

    def step():
        for _ in range(sps):
            one_scan()

    drain = reducer.drain

    for _ in range(n_rot):  # every table is scanned once before the clock starts (plans, automata, scan-level index)
        one_scan()
    drain()
    # the scan-level indexes are built off the query path (LC_OPT_LIKE_INDEX_ASYNC): the timed loop is the HOT-cache steady
    # state, so the builds this first pass kicked off are waited for (what a first evaluation costs is reported on its own:
    # first_evaluation_us, clickbench_protocol_us)
    t_builds = time.perf_counter()
    for sc in scans:
        sc.index_wait()
    index_builds_wait_ms = (time.perf_counter() - t_builds) * 1e3
    warm_steps = max(args.warmup, 1)
    for _ in range(warm_steps):
        step()
    drain()
    new_needle_us = None
    if args.workload == "url_like" and rank == 0:
        # ... and a needle this scan has not seen (its index exists): automata + plan + one evaluation
        e_new = lc.LiquidExpr.try_new("like", b"%yahoo%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
        torch.cuda.synchronize()
        t_new = time.perf_counter()
        scan.eval(e_new, mask.data_ptr(), 0, 0, stream)
        torch.cuda.synchronize()
        new_needle_us = (time.perf_counter() - t_new) * 1e6
    step_no[0] = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream around the SAME K steps: the device time of the timed region, from which the roofline's
    # per-launch kernel duration is taken (round 5's line quoted a flushed single-launch figure that, times the launches of a
    # step, exceeded the step)
    ev_loop0, ev_loop1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_loop0.record()
    for _ in range(args.steps):
        step()
    ev_loop1.record()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    loop_launch_ms = ev_loop0.elapsed_time(ev_loop1) / max(args.steps * sps, 1)  # per scan, launch gaps of one stream included
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    rows_t = torch.tensor([scan.rows], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(rows_t, op=dist.ReduceOp.SUM)
    rows_all = int(rows_t.item())
    step_no[0] = 0  # COUNT(*) of table 0 (the one the CPU oracle checks), through the same call
    one_scan()
    drain()
    torch.cuda.synchronize()
    hits = int(reducer.last().item())
    # the fused total must be the sum of the per-entry counts of the same predicate (separate launch, plain reduction)
    scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
    if os.environ.get("LC_DUMP_COUNTS"):  # kernel-instrumentation aid (profiling builds: make TIMING=1 / ABLATION=1)
        np.save(os.environ["LC_DUMP_COUNTS"], counts.cpu().numpy())
    local_hits = int(counts.sum(dtype=torch.int64).item())
    lh = torch.tensor([local_hits], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(lh, op=dist.ReduceOp.SUM)
    assert int(lh.item()) == hits, "fused COUNT(*) %d != sum of per-entry counts %d" % (hits, int(lh.item()))
    # the same for every column of the rotation (the timed loop scans them all): fused COUNT(*) == sum of per-entry counts
    rot_hits, rot_counts = [hits], [counts.cpu().numpy().astype(np.int64)]
    for r in range(1, n_rot):
        tot_r = torch.zeros((), dtype=torch.int64, device="cuda")
        scans[r].eval_count(expr, mask.data_ptr(), tot_r.data_ptr(), 0, 0, stream)
        scans[r].eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)
        torch.cuda.synchronize()
        c_r = counts.cpu().numpy().astype(np.int64)[: scans[r].entries]
        assert int(tot_r.item()) == int(c_r.sum()), "rotating column %d: fused COUNT(*) %d != per-entry sum %d" % (
            r, int(tot_r.item()), int(c_r.sum()))
        rot_hits.append(int(tot_r.item()))
        rot_counts.append(c_r)
    scan.eval(expr, mask.data_ptr(), 0, counts.data_ptr(), stream)  # (the mask / counts buffers hold column 0 again)
    torch.cuda.synchronize()
    steady_build_ms = None
    if n_rot > 1 and args.workload == "url_like":
        try:  # the index build of the rotation's last table: a build that is not the process's first (no kernel code load)
            steady_build_ms = round(float(scans[n_rot - 1].info().index_build_ms), 3)
        except Exception:  # noqa: BLE001
            steady_build_ms = None
    concurrent = None
    if rank == 0 and world == 1 and args.workload == "url_like" and n_rot >= 4 and not args.no_secondary and (
            args.secondary_set == "all" or "concurrent" in args.secondary_set.split(",")):
        try:  # (while the rotation's tables are resident)
            nt = min(8, n_rot)
            concurrent = secondary_concurrent_tables(cache, N, args, rot_ids[:nt], expr, sum(rot_hits[:nt]), int(scan.rows))
        except Exception as e:  # noqa: BLE001
            concurrent = {"error": "%s: %s" % (type(e).__name__, e)}
    for r in range(1, n_rot):  # the other tables of the rotation have done their work
        scans[r].close()
    like_stream = None
    if rank == 0 and world == 1 and args.workload == "url_like" and n_rot >= 7 and not args.no_secondary and not args.no_fingerprints \
            and not args.no_signatures and "k_like_flat" in scan.explain(expr):
        try:  # (while six of them are still resident)
            inf0 = scan.info()
            like_stream = secondary_like_stream(cache, lc, N, args, rot_ids[1:7], expr, rot_hits[1:7], torch, stream,
                                                int(inf0.index_bytes), 1)
        except Exception as e:  # noqa: BLE001
            like_stream = {"error": "%s: %s" % (type(e).__name__, e)}
    for r in range(1, n_rot):  # ... their HBM goes to the secondaries
        cache.evict(rot_ids[r])
    del scans[1:]
    if args.exchange == "mask" and world > 1:
        assert (int(gathered[0].numel()) if comm else sum(int(x.numel()) for x in gathered[0])) * 64 >= rows_all

    # roofline of the dominant kernel: HIP events on the launch stream, same launches as the timed region
    # (the same call as the timed loop's: no mask output when the loop asks for none)
    alg_bytes, own_bytes = scan.traffic_model(expr, False, no_mask=not mask_ptr)
    iters = max(5, args.steps)
    kernel_ms = scan.eval_timed(expr, mask_ptr, max(iters, 50), 0, counts.data_ptr(), stream)
    cold_ms = None if args.no_cold else scan.eval_timed_cold(expr, mask_ptr, max(5, iters // 2), FLUSH_BYTES, 0,
                                                             counts.data_ptr(), stream)
    with_mask = None
    if not mask_ptr and not args.no_cold:
        # the same predicate with the hit MASK written (what a mask consumer — a conjunction's next step — asks for)
        _, own_m = scan.traffic_model(expr, False)
        hot_m = scan.eval_timed(expr, mask.data_ptr(), max(iters, 50), 0, counts.data_ptr(), stream)
        cold_m = scan.eval_timed_cold(expr, mask.data_ptr(), max(5, iters // 2), FLUSH_BYTES, 0, counts.data_ptr(), stream)
        with_mask = {"kernel_ms_hot": hot_m, "kernel_ms_l3_cold": cold_m, "own_bytes": int(own_m),
                     "frac_l3_cold": own_m / (cold_m * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_hot": own_m / (hot_m * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # multi-GPU: what bounds a step — this rank's kernel against the exchange (8-byte all-reduce, overlapped with the next
    # scan by the pipelined reducer): kernel time of the slowest / fastest rank and the bare collective, measured here
    kt = torch.tensor([kernel_ms, -kernel_ms], dtype=torch.float64, device="cuda")
    exchange_us = None
    if world > 1:
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        xbuf = torch.zeros((), dtype=torch.int64, device="cuda")
        for _ in range(5):
            dist.all_reduce(xbuf)
        torch.cuda.synchronize()
        tx = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(xbuf)
        torch.cuda.synchronize()
        exchange_us = (time.perf_counter() - tx) / 50 * 1e6

    out = None
    if rank == 0:
        if args.workload == "url_like":
            tkey = "url_like_no_fingerprints" if args.no_fingerprints else (
                "url_like_no_signatures" if args.no_signatures else "url_like")
        else:
            tkey = "%s_gt_w%d" % (args.int_kind, args.int_bits)
        path = scan.explain(expr)
        if path.startswith("k_like_"):
            kernel = path.split(":")[0].split(" (")[0]
        traffic, traffic_src = measured_traffic(tkey, kernel)
        step_s = elapsed / args.steps / sps  # seconds per scan of one table
        out = {
            "metric": "filtered rows/s (+ GB/s scanned), ClickBench Q21 hot cache",
            "value": rows_all * sps / elapsed * args.steps,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "timed_region_s": elapsed,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "rows_per_gpu": int(scan.rows), "rows_all_gpus": rows_all,
                       "batch_rows": args.batch_size, "batches_per_gpu": int(scan.entries),
                       "distinct_per_batch": args.uniques,
                       "parallelism": "%s x%d, %s per step" % (
                           ("ONE %d-row table split into contiguous row-range shards (strong scaling)" % rows_all)
                           if scaling == "strong" else "a %d-row table per GPU (weak scaling)" % int(scan.rows), world,
                           "8-byte COUNT(*) all-reduce (overlapped with the next scan)" if args.exchange == "count"
                           else "COUNT(*) all-reduce + all-gather of the hit-mask segments"),
                       "predicate": ("URL LIKE '%%%s%%'" % args.needle) if args.workload == "url_like" else "col > literal",
                       "exchange_by": comm_used,
                       "hits": hits, "stage_seconds": round(t_stage, 2), "rotating_columns": n_rot,
                       "rotating_columns_hits": rot_hits,
                       "rotating_columns_count_equals_entry_counts": True,
                       "warmup_steps_run": warm_steps, "scans_per_step": sps, "us_per_scan": step_s * 1e6,
                       "cycle_read_bytes": int(read0) * n_rot,
                       "mask_written_in_timed_loop": bool(mask_ptr),
                       "step": "%d scans; scan i is one pass of the predicate (+ COUNT(*)) over resident table i %% %d of %d "
                               "rows: the cycle reads %.0f MB = %.1f x the 256 MiB Infinity Cache (L3-cold)" % (
                                   sps, n_rot, int(scan.rows), read0 * n_rot / 1e6, read0 * n_rot / (256 << 20))},
            # the bytes THIS implementation moves per second of wall clock (lc_scan_traffic_model's own-bytes figure) ...
            "gb_per_s_scanned": own_bytes * world / step_s / 1e9,
            # ... and what the scan is worth to the query: the reference algorithm's bytes per second of wall
            # clock — an equivalence, not a bandwidth (it exceeds the HBM peak where the index spares the kernel bytes)
            "reference_algorithm_equivalent_gbs": alg_bytes * world / step_s / 1e9,
            "first_evaluation_us": round(first_eval_us, 1),
            "new_needle_first_evaluation_us": None if new_needle_us is None else round(new_needle_us, 1),
            "roofline": roofline(kernel, kernel_ms, alg_bytes, own_bytes, cold_ms, traffic, traffic_src),
        }
        # The headline fraction is quoted on the timed region's OWN per-launch time (HIP events over the K steps: every launch
        # finds its table L3-cold thanks to the rotation, and the figure includes what a launch costs a stream): bytes per
        # launch / that.  The flushed single-launch and back-to-back figures stay beside it.
        rf = out["roofline"]
        if n_rot > 1 and not args.no_cold:
            rf["kernel_ms_flushed_single_launch"] = rf["kernel_ms"]
            rf["frac_flushed_single_launch"] = rf["frac"]
            rf["kernel_ms"] = loop_launch_ms
            rf["timing"] = "timed_region_hip_events_l3_cold_rotation"
            rf["achieved"] = own_bytes / (loop_launch_ms * 1e-3) / 1e9
            rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
            rf["reference_algorithm_equivalent_gbs"] = alg_bytes / (loop_launch_ms * 1e-3) / 1e9
            if rf.get("traffic"):
                rf["traffic_gbs"] = rf["traffic"] / (loop_launch_ms * 1e-3) / 1e9
                rf["frac_by_traffic"] = rf["traffic_gbs"] / HBM_PEAK_GBS
        if with_mask:
            out["same_predicate_with_mask_output"] = with_mask
        if world > 1:
            out["scaling_model"] = {
                "kernel_us_slowest_rank": float(kt[0].item()) * 1e3, "kernel_us_fastest_rank": -float(kt[1].item()) * 1e3,
                "exchange_us": exchange_us,
                "step_us": ms_per_step * 1e3,
                "note": "kernel = HIP events on rank-local launches (hot); exchange = one 8-byte all-reduce alone; the "
                        "pipelined reducer overlaps the exchange of step i with the scan of step i+1, so a step costs "
                        "max(kernel, exchange) plus launch gaps; below ~1,000 entries per GPU the kernel sits on its "
                        "launch floor (~5 us) and the curve flattens"}
        if world == 1 and not args.no_secondary:
            add_read_probe(out["roofline"], cache, N)  # (skipped in the profiling runs: their traces hold the scan kernels only)
        out["config"]["evaluation_path"] = path
        out["config"]["first_evaluation_us"] = out["first_evaluation_us"]
        inf = scan.info()
        out["config"]["index_bytes"] = int(inf.index_bytes) + int(inf.unigram_index_bytes)
        out["config"]["index_build_ms"] = round(float(inf.index_build_ms), 3)  # (the first build of the process: + kernel code load)
        if steady_build_ms:
            out["config"]["index_build_ms_steady"] = steady_build_ms
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        n_sample = args.cpu_batches or n_batches  # ~4 s (LIKE) / ~1 s (int) of single-thread CPU work at 100 M rows
        if args.workload == "url_like":
            extra = tuple((op, ("%" + n + "%").encode()) for _, op, n in NEEDLE_CLASSES if (op, n) != ("like", args.needle)) \
                if n_sample == n_batches and not args.no_fingerprints else ()
            v, rows_s, hits_s, dt, all_cores = cpu_baseline_url(cache, lc, N, args, rank, n_sample, pattern, threads, extra)
        else:
            v, rows_s, hits_s, dt, all_cores = cpu_baseline_int(cache, lc, N, args, rank, n_sample, literal, base)
        out["cpu_baseline"] = {"value": v, "unit": "rows/s", "cores": 1, "kind": "port", "hits": hits_s,
                               "sample": "first %d batches (%d rows) of the same column, %.1f s" % (n_sample, rows_s, dt)}
        out["cpu_baseline_all_cores"] = all_cores
        out["config"]["cpu_hits"] = hits_s
        if n_sample == n_batches:
            # the CPU restatement of the reference algorithm and the GPU scan saw the same bytes: same COUNT(*)
            assert hits_s == hits == all_cores["hits"], "GPU hits %d != CPU oracle hits %d" % (hits, hits_s)
            out["config"]["hits_match_cpu_oracle"] = True
            checks = all_cores.pop("_checks", None)
            if checks:
                # bit-for-bit: the GPU's hit mask and per-entry counts against the oracle's, over the WHOLE column, for
                # the headline needle and the extra ones (scalars only, so that they survive into the driver's record)
                import pyarrow as pa
                n_ok = n_bad = 0
                worst = None
                classes = {}
                labels = {(op, "%" + n + "%"): lab for lab, op, n in NEEDLE_CLASSES}
                hint = lc.CacheExpression.SUBSTRING_SEARCH
                for (op, pat), (cpu_total, cpu_mask, cpu_counts, cpu_s) in checks.items():
                    e2 = lc.LiquidExpr.try_new(op, pat.encode(), pa.string(), hint)
                    mask.zero_()
                    scan.eval(e2, mask.data_ptr(), 0, counts.data_ptr(), stream)
                    torch.cuda.synchronize()
                    g_mask = mask.cpu().numpy().view(np.uint64)[: cpu_mask.size]
                    g_counts = counts.cpu().numpy().view(np.uint32)[: cpu_counts.size]
                    ok = bool(np.array_equal(g_mask, cpu_mask)) and bool(np.array_equal(g_counts, cpu_counts))
                    n_ok += ok
                    n_bad += not ok
                    if not ok and worst is None:
                        worst = "%s %s: gpu %d cpu %d, %d entries differ" % (op, pat, int(g_counts.sum()), cpu_total,
                                                                             int((g_counts != cpu_counts).sum()))
                    lab = labels.get((op, pat))
                    if lab and not args.no_needle_classes:
                        it = max(3, iters // 4)
                        hot = scan.eval_timed(e2, mask.data_ptr(), it, 0, counts.data_ptr(), stream)
                        cold = None if args.no_cold else scan.eval_timed_cold(e2, mask.data_ptr(), 3, FLUSH_BYTES, 0,
                                                                              counts.data_ptr(), stream)
                        classes[lab] = {"predicate": "URL %s '%s'" % ("LIKE" if op == "like" else "NOT LIKE", pat),
                                        "needle_bytes": len(pat) - 2, "hits": int(g_counts.sum()),
                                        "hit_fraction": float(g_counts.sum()) / max(int(scan.rows), 1),
                                        "mask_equals_cpu_oracle": ok, "kernel_us_hot": round(hot * 1e3, 2),
                                        "kernel_us_l3_cold": None if cold is None else round(cold * 1e3, 2),
                                        "rows_per_s_l3_cold": None if cold is None else scan.rows / (cold * 1e-3),
                                        "cpu_oracle_all_cores_s": round(cpu_s, 3), "path": scan.explain(e2)}
                out["config"]["needles_checked"] = n_ok + n_bad
                out["config"]["needle_masks_all_match_cpu_oracle"] = n_bad == 0
                out["config"]["needle_entry_counts_all_match_cpu_oracle"] = n_bad == 0
                if worst:
                    out["config"]["needle_first_mismatch"] = worst
                if classes:
                    out["like_needle_classes"] = classes
                assert n_bad == 0, "GPU mask differs from the CPU oracle's: " + str(worst)
        all_cores.pop("_checks", None)
        if args.workload == "url_like" and n_rot > 1 and not args.no_fingerprints:
            # the other columns of the rotation (timed, too): per-entry counts of a spread sample of their batches against
            # the oracle's over the same bytes
            sample = sorted(set(range(0, n_batches, max(1, n_batches // 96))) | {n_batches - 1})
            ok_cols = 0
            for r in sorted(set(range(1, n_rot, max(1, (n_rot - 1) // 7)))):  # (<= 8 of them: each costs ~0.5 s of CPU)
                a2 = copy.copy(args)
                a2.seed = args.seed + 7919 * r
                want = oracle_url_sample_counts(cache, lc, N, a2, rank, 1000 + 16 * r + rank, sample, pattern, threads)
                got = rot_counts[r][sample]
                assert np.array_equal(got, want), "rotating column %d: per-entry counts differ from the oracle's on %d of %d " \
                    "sampled batches" % (r, int((got != want).sum()), len(sample))
                ok_cols += 1
            out["config"]["rotating_columns_checked_against_oracle"] = ok_cols
            out["config"]["rotating_columns_oracle_sample"] = "%d batches of each (every %dth + the last)" % (
                len(sample), max(1, n_batches // 96))

    if rank == 0 and world == 1 and not args.no_secondary:
        sec = {}
        if like_stream is not None:
            sec["mixed_table_like_stream"] = like_stream
        if concurrent is not None:
            sec["concurrent_table_scans"] = concurrent
        sec_rows = args.secondary_rows or args.rows
        groups = set(args.secondary_set.split(","))
        want = lambda g: "all" in groups or g in groups  # noqa: E731
        if args.workload == "url_like" and not args.no_q21 and want("q21"):
            try:
                sec["q21_pipeline"] = q21_pipeline(cache, lc, N, args, rank, n_batches, threads, scan, expr, torch, stream)
            except Exception as e:  # noqa: BLE001
                sec["q21_pipeline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.workload == "url_like" and want("rowgroup"):
            try:
                sec["rowgroup_granularity"] = secondary_rowgroup(cache, lc, N, args, ids, expr, hits, int(scan.rows))
            except Exception as e:  # noqa: BLE001
                sec["rowgroup_granularity"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if want("int"):
            sec.update(secondary_int_columns(cache, lc, N, args, sec_rows, threads, torch, stream, iters))
        if want("micro"):
            try:
                sec["micro"] = secondary_micro(cache, lc, N, args, sec_rows, threads, torch, stream, max(5, iters // 2),
                                               url_scan=scan if args.workload == "url_like" else None)
            except Exception as e:  # noqa: BLE001
                sec["micro"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            if want("q6"):
                sec["tpch_q6_pushdown"] = secondary_tpch_q6(cache, lc, N, args, sec_rows, threads, torch, stream, iters)
        except Exception as e:  # noqa: BLE001
            sec["tpch_q6_pushdown"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            sweep_rows = min(sec_rows, args.sweep_rows) if args.sweep_rows else sec_rows
            if want("sweep"):
                sec["clickbench_pushdown_sweep"] = secondary_clickbench_sweep(cache, lc, args, sweep_rows, threads, torch, stream,
                                                                          max(3, iters // 2))
        except Exception as e:  # noqa: BLE001
            sec["clickbench_pushdown_sweep"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            if want("staging"):
                sec["arrow_to_liquid_staging"] = secondary_transcode_rate(cache, lc, N, args, sec_rows, threads)
        except Exception as e:  # noqa: BLE001
            sec["arrow_to_liquid_staging"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.workload == "url_like" and not args.no_fingerprints and want("like"):
            sec.update(secondary_like_variants(lc, N, args, rank, n_batches, threads, torch, stream, iters, pattern))
        out["secondary"] = sec
    if args.workload == "url_like" and rank == 0:
        # a NEW scan over the same entries (a host that creates a scan per query): the scan-level index and the plans of the
        # scan just destroyed are adopted, so its first evaluation is records + automata + one launch
        scan.close()
        ids_np0 = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
        torch.cuda.synchronize()
        t_next = time.perf_counter()
        scan = cache.scan(ids_np0)  # (round 6: creation is timed too — the context's scan cache hands the kept scan back)
        scan.eval(expr, mask.data_ptr(), 0, 0, stream)
        torch.cuda.current_stream().synchronize()
        out["next_scan_first_evaluation_us"] = round((time.perf_counter() - t_next) * 1e6, 1)
        out["config"]["next_scan_first_evaluation_us"] = out["next_scan_first_evaluation_us"]
    if args.workload == "url_like" and rank == 0 and world == 1 and not args.no_secondary and not args.no_fingerprints:
        try:
            proto = clickbench_protocol(lc, N, args, rank, n_batches, threads, expr, hits, torch, stream, local_rank)
            out["clickbench_protocol"] = proto
            out["config"]["clickbench_protocol_us"] = {"run1": proto["run1_us"], "mean_last3": proto["mean_last3_us"],
                                                       "index_in_place_mean_last3": proto["index_in_place_mean_last3_us"]}
        except Exception as e:  # noqa: BLE001
            out["clickbench_protocol"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        out["config"]["index_builds_wait_ms"] = round(index_builds_wait_ms, 2)
        emit(out, args)
    scan.close()
    cache.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
