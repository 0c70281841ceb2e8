"""Where the first evaluation of `URL LIKE '%google%'` on a fresh scan goes (host wall clock, phases printed by a library built
with `make VARIANT=trace EXTRA=-DLC_TRACE_PHASES` when LC_LIB_PATH points at it): scan creation, first / second / third
evaluation, destroy and re-create through the scan cache — on a FRESH context (cold pools) and again on a second table of the
same context (warm pools).  usage: python scripts/first_eval_profile.py [--rows N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=99_997_497)
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows)])
    import copy
    import pyarrow as pa
    import torch
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    global T0
    T0 = time.perf_counter()
    stream = torch.cuda.current_stream().cuda_stream
    n_batches = (a.rows + args.batch_size - 1) // args.batch_size
    expr = lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    total = torch.zeros((), dtype=torch.int64, device="cuda")
    for table in range(2):
        a2 = copy.copy(args)
        a2.seed = args.seed + 7919 * table
        ids = bench.stage_url_column(cache, lc, N, a2, 0, n_batches, 16, file_id=300 + table)
        ids_np = np.ascontiguousarray(np.asarray([int(e) for e in ids], dtype=np.uint64))
        torch.cuda.synchronize()
        print("==== table %d (%s pools)" % (table, "cold" if table == 0 else "warm"), file=sys.stderr, flush=True)
        for run in range(4):
            t0 = time.perf_counter()
            sc = cache.scan(ids_np)
            t1 = time.perf_counter()
            sc.eval_count(expr, 0, total.data_ptr(), 0, 0, stream)
            t2 = time.perf_counter()
            got = int(total.item())  # (the query's stream only: a device-wide synchronise would wait for the builder's stream)
            t3 = time.perf_counter()
            k = N.LIKE_KERNEL_NAMES.get(int(sc.info().last_like_kernel))
            sc.close()
            t4 = time.perf_counter()
            print("---- (now %.1f us) run %d: create %.1f us, eval call %.1f us, wait+read %.1f us, close %.1f us, total %.1f us; %s; count %d" % (
                (time.perf_counter() - T0) * 1e6, run, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6, k, got), file=sys.stderr, flush=True)
            if run == 0:
                time.sleep(0.2)  # (let the builder finish: run 1 then shows the switch to the scan-level index)
    cache.close()


if __name__ == "__main__":
    main()
