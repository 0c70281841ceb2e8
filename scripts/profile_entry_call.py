"""Where one per-entry drop-in call (lc_eval_predicate, host buffers out) spends its time: calls it over the entries of a
small URL column through the library LC_LIB_PATH names.  With a `make VARIANT=prof EXTRA=-DLC_CALL_PROFILE` build the
library prints its own per-phase split at exit; this script adds the wall time per call as ctypes sees it.
usage: LC_LIB_PATH=liquid_cache_amd/variants/libliquid_cache_amd_prof.so python scripts/profile_entry_call.py [--rows N] [--rounds R]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--needle", default="google")
    a = ap.parse_args()
    args = bench.parse_args(["--rows", str(a.rows), "--needle", a.needle])
    import pyarrow as pa
    import liquid_cache_amd as lc
    from liquid_cache_amd import _native as N
    cache = lc.LiquidCacheBuilder.new().build()
    n_batches = (args.rows + args.batch_size - 1) // args.batch_size
    ids = [int(i) for i in bench.stage_url_column(cache, lc, N, args, 0, n_batches, 16)]
    expr = lc.LiquidExpr.try_new("like", ("%" + a.needle + "%").encode(), pa.string(), lc.CacheExpression.SUBSTRING_SEARCH)
    pred = expr.as_predicate()
    nb = (args.batch_size + 7) // 8 + 8
    values, validity = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    out_len, nullable = C.c_uint32(), C.c_int32()
    vp, qp = values.ctypes.data_as(C.c_void_p), validity.ctypes.data_as(C.c_void_p)
    fn = cache._lib.lc_eval_predicate

    def one_round():
        for e in ids:
            st = fn(cache._ctx, e, C.byref(pred), None, vp, qp, C.byref(out_len), C.byref(nullable))
            assert st == 0, st

    one_round()
    t0 = time.perf_counter()
    for _ in range(a.rounds):
        one_round()
    dt = time.perf_counter() - t0
    print("lc_eval_predicate through ctypes: %.2f us per call (%d entries x %d rounds)" % (dt / (len(ids) * a.rounds) * 1e6, len(ids), a.rounds),
          flush=True)
    cache.close()


if __name__ == "__main__":
    main()
