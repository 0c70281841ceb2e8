"""TPC-H Q6-shaped pushdown on the device: four chained conjuncts + one more over three columns of
a lineitem-shaped table, every mask the selection of the next predicate (LiquidCacheReader::build_predicate_filter).

    l_shipdate >= DATE '1994-01-01' AND l_shipdate < DATE '1995-01-01'
    AND l_discount >= 0.05 AND l_discount <= 0.07 AND l_quantity < 24

l_shipdate: Date32 uniform over [1992-01-02, 1998-12-01] (W=12); l_discount: Decimal128(15,2) in {0.00..0.10} (W=4);
l_quantity: Decimal128(15,2) in {1..50}.00 (W=13).  The expected COUNT(*) is accumulated with numpy while staging and
must match the device count exactly.  Prints one JSON line.  usage: python scripts/tpch_q6.py [--rows N] [--iters K]
"""
import argparse
import datetime
import decimal
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pyarrow as pa
import torch  # first: its HIP runtime must be the one the process initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liquid_cache_amd as lc  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=600_037_902)   # lineitem at SF100
p.add_argument("--iters", type=int, default=10)
p.add_argument("--batch-size", type=int, default=8192)
args = p.parse_args()
bs, rows = args.batch_size, args.rows
n_batches = (rows + bs - 1) // bs
epoch = datetime.date(1970, 1, 1)
d_lo, d_hi = (datetime.date(1992, 1, 2) - epoch).days, (datetime.date(1998, 12, 1) - epoch).days
d1, d2 = (datetime.date(1994, 1, 1) - epoch).days, (datetime.date(1995, 1, 1) - epoch).days
DEC = pa.decimal128(15, 2)


def dec_array(unscaled: np.ndarray) -> pa.Array:
    buf = np.zeros((len(unscaled), 2), np.int64)
    buf[:, 0] = unscaled
    return pa.Array.from_buffers(DEC, len(unscaled), [None, pa.py_buffer(buf)])


cache = lc.LiquidCacheBuilder.new().with_batch_size(bs).build()
ids = {c: [lc.ParquetArrayID.new(0, b // 54, c, b % 54) for b in range(n_batches)] for c in (10, 6, 4)}
threads = max(1, min(32, os.cpu_count() or 8))
expected = np.zeros(n_batches, np.int64)


def stage(chunk):
    rng = np.random.default_rng(1000 + chunk)
    for b in range(chunk, n_batches, threads):
        n = min(bs, rows - b * bs)
        ship = rng.integers(d_lo, d_hi + 1, size=n, dtype=np.int32)
        disc = rng.integers(0, 11, size=n, dtype=np.int64)
        qty = rng.integers(1, 51, size=n, dtype=np.int64) * 100
        expected[b] = int(((ship >= d1) & (ship < d2) & (disc >= 5) & (disc <= 7) & (qty < 2400)).sum())
        cache.insert(ids[10][b], pa.array(ship, type=pa.date32()))
        cache.insert(ids[6][b], dec_array(disc))
        cache.insert(ids[4][b], dec_array(qty))


t0 = time.perf_counter()
with ThreadPoolExecutor(max_workers=threads) as ex:
    list(ex.map(stage, range(threads)))
t_stage = time.perf_counter() - t0
s_ship, s_disc, s_qty = cache.scan(ids[10]), cache.scan(ids[6]), cache.scan(ids[4])
words = int(s_ship.mask_words)
masks = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(2)]
counts = torch.zeros(s_ship.entries, dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
E = lc.LiquidExpr.try_new
conj = [(s_ship, E(">=", datetime.date(1994, 1, 1), pa.date32())), (s_ship, E("<", datetime.date(1995, 1, 1), pa.date32())),
        (s_disc, E(">=", decimal.Decimal("0.05"), DEC)), (s_disc, E("<=", decimal.Decimal("0.07"), DEC)),
        (s_qty, E("<", decimal.Decimal("24.00"), DEC))]
assert all(e is not None for _, e in conj)


def run():
    sel = 0
    for i, (scan, expr) in enumerate(conj):
        out = masks[i & 1]
        scan.eval(expr, out.data_ptr(), sel, counts.data_ptr(), stream)
        sel = out.data_ptr()


for _ in range(2):
    run()
torch.cuda.synchronize()
got = counts.cpu().numpy().astype(np.int64)
assert got.tolist() == expected.tolist(), "device COUNT(*) per batch differs from numpy"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
widths = [12, 12, 4, 4, 13]
alg = sum(rows * w // 8 + (rows // 8 if i else 0) + rows // 8 for i, w in enumerate(widths))
print(json.dumps({"workload": "tpch_q6_pushdown", "rows": rows, "conjuncts": 5, "count": int(got.sum()),
                  "ms_per_pass": ms, "rows_per_s": rows / (ms * 1e-3), "algorithmic_bytes": alg,
                  "achieved_gbs": alg / (ms * 1e-3) / 1e9, "frac_of_8TBs": alg / (ms * 1e-3) / 1e9 / 8000.0,
                  "stage_seconds": round(t_stage, 1)}))
