"""PCIe-inclusive rate of the drop-in host-buffer boundary: lc_eval_predicate_batch over N entries with
host selections in and host BooleanArray buffers out, versus the device-resident scan of the same entries."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pyarrow as pa
import torch  # first: torch bundles its own HIP runtime, which must be the one the process initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liquid_cache_amd as lc  # noqa: E402
from liquid_cache_amd import _native as N  # noqa: E402

n_entries, rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 8192
L = N.load()
cache = lc.LiquidCacheBuilder.new().build()
rng = np.random.default_rng(0)
ids = []
buf = np.zeros(rows, np.int64)
for b in range(n_entries):
    L.lc_synth_int64_batch(7, b, rows, 62, 4_000_000_000_000_000_000 >> 2, buf.ctypes.data)
    eid = lc.ParquetArrayID.new(0, b // 54, 0, b % 54)
    cache.insert(eid, pa.array(buf))
    ids.append(int(eid))
expr = lc.LiquidExpr.try_new(">", (4_000_000_000_000_000_000 >> 2) + (1 << 61), pa.int64())
pred = expr.as_predicate()
sel = [np.packbits(rng.random(rows) < 0.5, bitorder="little") for _ in range(n_entries)]
outv = [np.zeros(rows // 8, np.uint8) for _ in range(n_entries)]
outn = [np.zeros(rows // 8, np.uint8) for _ in range(n_entries)]
vp = C.c_void_p
ids_a = (C.c_uint64 * n_entries)(*ids)
sel_a = (vp * n_entries)(*[s.ctypes.data for s in sel])
ov_a = (vp * n_entries)(*[o.ctypes.data for o in outv])
on_a = (vp * n_entries)(*[o.ctypes.data for o in outn])
lens = (C.c_uint32 * n_entries)()
nullable = (C.c_int32 * n_entries)()
st = (C.c_int32 * n_entries)()
for it in range(3):
    t0 = time.perf_counter()
    N.check(L.lc_eval_predicate_batch(cache.handle, n_entries, ids_a, C.byref(pred), sel_a, ov_a, on_a, lens, nullable, st),
            cache.handle)
    dt = time.perf_counter() - t0
print("host-buffer boundary (lc_eval_predicate_batch, %d entries x %d rows, selection in / BooleanArray out over PCIe): "
      "%.2f ms -> %.3g rows/s" % (n_entries, rows, dt * 1e3, n_entries * rows / dt))
# one call per entry, the way the reference's reader loop would call eval_predicate (liquid_cache_reader.rs:297-339)
k1 = min(n_entries, 512)
t0 = time.perf_counter()
for i in range(k1):
    one_len, one_null = C.c_uint32(), C.c_int32()
    N.check(L.lc_eval_predicate(cache.handle, ids[i], C.byref(pred), sel[i].ctypes.data_as(vp), outv[i].ctypes.data_as(vp),
                                outn[i].ctypes.data_as(vp), C.byref(one_len), C.byref(one_null)), cache.handle)
dt1 = time.perf_counter() - t0
print("one lc_eval_predicate call per entry: %.1f us per call -> %.3g rows/s" % (dt1 / k1 * 1e6, k1 * rows / dt1))
scan = cache.scan(ids)
mask = torch.zeros(int(scan.mask_words), dtype=torch.int64, device="cuda")
counts = torch.zeros(scan.entries, dtype=torch.int32, device="cuda")
ms = scan.eval_timed(expr, mask.data_ptr(), 20, 0, counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
print("device-resident scan of the same entries: %.3f ms -> %.3g rows/s" % (ms, n_entries * rows / (ms * 1e-3)))
