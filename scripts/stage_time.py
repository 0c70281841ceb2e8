import time, numpy as np, pyarrow as pa, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import liquid_cache_amd as lc
from liquid_cache_amd import _native as N
L = N.load()
rows = 8192
offs = np.zeros(rows + 1, np.int32); data = np.zeros(rows * 512, np.uint8)
arrs = []
for b in range(200):
    n = L.lc_synth_url_batch(42, b, rows, 2200, 159, offs.ctypes.data, data.ctypes.data, data.size)
    arrs.append(pa.StringArray.from_buffers(rows, pa.py_buffer(offs.copy()), pa.py_buffer(data[:n].copy())))
cache = lc.LiquidCacheBuilder.new().with_device(0).build()
hint = lc.CacheExpression.SUBSTRING_SEARCH
blobs = [cache.transcode(a, hint, path_id=7) for a in arrs[:1]]
t = time.perf_counter(); blobs = [cache.transcode(a, hint, path_id=7) for a in arrs]; t_tr = time.perf_counter() - t
ids = [lc.ParquetArrayID.new(1, 0, 3, b) for b in range(len(arrs))]
t = time.perf_counter()
for i, b in zip(ids, blobs): cache.stage([i], [b], [7])
t_st = time.perf_counter() - t
print("signatures=%s transcode %.2f ms/batch, stage %.2f ms/batch" % (os.environ.get("LC_NO_SIGNATURES", "0") in ("", "0"), t_tr / len(arrs) * 1e3, t_st / len(arrs) * 1e3))
ids2 = [lc.ParquetArrayID.new(2, 0, 3, b) for b in range(len(arrs))]
t = time.perf_counter()
cache.stage(ids2, blobs, [7] * len(blobs))
t_bulk = time.perf_counter() - t
print("bulk stage of %d entries in one call: %.2f ms/batch" % (len(blobs), t_bulk / len(blobs) * 1e3))
