import os, sys
import numpy as np
import pyarrow as pa
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import liquid_cache_amd as lc
import liquid_oracle as lo
from test_gpu_parity import _make_strings

rng = np.random.default_rng(4242)
hint = lc.CacheExpression.SUBSTRING_SEARCH
for lens in ([8192, 8192, 8192, 8192], [8192, 8191, 65, 1, 9000, 8192]):
    blobs, st = [], None
    for b, n in enumerate(lens):
        strs = _make_strings(rng, n, 1500, b != 5)
        if n > 100:
            strs[7] = "http://needle-once.example/only" + str(b)
        liquid, st = lo.encode_byte_view(strs, st=st, fingerprints=True)
        blobs.append(liquid)
    for mode in ("0", "1"):
        os.environ["LC_NO_POSTINGS"] = mode
        cache = lc.LiquidCacheBuilder.new().with_device(0).build()
        cache.set_symbol_table(7171, lo.symtab_bytes(st))
        ids = [lc.ParquetArrayID.new(71, 0, 2, b) for b in range(len(lens))]
        cache.stage(ids, blobs, [7171] * len(ids))
        scan = cache.scan(ids)
        for pat in (b"%needle-once%", b"%google%", b"%only%"):
            e = lc.LiquidExpr.try_new("like", pat, pa.string(), hint)
            print(lens, "nopost", mode, pat, scan.traffic_model(e), [cache.entry_info(i).device_bytes for i in ids][:3])
        scan.close()
        cache.close()
