"""Soak of the on-device byte-view transcoder against the host transcoder: random arrays (pool flavour, rows, distinct
values, null rate, slice, Arrow type incl. views, hint) in random groups per call; the Liquid bytes and the index blob of
every entry must be identical.  usage: python scripts/soak_bv_device.py [n_arrays] [seed]   (needs a GPU)"""
import os
import sys

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_data as fz  # noqa: E402
import liquid_cache_amd as lc  # noqa: E402

HINT = lc.CacheExpression.SUBSTRING_SEARCH


def make_array(rng):
    flavour = int(rng.integers(4))
    d = int(rng.choice([1, 2, 7, 60, 300, 1500, 4000]))
    pool = [fz._pool_urls, fz._pool_bytes, fz._pool_small_alphabet, fz._pool_escape_heavy][flavour](rng, d)
    n = int(rng.choice([0, 1, 5, 63, 64, 65, 1000, 4096, 8191, 8192, 12000]))
    rows = fz._zipf_rows(rng, pool, n) if rng.random() < 0.7 else [pool[int(k)] for k in rng.integers(len(pool), size=n)]
    null_rate = float(rng.choice([0.0, 0.0, 0.05, 0.5, 1.0]))
    vals = [None if rng.random() < null_rate else r for r in rows]
    is_text = flavour in (0, 2)
    t = rng.choice(["plain", "view"]) if True else "plain"
    if is_text and rng.random() < 0.6:
        typ = pa.string_view() if t == "view" else pa.string()
        vals = [None if v is None else v.decode("utf-8", "replace") for v in vals]
    else:
        typ = pa.binary_view() if t == "view" else pa.binary()
    arr = pa.array(vals, type=typ)
    if n > 10 and rng.random() < 0.3:
        a = int(rng.integers(0, n // 2))
        arr = arr.slice(a, int(rng.integers(1, n - a)))
    return arr


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    cache = lc.LiquidCacheBuilder.new().build()
    done = 0
    call = 0
    try:
        while done < total:
            m = int(rng.integers(1, 9))
            hint = HINT if rng.random() < 0.7 else None
            arrays = [make_array(rng) for _ in range(m)]
            path = 50_000 + call
            host = [lc.ParquetArrayID.new(1, call % 60000, 1, k) for k in range(m)]
            dev = [lc.ParquetArrayID.new(2, call % 60000, 1, k) for k in range(m)]
            device_first = rng.random() < 0.5   # whoever comes first trains the path's symbol table
            if device_first:
                cache.insert_device(dev, arrays, hint, path_ids=[path] * m)
            for e, a in zip(host, arrays):
                cache.insert(e, a, hint, path_id=path)
            if not device_first:
                cache.insert_device(dev, arrays, hint, path_ids=[path] * m)
            for k, (h, d, a) in enumerate(zip(host, dev, arrays)):
                hb, db = cache.entry_bytes(h), cache.entry_bytes(d)
                if hb != db or cache.entry_index_bytes(h) != cache.entry_index_bytes(d):
                    print("MISMATCH call %d array %d: type %s rows %d nulls %d hint %s device_first %s" % (
                        call, k, a.type, len(a), a.null_count, hint, device_first))
                    sys.exit(1)
                if len(a) and not cache.get(d).read().equals(a):
                    print("DECODE MISMATCH call %d array %d" % (call, k))
                    sys.exit(1)
            cache.evict(host + dev)
            done += m
            call += 1
        print("soak ok: %d arrays in %d calls, seed %d" % (done, call, seed))
    finally:
        cache.close()


if __name__ == "__main__":
    main()
