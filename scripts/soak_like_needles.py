"""Soak of the scan-level LIKE / NOT LIKE paths against plain Python truth (`needle in value`): a bench-shaped URL column,
many needles — 1 byte, 2 bytes, frequent ('/', 'http', 'ru/'), cut from the data at random lengths, absent — evaluated
through every index / kernel variant; per-row masks and per-entry counts must equal the truth.
usage: python scripts/soak_like_needles.py [batches] [needles] [seed]   (needs a GPU)"""
import os
import sys

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import liquid_cache_amd as lc  # noqa: E402
from liquid_cache_amd import _native as N  # noqa: E402

HINT = lc.CacheExpression.SUBSTRING_SEARCH
VARIANTS = {"default": {}, "no_signatures": dict(signatures=False), "no_row_lists": dict(row_lists=False),
            "k_str_pred_only": dict(like_path=1), "lean_every_needle": dict(like_pipeline_min_entries=1, like_path=3)}


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    n_needles = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rng = np.random.default_rng(seed)
    B = N.load_bench()
    bs = 8192
    offs = np.zeros(bs + 1, np.int32)
    data = np.zeros(bs * 512, np.uint8)
    batches, rows = [], []
    for b in range(n_batches):
        n = B.lc_synth_url_batch(seed, b, bs, 2200, 100, offs.ctypes.data, data.ctypes.data, data.size)
        raw = data[:n].tobytes()
        o = offs.copy()
        rows.append([raw[o[i]:o[i + 1]] for i in range(bs)])
        batches.append(pa.StringArray.from_buffers(bs, pa.py_buffer(o), pa.py_buffer(raw)))
    flat = [r for rr in rows for r in rr]
    needles = [b"q", b"/", b"z", b"ru", b"//", b"ru/", b"http", b".", b"google", b"mail", b"file", b"zzzzqqq", b"yandex.ru/search"]
    while len(needles) < n_needles:
        v = flat[int(rng.integers(len(flat)))]
        ln = int(rng.choice([1, 2, 3, 4, 6, 9, 15, 16, 17, 24, 33, 47, 48, 60]))
        if len(v) >= ln:
            a = int(rng.integers(0, len(v) - ln + 1))
            nd = v[a:a + ln]
            if not any(c in nd for c in b"%_\\"):
                needles.append(nd)
    truth = {}
    for nd in needles:
        hit = np.fromiter((nd in v for v in flat), dtype=np.bool_, count=len(flat))
        truth[nd] = hit
    checked = 0
    for name, opts in VARIANTS.items():
        cache = lc.LiquidCacheBuilder.new().with_index_options(**opts).build()
        try:
            ids = [lc.ParquetArrayID.new(3, b // 54, 13, b % 54) for b in range(n_batches)]
            for g in range(0, n_batches, 54):
                cache.insert_device(ids[g:g + 54], batches[g:g + 54], HINT)   # (entries transcoded on the device)
            scan = cache.scan(ids)
            for nd in needles:
                for op in ("like", "not_like"):
                    expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.string(), HINT)
                    for rep in range(2):  # (the second evaluation runs from the cached plan)
                        mask, counts = scan.eval_to_host(expr)
                        bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[: len(flat)].astype(np.bool_)
                        want = truth[nd] if op == "like" else ~truth[nd]
                        if not np.array_equal(bits, want) or not np.array_equal(counts, want.reshape(n_batches, bs).sum(axis=1)):
                            print("MISMATCH variant %s op %s needle %r rep %d: gpu %d truth %d" % (name, op, nd, rep, bits.sum(), want.sum()))
                            sys.exit(1)
                        checked += 1
            # comparisons against literals cut from the data (whole values, prefixes, a value plus a byte), and LIKE under a
            # random selection: the result is the selection AND the predicate
            import operator
            cmp_ops = {"==": operator.eq, "!=": operator.ne, "<": operator.lt, "<=": operator.le, ">": operator.gt, ">=": operator.ge}
            lits = []
            for _ in range(8):
                v = flat[int(rng.integers(len(flat)))]
                lits += [v, v[: max(1, len(v) // 2)], v + b"0"]
            lits += [b"", b"http://", b"https://zzzz"]
            for lit in lits:
                for opn, fn in cmp_ops.items():
                    expr = lc.LiquidExpr.try_new(opn, lit, pa.binary() if False else pa.string(), HINT)
                    mask, counts = scan.eval_to_host(expr)
                    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[: len(flat)].astype(np.bool_)
                    want = np.fromiter((fn(v, lit) for v in flat), dtype=np.bool_, count=len(flat))
                    if not np.array_equal(bits, want):
                        print("MISMATCH variant %s op %s literal %r: gpu %d truth %d" % (name, opn, lit, bits.sum(), want.sum()))
                        sys.exit(1)
                    checked += 1
            sel_bits = rng.random(len(flat)) < 0.3
            sel_words = np.packbits(sel_bits, bitorder="little").view(np.uint64)
            for nd in needles[:20]:
                for op in ("like", "not_like"):
                    expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.string(), HINT)
                    mask, counts = scan.eval_to_host(expr, sel_words)
                    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")[: len(flat)].astype(np.bool_)
                    want = (truth[nd] if op == "like" else ~truth[nd]) & sel_bits
                    if not np.array_equal(bits, want):
                        print("MISMATCH under selection: variant %s op %s needle %r" % (name, op, nd))
                        sys.exit(1)
                    checked += 1
            scan.close()
        finally:
            cache.close()
    print("soak ok: %d evaluations (%d needles x 2 operators x 2 + 27 literals x 6 comparisons + 40 under a selection, x %d variants) over %d rows" % (
        checked, len(needles), len(VARIANTS), len(flat)))


if __name__ == "__main__":
    main()
