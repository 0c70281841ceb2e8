"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bit-exact for every integer / byte / bitmap result (value bits are compared under validity: bits under null
slots are unspecified in Arrow).
"""
import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc

pytestmark = pytest.mark.gpu

INT_TYPES = [("int8", pa.int8()), ("int16", pa.int16()), ("int32", pa.int32()), ("int64", pa.int64()),
             ("uint8", pa.uint8()), ("uint16", pa.uint16()), ("uint32", pa.uint32()), ("uint64", pa.uint64()),
             ("date32", pa.date32()), ("date64", pa.date64()), ("timestamp[us]", pa.timestamp("us"))]
OPS = ["eq", "ne", "lt", "le", "gt", "ge"]


def _bool_result(arr: pa.BooleanArray):
    vals = arr.to_numpy(zero_copy_only=False)
    if arr.null_count:
        valid = ~np.asarray(arr.is_null().to_numpy(zero_copy_only=False), dtype=bool)
        return np.where(valid, vals, False).astype(bool), valid
    return vals.astype(bool), None


def _check_pred(cache, lo, eid, liquid, op, literal, dtype, selection, symtab=None, hint=None):
    expr = lc.LiquidExpr.try_new(op, literal, dtype, hint)
    assert expr is not None
    b = cache.eval_predicate(eid, expr)
    if selection is not None:
        b = b.with_selection(selection)
    got = b.read()
    want = lo.eval_predicate(liquid, lo.OP_NAMES[op], literal, selection, symtab=symtab)
    gv, gvalid = _bool_result(got)
    assert len(gv) == len(want.values), (op, literal, len(gv), len(want.values))
    if want.validity is None:
        assert gvalid is None or gvalid.all()
        assert gv.tolist() == want.values.tolist(), (op, literal)
    else:
        assert gvalid is not None or want.validity.all()
        if gvalid is not None:
            assert gvalid.tolist() == want.validity.tolist(), (op, literal)
        assert gv.tolist() == (want.values & want.validity).tolist(), (op, literal)


def _random_ints(rng, np_dtype, n, width_bits):
    info = np.iinfo(np_dtype)
    span = min((1 << width_bits) - 1, int(info.max) - int(info.min))
    room = int(info.max) - span - int(info.min)
    lo_ = int(info.min) + (int(rng.integers(0, 1 << 62)) % (room + 1))
    if span == 0:
        return np.full(n, lo_, dtype=np_dtype)
    vals = rng.integers(0, span, size=n, endpoint=True, dtype=np.uint64)
    out = (vals.astype(object) + lo_)
    return np.array(out.tolist(), dtype=np_dtype)


@pytest.mark.parametrize("name,dtype", INT_TYPES)
def test_integer_predicates_all_types(gpu_cache, oracle, name, dtype):
    lo = oracle
    rng = np.random.default_rng(hash(name) % 1000)
    np_dtype = lo.PHYS_NP[lo.PHYS[name]]
    bits = np.dtype(np_dtype).itemsize * 8
    widths = sorted({1, 3, 7, min(11, bits), min(17, bits), bits - 1, bits})
    eid = 0
    for W in widths:
        for n in (8192, 1000, 1, 2048 + 65):
            vals = _random_ints(rng, np_dtype, n, W)
            nullable = bool(rng.integers(2))
            valid = rng.random(n) < 0.8 if nullable else None
            liquid = lo.encode_primitive(lo.PHYS[name], vals, valid)
            eid += 1
            gpu_cache.stage([eid], [liquid], data_types=[dtype])
            for op in OPS:
                lit = int(vals[rng.integers(n)]) + int(rng.integers(-1, 2))
                lit = max(int(np.iinfo(np_dtype).min), min(int(np.iinfo(np_dtype).max), lit))
                sel = (rng.random(n) < rng.choice([0.01, 0.5, 0.99])) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, lit, dtype, sel)
            # literals outside the batch range exercise the packed-domain clamp
            for lit in (int(np.iinfo(np_dtype).min), int(np.iinfo(np_dtype).max)):
                _check_pred(gpu_cache, lo, eid, liquid, "gt", lit, dtype, None)
                _check_pred(gpu_cache, lo, eid, liquid, "le", lit, dtype, None)


def test_readme_example(gpu_cache, oracle):
    # reference README.md:43-88: [10..15] UInt64, `> 12` -> [F,F,F,T,T,T]
    arr = pa.array([10, 11, 12, 13, 14, 15], type=pa.uint64())
    gpu_cache.insert(lc.EntryID(42), arr)
    expr = lc.LiquidExpr.try_new(">", 12, pa.uint64())
    got = gpu_cache.eval_predicate(lc.EntryID(42), expr).read()
    assert got.to_pylist() == [False, False, False, True, True, True]
    got = gpu_cache.eval_predicate(lc.EntryID(42), expr).with_selection([True, False, True, False, True, False]).read()
    assert got.to_pylist() == [False, False, True]
    assert gpu_cache.eval_predicate(lc.EntryID(43), expr).read() is None  # not cached -> None


def test_all_null_and_empty(gpu_cache, oracle):
    lo = oracle
    liquid = lo.encode_primitive(lo.PHYS["int32"], np.zeros(100, np.int32), np.zeros(100, bool))
    gpu_cache.stage([1], [liquid])
    _check_pred(gpu_cache, lo, 1, liquid, "gt", 0, pa.int32(), None)
    _check_pred(gpu_cache, lo, 1, liquid, "eq", 0, pa.int32(), np.arange(100) % 3 == 0)
    liquid = lo.encode_primitive(lo.PHYS["int64"], np.zeros(0, np.int64))
    gpu_cache.stage([2], [liquid])
    _check_pred(gpu_cache, lo, 2, liquid, "gt", 0, pa.int64(), None)


def test_decimal_predicates(gpu_cache, oracle):
    lo = oracle
    rng = np.random.default_rng(5)
    unscaled = [int(x) for x in rng.integers(0, 11, size=8192)]
    unscaled[17] = None
    liquid = lo.encode_decimal(unscaled, precision=15, scale=2)
    gpu_cache.stage([1], [liquid])
    import decimal
    for op in OPS:
        for lit in ("0.05", "0.07", "0.00", "0.10", "-1.00", "99999.00"):
            expr = lc.LiquidExpr.try_new(op, decimal.Decimal(lit), pa.decimal128(15, 2))
            got = gpu_cache.eval_predicate(1, expr).read()
            want = lo.eval_predicate(liquid, lo.OP_NAMES[op], int(decimal.Decimal(lit) * 100))
            gv, gvalid = _bool_result(got)
            assert gvalid.tolist() == want.validity.tolist()
            assert gv.tolist() == (want.values & want.validity).tolist(), (op, lit)


@pytest.mark.parametrize("name,dtype,np_dtype", [("float32", pa.float32(), np.float32), ("float64", pa.float64(), np.float64)])
def test_float_predicates_alp(gpu_cache, oracle, name, dtype, np_dtype):
    """ALP float entries: the device compares in the packed domain (decode is monotone) and re-evaluates the exception
    rows; the oracle decodes the same Liquid bytes and compares like Arrow (totalOrder).  Bit-exact."""
    lo = oracle
    rng = np.random.default_rng(1234)
    eid = 0
    cases = []
    for n in (8192, 3001, 9):
        prices = (rng.integers(-50000, 50000, size=n) / 100.0).astype(np_dtype)        # 2 decimals: W ~ 17
        prices[rng.random(n) < 0.03] = np_dtype(np.pi)                                  # exceptions
        if n > 8:
            prices[1], prices[2], prices[3], prices[4] = np.nan, np.inf, -np.inf, -0.0
        cases.append((prices, rng.random(n) < 0.9))
        small = rng.integers(0, 8, size=n).astype(np_dtype)                             # tiny W, no exceptions
        cases.append((small, None))
        big = (rng.integers(-2**40, 2**40, size=n) * 1000.0).astype(np_dtype) if np_dtype == np.float64 else \
            (rng.integers(-2**22, 2**22, size=n) * 4.0).astype(np_dtype)               # wide W
        cases.append((big, rng.random(n) < 0.5))
        const = np.full(n, np_dtype(7.25))                                              # W = 0 region
        cases.append((const, None))
    for vals, valid in cases:
        n = len(vals)
        liquid = lo.encode_primitive(lo.PHYS[name], vals, valid)
        eid += 1
        gpu_cache.stage([eid], [liquid], data_types=[dtype])
        finite = vals[np.isfinite(vals)]
        lits = [float(finite[0]), float(finite[-1]), float(np.nextafter(finite[0], np_dtype(np.inf))),
                float(np.nextafter(finite[0], np_dtype(-np.inf))), float(np.median(finite)), 0.0, -0.0, float(np.pi),
                float("inf"), float("-inf"), float("nan"), 1e30, -1e30, float(np_dtype(np.pi))]
        for op in OPS:
            for lit in lits:
                sel = (rng.random(n) < rng.choice([0.05, 0.5])) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, lit, dtype, sel)


def _make_strings(rng, n, n_unique, with_nulls):
    hosts = ["google", "yandex", "mail", "goo", "gle", "oogle", "googl", "example", "ya", "g"]
    pool = []
    for i in range(n_unique):
        k = int(rng.integers(1, 5))
        parts = [hosts[int(rng.integers(len(hosts)))] for _ in range(k)]
        s = "http://" + ".".join(parts) + "/" + "x" * int(rng.integers(0, 40)) + str(i % 97)
        if rng.random() < 0.05:
            s += "ÿé" + "z" * int(rng.integers(200, 400))
        pool.append(s)
    keys = np.minimum((rng.zipf(1.3, size=n) - 1), n_unique - 1)
    out = [pool[k] for k in keys]
    if with_nulls:
        for i in rng.choice(n, size=max(1, n // 20), replace=False):
            out[int(i)] = None
    return out


@pytest.fixture()
def gpu_cache_no_signatures(product_lib, monkeypatch):
    """A context that stages byte views WITHOUT the bigram signature index (lc_ctx_set_option LC_OPT_SIGNATURE_INDEX = 0):
    the kernel then runs the reference's fingerprint filter."""
    cache = lc.LiquidCacheBuilder.new().with_index_options(signatures=False).build()
    yield cache
    cache.close()


@pytest.mark.parametrize("fingerprints", [True, False])
def test_string_predicates_without_signature_index(gpu_cache_no_signatures, oracle, fingerprints):
    test_string_predicates(gpu_cache_no_signatures, oracle, fingerprints)


@pytest.mark.parametrize("fingerprints", [True, False])
def test_string_predicates(gpu_cache, oracle, fingerprints):
    lo = oracle
    rng = np.random.default_rng(11 + int(fingerprints))
    hint = lc.CacheExpression.SUBSTRING_SEARCH
    eid = 0
    for n, d, nulls in ((8192, 2200, False), (8192, 300, True), (777, 500, True), (3, 3, False)):
        strs = _make_strings(rng, n, d, nulls)
        liquid, st = lo.encode_byte_view(strs, fingerprints=fingerprints)
        eid += 1
        path = 1000 + eid
        gpu_cache.set_symbol_table(path, lo.symtab_bytes(st))
        gpu_cache.stage([eid], [liquid], [path])
        nonnull = [s for s in strs if s is not None]
        needles = [nonnull[0], nonnull[-1], nonnull[0][:9], nonnull[0] + "x", "http://", "", "http://goo", "zzz",
                   nonnull[len(nonnull) // 2][:16]]
        for op in OPS:
            for needle in needles:
                sel = (rng.random(n) < 0.3) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, needle.encode(), pa.string(), sel, symtab=st)
        for op in ("like", "not_like"):
            for pat in ("%google%", "%goo%", "%zzzz%", "%g%", "%le.goo%", "%ÿé%", "%" + nonnull[0][-5:] + "%"):
                sel = (rng.random(n) < 0.3) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, pat.encode(), pa.string(), sel, symtab=st, hint=hint)
        # general patterns: Arrow `like` on the dictionary for entries without fingerprints; with fingerprints the
        # reference insists on %needle% (comparisons.rs:150-166) and the device answers LC_UNSUPPORTED
        general = ["http://goo%", "%le" + nonnull[0][-2:], "%goo%gle%", "h_tp%", "%\\%%", "http://%/x_" + "%", "%", "_%",
                   nonnull[0], nonnull[0][:-1] + "_", "%ÿ_z%", "__________%"]
        # (a %needle% of more than 63 bytes is a substring search like any other since round 4: automaton over its first 63
        # bytes, accepted values matched against the pattern)
        long_sub = ["%" + "o" * 70 + "%", "%" + max(nonnull, key=len)[2:72] + "%"]
        for op in ("like", "not_like"):
            for pat in long_sub:
                sel = (rng.random(n) < 0.3) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, pat.encode(), pa.string(), sel, symtab=st, hint=hint)
            for pat in general:
                if fingerprints:
                    with pytest.raises(lc.LiquidCacheError):
                        gpu_cache.eval_predicate(eid, lc.LiquidExpr.try_new(op, pat.encode(), pa.string(), hint)).read()
                    continue
                sel = (rng.random(n) < 0.3) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, pat.encode(), pa.string(), sel, symtab=st, hint=hint)


def test_string_large_dictionary_and_long_needles(gpu_cache, oracle):
    """Dictionaries above the byte-table limit (bitmap result variant, > 64 KB of LDS per workgroup), needles longer
    than the LDS automaton limit (table walked from global memory), escaped bytes inside the needle."""
    lo = oracle
    rng = np.random.default_rng(77)
    hint = lc.CacheExpression.SUBSTRING_SEARCH
    for eid, (n, d, fingerprints) in enumerate(((30000, 20000, True), (9000, 5000, False), (8192, 2200, True)), start=1):
        pool = _make_strings(rng, d, d, False)
        pool = [s + "#%d" % i for i, s in enumerate(pool)]       # all distinct
        keys = rng.integers(0, d, size=n)
        strs = [pool[k] for k in keys]
        for i in rng.choice(n, size=n // 50, replace=False):
            strs[int(i)] = None
        liquid, st = lo.encode_byte_view(strs, fingerprints=fingerprints)
        path = 2000 + eid
        gpu_cache.set_symbol_table(path, lo.symtab_bytes(st))
        gpu_cache.stage([eid], [liquid], [path])
        nonnull = [s for s in strs if s is not None]
        long_piece = max(nonnull, key=len)
        pats = ["%google%", "%" + nonnull[3][2:30] + "%", "%" + long_piece[5:60] + "%", "%ÿéz%", "%#1999%", "%e.g%"]
        for op in ("like", "not_like"):
            for pat in pats:
                sel = (rng.random(n) < 0.4) if rng.integers(2) else None
                _check_pred(gpu_cache, lo, eid, liquid, op, pat.encode(), pa.string(), sel, symtab=st, hint=hint)
        for op in OPS:
            for needle in (nonnull[0], nonnull[7][:12], "http://goo"):
                _check_pred(gpu_cache, lo, eid, liquid, op, needle.encode(), pa.string(), None, symtab=st)


def test_and_then(gpu_cache, oracle):
    lo = oracle
    rng = np.random.default_rng(3)
    # doc example datafusion/src/utils.rs:54-57
    L = np.array([c == "Y" for c in "NNYYYNNYYNYN"])
    R = np.array([c == "Y" for c in "YNYNYN"])
    got = lc.boolean_buffer_and_then(gpu_cache, L, R)
    assert "".join("Y" if x else "N" for x in got) == "NNYNYNNNYNNN"
    for n in (1, 63, 64, 65, 128, 8192, 8192 * 3 + 5):
        for p in (0.0, 0.02, 0.5, 1.0):
            left = rng.random(n) < p
            right = rng.random(int(left.sum())) < 0.5
            got = lc.boolean_buffer_and_then(gpu_cache, left, right)
            assert got.tolist() == lo.and_then(left, right).tolist(), (n, p)


def test_scan_conjunction_chain(gpu_cache, oracle):
    """Mask of one predicate is the selection of the next (build_predicate_filter), on the device."""
    lo = oracle
    rng = np.random.default_rng(21)
    n_batches, n = 9, 8192
    a = rng.integers(0, 3000, size=n_batches * n - 100, dtype=np.int32)
    b = rng.integers(-50, 50, size=n_batches * n - 100, dtype=np.int64)
    ids_a, ids_b = [], []
    for k in range(n_batches):
        sl = slice(k * n, min((k + 1) * n, len(a)))
        ia, ib = lc.ParquetArrayID.new(0, 0, 1, k), lc.ParquetArrayID.new(0, 0, 2, k)
        gpu_cache.insert(ia, pa.array(a[sl]))
        gpu_cache.insert(ib, pa.array(b[sl]))
        ids_a.append(ia)
        ids_b.append(ib)
    sa, sb = gpu_cache.scan(ids_a), gpu_cache.scan(ids_b)
    e1 = lc.LiquidExpr.try_new(">=", 1000, pa.int32())
    e2 = lc.LiquidExpr.try_new("<", 7, pa.int64())
    m1, c1 = sa.eval_to_host(e1)
    m2, c2 = sb.eval_to_host(e2, selection=m1)
    want = (a >= 1000) & (b < 7)
    assert int(c1.sum()) == int((a >= 1000).sum())
    assert int(c2.sum()) == int(want.sum())
    bits = np.unpackbits(m2.view(np.uint8), bitorder="little")
    got = np.concatenate([bits[int(sb.segment_offsets[k]) * 64: int(sb.segment_offsets[k]) * 64 +
                               (min((k + 1) * n, len(a)) - k * n)] for k in range(n_batches)]).astype(bool)
    assert got.tolist() == want.tolist()


# ---------------------------------------------------------------------------------------------------------------
# get().with_selection()  (LiquidCache::read_arrow_array -> LiquidArray::filter)
# ---------------------------------------------------------------------------------------------------------------
def _arrow_values_valid(arr: pa.Array):
    valid = ~np.asarray(arr.is_null().to_numpy(zero_copy_only=False), dtype=bool)
    return arr, valid


@pytest.mark.parametrize("name,dtype", INT_TYPES)
def test_get_with_selection_integers(gpu_cache, oracle, name, dtype):
    lo = oracle
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    np_dtype = lo.PHYS_NP[lo.PHYS[name]]
    bits = np.dtype(np_dtype).itemsize * 8
    eid = 100
    for W in sorted({1, 5, min(13, bits), bits - 1, bits}):
        for n in (8192, 1030, 1, 64):
            vals = _random_ints(rng, np_dtype, n, W)
            valid = rng.random(n) < 0.7 if rng.integers(2) else None
            liquid = lo.encode_primitive(lo.PHYS[name], vals, valid)
            eid += 1
            gpu_cache.stage([eid], [liquid], data_types=[dtype])
            for p in (None, 0.0, 0.03, 0.5, 1.0):
                sel = None if p is None else (rng.random(n) < p)
                g = gpu_cache.get(eid)
                got = (g.with_selection(sel) if sel is not None else g).read()
                want_vals, want_valid = lo.filter_fixed(liquid, sel if sel is not None else np.ones(n, bool))
                assert got.type == dtype
                assert len(got) == len(want_vals)
                storage = pa.from_numpy_dtype(np_dtype)
                gnp = got.view(storage).fill_null(0).to_numpy(zero_copy_only=False)
                gvalid = ~np.asarray(got.is_null().to_numpy(zero_copy_only=False), dtype=bool)
                if want_valid is None:
                    assert gvalid.all()
                    assert gnp.astype(np_dtype).tolist() == want_vals.tolist()
                else:
                    assert gvalid.tolist() == want_valid.tolist()
                    assert gnp.astype(np_dtype)[want_valid].tolist() == want_vals[want_valid].tolist()


def test_get_with_selection_reference_vectors(gpu_cache):
    # primitive_array.rs:928-944: [1,2,3,None,5] filter [T,F,T,F,T] -> [1,3,5]; README.md:43-60
    gpu_cache.insert(1, pa.array([1, 2, 3, None, 5], type=pa.int32()))
    assert gpu_cache.get(1).with_selection([True, False, True, False, True]).read().to_pylist() == [1, 3, 5]
    gpu_cache.insert(2, pa.array([10, 11, 12, 13, 14, 15], type=pa.uint64()))
    assert gpu_cache.get(2).with_selection([True, False, True, False, True, False]).read().to_pylist() == [10, 12, 14]
    # primitive_array.rs:953-968 all-null filter
    gpu_cache.insert(3, pa.array([None, None, None, None], type=pa.int32()))
    assert gpu_cache.get(3).with_selection([True, False, False, True]).read().to_pylist() == [None, None]
    assert gpu_cache.get(3).with_selection([False] * 4).read().to_pylist() == []
    assert gpu_cache.get(99).read() is None


def test_get_with_selection_floats_and_decimals(gpu_cache, oracle):
    lo = oracle
    import decimal
    rng = np.random.default_rng(77)
    for dt, pat in ((np.float32, pa.float32()), (np.float64, pa.float64())):
        for n in (8192, 3000, 5):
            base = rng.normal(size=n).astype(dt).round(2)
            base[rng.random(n) < 0.02] = dt(np.pi)          # forces ALP patches
            if n > 4:
                base[1], base[2] = np.nan, np.inf
            arr = pa.array(base, type=pat, mask=(rng.random(n) < 0.1))
            liquid = gpu_cache.transcode(arr)
            gpu_cache.stage([500 + n], [liquid], data_types=[pat])
            for p in (None, 0.2, 1.0):
                sel = None if p is None else (rng.random(n) < p)
                g = gpu_cache.get(500 + n)
                got = (g.with_selection(sel) if sel is not None else g).read()
                # the oracle decodes the same Liquid bytes (ALP maps -0.0 to +0.0 exactly like the reference:
                # float_array.rs:636 compares with `==`, so -0.0 is not patched)
                wv, valid = lo.filter_fixed(liquid, sel if sel is not None else np.ones(n, bool))
                assert got.type == pat and len(got) == len(wv)
                gv = got.to_numpy(zero_copy_only=False)
                assert (~np.asarray(got.is_null().to_numpy(zero_copy_only=False), dtype=bool)).tolist() == valid.tolist()
                bits = np.uint32 if dt == np.float32 else np.uint64
                assert gv[valid].view(bits).tolist() == wv[valid].view(bits).tolist()  # bit-exact
                want = arr if sel is None else arr.filter(pa.array(sel))
                np.testing.assert_array_equal(gv[valid], want.to_numpy(zero_copy_only=False)[valid])
    vals = [decimal.Decimal(int(x)) / 100 for x in rng.integers(0, 100000, size=5000)]
    vals[7] = None
    arr = pa.array(vals, type=pa.decimal128(15, 2))
    gpu_cache.insert(900, arr)
    sel = rng.random(5000) < 0.3
    assert gpu_cache.get(900).with_selection(sel).read().to_pylist() == arr.filter(pa.array(sel)).to_pylist()
    assert gpu_cache.get(900).read().to_pylist() == arr.to_pylist()


def test_scan_gather_fixed(gpu_cache, oracle):
    """get-with-selection over a whole scan: mask of a predicate -> compacted decoded values, row order."""
    rng = np.random.default_rng(31)
    n_batches, n = 7, 8192
    total = n_batches * n - 333
    for np_dt, pa_dt, lo_hi in ((np.int64, pa.int64(), (-2**40, 2**40)), (np.int32, pa.int32(), (0, 4000)),
                                (np.int32, pa.date32(), (7000, 12000)), (np.float64, pa.float64(), None)):
        if lo_hi is None:
            vals = (rng.integers(-10**6, 10**6, size=total) / 100.0).astype(np_dt)
        else:
            vals = rng.integers(lo_hi[0], lo_hi[1], size=total).astype(np_dt)
        ids = []
        for k in range(n_batches):
            eid = lc.ParquetArrayID.new(3, int(np.dtype(np_dt).itemsize) + (pa_dt == pa.date32()), len(ids), k)
            gpu_cache.insert(eid, pa.array(vals[k * n: min((k + 1) * n, total)], type=pa_dt))
            ids.append(eid)
        scan = gpu_cache.scan(ids)
        got, offs = scan.gather_fixed_to_host(np_dt)                       # no selection: everything, in order
        assert got.view(np.uint8).tobytes() == vals.view(np.uint8).tobytes()
        assert offs.tolist() == [min(k * n, total) for k in range(n_batches + 1)]
        for p_sel in (0.13, 0.002):   # dense blocks are staged through LDS, sparse ones fetch single words
            keep = rng.random(total) < p_sel
            words = np.zeros(int(scan.mask_words), np.uint64)
            for k in range(n_batches):
                seg = keep[k * n: min((k + 1) * n, total)]
                packed = np.packbits(seg, bitorder="little")
                w0 = int(scan.segment_offsets[k])
                words[w0: w0 + (len(seg) + 63) // 64].view(np.uint8)[: len(packed)] = packed
            got, offs = scan.gather_fixed_to_host(np_dt, selection=words)
            assert got.view(np.uint8).tobytes() == vals[keep].view(np.uint8).tobytes()
            assert int(offs[-1]) == int(keep.sum())
        if pa_dt == pa.date32():
            lo = oracle
            got, _ = scan.gather_fixed_to_host(np_dt, selection=words, date_field=lc.Date32Field.MONTH)
            want = [lo.date_lossy_days(1, lo.date_component(1, int(d))) for d in vals[keep]]
            assert got.tolist() == want
        scan.close()


def test_scan_gather_bytes(gpu_cache, oracle):
    """get-with-selection over a whole byte-view scan (device resident): selected rows' values in row order."""
    rng = np.random.default_rng(41)
    n_batches, n = 6, 8192
    total = n_batches * n - 777
    pool = _make_strings(rng, 3000, 3000, False) + ["", "ÿ", "a" * 700]
    keys = rng.integers(0, len(pool), size=total)
    strs = [pool[k] for k in keys]
    for i in rng.choice(total, size=total // 40, replace=False):
        strs[int(i)] = None
    ids = []
    for k in range(n_batches):
        eid = lc.ParquetArrayID.new(9, 0, 7, k)
        gpu_cache.insert(eid, pa.array(strs[k * n: min((k + 1) * n, total)], type=pa.string()),
                         lc.CacheExpression.SUBSTRING_SEARCH if k % 2 else None)
        ids.append(eid)
    scan = gpu_cache.scan(ids)
    want_all = [None if s is None else s.encode() for s in strs]
    assert scan.gather_bytes_to_host() == want_all                       # no selection: everything
    for p_sel in (0.2, 0.001, 0.0):
        keep = rng.random(total) < p_sel
        words = np.zeros(int(scan.mask_words), np.uint64)
        for k in range(n_batches):
            seg = keep[k * n: min((k + 1) * n, total)]
            packed = np.packbits(seg, bitorder="little")
            w0 = int(scan.segment_offsets[k])
            words[w0: w0 + (len(seg) + 63) // 64].view(np.uint8)[: len(packed)] = packed
        assert scan.gather_bytes_to_host(selection=words) == [w for w, kp in zip(want_all, keep) if kp]
    # the projection step of a filter: LIKE mask -> the matching URLs
    m, c = scan.eval_to_host(lc.LiquidExpr.try_new("like", b"%google%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH))
    got = scan.gather_bytes_to_host(selection=m)
    assert got == [w for w in want_all if w is not None and b"google" in w] and len(got) == int(c.sum())
    scan.close()


def test_get_with_date_part_hint(gpu_cache, oracle):
    """cache.get(id).with_expression_hint(extract_date32(field)) == SqueezedDate32Array's lossy reconstruction."""
    lo = oracle
    # reference test cache/core.rs:1053-1084: [2, 366, null, 465] / Year -> [0, 365, null, 365]
    arr = pa.array([2, 366, None, 465], type=pa.date32())
    gpu_cache.insert(lc.EntryID(42), arr)
    got = gpu_cache.get(lc.EntryID(42)).with_expression_hint(lc.CacheExpression.extract_date32(lc.Date32Field.YEAR)).read()
    assert got.type == pa.date32()
    assert got.view(pa.int32()).to_pylist() == [0, 365, None, 365]
    rng = np.random.default_rng(8)
    n = 5000
    days = rng.integers(-800000, 800000, size=n).astype(np.int32)   # years -220 .. 4160, incl. negative days
    days[:6] = [0, -1, 59, 60, 11016, -719468]
    mask = rng.random(n) < 0.1
    eid = 100
    for field in (lc.Date32Field.YEAR, lc.Date32Field.MONTH, lc.Date32Field.DAY, lc.Date32Field.DAY_OF_WEEK):
        hint = lc.CacheExpression.extract_date32(field)
        want_days = np.array([lo.date_lossy_days(field, lo.date_component(field, int(d))) for d in days], dtype=np.int64)
        eid += 1
        gpu_cache.insert(eid, pa.array(days, type=pa.date32(), mask=mask))
        for sel in (None, rng.random(n) < 0.3):
            g = gpu_cache.get(eid).with_expression_hint(hint)
            got = (g.with_selection(sel) if sel is not None else g).read()
            keep = np.ones(n, bool) if sel is None else sel
            assert got.type == pa.date32()
            assert got.is_null().to_pylist() == mask[keep].tolist()
            gv = got.view(pa.int32()).to_numpy(zero_copy_only=False)
            assert gv[~mask[keep]].astype(np.int64).tolist() == want_days[keep][~mask[keep]].tolist()
        for unit_code, unit, tpd in ((0, "s", 86400), (1, "ms", 86400000), (2, "us", 86400000000), (3, "ns", 86400000000000)):
            span = 100000 if unit == "ns" else 800000                  # keep ns inside i64
            ts = rng.integers(-span, span, size=n).astype(np.int64) * tpd + rng.integers(0, tpd, size=n)
            want = np.array([lo.date_lossy_days(field, lo.date_component(field, lo.timestamp_to_days(int(t), unit_code)))
                             for t in ts], dtype=np.int64) * tpd
            eid += 1
            gpu_cache.insert(eid, pa.array(ts, type=pa.timestamp(unit)))
            sel = rng.random(n) < 0.5
            got = gpu_cache.get(eid).with_expression_hint(hint).with_selection(sel).read()
            assert got.type == pa.timestamp(unit)
            assert got.view(pa.int64()).to_pylist() == want[sel].tolist()
    # not a date-like entry: the C ABI answers LC_UNSUPPORTED (the reference only squeezes Date32 / Timestamp)
    gpu_cache.insert(999, pa.array([1, 2, 3], type=pa.int32()))
    with pytest.raises(lc.LiquidCacheError):
        gpu_cache.get(999).with_expression_hint(lc.CacheExpression.extract_date32("year")).read()


@pytest.mark.parametrize("arrow_type", [pa.string(), pa.binary(), pa.string_view()])
def test_get_with_selection_strings(gpu_cache, oracle, arrow_type):
    rng = np.random.default_rng(91)
    for n, d, nulls in ((8192, 2200, True), (500, 400, False), (2, 2, False)):
        strs = _make_strings(rng, n, d, nulls)
        arr = pa.array([None if s is None else (s.encode() if pa.types.is_binary(arrow_type) else s) for s in strs],
                       type=arrow_type)
        eid = lc.ParquetArrayID.new(1, n % 7, 3, 0)
        gpu_cache.insert(eid, arr, lc.CacheExpression.SUBSTRING_SEARCH)
        for p in (None, 0.0, 0.01, 0.4, 1.0):
            sel = None if p is None else (rng.random(n) < p)
            g = gpu_cache.get(eid)
            got = (g.with_selection(sel) if sel is not None else g).read()
            plain = arr.cast(pa.string()) if pa.types.is_string_view(arrow_type) else arr
            want = plain if sel is None else plain.filter(pa.array(sel))
            assert got.type == arrow_type
            assert got.to_pylist() == want.to_pylist()
