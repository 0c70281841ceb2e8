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
