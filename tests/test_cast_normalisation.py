"""Host logic: `CAST(col) OP literal` / `to_timestamp_seconds(col) OP literal` -> `col OP' literal'` (LiquidExpr.try_new(lhs=...)).

The reference accepts these column-like forms (src/core/src/cache/liquid_expr.rs:150-174) and evaluates them by running the
cast on the decoded array (eval_predicate_on_array).  The rewrite must therefore be the same predicate for EVERY column value:
checked here against pyarrow's cast + compare over value sets that hold every boundary.  No GPU needed.
"""
import datetime

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import liquid_cache_amd as lc
from liquid_cache_amd import _native as N
from liquid_cache_amd.cache import _normalise_lhs

OPS = {N.OP_EQ: pc.equal, N.OP_NE: pc.not_equal, N.OP_LT: pc.less, N.OP_LE: pc.less_equal, N.OP_GT: pc.greater,
       N.OP_GE: pc.greater_equal}


import operator

PYOPS = {N.OP_EQ: operator.eq, N.OP_NE: operator.ne, N.OP_LT: operator.lt, N.OP_LE: operator.le, N.OP_GT: operator.gt,
         N.OP_GE: operator.ge}


def _stored(arr):
    """The integers (or floats) the column stores: what the device kernels compare."""
    t = arr.type
    if pa.types.is_date32(t):
        arr = arr.view(pa.int32())
    elif pa.types.is_date64(t) or pa.types.is_timestamp(t):
        arr = arr.view(pa.int64())
    return arr.to_pylist()


def _eval(arr, code, lit):
    """`col OP lit` over the stored values with Python's exact integers (a literal outside the column type's range is legal:
    the library classifies it as below / above every value)."""
    return [None if v is None else bool(PYOPS[code](v, lit)) for v in _stored(arr)]


def _check(arr, lhs, t_out, literals):
    n_checked = 0
    casted = pc.cast(arr, t_out)
    for code, fn in OPS.items():
        for lit in literals:
            want = fn(casted, pa.scalar(lit, type=t_out)).to_pylist()
            norm = _normalise_lhs(code, lit, arr.type, lhs)
            assert norm is not None, (arr.type, t_out, code, lit)
            c2, l2, constant = norm
            if constant is not None:
                got = [None if v is None else constant for v in arr.to_pylist()]
            else:
                got = _eval(arr, c2, l2)
            assert got == want, (str(arr.type), str(t_out), code, lit, c2, l2)
            # and the expression object exists (the literal fits the C ABI's literal forms)
            assert lc.LiquidExpr.try_new(code, lit, arr.type, None, lhs) is not None
            n_checked += 1
    return n_checked


def test_integer_widening_and_float_casts():
    i8 = pa.array(list(range(-128, 128)) + [None], type=pa.int8())
    u16 = pa.array([0, 1, 2, 255, 256, 65534, 65535, None], type=pa.uint16())
    i32 = pa.array([-2**31, -2**31 + 1, -1000001, -13, -12, -1, 0, 1, 12, 13, 999999, 2**31 - 2, 2**31 - 1, None], type=pa.int32())
    col = lc.Column()
    n = 0
    n += _check(i8, lc.Cast(col, pa.int64()), pa.int64(), [-129, -128, -1, 0, 5, 127, 128, 10**12, -10**12])
    n += _check(u16, lc.Cast(col, pa.int32()), pa.int32(), [-1, 0, 1, 255, 65535, 65536, 2**31 - 1])
    n += _check(i32, lc.Cast(col, pa.int64()), pa.int64(), [-2**40, -2**31, -12, 0, 13, 2**31 - 1, 2**31, 2**40])
    n += _check(i32, lc.Cast(col, pa.float64()), pa.float64(),
                [-12.5, -12.0, -0.5, 0.0, 0.5, 12.0, 12.5, 12.999, 13.0, 1e15, -1e15, float("inf"), float("-inf"), 2.0**31, -(2.0**31) - 0.5])
    n += _check(i8, lc.Cast(col, pa.float32()), pa.float32(), [-128.5, -0.25, 0.0, 3.0, 126.5, 127.0, 127.5, 1e6])
    n += _check(u16, lc.Cast(lc.Cast(col, pa.int32()), pa.float64()), pa.float64(), [-0.5, 0.0, 255.5, 65535.0, 65535.5])
    assert n > 300


def test_date_and_timestamp_casts():
    d = pa.array([-25567, -1, 0, 1, 8036, 8037, 19000, 19001, 50000, None], type=pa.date32())
    col = lc.Column()
    ts_s = [datetime.datetime(1970, 1, 1), datetime.datetime(1992, 1, 2), datetime.datetime(1992, 1, 2, 0, 0, 1),
            datetime.datetime(1992, 1, 1, 23, 59, 59), datetime.datetime(2022, 1, 8, 12), datetime.datetime(1969, 12, 31, 23, 59, 59),
            datetime.datetime(1900, 1, 1)]
    n = 0
    for unit in ("s", "ms", "us"):
        n += _check(d, lc.Cast(col, pa.timestamp(unit)), pa.timestamp(unit), ts_s)
    n += _check(d, lc.Cast(col, pa.date64()), pa.date64(), [datetime.date(1992, 1, 2), datetime.date(1970, 1, 1), datetime.date(1969, 12, 31)])
    t_s = pa.array([0, 1, 59, 60, 1_000_000_000, -1, None], type=pa.timestamp("s"))
    n += _check(t_s, lc.Cast(col, pa.timestamp("ms")), pa.timestamp("ms"),
                [datetime.datetime(1970, 1, 1, 0, 0, 1), datetime.datetime(1970, 1, 1, 0, 0, 0, 500000), datetime.datetime(2001, 9, 9, 1, 46, 40)])
    # to_timestamp_seconds(Int64 column): the integer read as seconds
    secs = pa.array([0, 1, 1_000_000_000, -5, None], type=pa.int64())
    as_ts = pc.cast(secs, pa.timestamp("s"))
    for code, fn in OPS.items():
        for lit in (datetime.datetime(2001, 9, 9, 1, 46, 40), datetime.datetime(1970, 1, 1)):
            want = fn(as_ts, pa.scalar(lit, type=pa.timestamp("s"))).to_pylist()
            c2, l2, constant = _normalise_lhs(code, lit, secs.type, lc.ToTimestampSeconds(lc.Column()))
            assert constant is None
            assert _eval(secs, c2, l2) == want
            n += 1
    assert n > 150


def test_float32_widening():
    rng = np.random.default_rng(1)
    base = rng.normal(size=400).astype(np.float32)
    lits = [0.1, -0.1, 0.5, float(base[3]), float(np.nextafter(base[3], np.float32(1))), 1e-46, 3.5e38, 1e39, -1e39]
    vals = np.concatenate([base, np.float32(lits).astype(np.float32),
                           np.nextafter(np.float32(lits), np.float32(np.inf)).astype(np.float32),
                           np.nextafter(np.float32(lits), np.float32(-np.inf)).astype(np.float32),
                           np.array([np.inf, -np.inf, 0.0, -0.0], np.float32)])
    arr = pa.array(vals, type=pa.float32())
    casted = pc.cast(arr, pa.float64())
    lhs = lc.Cast(lc.Column(), pa.float64())
    for code, fn in OPS.items():
        for lit in lits:
            want = fn(casted, pa.scalar(lit, type=pa.float64())).to_pylist()
            norm = _normalise_lhs(code, lit, arr.type, lhs)
            assert norm is not None
            c2, l2, constant = norm
            if constant is not None:
                got = [constant] * len(vals)
            else:
                got = OPS[c2](arr, pa.scalar(np.float32(l2), type=pa.float32())).to_pylist()
            assert got == want, (code, lit, c2, l2)


def test_forms_without_an_exact_rewrite_are_rejected():
    col = lc.Column()
    assert lc.LiquidExpr.try_new(">", 5, pa.int64(), None, lc.Cast(col, pa.int32())) is None        # narrowing
    assert lc.LiquidExpr.try_new(">", 5.5, pa.int64(), None, lc.Cast(col, pa.float64())) is None    # Int64 -> Double rounds
    assert lc.LiquidExpr.try_new(">", 5.5, pa.int32(), None, lc.Cast(col, pa.float32())) is None    # Int32 -> Float rounds
    assert lc.LiquidExpr.try_new(">", 5, pa.int32(), None, lc.ToTimestampSeconds(col)) is None      # not an Int64 column
    assert lc.LiquidExpr.try_new("=", "x", pa.string(), None, lc.ToTimestampSeconds(col)) is None
    assert lc.LiquidExpr.try_new(">", float("nan"), pa.int32(), None, lc.Cast(col, pa.float64())) is None
    # byte-like columns: casts between the byte types are the same bytes
    e = lc.LiquidExpr.try_new("=", "x", pa.string(), None, lc.Cast(col, pa.string_view()))
    assert e is not None and e.lit_bytes == b"x"
    e = lc.LiquidExpr.try_new("like", "%x%", pa.string(), lc.CacheExpression.SUBSTRING_SEARCH, lc.Cast(lc.Cast(col, pa.binary()), pa.string_view()))
    assert e is not None
    # an un-wrapped Column() is the plain form
    assert lc.LiquidExpr.try_new(">", 5, pa.int32(), None, lc.Column()).lit_bytes == (5).to_bytes(8, "little")
