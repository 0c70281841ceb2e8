"""Host-side mirror of the reference's cache API for the decode + predicate-pushdown path.

Names, argument meaning and error behaviour follow the reference (paths relative to its repository root):

    LiquidCacheBuilder / LiquidCache::{insert,get,eval_predicate}   src/core/src/cache/builders.rs:32-356,
                                                                     src/core/src/cache/core.rs:122-142
    EntryID                                                          src/core/src/cache/utils.rs:70-84
    ParquetArrayID (file16|rg16|col16|batch16)                       src/datafusion/src/cache/id.rs:7-30
    LiquidExpr::try_new                                              src/core/src/cache/liquid_expr.rs:33-202
    CacheExpression                                                  src/core/src/cache/expressions.rs:36-51
    boolean_buffer_and_then                                          src/datafusion/src/utils.rs:62-83

All arithmetic on cached data runs in the HIP library behind include/liquid_cache_amd.h; this module only
marshals Arrow buffers.  `None` results mean exactly what they mean in the reference: entry not cached /
expression not supported ("caller falls back").
"""
from __future__ import annotations

import ctypes as C
import weakref
import datetime
import decimal
from typing import Iterable, Optional, Sequence, Union

import numpy as np
import pyarrow as pa

from . import _native as N
from ._native import LiquidCacheError

__all__ = ["EntryID", "ParquetArrayID", "CacheExpression", "LiquidExpr", "LiquidCacheBuilder", "LiquidCache",
           "Scan", "LiquidCacheError", "boolean_buffer_and_then"]


class EntryID(int):
    """Opaque cache key (reference: `EntryID(usize)`)."""

    def __new__(cls, value: int):
        if value < 0 or value >= 1 << 64:
            raise ValueError("EntryID must fit usize")
        return super().__new__(cls, value)


class ParquetArrayID:
    """file16 | rg16 | col16 | batch16 packing of the DataFusion layer (datafusion/src/cache/id.rs:15-21)."""

    @staticmethod
    def new(file_id: int, row_group_id: int, column_id: int, batch_id: int) -> EntryID:
        for v in (file_id, row_group_id, column_id, batch_id):
            if not 0 <= v <= 0xFFFF:
                raise ValueError("ParquetArrayID fields are 16 bit")
        return EntryID(file_id << 48 | row_group_id << 32 | column_id << 16 | batch_id)

    @staticmethod
    def column_access_path(entry_id: int) -> int:
        """(file, row group, column) — the unit that shares one FSST compressor (id.rs:141-146)."""
        return int(entry_id) >> 16


class CacheExpression:
    """Squeeze / encoding hints (only the ones that influence this path)."""
    SUBSTRING_SEARCH = N.HINT_SUBSTRING_SEARCH
    PREDICATE_COLUMN = N.HINT_PREDICATE_COLUMN

    @staticmethod
    def substring_search() -> int:
        return N.HINT_SUBSTRING_SEARCH

    @staticmethod
    def extract_date32(field) -> "ExtractDate32":
        """CacheExpression::extract_date32(Date32Field) (cache/expressions.rs:82-84)."""
        return ExtractDate32(field)


class Date32Field:
    """liquid_array::Date32Field (squeezed_date32_array.rs:26-38)."""
    YEAR, MONTH, DAY, DAY_OF_WEEK = 0, 1, 2, 3
    _NAMES = {"year": 0, "month": 1, "day": 2, "dow": 3, "dayofweek": 3, "day_of_week": 3}


class ExtractDate32:
    def __init__(self, field):
        self.field = Date32Field._NAMES[field.lower()] if isinstance(field, str) else int(field)


_OPS = {"=": N.OP_EQ, "==": N.OP_EQ, "eq": N.OP_EQ, "!=": N.OP_NE, "<>": N.OP_NE, "ne": N.OP_NE, "noteq": N.OP_NE,
        "<": N.OP_LT, "lt": N.OP_LT, "<=": N.OP_LE, "le": N.OP_LE, "lteq": N.OP_LE, ">": N.OP_GT, "gt": N.OP_GT,
        ">=": N.OP_GE, "ge": N.OP_GE, "gteq": N.OP_GE, "like": N.OP_LIKE, "not like": N.OP_NOT_LIKE,
        "not_like": N.OP_NOT_LIKE}


def _is_byte_like(t: pa.DataType) -> bool:
    if pa.types.is_dictionary(t):
        return _is_byte_like(t.value_type)
    return (pa.types.is_string(t) or pa.types.is_binary(t) or pa.types.is_string_view(t)
            or pa.types.is_binary_view(t))


def _is_numeric_like(t: pa.DataType) -> bool:
    return (pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_date(t) or pa.types.is_decimal(t)
            or (pa.types.is_timestamp(t) and t.tz is None))


class Column:
    """The left-hand side of a predicate as the reference's `is_column_like` sees it (liquid_expr.rs:150-163): the column
    itself or the column under Cast / CastColumn / TryCast wrappers; `ToTimestampSeconds` is the one scalar function the
    reference accepts around it (`is_to_timestamp_seconds_column`, :165-174)."""

    def __init__(self):
        self.inner = None


class Cast(Column):
    """CAST(inner AS to_type) (CastExpr / CastColumnExpr; `try_cast=True`: TryCastExpr)."""

    def __init__(self, inner: Column, to_type: pa.DataType, try_cast: bool = False):
        self.inner, self.to_type, self.try_cast = inner, to_type, try_cast


class ToTimestampSeconds(Column):
    """to_timestamp_seconds(inner): an Int64 of seconds since the epoch read as Timestamp(Second)."""

    def __init__(self, inner: Column):
        self.inner = inner


_BIG = (1 << 64) - 1  # beyond every signed column type: `col = _BIG` is constant false, `col != _BIG` constant true


def _int_range(t: pa.DataType):
    if pa.types.is_date32(t):
        return -(1 << 31), (1 << 31) - 1
    if pa.types.is_date64(t) or pa.types.is_timestamp(t):
        return -(1 << 63), (1 << 63) - 1
    if pa.types.is_signed_integer(t):
        return -(1 << (t.bit_width - 1)), (1 << (t.bit_width - 1)) - 1
    if pa.types.is_unsigned_integer(t):
        return 0, (1 << t.bit_width) - 1
    return None


def _as_int_in(v, t: pa.DataType):
    """The literal as the integer a column of type `t` stores (days, ticks, the integer itself); None: not exact."""
    if isinstance(v, bool):
        return None
    if pa.types.is_date32(t) and isinstance(v, datetime.date) and not isinstance(v, datetime.datetime):
        return (v - datetime.date(1970, 1, 1)).days
    if pa.types.is_date64(t) and isinstance(v, datetime.date) and not isinstance(v, datetime.datetime):
        return (v - datetime.date(1970, 1, 1)).days * 86_400_000
    if pa.types.is_timestamp(t) and isinstance(v, datetime.datetime):
        return int(pa.scalar(v, type=t).value)
    if isinstance(v, (int, np.integer)):
        return int(v)
    return None


_TICKS = {"s": 1, "ms": 1000, "us": 1_000_000, "ns": 1_000_000_000}


def _scaled(code: int, num, k: int):
    """`x * k OP num` (k > 0, x an integer) as `x OP' lit'`; num: int, Fraction-like float handled by the caller."""
    import math
    from fractions import Fraction
    q = Fraction(num) / k
    integral = q.denominator == 1
    if code == N.OP_EQ:
        return (N.OP_EQ, int(q)) if integral else (N.OP_EQ, None)    # None: constant (false for =, true for !=)
    if code == N.OP_NE:
        return (N.OP_NE, int(q)) if integral else (N.OP_NE, None)
    if code == N.OP_GT:
        return N.OP_GT, math.floor(q)
    if code == N.OP_GE:
        return N.OP_GE, math.ceil(q)
    if code == N.OP_LT:
        return N.OP_LT, math.ceil(q)
    if code == N.OP_LE:
        return N.OP_LE, math.floor(q)
    return None


def _unwrap_cast(code: int, lit, t_in: pa.DataType, t_out: pa.DataType):
    """`CAST(x AS t_out) OP lit` as `x OP' lit'` with x of type t_in — only where the two are the same predicate for EVERY x
    (the reference evaluates the cast itself, eval_predicate_on_array).  Returns (code', lit') or None (unsupported:
    the caller keeps the reference's CPU path).  lit' is None for a constant outcome."""
    import math
    if _is_byte_like(t_in) and _is_byte_like(t_out):
        return code, lit  # Utf8 / Utf8View / Binary / BinaryView: the same bytes
    if code > N.OP_GE:
        return None
    r_in = _int_range(t_in)
    if r_in is None:
        if pa.types.is_float32(t_in) and pa.types.is_float64(t_out) and isinstance(lit, (int, float, np.floating)):
            # every f32 is an f64: compare against the f32 neighbours of the literal
            x = float(lit)
            if math.isnan(x):
                return None
            f = np.float32(x)
            fx = float(f)  # (a literal beyond the f32 range rounds to +-inf: its f32 neighbours are +-max and +-inf)
            if fx == x:
                return code, fx
            lo = fx if fx < x else float(np.nextafter(f, np.float32(-np.inf)))   # largest f32 below x
            hi = fx if fx > x else float(np.nextafter(f, np.float32(np.inf)))    # smallest f32 above x
            if code == N.OP_EQ or code == N.OP_NE:
                return code, None
            if code in (N.OP_GT, N.OP_GE):
                return N.OP_GE, hi
            return N.OP_LE, lo
        return None
    lo_in, hi_in = r_in
    r_out = _int_range(t_out)
    int_like_in = pa.types.is_integer(t_in)
    if r_out is not None:
        # integer-like to integer-like: same stored integer (widening), or the integer times a constant (date / time units)
        k = None
        if int_like_in and pa.types.is_integer(t_out):
            k = 1
        elif pa.types.is_date32(t_in) and pa.types.is_date64(t_out):
            k = 86_400_000
        elif pa.types.is_date32(t_in) and pa.types.is_timestamp(t_out) and t_out.tz is None:
            k = 86_400 * _TICKS[t_out.unit]
        elif pa.types.is_date32(t_in) and pa.types.is_integer(t_out):
            k = 1
        elif pa.types.is_timestamp(t_in) and pa.types.is_timestamp(t_out) and t_in.tz is None and t_out.tz is None \
                and _TICKS[t_out.unit] % _TICKS[t_in.unit] == 0:
            k = _TICKS[t_out.unit] // _TICKS[t_in.unit]
        elif pa.types.is_integer(t_in) and t_in.bit_width == 64 and pa.types.is_timestamp(t_out) and t_out.tz is None:
            k = 1  # Int64 reinterpreted as ticks
        if k is None:
            return None
        if k == 1 and not (r_out[0] <= lo_in and hi_in <= r_out[1]):
            return None  # narrowing: the cast itself can fail / wrap
        if k != 1 and (lo_in * k < r_out[0] or hi_in * k > r_out[1]):
            # the product can leave the target type: accepted for dates and timestamps going to s / ms / us (they overflow
            # beyond +-290,000 years, where the cast itself fails, in the reference as well), never to nanoseconds (year 2262)
            if not (pa.types.is_date32(t_in) or pa.types.is_timestamp(t_in)) or (pa.types.is_timestamp(t_out) and t_out.unit == "ns"):
                return None
        li = _as_int_in(lit, t_out)
        if li is None:
            return None
        return _scaled(code, li, k)
    if pa.types.is_floating(t_out) and (int_like_in or pa.types.is_date32(t_in)):
        # every value must convert exactly: |x| < 2^24 for f32, 2^53 for f64
        exact_bits = 24 if t_out.bit_width == 32 else 53
        if max(abs(lo_in), abs(hi_in)) > (1 << exact_bits):
            return None
        if not isinstance(lit, (int, float, np.floating, np.integer)) or isinstance(lit, bool):
            return None
        x = float(np.float32(lit)) if t_out.bit_width == 32 else float(lit)
        if math.isnan(x):
            return None
        if math.isinf(x):
            return code, (_BIG if x > 0 else -(1 << 63))
        from fractions import Fraction
        return _scaled(code, Fraction(x), 1)
    return None


def _normalise_lhs(code: int, literal, column_type: pa.DataType, lhs: Column):
    """Peel the wrappers of `lhs` off the predicate, outermost first.  Returns (code, literal in the column's own domain,
    constant) or None.  constant: None, or the outcome for every valid row."""
    chain = []
    node = lhs
    while node is not None and not type(node) is Column:
        chain.append(node)
        node = node.inner
    if node is None:
        return None
    # the type every wrapper sees: innermost first
    t = column_type
    typed = []
    for w in reversed(chain):
        if isinstance(w, ToTimestampSeconds):
            if not (pa.types.is_integer(t) and t.bit_width == 64):
                return None
            t_out = pa.timestamp("s")
        else:
            t_out = w.to_type
        typed.append((t, t_out))
        t = t_out
    for t_in, t_out in reversed(typed):  # outermost first
        r = _unwrap_cast(code, literal, t_in, t_out)
        if r is None:
            return None
        code, literal = r
        if literal is None:  # constant outcome
            return code, None, code == N.OP_NE
    return code, literal, None


class LiquidExpr:
    """A predicate validated for evaluation on Liquid data: `column OP literal`, `column [NOT] LIKE pattern`
    or a boolean literal on byte-like columns.  `try_new` returns None where the reference's
    `LiquidExpr::try_new` returns None (liquid_expr.rs:65-148).

    `lhs` (round 5): the column-like forms the reference also accepts — Cast / CastColumn / TryCast wrappers and
    `to_timestamp_seconds(col)` (liquid_expr.rs:150-174).  The reference evaluates such a predicate by running the cast on
    the decoded array (eval_predicate_on_array); here the wrappers are peeled off on the host and the literal is moved
    into the column's own domain with the exact rounding rule per operator (`CAST(d AS TIMESTAMP) > t` is `d > floor(t /
    86400 s)`, `CAST(i AS DOUBLE) >= 12.5` is `i >= 13`, `= 12.5` is constant false ...), so that the device kernels
    evaluate `col OP literal'` on the encoded data.  Forms without an exact rewrite (narrowing casts, Int64 -> Double,
    time zones) return None: the caller keeps the reference's CPU path, exactly as for an unsupported expression."""

    def __init__(self, op: int, lit_tag: int, lit_bytes: bytes):
        self.op, self.lit_tag, self.lit_bytes = op, lit_tag, lit_bytes
        self._buf = C.create_string_buffer(lit_bytes, max(len(lit_bytes), 1))

    def as_predicate(self) -> N.Predicate:
        return N.Predicate(self.op, self.lit_tag, C.cast(self._buf, C.c_void_p), len(self.lit_bytes))

    @staticmethod
    def try_new(op: Union[str, int, None], literal, data_type: pa.DataType,
                expression_hint: Optional[int] = None, lhs: Optional[Column] = None) -> Optional["LiquidExpr"]:
        if op is None:  # Literal(Boolean) (liquid_expr.rs:78-80)
            if isinstance(literal, bool) and _is_byte_like(data_type):
                return LiquidExpr(N.OP_EQ, N.LIT_BOOL, bytes([1 if literal else 0]))
            return None
        code = _OPS.get(op.lower()) if isinstance(op, str) else op
        if code is None:
            return None
        if lhs is not None and type(lhs) is not Column:
            if _is_byte_like(data_type) and isinstance(lhs, ToTimestampSeconds):
                return None  # (byte-like columns take is_column_like only, liquid_expr.rs:92-94)
            norm = _normalise_lhs(code, literal, data_type, lhs)
            if norm is None:
                return None
            code, literal, constant = norm
            if constant is not None:
                # every valid row answers `constant`: a literal outside the column type's range says exactly that
                lo_hi = _int_range(data_type)
                if lo_hi is None:
                    if not pa.types.is_floating(data_type):
                        return None
                    # floats: no value equals NaN under IEEE, but Arrow's total order does — use an interval that is empty
                    return None
                code = N.OP_NE if constant else N.OP_EQ
                literal = -1 if lo_hi[0] == 0 else _BIG
        if _is_byte_like(data_type):
            if isinstance(literal, str):
                literal = literal.encode("utf-8")
            if not isinstance(literal, (bytes, bytearray)):
                return None
            if code in (N.OP_LIKE, N.OP_NOT_LIKE) and expression_hint != N.HINT_SUBSTRING_SEARCH:
                return None  # LIKE needs the SubstringSearch hint (liquid_expr.rs:106-109, 135-137)
            return LiquidExpr(code, N.LIT_BYTES, bytes(literal))
        if _is_numeric_like(data_type):
            if code > N.OP_GE:
                return None
            enc = _encode_numeric_literal(literal, data_type)
            if enc is None:
                return None
            return LiquidExpr(code, enc[0], enc[1])
        return None


def _encode_numeric_literal(v, t: pa.DataType):
    try:
        if pa.types.is_floating(t):
            if t.bit_width == 32:
                return N.LIT_F32, np.float32(v).tobytes()
            return N.LIT_F64, np.float64(v).tobytes()
        if pa.types.is_decimal(t):
            d = v if isinstance(v, decimal.Decimal) else decimal.Decimal(str(v))
            unscaled = int(d.scaleb(t.scale).to_integral_value())
            if d.scaleb(t.scale) != unscaled:
                return None
            return N.LIT_I128, (unscaled & ((1 << 128) - 1)).to_bytes(16, "little")
        if pa.types.is_date32(t) and isinstance(v, datetime.date):
            v = (v - datetime.date(1970, 1, 1)).days
        if pa.types.is_date64(t) and isinstance(v, datetime.date):
            v = (v - datetime.date(1970, 1, 1)).days * 86_400_000
        if pa.types.is_timestamp(t) and isinstance(v, datetime.datetime):
            v = pa.scalar(v, type=t).value
        iv = int(v)
        if pa.types.is_unsigned_integer(t):
            if iv < 0:
                return N.LIT_I64, int(max(iv, -(1 << 63))).to_bytes(8, "little", signed=True)
            return N.LIT_U64, min(iv, (1 << 64) - 1).to_bytes(8, "little")
        if iv > (1 << 63) - 1:
            return N.LIT_U64, min(iv, (1 << 64) - 1).to_bytes(8, "little")
        return N.LIT_I64, max(iv, -(1 << 63)).to_bytes(8, "little", signed=True)
    except (TypeError, ValueError, decimal.InvalidOperation, pa.ArrowInvalid):
        return None


def _selection_bytes(selection, n: Optional[int] = None) -> np.ndarray:
    """BooleanBuffer -> LSB-first bitmap bytes (a BooleanBuffer is not nullable)."""
    if isinstance(selection, pa.BooleanArray):
        if selection.null_count:
            raise ValueError("selection must not contain nulls")
        b = selection.to_numpy(zero_copy_only=False)
    else:
        b = np.asarray(selection, dtype=bool)
    if n is not None and len(b) != n:
        raise ValueError(f"selection has {len(b)} bits, entry has {n} rows")
    out = np.packbits(b, bitorder="little") if b.size else np.zeros(0, np.uint8)
    return np.ascontiguousarray(np.concatenate([out, np.zeros(8, np.uint8)]))


def _bits_to_bool(buf: np.ndarray, n: int) -> np.ndarray:
    if n == 0:
        return np.zeros(0, dtype=bool)
    return np.unpackbits(buf, bitorder="little")[:n].astype(bool)


class LiquidCacheBuilder:
    """`LiquidCacheBuilder::new()...build()` — options that belong to the cache manager (policies, disk store) are
    out of scope here; batch size and the memory budget are kept."""

    # process-wide defaults of `options` (a test session pins LC_OPT_LIKE_INDEX_ASYNC = 0 here so that WHICH kernel answers an
    # evaluation does not depend on how far a background build has come; the library itself reads no environment variable)
    default_options: dict = {}

    def __init__(self):
        self.batch_size = 8192
        self.max_memory_bytes = 0  # 0: bounded by HBM only (the reference default is 1 GiB of host RAM)
        self.device: Optional[int] = None
        self.host_only = False
        # lc_ctx_set_option(option, value) pairs applied right after the context is created
        self.options = dict(LiquidCacheBuilder.default_options)

    def with_option(self, option: int, value: int) -> "LiquidCacheBuilder":
        """Any lc_ctx_set_option pair (N.OPT_*), applied when the context is created."""
        self.options[int(option)] = int(value)
        return self

    @staticmethod
    def new() -> "LiquidCacheBuilder":
        return LiquidCacheBuilder()

    def with_batch_size(self, batch_size: int) -> "LiquidCacheBuilder":
        self.batch_size = batch_size
        return self

    def with_max_memory_bytes(self, n: int) -> "LiquidCacheBuilder":
        self.max_memory_bytes = n
        return self

    def with_device(self, device: int) -> "LiquidCacheBuilder":
        self.device = device
        return self

    def with_host_only(self) -> "LiquidCacheBuilder":
        """Transcoding / symbol tables only (no GPU): every staging or evaluation call raises."""
        self.host_only = True
        return self

    def with_index_options(self, signatures: Optional[bool] = None, row_lists: Optional[bool] = None,
                           host_built: Optional[bool] = None, like_pipeline_min_entries: Optional[int] = None,
                           like_path: Optional[int] = None) -> "LiquidCacheBuilder":
        """The device-side acceleration structures of substring-search byte views (include/liquid_cache_amd.h,
        lc_ctx_set_option): bigram signature index, inverted row lists, host-built index.  Results never depend on them."""
        if signatures is not None:
            self.options[N.OPT_SIGNATURE_INDEX] = int(bool(signatures))
        if row_lists is not None:
            self.options[N.OPT_ROW_LISTS] = int(bool(row_lists))
        if host_built is not None:
            self.options[N.OPT_HOST_BUILT_INDEX] = int(bool(host_built))
        if like_pipeline_min_entries is not None:
            self.options[N.OPT_LIKE_PIPELINE_MIN_ENTRIES] = int(like_pipeline_min_entries)
        if like_path is not None:
            self.options[N.OPT_LIKE_PATH] = int(like_path)
        return self

    def build(self) -> "LiquidCache":
        return LiquidCache(self)


class _Get:
    def __init__(self, cache: "LiquidCache", entry_id: int):
        self._cache, self._id, self._sel, self._hint = cache, entry_id, None, None

    def with_selection(self, selection) -> "_Get":
        self._sel = selection
        return self

    def with_expression_hint(self, hint) -> "_Get":
        """builders.rs:242-246; only ExtractDate32 changes what a read returns (core.rs:725-745)."""
        self._hint = hint
        return self

    def read(self) -> Optional[pa.Array]:
        field = self._hint.field if isinstance(self._hint, ExtractDate32) else None
        return self._cache._read_arrow_array(self._id, self._sel, date_field=field)


class _EvaluatePredicate:
    def __init__(self, cache: "LiquidCache", entry_id: int, expr: LiquidExpr):
        self._cache, self._id, self._expr, self._sel = cache, entry_id, expr, None

    def with_selection(self, selection) -> "_EvaluatePredicate":
        self._sel = selection
        return self

    def read(self) -> Optional[pa.BooleanArray]:
        return self._cache._eval_predicate_internal(self._id, self._sel, self._expr)


class LiquidCache:
    def __init__(self, builder: LiquidCacheBuilder):
        self._lib = N.load()
        self._batch_size = builder.batch_size
        ctx = C.c_void_p()
        if builder.host_only:
            st = self._lib.lc_ctx_create(None, 0, builder.max_memory_bytes, C.byref(ctx))
        elif builder.device is None:
            st = self._lib.lc_ctx_create(None, 1, builder.max_memory_bytes, C.byref(ctx))
        else:
            dev = (C.c_int32 * 1)(builder.device)
            st = self._lib.lc_ctx_create(dev, 1, builder.max_memory_bytes, C.byref(ctx))
        N.check(st, None)
        self._ctx = ctx
        self._types = {}
        for opt, val in builder.options.items():
            N.check(self._lib.lc_ctx_set_option(self._ctx, opt, val), self._ctx)

    def set_option(self, option: int, value: int):
        """lc_ctx_set_option: evaluation options may change at any time (N.OPT_LIKE_PATH, N.OPT_LIKE_INDEX_BUDGET_BYTES, ...)."""
        N.check(self._lib.lc_ctx_set_option(self._ctx, int(option), int(value)), self._ctx)

    # -- lifecycle ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            # scans hold device buffers of this context: destroy the ones still open before the context goes away
            for ref in list(getattr(self, "_scans", [])):
                scan = ref()
                if scan is not None:
                    scan.close()
            self._scans = []
            self._lib.lc_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def batch_size(self) -> int:
        return self._batch_size

    @property
    def handle(self) -> C.c_void_p:
        return self._ctx

    def device_info(self) -> N.DeviceInfo:
        info = N.DeviceInfo()
        N.check(self._lib.lc_device_info_get(self._ctx, C.byref(info)), self._ctx)
        return info

    # -- staging -----------------------------------------------------------------------------------
    def set_symbol_table(self, path_id: int, table_bytes: bytes):
        buf = (C.c_uint8 * len(table_bytes)).from_buffer_copy(table_bytes)
        N.check(self._lib.lc_symtab_set(self._ctx, path_id, buf, len(table_bytes)), self._ctx)

    def symbol_table(self, path_id: int) -> Optional[bytes]:
        out, ln = C.c_void_p(), C.c_size_t()
        st = self._lib.lc_symtab_get(self._ctx, path_id, C.byref(out), C.byref(ln))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        data = C.string_at(out, ln.value)
        self._lib.lc_free(out)
        return data

    def stage(self, entry_ids: Sequence[int], liquid_bytes: Sequence[bytes], path_ids: Optional[Sequence[int]] = None,
              data_types: Optional[Sequence[pa.DataType]] = None, index_bytes: Optional[Sequence[Optional[bytes]]] = None):
        """Stage serialized LiquidArrays (`LiquidArray::to_bytes()` output) in HBM; `index_bytes[i]` (optional) is the
        acceleration index `entry_index_bytes` returned for the same bytes (lc_stage_indexed)."""
        n = len(entry_ids)
        if index_bytes is not None:
            return self._stage_indexed(entry_ids, liquid_bytes, path_ids, data_types, index_bytes)
        ids = (C.c_uint64 * n)(*[int(e) for e in entry_ids])
        keep = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in liquid_bytes]
        ptrs = (C.c_void_p * n)(*[C.cast(k, C.c_void_p) for k in keep])
        lens = (C.c_size_t * n)(*[len(b) for b in liquid_bytes])
        if path_ids is None:
            path_ids = [ParquetArrayID.column_access_path(e) for e in entry_ids]
        pids = (C.c_uint64 * n)(*[int(p) for p in path_ids])
        N.check(self._lib.lc_stage(self._ctx, n, ids, ptrs, lens, pids), self._ctx)
        if data_types is not None:
            for e, t in zip(entry_ids, data_types):
                self._types[int(e)] = t

    def _stage_indexed(self, entry_ids, liquid_bytes, path_ids, data_types, index_bytes):
        n = len(entry_ids)
        ids = (C.c_uint64 * n)(*[int(e) for e in entry_ids])
        keep = [(C.c_uint8 * len(b)).from_buffer_copy(b) for b in liquid_bytes]
        ptrs = (C.c_void_p * n)(*[C.cast(k, C.c_void_p) for k in keep])
        lens = (C.c_size_t * n)(*[len(b) for b in liquid_bytes])
        if path_ids is None:
            path_ids = [ParquetArrayID.column_access_path(e) for e in entry_ids]
        pids = (C.c_uint64 * n)(*[int(p) for p in path_ids])
        ikeep = [None if b is None else (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0") for b in index_bytes]
        iptrs = (C.c_void_p * n)(*[None if k is None else C.cast(k, C.c_void_p) for k in ikeep])
        ilens = (C.c_size_t * n)(*[0 if b is None else len(b) for b in index_bytes])
        N.check(self._lib.lc_stage_indexed(self._ctx, n, ids, ptrs, lens, pids, iptrs, ilens), self._ctx)
        if data_types is not None:
            for e, t in zip(entry_ids, data_types):
                self._types[int(e)] = t

    def entry_index_bytes(self, entry_id: int) -> Optional[bytes]:
        """The entry's device-side acceleration index as an opaque blob (lc_entry_index_to_bytes); b"" if it has none."""
        out, ln = C.c_void_p(), C.c_size_t()
        st = self._lib.lc_entry_index_to_bytes(self._ctx, int(entry_id), C.byref(out), C.byref(ln))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        data = C.string_at(out, ln.value) if ln.value else b""
        if out.value:
            self._lib.lc_free(out)
        return data

    def transcode(self, array: pa.Array, squeeze_hint: Optional[int] = None, path_id: int = 0) -> Optional[bytes]:
        """Arrow -> Liquid bytes (transcode_liquid_inner_with_hint); None for types that stay Arrow."""
        c_arr, c_schema = N.ArrowArray(), N.ArrowSchema()
        array._export_to_c(C.addressof(c_arr), C.addressof(c_schema))
        out, ln = C.c_void_p(), C.c_size_t()
        try:
            st = self._lib.lc_transcode_arrow(self._ctx, C.addressof(c_arr), C.addressof(c_schema),
                                              squeeze_hint or N.HINT_NONE, path_id, C.byref(out), C.byref(ln))
        finally:
            _release(c_arr, c_schema)
        if st == N.LC_UNSUPPORTED:
            return None
        N.check(st, self._ctx)
        data = C.string_at(out, ln.value)
        self._lib.lc_free(out)
        return data

    def insert(self, entry_id: int, batch_to_cache: pa.Array, squeeze_hint: Optional[int] = None,
               path_id: Optional[int] = None):
        """`cache.insert(entry_id, array)` with eager transcoding (bench mode `liquid_eager_transcode`)."""
        if isinstance(batch_to_cache, pa.ChunkedArray):
            batch_to_cache = batch_to_cache.combine_chunks()
        pid = ParquetArrayID.column_access_path(entry_id) if path_id is None else path_id
        c_arr, c_schema = N.ArrowArray(), N.ArrowSchema()
        batch_to_cache._export_to_c(C.addressof(c_arr), C.addressof(c_schema))
        try:
            st = self._lib.lc_insert_arrow(self._ctx, int(entry_id), C.addressof(c_arr), C.addressof(c_schema),
                                           squeeze_hint or N.HINT_NONE, pid)
        finally:
            _release(c_arr, c_schema)
        N.check(st, self._ctx)
        self._types[int(entry_id)] = batch_to_cache.type

    def insert_batch(self, entry_ids: Sequence[int], arrays: Sequence[pa.Array], squeeze_hint: Optional[int] = None,
                     path_ids: Optional[Sequence[int]] = None):
        """`cache.insert` for the batches of one row group in ONE call (lc_insert_arrow_batch): transcoded one after the
        other, staged together (one upload, one signature-builder launch)."""
        n = len(entry_ids)
        c_arrs = [N.ArrowArray() for _ in range(n)]
        c_schemas = [N.ArrowSchema() for _ in range(n)]
        arrays = [a.combine_chunks() if isinstance(a, pa.ChunkedArray) else a for a in arrays]
        try:
            for a, ca, cs in zip(arrays, c_arrs, c_schemas):
                a._export_to_c(C.addressof(ca), C.addressof(cs))
            ids = (C.c_uint64 * n)(*[int(e) for e in entry_ids])
            ap = (C.c_void_p * n)(*[C.addressof(x) for x in c_arrs])
            sp = (C.c_void_p * n)(*[C.addressof(x) for x in c_schemas])
            hints = (C.c_int32 * n)(*([squeeze_hint or N.HINT_NONE] * n))
            pids = (C.c_uint64 * n)(*[int(ParquetArrayID.column_access_path(e) if path_ids is None else path_ids[i])
                                      for i, e in enumerate(entry_ids)])
            st = self._lib.lc_insert_arrow_batch(self._ctx, n, ids, ap, sp, hints, pids)
        finally:
            for ca, cs in zip(c_arrs, c_schemas):
                _release(ca, cs)
        N.check(st, self._ctx)
        for e, a in zip(entry_ids, arrays):
            self._types[int(e)] = a.type

    def insert_device(self, entry_ids: Sequence[int], arrays: Sequence[pa.Array], squeeze_hint: Optional[int] = None,
                      path_ids: Optional[Sequence[int]] = None):
        """`cache.insert` for a batch of arrays with the transcoding done ON THE DEVICE (lc_insert_arrow_batch_device):
        raw values cross PCIe once; min / max, ALP, FastLanes packing and — for Utf8 / Binary arrays — the dictionary,
        FSST compression, prefix keys, fingerprints and compact offsets run as kernels.  Raises
        LiquidCacheError(LC_UNSUPPORTED) for other types (use `insert`)."""
        n = len(entry_ids)
        c_arrs = [N.ArrowArray() for _ in range(n)]
        c_schemas = [N.ArrowSchema() for _ in range(n)]
        try:
            for a, ca, cs in zip(arrays, c_arrs, c_schemas):
                if isinstance(a, pa.ChunkedArray):
                    a = a.combine_chunks()
                a._export_to_c(C.addressof(ca), C.addressof(cs))
            ids = (C.c_uint64 * n)(*[int(e) for e in entry_ids])
            ap = (C.c_void_p * n)(*[C.addressof(x) for x in c_arrs])
            sp = (C.c_void_p * n)(*[C.addressof(x) for x in c_schemas])
            hints = (C.c_int32 * n)(*([squeeze_hint or N.HINT_NONE] * n))
            pids = (C.c_uint64 * n)(*[int(ParquetArrayID.column_access_path(e) if path_ids is None else path_ids[i])
                                      for i, e in enumerate(entry_ids)])
            st = self._lib.lc_insert_arrow_batch_device(self._ctx, n, ids, ap, sp, hints, pids)
        finally:
            for ca, cs in zip(c_arrs, c_schemas):
                _release(ca, cs)
        N.check(st, self._ctx)
        for e, a in zip(entry_ids, arrays):
            self._types[int(e)] = a.type

    def entry_bytes(self, entry_id: int) -> Optional[bytes]:
        """`LiquidArray::to_bytes()` of a staged fixed-width entry, rebuilt from HBM (lc_entry_to_liquid_bytes)."""
        out, ln = C.c_void_p(), C.c_size_t()
        st = self._lib.lc_entry_to_liquid_bytes(self._ctx, int(entry_id), C.byref(out), C.byref(ln))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        data = C.string_at(out, ln.value)
        self._lib.lc_free(out)
        return data

    def squeeze_date(self, entry_ids: Sequence[int], field):
        """Squeeze Date32 / Timestamp entries to one calendar component in HBM (LiquidPrimitiveArray::squeeze with an
        ExtractDate32 hint).  Afterwards only `get(...).with_expression_hint(extract_date32(field))` is served; other
        reads raise LiquidCacheError(LC_NEEDS_BACKING)."""
        f = field.field if isinstance(field, ExtractDate32) else (Date32Field._NAMES[field.lower()] if isinstance(field, str) else int(field))
        ids = (C.c_uint64 * len(entry_ids))(*[int(e) for e in entry_ids])
        N.check(self._lib.lc_squeeze_date(self._ctx, len(entry_ids), ids, f), self._ctx)

    def squeeze_clamp(self, entry_ids: Sequence[int]) -> int:
        """Squeeze integer entries to half their bit width with the Clamp policy (lc_squeeze_clamp); returns how many
        qualified.  Reads / predicates that need the clamped-away bits raise LiquidCacheError(LC_NEEDS_BACKING)."""
        ids = (C.c_uint64 * len(entry_ids))(*[int(e) for e in entry_ids])
        k = C.c_uint64()
        N.check(self._lib.lc_squeeze_clamp(self._ctx, len(entry_ids), ids, C.byref(k)), self._ctx)
        return int(k.value)

    def squeeze_quantize(self, entry_ids: Sequence[int]) -> int:
        """Squeeze integer entries to half their bit width with the Quantize policy (lc_squeeze_quantize, the
        reference's default IntegerSqueezePolicy); returns how many qualified.  Predicates the literal's bucket cannot
        decide, and every read, raise LiquidCacheError(LC_NEEDS_BACKING)."""
        ids = (C.c_uint64 * len(entry_ids))(*[int(e) for e in entry_ids])
        k = C.c_uint64()
        N.check(self._lib.lc_squeeze_quantize(self._ctx, len(entry_ids), ids, C.byref(k)), self._ctx)
        return int(k.value)

    def evict(self, entry_ids: Iterable[int]):
        ids = [int(e) for e in entry_ids]
        arr = (C.c_uint64 * len(ids))(*ids)
        N.check(self._lib.lc_evict(self._ctx, len(ids), arr), self._ctx)
        for e in ids:
            self._types.pop(e, None)

    def entry_info(self, entry_id: int) -> Optional[N.EntryInfo]:
        info = N.EntryInfo()
        st = self._lib.lc_entry_info_get(self._ctx, int(entry_id), C.byref(info))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        return info

    def is_cached(self, entry_id: int) -> bool:
        return self.entry_info(entry_id) is not None

    # -- the two hot-path calls ----------------------------------------------------------------------
    def get(self, entry_id: int) -> _Get:
        return _Get(self, int(entry_id))

    def eval_predicate(self, entry_id: int, predicate: LiquidExpr) -> _EvaluatePredicate:
        return _EvaluatePredicate(self, int(entry_id), predicate)

    def _eval_predicate_internal(self, entry_id: int, selection, expr: LiquidExpr) -> Optional[pa.BooleanArray]:
        info = self.entry_info(entry_id)
        if info is None:
            return None
        n = info.len
        sel = _selection_bytes(selection, n) if selection is not None else None
        nb = (n + 7) // 8 + 8
        values = np.zeros(nb, np.uint8)
        validity = np.zeros(nb, np.uint8)
        out_len, nullable = C.c_uint32(), C.c_int32()
        pred = expr.as_predicate()
        st = self._lib.lc_eval_predicate(self._ctx, entry_id, C.byref(pred),
                                         sel.ctypes.data_as(C.c_void_p) if sel is not None else None,
                                         values.ctypes.data_as(C.c_void_p), validity.ctypes.data_as(C.c_void_p),
                                         C.byref(out_len), C.byref(nullable))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        k = out_len.value
        v = _bits_to_bool(values, k)
        if nullable.value:
            valid = _bits_to_bool(validity, k)
            return pa.array(v, type=pa.bool_(), mask=~valid)
        return pa.array(v, type=pa.bool_())

    def eval_predicate_or(self, entry_ids: Sequence[int], exprs: Sequence[LiquidExpr], selection=None
                          ) -> Optional[pa.BooleanArray]:
        """A predicate that is an OR over the columns of one batch: entry_ids[i] / exprs[i] are its column-literal
        disjuncts (CachedRowGroup::evaluate_selection_with_predicate, cache/mod.rs:111-150).  Kleene OR of the per-column
        results over the same selection; None when an entry is not cached."""
        infos = [self.entry_info(e) for e in entry_ids]
        if any(i is None for i in infos):
            return None
        n = infos[0].len
        sel = _selection_bytes(selection, n) if selection is not None else None
        nb = (n + 7) // 8 + 8
        values, validity = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
        out_len, nullable = C.c_uint32(), C.c_int32()
        ids = (C.c_uint64 * len(entry_ids))(*[int(e) for e in entry_ids])
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        st = self._lib.lc_eval_predicate_or(self._ctx, len(entry_ids), ids, preds,
                                            sel.ctypes.data_as(C.c_void_p) if sel is not None else None,
                                            values.ctypes.data_as(C.c_void_p), validity.ctypes.data_as(C.c_void_p),
                                            C.byref(out_len), C.byref(nullable))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        k = out_len.value
        v = _bits_to_bool(values, k)
        if nullable.value:
            return pa.array(v, type=pa.bool_(), mask=~_bits_to_bool(validity, k))
        return pa.array(v, type=pa.bool_())

    def _read_arrow_array(self, entry_id: int, selection, date_field: Optional[int] = None) -> Optional[pa.Array]:
        info = self.entry_info(entry_id)
        if info is None:
            return None
        sel = _selection_bytes(selection, info.len) if selection is not None else None
        c_arr, c_schema = N.ArrowArray(), N.ArrowSchema()
        sel_ptr = sel.ctypes.data_as(C.c_void_p) if sel is not None else None
        if date_field is None:
            st = self._lib.lc_get_with_selection(self._ctx, entry_id, sel_ptr, C.addressof(c_arr), C.addressof(c_schema))
        else:
            st = self._lib.lc_get_date_part_with_selection(self._ctx, entry_id, sel_ptr, int(date_field),
                                                           C.addressof(c_arr), C.addressof(c_schema))
        if st == N.LC_NOT_STAGED:
            return None
        N.check(st, self._ctx)
        arr = pa.Array._import_from_c(C.addressof(c_arr), C.addressof(c_schema))
        want = self._types.get(int(entry_id))
        if want is not None and arr.type != want:
            arr = arr.cast(want)
        return arr

    # -- scans ---------------------------------------------------------------------------------------
    def scan(self, entry_ids: Sequence[int]) -> "Scan":
        return Scan(self, entry_ids)

    def eval_predicate_row_groups(self, entry_ids, group_ends, exprs, want_mask: bool = False):
        """lc_eval_predicate_row_groups: the predicate over MANY row groups of one column in one call, host in / host out
        (row group g = entry_ids[group_ends[g-1]:group_ends[g]]).  Returns (per-row-group hit counts, total, mask words or
        None); the scan behind the call comes from the context's scan cache."""
        if isinstance(exprs, LiquidExpr):
            exprs = [exprs]
        if isinstance(entry_ids, np.ndarray) and entry_ids.dtype == np.uint64:
            ids = np.ascontiguousarray(entry_ids)
        else:
            ids = np.ascontiguousarray(np.asarray([int(e) for e in entry_ids], dtype=np.uint64))
        ends = np.ascontiguousarray(group_ends, dtype=np.uint32)
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        counts = np.zeros(max(len(ends), 1), np.uint64)
        total = C.c_uint64(0)
        mask, words = None, 0
        if want_mask:
            infos = [self.entry_info(int(e)) for e in ids]
            if any(i is None for i in infos):
                raise N.LiquidCacheError(N.LC_NOT_STAGED, "an entry of the list is not staged")
            words = int(sum((int(i.len) + 63) // 64 for i in infos))
            mask = np.zeros(max(words, 1), np.uint64)
        st = N.check(self._lib.lc_eval_predicate_row_groups(
            self._ctx, len(ids), ids.ctypes.data_as(C.POINTER(C.c_uint64)), len(ends), ends.ctypes.data_as(C.POINTER(C.c_uint32)),
            preds, len(exprs), counts.ctypes.data_as(C.POINTER(C.c_uint64)),
            mask.ctypes.data_as(C.POINTER(C.c_uint64)) if want_mask else None, words, C.byref(total)), self._ctx)
        if st == N.LC_NOT_STAGED:
            raise N.LiquidCacheError(st, "an entry of the list is not staged")
        return counts[:len(ends)], int(total.value), (mask[:words] if want_mask else None)


def _release(c_arr: N.ArrowArray, c_schema: N.ArrowSchema):
    for s in (c_arr, c_schema):
        if s.release:
            C.CFUNCTYPE(None, C.c_void_p)(s.release)(C.addressof(s))


class Scan:
    """Device-resident column scan (lc_scan_*): ordered entries of one column, hit mask kept in HBM."""

    def __init__(self, cache: LiquidCache, entry_ids: Sequence[int]):
        self._cache = cache
        self._lib = cache._lib
        if isinstance(entry_ids, np.ndarray) and entry_ids.dtype == np.uint64:
            ids = np.ascontiguousarray(entry_ids)  # (a host that creates a scan per query keeps its id arrays)
        else:
            ids = np.ascontiguousarray(np.asarray([int(e) for e in entry_ids], dtype=np.uint64))
        h = C.c_void_p()
        st = N.check(self._lib.lc_scan_create(cache.handle, len(ids), ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              C.byref(h)), cache.handle)
        if st == N.LC_NOT_STAGED:  # (the reference's `None`: a scan needs every entry of its list)
            raise N.LiquidCacheError(st, "an entry of the list is not staged")
        self._h = h
        if not hasattr(cache, "_scans"):
            cache._scans = []
        cache._scans.append(weakref.ref(self))
        self.entries = len(ids)
        self.rows = self._lib.lc_scan_rows(h)
        self.mask_words = self._lib.lc_scan_mask_words(h)
        seg = self._lib.lc_scan_segment_offsets(h)
        self.segment_offsets = np.ctypeslib.as_array(seg, shape=(self.entries + 1,)).copy()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lc_scan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> N.ScanInfo:
        """lc_scan_info_get: entries, rows, HBM of the entries and of the scan-level LIKE indexes."""
        out = N.ScanInfo()
        N.check(self._lib.lc_scan_info_get(self._h, C.byref(out)), self._cache.handle)
        return out

    def index_wait(self):
        """lc_scan_index_wait: block until the index builds in flight for this scan are in place (LC_OPT_LIKE_INDEX_ASYNC)."""
        N.check(self._lib.lc_scan_index_wait(self._h), self._cache.handle)

    def algorithmic_bytes(self, expr: LiquidExpr, with_selection: bool = False) -> int:
        pred = expr.as_predicate()
        return self._lib.lc_scan_algorithmic_bytes(self._h, C.byref(pred), int(with_selection))

    def explain(self, expr: LiquidExpr) -> str:
        """Which kernels evaluate `expr` over this scan (lc_scan_explain)."""
        buf = C.create_string_buffer(512)
        pred = expr.as_predicate()
        N.check(self._lib.lc_scan_explain(self._h, C.byref(pred), buf, 512), self._cache.handle)
        return buf.value.decode()

    def traffic_model(self, expr: LiquidExpr, with_selection: bool = False, no_mask: bool = False, hit_list: bool = False):
        """(algorithmic bytes of the reference algorithm, bytes this kernel itself moves) for one evaluation.  `no_mask` /
        `hit_list`: the evaluation is a COUNT(*) / hit-list call without a mask output (LC_TRAFFIC_NO_MASK / _HIT_LIST)."""
        pred = expr.as_predicate()
        alg, own = C.c_uint64(), C.c_uint64()
        flags = (1 if with_selection else 0) | (2 if no_mask else 0) | (4 if hit_list else 0)
        N.check(self._lib.lc_scan_traffic_model(self._h, C.byref(pred), flags, C.byref(alg),
                                                C.byref(own)), self._cache.handle)
        return int(alg.value), int(own.value)

    def eval(self, expr: LiquidExpr, mask_out_ptr: int, selection_ptr: int = 0, counts_ptr: int = 0,
             stream: int = 0):
        """Asynchronous: raw device pointers (e.g. torch tensor .data_ptr()) and a hipStream_t handle."""
        pred = expr.as_predicate()
        N.check(self._lib.lc_scan_eval(self._cache.handle, self._h, C.byref(pred), C.c_void_p(selection_ptr or None),
                                       C.c_void_p(mask_out_ptr), C.c_void_p(counts_ptr or None),
                                       C.c_void_p(stream or None)), self._cache.handle)

    def eval_and(self, exprs: Sequence[LiquidExpr], mask_out_ptr: int, selection_ptr: int = 0, counts_ptr: int = 0,
                 stream: int = 0) -> bool:
        """One pass for a conjunction of one or two predicates on this column (lc_scan_eval_and).  Returns False when
        the pair cannot be fused (the caller chains two `eval` calls instead)."""
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        st = self._lib.lc_scan_eval_and(self._cache.handle, self._h, preds, len(exprs), C.c_void_p(selection_ptr or None),
                                        C.c_void_p(mask_out_ptr), C.c_void_p(counts_ptr or None),
                                        C.c_void_p(stream or None))
        if st == N.LC_UNSUPPORTED:
            return False
        N.check(st, self._cache.handle)
        return True

    def aggregate(self, out_ptr: int, selection_ptr: int = 0, stream: int = 0):
        """COUNT / SUM / MIN / MAX of the valid selected rows (lc_scan_aggregate): six u64 at `out_ptr` on the device,
        {count, sum lo, sum hi, min, max, 0}.  Asynchronous."""
        N.check(self._lib.lc_scan_aggregate(self._cache.handle, self._h, C.c_void_p(selection_ptr or None),
                                            C.c_void_p(out_ptr), C.c_void_p(stream or None)), self._cache.handle)

    def sum_product(self, other: "Scan", out_ptr: int, selection_ptr: int = 0, stream: int = 0):
        """SUM(self * other) over the selected rows valid in both columns (lc_scan_sum_product): {count, sum lo, sum hi,
        0, 0, 0} as six u64 at `out_ptr`.  Asynchronous."""
        N.check(self._lib.lc_scan_sum_product(self._cache.handle, self._h, other._h, C.c_void_p(selection_ptr or None),
                                              C.c_void_p(out_ptr), C.c_void_p(stream or None)), self._cache.handle)

    def sum_product_to_host(self, other: "Scan", selection_ptr: int = 0) -> dict:
        lib, ctx = self._lib, self._cache.handle
        buf = C.c_void_p()
        N.check(lib.lc_device_alloc(ctx, 48, C.byref(buf)), ctx)
        try:
            self.sum_product(other, buf.value, selection_ptr)
            host = (C.c_uint64 * 6)()
            N.check(lib.lc_device_to_host(ctx, C.cast(host, C.c_void_p), buf, 48, None), ctx)
        finally:
            lib.lc_device_free(ctx, buf)
        total = (int(host[2]) << 64) | int(host[1])
        if total >= 1 << 127:
            total -= 1 << 128
        return {"count": int(host[0]), "sum": total}

    def aggregate_to_host(self, selection_ptr: int = 0, signed: bool = True) -> dict:
        """Convenience: run `aggregate` and return {count, sum, min, max} as Python ints (None when count == 0)."""
        lib, ctx = self._lib, self._cache.handle
        buf = C.c_void_p()
        N.check(lib.lc_device_alloc(ctx, 48, C.byref(buf)), ctx)
        try:
            self.aggregate(buf.value, selection_ptr)
            host = (C.c_uint64 * 6)()
            N.check(lib.lc_device_to_host(ctx, C.cast(host, C.c_void_p), buf, 48, None), ctx)
        finally:
            lib.lc_device_free(ctx, buf)
        count, lo, hi, mn, mx = (int(host[i]) for i in range(5))
        total = (hi << 64) | lo
        if total >= 1 << 127:
            total -= 1 << 128
        if signed:
            mn = mn - (1 << 64) if mn >= 1 << 63 else mn
            mx = mx - (1 << 64) if mx >= 1 << 63 else mx
        return {"count": count, "sum": total if count else None, "min": mn if count else None, "max": mx if count else None}

    def eval_count(self, exprs, mask_out_ptr: int, total_out_ptr: int, selection_ptr: int = 0, counts_ptr: int = 0,
                   stream: int = 0):
        """Predicate pass (one predicate or a fusable pair) whose kernel also produces the COUNT(*) of the launch in
        `total_out_ptr` (one u64 on the device): lc_scan_eval_count."""
        if isinstance(exprs, LiquidExpr):
            exprs = [exprs]
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        N.check(self._lib.lc_scan_eval_count(self._cache.handle, self._h, preds, len(exprs),
                                             C.c_void_p(selection_ptr or None), C.c_void_p(mask_out_ptr),
                                             C.c_void_p(counts_ptr or None), C.c_void_p(total_out_ptr),
                                             C.c_void_p(stream or None)), self._cache.handle)

    def eval_count_groups(self, exprs, group_ends, group_counts_ptr: int, mask_out_ptr: int = 0, total_out_ptr: int = 0,
                          selection_ptr: int = 0, counts_ptr: int = 0, stream: int = 0):
        """lc_scan_eval_count_groups: COUNT(*) per ROW GROUP (group g = entries [group_ends[g-1], group_ends[g]) of the scan)
        from one evaluation launch; `group_counts_ptr`: len(group_ends) u64 on the device."""
        if isinstance(exprs, LiquidExpr):
            exprs = [exprs]
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        ends = np.ascontiguousarray(group_ends, dtype=np.uint32)
        N.check(self._lib.lc_scan_eval_count_groups(self._cache.handle, self._h, preds, len(exprs),
                                                    C.c_void_p(selection_ptr or None), len(ends),
                                                    ends.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(group_counts_ptr),
                                                    C.c_void_p(mask_out_ptr or None), C.c_void_p(counts_ptr or None),
                                                    C.c_void_p(total_out_ptr or None), C.c_void_p(stream or None)),
                self._cache.handle)

    @staticmethod
    def eval_or(scans: Sequence["Scan"], exprs: Sequence[LiquidExpr], mask_out_ptr: int, selection_ptr: int = 0,
                counts_ptr: int = 0, valid_out_ptr: int = 0, stream: int = 0):
        """Multi-column OR over whole scans (lc_scan_eval_or): scans[i] / exprs[i] are the disjuncts, all over the same
        row ranges; the same scan may appear twice (`col IN (a, b)`)."""
        cache = scans[0]._cache
        hs = (C.c_void_p * len(scans))(*[s._h for s in scans])
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        N.check(cache._lib.lc_scan_eval_or(cache.handle, len(scans), hs, preds, C.c_void_p(selection_ptr or None),
                                           C.c_void_p(mask_out_ptr), C.c_void_p(valid_out_ptr or None),
                                           C.c_void_p(counts_ptr or None), C.c_void_p(stream or None)), cache.handle)

    def eval_timed_cold(self, expr: LiquidExpr, mask_out_ptr: int, iters: int, flush_bytes: int = 1 << 30,
                        selection_ptr: int = 0, counts_ptr: int = 0, stream: int = 0) -> float:
        """Average kernel milliseconds per evaluation with the Infinity Cache flushed before every launch (bench aid:
        lc_bench_eval_timed of libliquid_cache_amd_bench.so, over the public scan API)."""
        return self._bench_timed(expr, mask_out_ptr, iters, flush_bytes, selection_ptr, counts_ptr, stream)

    def eval_timed(self, expr: LiquidExpr, mask_out_ptr: int, iters: int, selection_ptr: int = 0,
                   counts_ptr: int = 0, stream: int = 0) -> float:
        """Average kernel-side milliseconds per evaluation, HIP events on `stream`, launches back to back (bench aid)."""
        return self._bench_timed(expr, mask_out_ptr, iters, 0, selection_ptr, counts_ptr, stream)

    def _bench_timed(self, expr, mask_out_ptr, iters, flush_bytes, selection_ptr, counts_ptr, stream) -> float:
        pred = expr.as_predicate()
        ms = C.c_float()
        B = N.load_bench()
        N.check(B.lc_bench_eval_timed(self._cache.handle, self._h, C.cast(C.byref(pred), C.c_void_p),
                                      C.c_void_p(selection_ptr or None), C.c_void_p(mask_out_ptr),
                                      C.c_void_p(counts_ptr or None), C.c_void_p(stream or None), iters, flush_bytes,
                                      C.byref(ms)), self._cache.handle)
        return ms.value

    def gather_fixed(self, values_out_ptr: int, capacity_bytes: int, row_offsets_ptr: int, selection_ptr: int = 0,
                     stream: int = 0):
        """get-with-selection over the whole scan (fixed-width columns): decoded values of the selected rows, compacted
        in row order into `values_out_ptr` (device); `row_offsets_ptr` (entries+1 u64, device) receives the exclusive
        prefix sum of per-entry selected counts.  Asynchronous on `stream`."""
        N.check(self._lib.lc_scan_gather_fixed(self._cache.handle, self._h, C.c_void_p(selection_ptr or None),
                                               C.c_void_p(values_out_ptr), capacity_bytes, C.c_void_p(row_offsets_ptr),
                                               C.c_void_p(stream or None)), self._cache.handle)

    def date_part(self, values_ptr: int, n_values: int, field: int, stream: int = 0):
        """ExtractDate32 over gathered Date32 / Timestamp values, in place (lc_scan_date_part)."""
        N.check(self._lib.lc_scan_date_part(self._cache.handle, self._h, C.c_void_p(values_ptr), n_values, int(field),
                                            C.c_void_p(stream or None)), self._cache.handle)

    def gather_bytes_plan(self, row_offsets_ptr: int, row_refs_ptr: int, value_offsets_ptr: int, capacity_rows: int,
                          selection_ptr: int = 0, row_valid_ptr: int = 0, stream: int = 0):
        """First half of get-with-selection over a byte-view scan: (selected rows k, total decoded bytes)."""
        k, nbytes = C.c_uint64(), C.c_uint64()
        N.check(self._lib.lc_scan_gather_bytes_plan(self._cache.handle, self._h, C.c_void_p(selection_ptr or None),
                                                    C.c_void_p(row_offsets_ptr), C.c_void_p(row_refs_ptr),
                                                    C.c_void_p(value_offsets_ptr), C.c_void_p(row_valid_ptr or None),
                                                    capacity_rows, C.byref(k), C.byref(nbytes),
                                                    C.c_void_p(stream or None)), self._cache.handle)
        return int(k.value), int(nbytes.value)

    def gather_bytes(self, row_refs_ptr: int, value_offsets_ptr: int, rows: int, data_ptr: int, stream: int = 0):
        N.check(self._lib.lc_scan_gather_bytes(self._cache.handle, self._h, C.c_void_p(row_refs_ptr),
                                               C.c_void_p(value_offsets_ptr), rows, C.c_void_p(data_ptr),
                                               C.c_void_p(stream or None)), self._cache.handle)

    def gather_bytes_async(self, row_offsets_ptr: int, row_refs_ptr: int, value_offsets_ptr: int, capacity_rows: int,
                           data_ptr: int, capacity_bytes: int, selection_ptr: int = 0, row_valid_ptr: int = 0,
                           stream: int = 0):
        """Plan + fill in one asynchronous call (lc_scan_gather_bytes_async): no host round trip."""
        N.check(self._lib.lc_scan_gather_bytes_async(self._cache.handle, self._h, C.c_void_p(selection_ptr or None),
                                                     C.c_void_p(row_offsets_ptr), C.c_void_p(row_refs_ptr),
                                                     C.c_void_p(value_offsets_ptr), C.c_void_p(row_valid_ptr or None),
                                                     capacity_rows, C.c_void_p(data_ptr), capacity_bytes,
                                                     C.c_void_p(stream or None)), self._cache.handle)

    def gather_bytes_to_host(self, selection: Optional[np.ndarray] = None):
        """Convenience for tests: list of bytes / None for the selected rows, in row order."""
        lib, ctx = self._lib, self._cache.handle
        cap = int(self.rows) if selection is None else int(np.unpackbits(np.ascontiguousarray(selection, np.uint64).view(np.uint8)).sum())
        cap = max(cap, 1)
        ptrs = [C.c_void_p() for _ in range(6)]
        d_ro, d_refs, d_vo, d_valid, d_sel, d_data = ptrs
        N.check(lib.lc_device_alloc(ctx, (self.entries + 1) * 8, C.byref(d_ro)), ctx)
        N.check(lib.lc_device_alloc(ctx, cap * 8, C.byref(d_refs)), ctx)
        N.check(lib.lc_device_alloc(ctx, (cap + 1) * 8, C.byref(d_vo)), ctx)
        N.check(lib.lc_device_alloc(ctx, cap, C.byref(d_valid)), ctx)
        try:
            if selection is not None:
                sel = np.ascontiguousarray(selection, dtype=np.uint64)
                N.check(lib.lc_device_alloc(ctx, max(sel.size, 1) * 8, C.byref(d_sel)), ctx)
                N.check(lib.lc_host_to_device(ctx, d_sel, sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
            k, nbytes = self.gather_bytes_plan(d_ro.value, d_refs.value, d_vo.value, cap, d_sel.value or 0, d_valid.value)
            N.check(lib.lc_device_alloc(ctx, max(nbytes, 1), C.byref(d_data)), ctx)
            self.gather_bytes(d_refs.value, d_vo.value, k, d_data.value)
            offs = np.zeros(k + 1, np.uint64)
            valid = np.zeros(max(k, 1), np.uint8)
            data = np.zeros(max(nbytes, 1), np.uint8)
            N.check(lib.lc_device_to_host(ctx, offs.ctypes.data_as(C.c_void_p), d_vo, (k + 1) * 8, None), ctx)
            N.check(lib.lc_device_to_host(ctx, valid.ctypes.data_as(C.c_void_p), d_valid, k, None), ctx)
            N.check(lib.lc_device_to_host(ctx, data.ctypes.data_as(C.c_void_p), d_data, nbytes, None), ctx)
            # the one-call asynchronous form must produce the same buffers
            N.check(lib.lc_device_memset(ctx, d_data, 0, max(nbytes, 1), None), ctx)
            N.check(lib.lc_device_memset(ctx, d_vo, 0xEE, (cap + 1) * 8, None), ctx)
            self.gather_bytes_async(d_ro.value, d_refs.value, d_vo.value, cap, d_data.value, max(nbytes, 1),
                                    d_sel.value or 0, d_valid.value)
            offs2 = np.zeros(k + 1, np.uint64)
            data2 = np.zeros(max(nbytes, 1), np.uint8)
            N.check(lib.lc_device_to_host(ctx, offs2.ctypes.data_as(C.c_void_p), d_vo, (k + 1) * 8, None), ctx)
            N.check(lib.lc_device_to_host(ctx, data2.ctypes.data_as(C.c_void_p), d_data, nbytes, None), ctx)
            assert offs2.tolist() == offs.tolist() and data2[:nbytes].tobytes() == data[:nbytes].tobytes(), \
                "lc_scan_gather_bytes_async differs from plan + fill"
        finally:
            for p in ptrs:
                if p.value:
                    lib.lc_device_free(ctx, p)
        raw = data.tobytes()
        return [raw[int(offs[i]): int(offs[i + 1])] if valid[i] else None for i in range(k)]

    def gather_fixed_to_host(self, np_dtype, selection: Optional[np.ndarray] = None, date_field: Optional[int] = None):
        """Convenience for tests: (values ndarray of the selected rows, row offsets)."""
        lib, ctx = self._lib, self._cache.handle
        width = np.dtype(np_dtype).itemsize
        k_max = int(self.rows)
        d_vals, d_offs, d_sel = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(lib.lc_device_alloc(ctx, max(k_max, 1) * width + 64, C.byref(d_vals)), ctx)
        N.check(lib.lc_device_alloc(ctx, (self.entries + 1) * 8, C.byref(d_offs)), ctx)
        try:
            if selection is not None:
                sel = np.ascontiguousarray(selection, dtype=np.uint64)
                N.check(lib.lc_device_alloc(ctx, max(sel.size, 1) * 8, C.byref(d_sel)), ctx)
                N.check(lib.lc_host_to_device(ctx, d_sel, sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
            self.gather_fixed(d_vals.value, max(k_max, 1) * width, d_offs.value, d_sel.value or 0)
            offs = np.zeros(self.entries + 1, np.uint64)
            N.check(lib.lc_device_to_host(ctx, offs.ctypes.data_as(C.c_void_p), d_offs, offs.size * 8, None), ctx)
            k = int(offs[-1])
            if date_field is not None:
                self.date_part(d_vals.value, k, date_field)
            vals = np.zeros(max(k, 1), np_dtype)
            N.check(lib.lc_device_to_host(ctx, vals.ctypes.data_as(C.c_void_p), d_vals, k * width, None), ctx)
        finally:
            for p in (d_vals, d_offs, d_sel):
                if p.value:
                    lib.lc_device_free(ctx, p)
        return vals[:k], offs

    def group_partials(self, value_scan: Optional["Scan"], partials_ptr: int, capacity: int, n_ptr: int,
                       selection_ptr: int = 0, want_max: bool = False, stream: int = 0):
        """Partial GROUP BY this (byte-view) column with COUNT(*) and MIN / MAX of `value_scan` (lc_scan_group_partials)."""
        N.check(self._lib.lc_scan_group_partials(self._cache.handle, self._h, value_scan._h if value_scan is not None else None,
                                                 1 if want_max else 0, selection_ptr or None, partials_ptr or None, capacity,
                                                 n_ptr, stream or None), self._cache.handle)

    def group_partials_to_host(self, value_scan: Optional["Scan"] = None, selection: Optional[np.ndarray] = None,
                               want_max: bool = False, capacity: int = 1 << 16):
        """Convenience for tests: ndarray of (entry, group_row, count, best_row) u32 records (retries when `capacity` was
        too small)."""
        lib, ctx = self._lib, self._cache.handle
        d_sel, d_n = C.c_void_p(), C.c_void_p()
        N.check(lib.lc_device_alloc(ctx, 8, C.byref(d_n)), ctx)
        try:
            if selection is not None:
                sel = np.ascontiguousarray(selection, dtype=np.uint64)
                N.check(lib.lc_device_alloc(ctx, max(sel.size, 1) * 8, C.byref(d_sel)), ctx)
                N.check(lib.lc_host_to_device(ctx, d_sel, sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
            while True:
                d_p = C.c_void_p()
                N.check(lib.lc_device_alloc(ctx, max(capacity, 1) * 16, C.byref(d_p)), ctx)
                try:
                    self.group_partials(value_scan, d_p.value, capacity, d_n.value, d_sel.value or 0, want_max)
                    n = np.zeros(1, np.uint64)
                    N.check(lib.lc_device_to_host(ctx, n.ctypes.data_as(C.c_void_p), d_n, 8, None), ctx)
                    k = int(n[0])
                    if k <= capacity:
                        out = np.zeros((max(k, 1), 4), np.uint32)
                        if k:
                            N.check(lib.lc_device_to_host(ctx, out.ctypes.data_as(C.c_void_p), d_p, k * 16, None), ctx)
                        return out[:k]
                    capacity = k
                finally:
                    lib.lc_device_free(ctx, d_p)
        finally:
            for p in (d_sel, d_n):
                if p.value:
                    lib.lc_device_free(ctx, p)

    # ---- sparse results: hit lists (lc_scan_eval_hits / lc_scan_gather_*_hits) -----------------------------------------
    def eval_hits(self, exprs, hits_ptr: int, capacity: int, n_hits_ptr: int, selection_ptr: int = 0, hit_first_ptr: int = 0,
                  counts_ptr: int = 0, total_out_ptr: int = 0, stream: int = 0, counters_zeroed: bool = False,
                  partitioned: bool = False):
        """The predicate's hit rows as (entry << 32 | row) u64 records instead of a mask (lc_scan_eval_hits).  Asynchronous.
        `counters_zeroed`: the caller zeroed *n_hits (LC_HITS_COUNTERS_ZEROED: no memset kernel in front of the call).
        `partitioned` (LC_HITS_PARTITIONED): 16 partitions of capacity / 16 records, n_hits_ptr -> 16 x 16 u64 (a counter per
        128-byte line); the consuming calls take the same flag."""
        if isinstance(exprs, LiquidExpr):
            exprs = [exprs]
        preds = (N.Predicate * len(exprs))(*[e.as_predicate() for e in exprs])
        N.check(self._lib.lc_scan_eval_hits(self._cache.handle, self._h, preds, len(exprs), C.c_void_p(selection_ptr or None),
                                            C.c_void_p(hits_ptr), capacity, C.c_void_p(n_hits_ptr),
                                            C.c_void_p(hit_first_ptr or None), C.c_void_p(counts_ptr or None),
                                            C.c_void_p(total_out_ptr or None), (1 if counters_zeroed else 0) | (4 if partitioned else 0),
                                            C.c_void_p(stream or None)), self._cache.handle)

    def filter_hits(self, expr: LiquidExpr, hits_in_ptr: int, n_in_ptr: int, capacity_in: int, hits_out_ptr: int,
                    capacity_out: int, n_out_ptr: int, stream: int = 0, counters_zeroed: bool = False, partitioned: bool = False):
        """The next conjunct over the rows of a hit list (lc_scan_filter_hits): records whose row satisfies `expr` on this
        scan's column are appended to `hits_out`.  Asynchronous."""
        pred = expr.as_predicate()
        N.check(self._lib.lc_scan_filter_hits(self._cache.handle, self._h, C.byref(pred), C.c_void_p(hits_in_ptr),
                                              C.c_void_p(n_in_ptr), capacity_in, C.c_void_p(hits_out_ptr), capacity_out,
                                              C.c_void_p(n_out_ptr), (1 if counters_zeroed else 0) | (4 if partitioned else 0),
                                              C.c_void_p(stream or None)), self._cache.handle)

    def filter_hits_to_host(self, expr: LiquidExpr, hits: np.ndarray) -> np.ndarray:
        """Convenience for tests: the records of `hits` that survive `expr`, as written."""
        lib, ctx = self._lib, self._cache.handle
        hits = np.ascontiguousarray(hits, dtype=np.uint64)
        k = int(hits.size)
        ptrs = []
        try:
            d_in = self._to_dev(hits); ptrs.append(d_in)
            d_nin = self._to_dev(np.array([k], np.uint64)); ptrs.append(d_nin)
            d_out = self._dev(max(k, 1) * 8); ptrs.append(d_out)
            d_nout = self._dev(8); ptrs.append(d_nout)
            self.filter_hits(expr, d_in.value, d_nin.value, max(k, 1), d_out.value, max(k, 1), d_nout.value)
            N.check(lib.lc_stream_synchronize(ctx, None), ctx)
            n = int(self._from_dev(d_nout, np.uint64, 1)[0])
            out = self._from_dev(d_out, np.uint64, n)
        finally:
            for p in ptrs:
                if p.value:
                    lib.lc_device_free(ctx, p)
        return out

    def mask_to_hits(self, mask_ptr: int, hits_ptr: int, capacity: int, n_hits_ptr: int, hit_first_ptr: int = 0, stream: int = 0,
                     partitioned: bool = False):
        N.check(self._lib.lc_scan_mask_to_hits(self._cache.handle, self._h, C.c_void_p(mask_ptr), C.c_void_p(hits_ptr), capacity,
                                               C.c_void_p(n_hits_ptr), C.c_void_p(hit_first_ptr or None), 4 if partitioned else 0,
                                               C.c_void_p(stream or None)), self._cache.handle)

    def hits_compact(self, hits_ptr: int, n_hits_ptr: int, capacity: int, hits_out_ptr: int, capacity_out: int, n_out_ptr: int,
                     stream: int = 0):
        """lc_hits_compact: the partitions of a list, in partition order, as one contiguous list (+ its count)."""
        N.check(self._lib.lc_hits_compact(self._cache.handle, C.c_void_p(hits_ptr), C.c_void_p(n_hits_ptr), capacity,
                                          C.c_void_p(hits_out_ptr), capacity_out, C.c_void_p(n_out_ptr), C.c_void_p(stream or None)),
                self._cache.handle)

    def gather_fixed_hits(self, hits_ptr: int, n_hits_ptr: int, capacity_rows: int, values_out_ptr: int, row_valid_ptr: int = 0,
                          stream: int = 0, partitioned: bool = False):
        """get().with_selection() of a fixed-width column for the rows of a hit list, one launch."""
        N.check(self._lib.lc_scan_gather_fixed_hits(self._cache.handle, self._h, C.c_void_p(hits_ptr), C.c_void_p(n_hits_ptr),
                                                    capacity_rows, C.c_void_p(values_out_ptr), C.c_void_p(row_valid_ptr or None),
                                                    4 if partitioned else 0, C.c_void_p(stream or None)), self._cache.handle)

    def gather_bytes_hits(self, hits_ptr: int, n_hits_ptr: int, capacity_rows: int, views_ptr: int, data_ptr: int,
                          capacity_bytes: int, n_bytes_ptr: int, row_valid_ptr: int = 0, stream: int = 0,
                          counters_zeroed: bool = False, slotted: bool = False, partitioned: bool = False):
        """The same for a byte-view column: Arrow BinaryView records (16 bytes per row) + one data buffer, one launch.
        `slotted` (LC_GATHER_SLOTTED): record i's bytes at i * 128 when they fit, longer values behind capacity_rows * 128."""
        N.check(self._lib.lc_scan_gather_bytes_hits(self._cache.handle, self._h, C.c_void_p(hits_ptr), C.c_void_p(n_hits_ptr),
                                                    capacity_rows, C.c_void_p(views_ptr), C.c_void_p(row_valid_ptr or None),
                                                    C.c_void_p(data_ptr or None), capacity_bytes, C.c_void_p(n_bytes_ptr),
                                                    (1 if counters_zeroed else 0) | (2 if slotted else 0) | (4 if partitioned else 0),
                                                    C.c_void_p(stream or None)), self._cache.handle)

    def _dev(self, nbytes: int) -> C.c_void_p:
        p = C.c_void_p()
        N.check(self._lib.lc_device_alloc(self._cache.handle, max(int(nbytes), 8), C.byref(p)), self._cache.handle)
        return p

    def _to_dev(self, arr: np.ndarray) -> C.c_void_p:
        p = self._dev(arr.nbytes)
        if arr.nbytes:
            N.check(self._lib.lc_host_to_device(self._cache.handle, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, None),
                    self._cache.handle)
        return p

    def _from_dev(self, p: C.c_void_p, dtype, count: int) -> np.ndarray:
        out = np.zeros(max(int(count), 1), dtype)
        if count:
            N.check(self._lib.lc_device_to_host(self._cache.handle, out.ctypes.data_as(C.c_void_p), p,
                                                int(count) * out.itemsize, None), self._cache.handle)
        return out[: int(count)]

    def eval_hits_to_host(self, exprs, selection: Optional[np.ndarray] = None, capacity: Optional[int] = None,
                          from_mask: bool = False):
        """Convenience for tests: (hit records as they were written, n_hits, per-entry counts, COUNT(*), first-hit index per
        entry).  `from_mask`: evaluate to a mask and list it with lc_scan_mask_to_hits instead (the two must agree)."""
        lib, ctx = self._lib, self._cache.handle
        cap = int(self.rows) if capacity is None else int(capacity)
        ptrs = []
        try:
            d_hits = self._dev(max(cap, 1) * 8); ptrs.append(d_hits)
            d_n = self._dev(8); ptrs.append(d_n)
            d_first = self._to_dev(np.full(max(self.entries, 1), 0xFFFFFFFF, np.uint32)); ptrs.append(d_first)
            d_counts = self._dev(max(self.entries, 1) * 4); ptrs.append(d_counts)
            d_total = self._to_dev(np.zeros(1, np.uint64)); ptrs.append(d_total)
            d_sel = None
            if selection is not None:
                d_sel = self._to_dev(np.ascontiguousarray(selection, dtype=np.uint64)); ptrs.append(d_sel)
            if from_mask:
                d_mask = self._dev(max(int(self.mask_words), 1) * 8); ptrs.append(d_mask)
                self.eval_count(exprs, d_mask.value, d_total.value, d_sel.value if d_sel else 0, d_counts.value)
                self.mask_to_hits(d_mask.value, d_hits.value, cap, d_n.value, d_first.value)
            else:
                self.eval_hits(exprs, d_hits.value, cap, d_n.value, d_sel.value if d_sel else 0, d_first.value, d_counts.value,
                               d_total.value)
            N.check(lib.lc_stream_synchronize(ctx, None), ctx)
            n = int(self._from_dev(d_n, np.uint64, 1)[0])
            hits = self._from_dev(d_hits, np.uint64, min(n, cap))
            counts = self._from_dev(d_counts, np.uint32, self.entries)
            total = int(self._from_dev(d_total, np.uint64, 1)[0])
            first = self._from_dev(d_first, np.uint32, self.entries)
        finally:
            for p in ptrs:
                if p.value:
                    lib.lc_device_free(ctx, p)
        return hits, n, counts, total, first

    def gather_bytes_hits_to_host(self, hits: np.ndarray, capacity_bytes: Optional[int] = None, slotted: bool = False):
        """Convenience for tests: list of bytes / None, row i = record i of `hits` — decoded from the BinaryView records."""
        lib, ctx = self._lib, self._cache.handle
        hits = np.ascontiguousarray(hits, dtype=np.uint64)
        k = int(hits.size)
        ptrs = []
        try:
            d_hits = self._to_dev(hits); ptrs.append(d_hits)
            d_n = self._to_dev(np.array([k], np.uint64)); ptrs.append(d_n)
            d_views = self._dev(max(k, 1) * 16); ptrs.append(d_views)
            d_valid = self._dev(max(k, 1)); ptrs.append(d_valid)
            d_nb = self._dev(8); ptrs.append(d_nb)
            cap_b = 1 << 16 if capacity_bytes is None else int(capacity_bytes)
            slots = max(k, 1) * 128 if slotted else 0  # (LC_GATHER_SLOTTED: the slots, then the values that fit none)
            cap_b = max(cap_b, slots)
            while True:
                d_data = self._dev(max(cap_b, 1))
                try:
                    self.gather_bytes_hits(d_hits.value, d_n.value, max(k, 1), d_views.value, d_data.value, cap_b, d_nb.value,
                                           d_valid.value, slotted=slotted)
                    N.check(lib.lc_stream_synchronize(ctx, None), ctx)
                    need = slots + int(self._from_dev(d_nb, np.uint64, 1)[0])
                    if need <= cap_b:
                        data = self._from_dev(d_data, np.uint8, need).tobytes()
                        break
                    cap_b = need
                finally:
                    lib.lc_device_free(ctx, d_data)
            views = self._from_dev(d_views, np.uint8, k * 16).reshape(-1, 16) if k else np.zeros((0, 16), np.uint8)
            valid = self._from_dev(d_valid, np.uint8, k)
        finally:
            for p in ptrs:
                if p.value:
                    lib.lc_device_free(ctx, p)
        out = []
        for i in range(k):
            if not valid[i]:
                assert not views[i].any(), "a null row's view must be all zero"
                out.append(None)
                continue
            ln = int(views[i, :4].view(np.int32)[0])
            if ln <= 12:
                assert not views[i, 4 + ln:].any(), "inline views are zero padded"
                out.append(views[i, 4: 4 + ln].tobytes())
            else:
                buf, off = (int(x) for x in views[i, 8:16].view(np.int32))
                assert buf == 0
                if slotted:
                    assert off == i * 128 if ln <= 128 else off >= slots, "slotted gather: value outside its place"
                v = data[off: off + ln]
                assert v[:4] == views[i, 4:8].tobytes(), "view prefix differs from the value"
                out.append(v)
        return out

    def gather_fixed_hits_to_host(self, hits: np.ndarray, np_dtype):
        """Convenience for tests: (values, validity) for the records of `hits`."""
        lib, ctx = self._lib, self._cache.handle
        hits = np.ascontiguousarray(hits, dtype=np.uint64)
        k = int(hits.size)
        width = np.dtype(np_dtype).itemsize
        ptrs = []
        try:
            d_hits = self._to_dev(hits); ptrs.append(d_hits)
            d_n = self._to_dev(np.array([k], np.uint64)); ptrs.append(d_n)
            d_vals = self._dev(max(k, 1) * width + 64); ptrs.append(d_vals)
            d_valid = self._dev(max(k, 1)); ptrs.append(d_valid)
            self.gather_fixed_hits(d_hits.value, d_n.value, max(k, 1), d_vals.value, d_valid.value)
            N.check(lib.lc_stream_synchronize(ctx, None), ctx)
            vals = self._from_dev(d_vals, np_dtype, k)
            valid = self._from_dev(d_valid, np.uint8, k)
        finally:
            for p in ptrs:
                if p.value:
                    lib.lc_device_free(ctx, p)
        return vals, valid.astype(bool)

    def eval_to_host(self, expr: LiquidExpr, selection: Optional[np.ndarray] = None):
        """Convenience for tests: runs the scan and returns (mask words as uint64 ndarray, per-entry counts)."""
        lib, ctx = self._lib, self._cache.handle
        words = max(int(self.mask_words), 1)
        d_mask, d_counts, d_sel = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(lib.lc_device_alloc(ctx, words * 8, C.byref(d_mask)), ctx)
        N.check(lib.lc_device_alloc(ctx, max(self.entries, 1) * 4, C.byref(d_counts)), ctx)
        try:
            if selection is not None:
                sel = np.ascontiguousarray(selection, dtype=np.uint64)
                assert sel.size == self.mask_words
                N.check(lib.lc_device_alloc(ctx, words * 8, C.byref(d_sel)), ctx)
                N.check(lib.lc_host_to_device(ctx, d_sel, sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
            pred = expr.as_predicate()
            N.check(lib.lc_scan_eval(ctx, self._h, C.byref(pred), d_sel if selection is not None else None, d_mask,
                                     d_counts, None), ctx)
            mask = np.zeros(words, np.uint64)
            counts = np.zeros(max(self.entries, 1), np.uint32)
            N.check(lib.lc_device_to_host(ctx, mask.ctypes.data_as(C.c_void_p), d_mask, words * 8, None), ctx)
            N.check(lib.lc_device_to_host(ctx, counts.ctypes.data_as(C.c_void_p), d_counts, counts.size * 4, None), ctx)
        finally:
            for p in (d_mask, d_counts, d_sel):
                if p.value:
                    lib.lc_device_free(ctx, p)
        return mask[: int(self.mask_words)], counts[: self.entries]


def boolean_buffer_and_then(cache: LiquidCache, left, right) -> np.ndarray:
    """`boolean_buffer_and_then(left, right)`; popcount(left) must equal len(right)."""
    lb = np.asarray(left, dtype=bool)
    rb = np.asarray(right, dtype=bool)
    if int(lb.sum()) != len(rb):
        raise ValueError("the right selection must have as many bits as the left selection has set bits")
    l_bytes, r_bytes = _selection_bytes(lb), _selection_bytes(rb)
    out = np.zeros(len(l_bytes), np.uint8)
    N.check(cache._lib.lc_mask_and_then(cache.handle, l_bytes.ctypes.data_as(C.c_void_p), len(lb),
                                        r_bytes.ctypes.data_as(C.c_void_p), len(rb),
                                        out.ctypes.data_as(C.c_void_p)), cache.handle)
    return _bits_to_bool(out, len(lb))
