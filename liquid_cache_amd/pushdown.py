"""Host-side mirror of the reference's predicate-pushdown orchestration for one row range of a table.

    build_row_filter / get_priority          src/datafusion/src/reader/plantime/row_filter.rs:428-515
    LiquidCacheReader::build_predicate_filter src/datafusion/src/reader/runtime/liquid_cache_reader.rs:297-339
    CachedRowGroup::evaluate_selection_with_predicate (single column / multi-column OR)
                                              src/datafusion/src/cache/mod.rs:96-167

The reference splits a filter into conjuncts, orders them (Eq / NotEq, LIKE, NOT LIKE, ranges, everything else) and
evaluates them one after the other per 8192-row batch, every result narrowing the selection of the next
(`boolean_buffer_and_then`), leaving a batch as soon as its selection is empty.  Here the same plan runs over whole
column scans on the device: the hit mask of conjunct k (= predicate AND valid AND selected) IS the selection of
conjunct k + 1, batches without a selected row are skipped by the kernels themselves, and nothing returns to the host
between conjuncts.  Two adjacent range conjuncts on the same fixed-width column (`c >= a AND c < b`) ride in one pass
(lc_scan_eval_and); a disjunction over columns, or an IN list, is one Kleene-OR step (lc_scan_eval_or).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import pyarrow as pa

from . import _native as N
from .cache import LiquidExpr, Scan

_RANGE_OPS = {N.OP_LT, N.OP_LE, N.OP_GT, N.OP_GE}
_FUSABLE_OPS = _RANGE_OPS | {N.OP_EQ}


@dataclass
class Conjunct:
    """`column OP literal` (or `column [NOT] LIKE pattern`)."""
    column: str
    op: str
    literal: object


@dataclass
class AnyOf:
    """A disjunction of column-literal terms: a multi-column OR (cache/mod.rs:111-150) or an IN list on one column."""
    terms: List[Conjunct]


@dataclass
class Column:
    """A column of the row range as the executor sees it."""
    scan: Scan
    dtype: pa.DataType
    hint: Optional[int] = None
    fixed_width: bool = True


def get_priority(c: Union[Conjunct, AnyOf]) -> int:
    """row_filter.rs:499-515: Eq/NotEq 0, LIKE 1, NOT LIKE 2, ranges 3, other binary expressions (OR) 4, the rest 5."""
    if isinstance(c, AnyOf):
        # `a = 1 OR b = 2` is a BinaryExpr with Operator::Or (priority 4); an IN list is not a BinaryExpr (5)
        return 4 if len({t.column for t in c.terms}) > 1 else 5
    code = _op_code(c.op)
    if code in (N.OP_EQ, N.OP_NE):
        return 0
    if code == N.OP_LIKE:
        return 1
    if code == N.OP_NOT_LIKE:
        return 2
    return 3


def _op_code(op: str) -> int:
    from .cache import _OPS
    return _OPS[op.lower()]


class LiquidRowFilter:
    """The ordered conjuncts of one pushed-down filter (LiquidRowFilter::new, row_filter.rs:481-496).  The reference sorts
    with `sort_unstable_by` on the priority; ties keep the query's order here (any order gives the same result)."""

    def __init__(self, conjuncts: Sequence[Union[Conjunct, AnyOf]]):
        self.predicates = sorted(conjuncts, key=get_priority)

    def __len__(self):
        return len(self.predicates)


@dataclass
class _Step:
    kind: str                       # "and" (one column, 1-2 predicates) or "or"
    scans: List[Scan]
    exprs: List[LiquidExpr]
    text: str = ""


class PushdownExecutor:
    """Evaluates a LiquidRowFilter over the column scans of one row range, device resident."""

    def __init__(self, columns: Dict[str, Column], fuse_ranges: bool = True):
        self.columns = columns
        self.fuse_ranges = fuse_ranges
        words = {int(c.scan.mask_words) for c in columns.values()}
        if len(words) > 1:
            raise ValueError("all columns of a row range must cover the same rows (same mask layout)")

    def _expr(self, c: Conjunct) -> LiquidExpr:
        col = self.columns[c.column]
        e = LiquidExpr.try_new(c.op, c.literal, col.dtype, col.hint)
        if e is None:
            raise ValueError("not a LiquidExpr (the reference would materialise and evaluate with Arrow): %r" % (c,))
        return e

    def plan(self, row_filter: LiquidRowFilter) -> List[_Step]:
        steps: List[_Step] = []
        preds = list(row_filter.predicates)
        i = 0
        while i < len(preds):
            p = preds[i]
            if isinstance(p, AnyOf):
                steps.append(_Step("or", [self.columns[t.column].scan for t in p.terms], [self._expr(t) for t in p.terms],
                                   " OR ".join("%s %s %r" % (t.column, t.op, t.literal) for t in p.terms)))
                i += 1
                continue
            col = self.columns[p.column]
            group = [p]
            nxt = preds[i + 1] if i + 1 < len(preds) else None
            if (self.fuse_ranges and col.fixed_width and isinstance(nxt, Conjunct) and nxt.column == p.column
                    and _op_code(p.op) in _FUSABLE_OPS and _op_code(nxt.op) in _FUSABLE_OPS):
                group.append(nxt)
            steps.append(_Step("and", [col.scan], [self._expr(g) for g in group],
                               " AND ".join("%s %s %r" % (g.column, g.op, g.literal) for g in group)))
            i += len(group)
        return steps

    def compile(self, row_filter: LiquidRowFilter) -> "CompiledFilter":
        """Marshal the plan once: `CompiledFilter.run` is then ONE call into the library (lc_scan_eval_filter) per
        evaluation — what a Rust caller pays, instead of Python rebuilding expressions and ctypes structs per pass."""
        return CompiledFilter(self, self.plan(row_filter))

    def evaluate(self, row_filter: LiquidRowFilter, mask_ptrs: Sequence[int], counts_ptr: int = 0,
                 selection_ptr: int = 0, stream: int = 0) -> int:
        """Asynchronous.  `mask_ptrs`: two device buffers of `mask_words` u64 used alternately; returns the pointer that
        holds the final hit mask (== selection_ptr when the filter is empty); counts_ptr receives the per-entry hits of
        the LAST conjunct, i.e. of the whole filter."""
        sel = selection_ptr
        for k, st in enumerate(self.plan(row_filter)):
            out = mask_ptrs[k & 1]
            if out == sel:
                out = mask_ptrs[(k + 1) & 1]
            if st.kind == "or":
                Scan.eval_or(st.scans, st.exprs, out, sel, counts_ptr, 0, stream)
            elif len(st.exprs) == 2:
                if not st.scans[0].eval_and(st.exprs, out, sel, counts_ptr, stream):
                    # not fusable after all (e.g. a byte-view column): chain the two passes
                    other = mask_ptrs[0] if out == mask_ptrs[1] else mask_ptrs[1]
                    if other == sel:
                        raise RuntimeError("chaining an unfusable pair needs a third mask buffer")
                    st.scans[0].eval(st.exprs[0], other, sel, counts_ptr, stream)
                    st.scans[0].eval(st.exprs[1], out, other, counts_ptr, stream)
            else:
                st.scans[0].eval(st.exprs[0], out, sel, counts_ptr, stream)
            sel = out
        return sel

    def evaluate_to_host(self, row_filter: LiquidRowFilter, selection: Optional[np.ndarray] = None):
        """Convenience for tests: (final mask words, per-entry counts) as numpy arrays."""
        any_col = next(iter(self.columns.values()))
        scan = any_col.scan
        lib, ctx = scan._lib, scan._cache.handle
        words = max(int(scan.mask_words), 1)
        bufs = [C.c_void_p() for _ in range(4)]
        try:
            for b in bufs[:2]:
                N.check(lib.lc_device_alloc(ctx, words * 8, C.byref(b)), ctx)
            N.check(lib.lc_device_alloc(ctx, max(scan.entries, 1) * 4, C.byref(bufs[2])), ctx)
            sel_ptr = 0
            if selection is not None:
                sel = np.ascontiguousarray(selection, dtype=np.uint64)
                N.check(lib.lc_device_alloc(ctx, words * 8, C.byref(bufs[3])), ctx)
                N.check(lib.lc_host_to_device(ctx, bufs[3], sel.ctypes.data_as(C.c_void_p), sel.size * 8, None), ctx)
                sel_ptr = bufs[3].value
            final = self.evaluate(row_filter, [bufs[0].value, bufs[1].value], bufs[2].value, sel_ptr)
            mask = np.zeros(words, np.uint64)
            counts = np.zeros(max(scan.entries, 1), np.uint32)
            if final:
                N.check(lib.lc_device_to_host(ctx, mask.ctypes.data_as(C.c_void_p), C.c_void_p(final), words * 8, None), ctx)
                N.check(lib.lc_device_to_host(ctx, counts.ctypes.data_as(C.c_void_p), bufs[2], counts.size * 4, None), ctx)
            else:  # empty filter and no selection: every row passes
                mask[:] = ~np.uint64(0)
        finally:
            for b in bufs:
                if b.value:
                    lib.lc_device_free(ctx, b)
        return mask[: int(scan.mask_words)], counts[: scan.entries]


class CompiledFilter:
    """The steps of a LiquidRowFilter as the C ABI takes them (lc_filter_step[]); keeps the ctypes objects alive."""

    def __init__(self, executor: Optional[PushdownExecutor], steps: List[_Step]):
        any_scan = next(iter(executor.columns.values())).scan if executor is not None else steps[0].scans[0]
        self._lib, self._ctx = any_scan._lib, any_scan._cache.handle
        self.steps = steps
        self._keep = []
        arr = (N.FilterStep * max(len(steps), 1))()
        for k, st in enumerate(steps):
            preds = (N.Predicate * len(st.exprs))(*[e.as_predicate() for e in st.exprs])
            scans = (C.c_void_p * len(st.scans))(*[s._h for s in st.scans])
            self._keep += [preds, scans, st.exprs]
            arr[k].kind = 1 if st.kind == "or" else 0
            arr[k].n_terms = len(st.exprs)
            arr[k].scans = C.cast(scans, C.POINTER(C.c_void_p))
            arr[k].preds = C.cast(preds, C.c_void_p)
        self._arr = arr
        self._n = len(steps)
        self._final = C.c_void_p()

    @classmethod
    def from_conjunction(cls, scan_exprs: Sequence) -> "CompiledFilter":
        """[(scan, [expr] or [expr, expr]), ...]: predicates (or fusable pairs) on the given column scans, in order."""
        return cls(None, [_Step("and", [scan], list(exprs)) for scan, exprs in scan_exprs])

    def run(self, mask_a_ptr: int, mask_b_ptr: int, counts_ptr: int = 0, selection_ptr: int = 0, total_ptr: int = 0,
            stream: int = 0) -> int:
        """Asynchronous.  Returns the device pointer that holds the filter's hit mask."""
        N.check(self._lib.lc_scan_eval_filter(self._ctx, self._n, self._arr, C.c_void_p(selection_ptr or None),
                                              C.c_void_p(mask_a_ptr), C.c_void_p(mask_b_ptr),
                                              C.c_void_p(counts_ptr or None), C.c_void_p(total_ptr or None),
                                              C.byref(self._final), C.c_void_p(stream or None)), self._ctx)
        return int(self._final.value or 0)
