"""GPU parity of BASELINE.json config 5: the pushed-down predicates of every ClickBench query over a synthetic hits-shaped
table (4 batches), device chain vs the oracle's evaluation of the same Liquid bytes; and the multi-column OR (Kleene) of
CachedRowGroup::evaluate_selection_with_predicate (src/datafusion/src/cache/mod.rs:111-150) against a Kleene OR of the
oracle's per-column results."""
import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc
from liquid_cache_amd import clickbench as cb
from liquid_cache_amd.pushdown import AnyOf, Conjunct, LiquidRowFilter, PushdownExecutor

pytestmark = pytest.mark.gpu
ROWS, BS = 3 * 8192 + 1234, 8192
OPS = {"=": "eq", "!=": "ne", "<": "lt", "<=": "le", ">": "gt", ">=": "ge", "like": "like", "not like": "not_like"}


def _bits(mask, scan, b, rows):
    w0 = int(scan.segment_offsets[b])
    return np.unpackbits(mask[w0: w0 + (rows + 63) // 64].view(np.uint8), bitorder="little")[:rows].astype(bool)


@pytest.fixture(scope="module")
def hits(product_lib):
    cache = lc.LiquidCacheBuilder.new().build()
    columns, ids, arrays = cb.stage_hits(cache, ROWS, seed=11, batch_size=BS, row_group_batches=2, threads=4, keep_arrays=True)
    from oracle import liquid_oracle as lo
    lo.build()
    liquids, symtabs = {}, {}
    for nm, arrs in arrays.items():
        liquids[nm] = []
        for b, arr in enumerate(arrs):
            path = lc.ParquetArrayID.column_access_path(ids[nm][b])
            liquids[nm].append(cache.transcode(arr, cb.SCHEMA[nm][2], path))
            if nm in cb._STRINGS and path not in symtabs:
                symtabs[path] = lo.symtab_load(cache.symbol_table(path))
    yield cache, columns, ids, arrays, liquids, symtabs, lo
    for c in columns.values():
        c.scan.close()
    cache.close()


def _oracle_term(h, t: Conjunct, b):
    cache, columns, ids, arrays, liquids, symtabs, lo = h
    lit = t.literal.encode() if isinstance(t.literal, str) else t.literal
    st = symtabs.get(lc.ParquetArrayID.column_access_path(ids[t.column][b]))
    r = lo.eval_predicate(liquids[t.column][b], lo.OP_NAMES[OPS[t.op]], lit, None, symtab=st)
    return r.values if r.validity is None else (r.values & r.validity)


def _oracle_filter(h, conjuncts, b, rows):
    sel = np.ones(rows, bool)
    for p in LiquidRowFilter(conjuncts).predicates:
        if isinstance(p, AnyOf):
            hit = np.zeros(rows, bool)
            for t in p.terms:
                hit |= _oracle_term(h, t, b)
        else:
            hit = _oracle_term(h, p, b)
        sel &= hit
    return sel


@pytest.mark.parametrize("fuse", [True, False])
def test_every_clickbench_pushdown_matches_the_oracle(hits, fuse):
    cache, columns, ids, arrays, liquids, symtabs, lo = hits
    ex = PushdownExecutor(columns, fuse_ranges=fuse)
    n_batches = (ROWS + BS - 1) // BS
    total_hits = 0
    for q in range(cb.N_QUERIES):
        conj = cb.QUERIES.get(q)
        if not conj:
            continue
        mask, counts = ex.evaluate_to_host(LiquidRowFilter(conj))
        scan = columns[(conj[0].terms[0] if isinstance(conj[0], AnyOf) else conj[0]).column].scan
        for b in range(n_batches):
            rows = min(BS, ROWS - b * BS)
            want = _oracle_filter(hits, conj, b, rows)
            assert np.array_equal(_bits(mask, scan, b, rows), want), (q, b)
            assert int(counts[b]) == int(want.sum()), (q, b)
            total_hits += int(want.sum())
    assert total_hits > 0


def test_sweep_with_an_input_selection_and_correlated_hits(hits):
    """Same chain under a caller selection (rows already pruned by an earlier filter), and a conjunction crafted to have
    survivors in every batch (the independent synthetic columns make q40 / q41 empty)."""
    cache, columns, ids, arrays, liquids, symtabs, lo = hits
    ex = PushdownExecutor(columns)
    rng = np.random.default_rng(2)
    n_batches = (ROWS + BS - 1) // BS
    scan = columns["IsRefresh"].scan
    keep = [rng.random(min(BS, ROWS - b * BS)) < 0.3 for b in range(n_batches)]
    words = np.zeros(int(scan.mask_words), np.uint64)
    for b, seg in enumerate(keep):
        packed = np.packbits(seg, bitorder="little")
        w0 = int(scan.segment_offsets[b])
        words[w0: w0 + (len(seg) + 63) // 64].view(np.uint8)[: len(packed)] = packed
    conj = [Conjunct("IsRefresh", "=", 0), Conjunct("EventDate", ">=", cb._days("2013-07-01")),
            Conjunct("EventDate", "<=", cb._days("2013-07-31")), Conjunct("SearchPhrase", "!=", ""),
            AnyOf([Conjunct("TraficSourceID", "=", -1), Conjunct("AdvEngineID", "!=", 0), Conjunct("Title", "like", "%Google%")]),
            Conjunct("URL", "not like", "%yandex%")]
    mask, counts = ex.evaluate_to_host(LiquidRowFilter(conj), selection=words)
    got_total = 0
    for b in range(n_batches):
        rows = min(BS, ROWS - b * BS)
        want = _oracle_filter(hits, conj, b, rows) & keep[b]
        assert np.array_equal(_bits(mask, scan, b, rows), want), b
        got_total += int(want.sum())
    assert got_total > 0 and int(counts.sum()) == got_total


def _kleene_or(results):
    """arrow or_kleene over (values, validity) pairs."""
    val = np.zeros_like(results[0][0])
    true_any = np.zeros_like(val)
    all_valid = np.ones_like(val)
    for v, m in results:
        true_any |= v & m
        all_valid &= m
    return true_any, all_valid | true_any


def test_multi_column_or_is_kleene(gpu_cache, oracle):
    """cache.eval_predicate_or on nullable columns, with and without a selection: values AND validity equal arrow's
    or_kleene of the per-column BooleanArrays (true OR null = true, false OR null = null)."""
    lo = oracle
    rng = np.random.default_rng(6)
    n = 5000
    a = rng.integers(0, 50, size=n).astype(np.int32)
    b = rng.integers(0, 1000, size=n).astype(np.int64)
    s = np.array(["http://%s/%d" % (rng.choice(["google.com", "yandex.ru", "mail.ru"]), rng.integers(100)) for _ in range(n)])
    va, vb, vs = rng.random(n) < 0.8, rng.random(n) < 0.7, rng.random(n) < 0.9
    arrs = [pa.array(a, mask=~va), pa.array(b, mask=~vb), pa.array(s.tolist(), mask=~vs, type=pa.string())]
    ids = [lc.ParquetArrayID.new(12, 0, c, 0) for c in (1, 2, 3)]
    hint = lc.CacheExpression.SUBSTRING_SEARCH
    for e, arr, h in zip(ids, arrs, (None, None, hint)):
        gpu_cache.insert(e, arr, h)
    path = lc.ParquetArrayID.column_access_path(ids[2])
    liquids = [gpu_cache.transcode(arrs[0]), gpu_cache.transcode(arrs[1]), gpu_cache.transcode(arrs[2], hint, path)]
    st = lo.symtab_load(gpu_cache.symbol_table(path))
    exprs = [lc.LiquidExpr.try_new("=", 7, pa.int32()), lc.LiquidExpr.try_new("<", 100, pa.int64()),
             lc.LiquidExpr.try_new("like", "%google%", pa.string(), hint)]
    oracle_args = [(lo.EQ, 7, None), (lo.LT, 100, None), (lo.LIKE, b"%google%", st)]
    for sel in (None, rng.random(n) < 0.4, np.zeros(n, bool)):
        for cols in ((0, 1), (0, 2), (0, 1, 2), (2, 2)):
            got = gpu_cache.eval_predicate_or([ids[c] for c in cols], [exprs[c] for c in cols], sel)
            res = []
            for c in cols:
                op, lit, stc = oracle_args[c]
                r = lo.eval_predicate(liquids[c], op, lit, sel, symtab=stc)
                res.append((r.values, r.validity if r.validity is not None else np.ones_like(r.values)))
            want_v, want_m = _kleene_or(res)
            gv = np.asarray(got.to_numpy(zero_copy_only=False), dtype=object)
            gm = ~np.asarray(got.is_null().to_numpy(zero_copy_only=False), dtype=bool)
            assert len(gm) == len(want_m)
            assert gm.tolist() == want_m.tolist(), (cols, sel is None)
            assert [bool(x) for x, m in zip(gv, gm) if m] == [bool(x) for x, m in zip(want_v, want_m) if m]
    assert gpu_cache.eval_predicate_or([ids[0], 999], exprs[:2]) is None       # an uncached column -> None


@pytest.mark.parametrize("fuse", [True, False])
def test_compiled_filter_is_one_call_and_gives_the_same_masks(hits, fuse):
    """lc_scan_eval_filter (the whole LiquidRowFilter in one C call) against the step-by-step executor for every query,
    with and without an input selection; the fused COUNT(*) of the last predicate kernel equals the popcount."""
    import ctypes as C
    from liquid_cache_amd import _native as N
    cache, columns, ids, arrays, liquids, symtabs, lo = hits
    ex = PushdownExecutor(columns, fuse_ranges=fuse)
    scan = columns["IsRefresh"].scan
    lib, ctx = scan._lib, scan._cache.handle
    words = int(scan.mask_words)
    rng = np.random.default_rng(5)
    sel = rng.integers(0, 1 << 63, size=words, dtype=np.uint64) | (rng.integers(0, 2, size=words, dtype=np.uint64) << np.uint64(63))
    bufs = [C.c_void_p() for _ in range(5)]
    try:
        for b in bufs[:3]:
            N.check(lib.lc_device_alloc(ctx, words * 8, C.byref(b)), ctx)
        N.check(lib.lc_device_alloc(ctx, scan.entries * 4, C.byref(bufs[3])), ctx)
        N.check(lib.lc_device_alloc(ctx, 8, C.byref(bufs[4])), ctx)
        N.check(lib.lc_host_to_device(ctx, bufs[2], sel.ctypes.data_as(C.c_void_p), words * 8, None), ctx)
        for q, conj in cb.QUERIES.items():
            rf = LiquidRowFilter(conj)
            cf = ex.compile(rf)
            for use_sel in (False, True):
                want_mask, want_counts = ex.evaluate_to_host(rf, selection=sel if use_sel else None)
                last_is_pred = cf.steps[-1].kind != "or"
                final = cf.run(bufs[0].value, bufs[1].value, bufs[3].value, bufs[2].value if use_sel else 0,
                               bufs[4].value if last_is_pred else 0)
                assert final in (bufs[0].value, bufs[1].value)
                got = np.zeros(words, np.uint64)
                cnt = np.zeros(scan.entries, np.uint32)
                tot = np.zeros(1, np.uint64)
                N.check(lib.lc_device_to_host(ctx, got.ctypes.data_as(C.c_void_p), C.c_void_p(final), words * 8, None), ctx)
                N.check(lib.lc_device_to_host(ctx, cnt.ctypes.data_as(C.c_void_p), bufs[3], scan.entries * 4, None), ctx)
                assert np.array_equal(got, want_mask), (q, use_sel)
                assert np.array_equal(cnt, want_counts), (q, use_sel)
                if last_is_pred:
                    N.check(lib.lc_device_to_host(ctx, tot.ctypes.data_as(C.c_void_p), bufs[4], 8, None), ctx)
                    assert int(tot[0]) == int(want_counts.sum()), (q, use_sel)
        # no steps: the selection itself is the result
        empty = ex.compile(LiquidRowFilter([]))
        assert empty.run(bufs[0].value, bufs[1].value, 0, bufs[2].value) == bufs[2].value
    finally:
        for b in bufs:
            if b.value:
                lib.lc_device_free(ctx, b)
