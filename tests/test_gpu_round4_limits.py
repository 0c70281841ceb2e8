"""Round 4: the limits of the byte-view path against the oracle.

* needles of 48..63 bytes run the folded automaton from LDS like shorter ones (a 49-64 KB image: the workgroup's LDS limit
  is raised past the default 64 KB), on every LIKE kernel; for longer ones the automaton runs over the first 63 bytes and
  the dictionary values it accepts are matched against the whole pattern (one case: the first 64 bytes of a value followed
  by a byte it does not have) — all equal to the oracle's decode + memmem (byte_view_array/comparisons.rs:598-651);
* entries of more than 8,192 rows (batch sizes of 16,384 .. 65,535: builders.rs:68-71 takes any) carry inverted row lists
  and run the scan-level index kernel; entries beyond that (100,000 and 150,001 rows here) are evaluated by the general
  kernels — LIKE, Eq and ordering predicates, per-entry calls and get-with-selection.
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

import liquid_cache_amd as lc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_data as fz  # noqa: E402

pytestmark = pytest.mark.gpu
HINT = lc.CacheExpression.SUBSTRING_SEARCH


def _want_full(lo, liquid, st, op, literal, sel, n):
    r = lo.eval_predicate(liquid, op, literal, sel, symtab=st)
    v = r.values if r.validity is None else (r.values & r.validity)
    hit = np.zeros(n, bool)
    if sel is None:
        hit[:] = v
    else:
        hit[np.flatnonzero(sel)] = v
    return hit


def _long_value_pool(rng, d):
    """URL-like values of 30..220 bytes, a third of them sharing long stretches (so that long needles hit several)."""
    stems = [b"http://example.com/catalog/section-%d/subsection/%d/item.html?ref=" % (i, i * 7) for i in range(6)]
    stems += [b"https://yandex.ru/search/?text=" + bytes(rng.integers(97, 123, size=70).astype(np.uint8)) for _ in range(3)]
    out = []
    for i in range(d):
        s = stems[int(rng.integers(len(stems)))] if i % 3 else b""
        tail = bytes(rng.integers(33, 127, size=int(rng.integers(10, 120))).astype(np.uint8)).replace(b"%", b"p").replace(b"_", b"u").replace(b"\\", b"b")
        out.append(s + tail + b"#%d" % i)
    return out


def _stage_entries(cache, lo, specs, rng, file_id, extra=()):
    """specs: (rows, distinct, nulls) per entry, ONE symbol table; returns (ids, [(rows, liquid, st)]).  `extra`: values every
    entry's dictionary holds besides its own."""
    pools = []
    for n, d, nulls in specs:
        pool = list(extra) + _long_value_pool(rng, max(d - len(extra), 1))
        d = len(pool)
        rows = [pool[i] for i in range(min(d, n))] + [pool[int(k)] for k in rng.integers(0, d, size=max(0, n - d))]
        rows = [rows[int(i)] for i in rng.permutation(len(rows))]
        if nulls:
            for i in rng.choice(n, size=max(1, n // 20), replace=False):
                rows[int(i)] = None
        pools.append(rows)
    train = [v for rows in pools for v in rows if v is not None][:20000]
    o, dt, _ = lo.strings_to_arrow(train)
    st = lo.fsst_train(o, dt)
    path = 9000 + file_id
    cache.set_symbol_table(path, lo.symtab_bytes(st))
    ids, flat = [], []
    for e_i, rows in enumerate(pools):
        liquid, _ = lo.encode_byte_view(rows, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
        eid = lc.ParquetArrayID.new(file_id, 0, 5, e_i)
        cache.stage([eid], [liquid], [path])
        ids.append(eid)
        flat.append((rows, liquid, st))
    return ids, flat


def _check_scan(lo, scan, flat, needles, rng, tag, ops=("like", "not_like")):
    lens = [len(c[0]) for c in flat]
    offs = scan.segment_offsets
    n_checked = 0
    for qi, nd in enumerate(needles):
        for op in ops:
            with_sel = (qi + (op == "like")) % 2 == 0
            sels, words = [None] * len(flat), None
            if with_sel:
                words = np.zeros(int(scan.mask_words), np.uint64)
                for b, n in enumerate(lens):
                    se = rng.random(n) < [0.03, 0.5, 0.95][b % 3]
                    sels[b] = se
                    packed = np.packbits(se, bitorder="little")
                    words[int(offs[b]): int(offs[b + 1])].view(np.uint8)[: len(packed)] = packed
            lit = b"%" + nd + b"%" if "like" in op else nd
            expr = lc.LiquidExpr.try_new(op, lit, pa.binary(), HINT)
            mask, counts = scan.eval_to_host(expr, selection=words)
            bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
            for b, (rows, liquid, st) in enumerate(flat):
                got = bits[int(offs[b]) * 64: int(offs[b]) * 64 + lens[b]].astype(bool)
                want = _want_full(lo, liquid, st, lo.OP_NAMES[op], lit, sels[b], lens[b])
                assert np.array_equal(got, want), (tag, b, op, len(nd), nd[:20], with_sel, int(got.sum()), int(want.sum()))
                assert int(counts[b]) == int(want.sum()), (tag, b, op, len(nd))
                assert not bits[int(offs[b]) * 64 + lens[b]: int(offs[b + 1]) * 64].any()
                n_checked += 1
    return n_checked


@pytest.mark.parametrize("like_path", [0, 4, 3, 1, 5])
def test_needles_of_48_to_63_bytes_and_longer(product_lib, oracle, like_path):
    lo = oracle
    rng = np.random.default_rng(4863)
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = _stage_entries(cache, lo, [(4000, 900, False), (8192, 2000, True), (700, 650, False), (8192, 1500, False)],
                                   rng, 21)
        scan = cache.scan(ids)
        values = [v for rows, _, _ in flat for v in rows if v is not None and len(v) >= 100]
        needles = []
        for k, ln in enumerate((48, 49, 55, 62, 63, 64, 65, 90)):
            v = values[int(rng.integers(len(values)))]
            a = int(rng.integers(0, len(v) - ln))
            needles.append(v[a:a + ln])
            if k % 2 == 0:  # one that is in no value: the last byte replaced
                needles.append(v[a:a + ln - 1] + b"\x01")
        needles.append(b"catalog/section-3/subsection/21/item.html?ref=" + b"x" * 8)  # absent, 54 bytes
        needles.append(b"http://example.com/catalog/section-3/subsection/21/item")    # 55 bytes shared by many values
        assert {len(n) for n in needles} >= {48, 55, 63, 64, 90}
        n = _check_scan(lo, scan, flat, needles, rng, "path %d" % like_path)
        assert n >= len(flat) * 2 * len(needles)
        how = scan.explain(lc.LiquidExpr.try_new("like", b"%" + needles[4] + b"%", pa.binary(), HINT))  # 55 bytes
        if like_path == 4:
            assert how.startswith("k_like_flat"), how
        if like_path == 3:
            assert how.startswith("k_like_lean"), how
        if like_path == 5:
            assert how.startswith("k_like_scanall"), how
        scan.close()
    finally:
        cache.close()


# (rows, distinct, nulls): batch sizes of 16,384 .. 65,535 rows next to ordinary ones; 65,536 rows carry no row lists (their
# u16 offsets could not count the valid rows) and keep the whole scan on k_str_pred
BIG_SPECS = {
    "16k": [(16384, 3000, False), (16384, 2500, True), (8192, 2000, False), (16384, 90, False)],
    "40k_65535": [(40000, 6000, True), (65535, 5000, False), (100, 60, False), (65535, 8000, True)],
    "65536": [(65536, 4000, False), (8192, 2000, True)],
}


@pytest.mark.parametrize("like_path", [0, 4, 1, 5])
@pytest.mark.parametrize("spec", sorted(BIG_SPECS))
def test_entries_of_more_than_8192_rows(product_lib, oracle, spec, like_path):
    lo = oracle
    rng = np.random.default_rng(len(spec) * 131 + like_path)
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = _stage_entries(cache, lo, BIG_SPECS[spec], rng, 31)
        scan = cache.scan(ids)
        values = [v for rows, _, _ in flat for v in rows if v is not None]
        needles = [b"example.com/catalog", b"yandex.ru/search", b"section-3", b"#77", b"zzzzqqq", b"ref="]
        for ln in (5, 9, 20):
            v = values[int(rng.integers(len(values)))]
            a = int(rng.integers(0, len(v) - ln))
            needles.append(v[a:a + ln])
        n = _check_scan(lo, scan, flat, needles, rng, "%s path %d" % (spec, like_path))
        assert n >= len(flat) * 2 * len(needles)
        how = scan.explain(lc.LiquidExpr.try_new("like", b"%section-3%", pa.binary(), HINT))
        if like_path == 4 and spec != "65536":
            assert how.startswith("k_like_flat"), how
        if spec == "65536" and like_path in (0, 4):
            assert how.startswith("k_str_pred"), how
        # the rows come back the same through get-with-selection (the row lists are not involved, the entry size is)
        rows, liquid, st = flat[0]
        sel = rng.random(len(rows)) < 0.001
        got = cache.get(ids[0]).with_selection(sel).read()
        want = [rows[int(i)] for i in np.flatnonzero(sel)]
        assert got.to_pylist() == want
        scan.close()
    finally:
        cache.close()


@pytest.mark.parametrize("like_path", [0, 1, 5])
def test_entries_of_more_than_65536_rows(product_lib, oracle, like_path):
    lo = oracle
    rng = np.random.default_rng(65537 + like_path)
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = _stage_entries(cache, lo, [(100000, 7000, True), (8192, 2000, False), (150001, 300, False)], rng, 41)
        scan = cache.scan(ids)
        values = [v for rows, _, _ in flat for v in rows if v is not None]
        needles = [b"example.com/catalog", b"section-3", b"#77", b"zzzzqqq"]
        v = values[int(rng.integers(len(values)))]
        needles.append(v[3:14])
        n = _check_scan(lo, scan, flat, needles, rng, "path %d" % like_path)
        lits = [values[int(rng.integers(len(values)))] for _ in range(2)] + [b"http://m", b""]
        n += _check_scan(lo, scan, flat, lits, rng, "path %d cmp" % like_path, ops=("eq", "lt", "ge"))
        assert n >= len(flat) * (2 * len(needles) + 3 * len(lits))
        # per-entry drop-in calls on the large entries
        for b in (0, 2):
            rows, liquid, st = flat[b]
            sel = rng.random(len(rows)) < 0.3
            expr = lc.LiquidExpr.try_new("like", b"%section-3%", pa.binary(), HINT)
            got = cache.eval_predicate(ids[b], expr).with_selection(sel).read()
            r = lo.eval_predicate(liquid, lo.OP_NAMES["like"], b"%section-3%", sel, symtab=st)
            want = pa.array(r.values, mask=None if r.validity is None else ~r.validity)
            assert got.equals(want), (b, like_path)
            sel2 = rng.random(len(rows)) < 0.0005
            sel2[-1] = True
            back = cache.get(ids[b]).with_selection(sel2).read()
            assert back.to_pylist() == [rows[int(i)] for i in np.flatnonzero(sel2)]
        scan.close()
    finally:
        cache.close()


@pytest.mark.parametrize("like_path", [0, 4, 1])
def test_string_equality_through_the_scan_level_index(product_lib, oracle, like_path):
    """`=` / `<>` with a literal of 2 bytes or more: k_like_flat finds the dictionary values that CONTAIN the literal (its first
    63 bytes) and keeps those of its length (longer literals: compared byte by byte; byte_view_array/comparisons.rs:21-151
    compares the decoded value).  Literals that are values, that
    are proper substrings of values (of values under and over 255 bytes: the prefix key's length byte saturates there), that
    extend a value, that are absent; with and without a selection; against the oracle."""
    lo = oracle
    rng = np.random.default_rng(2163 + like_path)
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        base = b"http://example.org/equal/literal-of-forty-bytes"
        extra = [base, base + b"/longer", base + b"x" * 300, b"ab", b"abc", b"zab", base[:20]]
        ids, flat = _stage_entries(cache, lo, [(8192, 1500, False), (5000, 900, True), (8192, 2000, False), (300, 200, True),
                                               (16384, 2500, False)], rng, 51, extra=extra)
        scan = cache.scan(ids)
        values = [v for rows, _, _ in flat for v in rows if v is not None]
        lits = list(extra) + [base + b"/longe", base[1:], b"bc", b"zzzzqqq", b"example.org/equal"]
        lits += [values[int(i)] for i in rng.integers(0, len(values), size=8)]  # (most are longer than the 63-byte automaton)
        longv = max(values, key=len)
        lits += [longv, longv[:-1] + b"\x01", longv[:70], longv[:63], longv[:64]]
        lits = [x for x in lits if 2 <= len(x)]
        n = _check_scan(lo, scan, flat, lits, rng, "path %d" % like_path, ops=("eq", "ne"))
        assert n >= len(flat) * 2 * len(lits)
        expr = lc.LiquidExpr.try_new("=", base, pa.binary(), HINT)
        scan.eval_to_host(expr)  # (the plans of a scan are a small LRU: this literal's is the newest again)
        how = scan.explain(expr)
        if like_path in (0, 4):
            assert how.startswith("k_like_flat (string equality"), how
        else:
            assert how.startswith("k_str_pred"), how
        scan.close()
    finally:
        cache.close()
