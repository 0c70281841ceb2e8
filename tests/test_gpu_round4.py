"""Round 4: the scan-level signature index (k_like_flat) against the oracle.

A scan whose entries SHARE symbol tables in runs (row groups), so that a wave's group really holds several entries:
dictionaries from one value to 2,500, row counts that are no multiple of 64, all-null entries inside a group, a group cut
by the 128-word limit and by a symbol-table change.  LIKE / NOT LIKE, with and without a selection, through k_like_flat for
every needle, through the automatic plan, and through k_like_lean — all equal to the oracle's per-entry masks and counts
(reference: byte_view_array/comparisons.rs:159-183, 598-651).
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

# (rows, distinct values, nulls) per entry; None = an all-null entry
RUNS = [
    [(8192, 2200, False), (8192, 2250, True), (8192, 2150, False), (8192, 2200, False), (10, 8, False)],
    [(65, 40, True), None, (700, 300, False), (1500, 1, False), (64, 64, False), (129, 3, True), (2049, 900, False),
     (8192, 2500, True), (8192, 2500, False), (8192, 2500, False), (8192, 2500, False), (1, 1, False)],
    [(8192, 5000, False), (8192, 4100, True), (8192, 60, False), (100, 100, False)],
]


def _want_full(lo, liquid, st, op, literal, sel, n):
    r = lo.eval_predicate(liquid, op, literal, sel, symtab=st)
    v = r.values if r.validity is None else (r.values & r.validity)
    hit = np.zeros(n, bool)
    if sel is None:
        hit[:] = v
    else:
        hit[np.flatnonzero(sel)] = v
    return hit


@pytest.fixture(scope="module")
def grouped_cases(oracle):
    lo = oracle
    rng = np.random.default_rng(77)
    runs = []
    for r_i, run in enumerate(RUNS):
        pools = []
        for e_i, spec in enumerate(run):
            if spec is None:
                pools.append(None)
                continue
            n, d, nulls = spec
            pool = fz._pool_urls(rng, d) if (e_i + r_i) % 3 else fz._pool_escape_heavy(rng, d)
            pool = list(dict.fromkeys(pool))
            while len(pool) < d:  # distinct values, exactly d of them
                pool.append(pool[int(rng.integers(len(pool)))] + b"#%d" % len(pool))
            pool = pool[:d]
            rows = [pool[i] for i in range(min(d, n))] + [pool[int(k)] for k in rng.integers(0, d, size=max(0, n - d))]
            rows = [rows[int(i)] for i in rng.permutation(len(rows))]
            if nulls:
                for i in rng.choice(n, size=max(1, n // 15), replace=False):
                    rows[int(i)] = None
            pools.append(rows)
        train = [v for rows in pools if rows for v in rows if v is not None][:20000] or [b""]
        o, dt, _ = lo.strings_to_arrow(train)
        st = lo.fsst_train(o, dt)
        entries = []
        for rows in pools:
            if rows is None:
                rows = [None] * 37
            liquid, _ = lo.encode_byte_view(rows, st=st, fingerprints=True, arrow_type=lo.BT_BINARY)
            entries.append((rows, liquid))
        runs.append((st, entries))
    return runs


@pytest.mark.parametrize("like_path", [4, 0, 3])
def test_flat_index_groups_against_oracle(product_lib, oracle, grouped_cases, like_path):
    lo = oracle
    cache = lc.LiquidCacheBuilder.new().with_index_options(like_pipeline_min_entries=1, like_path=like_path or None).build()
    try:
        ids, flat = [], []
        for r_i, (st, entries) in enumerate(grouped_cases):
            path = 7000 + r_i
            cache.set_symbol_table(path, lo.symtab_bytes(st))
            for e_i, (rows, liquid) in enumerate(entries):
                eid = lc.ParquetArrayID.new(11, r_i, 5, e_i)
                cache.stage([eid], [liquid], [path])
                ids.append(eid)
                flat.append((rows, liquid, st))
        scan = cache.scan(ids)
        lens = [len(c[0]) for c in flat]
        offs = scan.segment_offsets
        rng = np.random.default_rng(99)
        needles = [b"google", b"mail", b"go", b"a", b"yandex.google", b"search1", b"zzzzqqq", b"http://", b"tours4",
                   b"index.php?id=1", b"\xd0\xbf\xd0\xbe", b"#21", b"gm", b"ogle.goo.gle", b"//"]
        for rows, _, st in (flat[1], flat[7], flat[-3]):
            needles += fz.make_needles(rng, rows, st, 3, for_like=True)
        n_checked = 0
        for qi, nd in enumerate(needles):
            for op, with_sel in (("like", qi % 2 == 0), ("not_like", qi % 3 == 0), ("like", qi % 2 == 1)):
                sels, words = [None] * len(flat), None
                if with_sel:
                    words = np.zeros(int(scan.mask_words), np.uint64)
                    for b, n in enumerate(lens):
                        se = rng.random(n) < [0.02, 0.5, 0.97][b % 3]
                        if b % 7 == 3:
                            se[:] = False
                        sels[b] = se
                        packed = np.packbits(se, bitorder="little")
                        words[int(offs[b]): int(offs[b + 1])].view(np.uint8)[: len(packed)] = packed
                expr = lc.LiquidExpr.try_new(op, b"%" + nd + b"%", pa.binary(), HINT)
                mask, counts = scan.eval_to_host(expr, selection=words)
                bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
                for b, (rows, liquid, st) in enumerate(flat):
                    got = bits[int(offs[b]) * 64: int(offs[b]) * 64 + lens[b]].astype(bool)
                    want = _want_full(lo, liquid, st, lo.OP_NAMES[op], b"%" + nd + b"%", sels[b], lens[b])
                    assert np.array_equal(got, want), (like_path, b, op, nd, with_sel, int(got.sum()), int(want.sum()))
                    assert int(counts[b]) == int(want.sum()), (like_path, b, op, nd)
                    tail = bits[int(offs[b]) * 64 + lens[b]: int(offs[b + 1]) * 64]
                    assert not tail.any()
                    n_checked += 1
        assert n_checked >= len(flat) * 3 * len(needles)
        # the evaluation path is what was asked for
        how = scan.explain(lc.LiquidExpr.try_new("like", b"%google%", pa.binary(), HINT))
        if like_path == 4:
            assert how.startswith("k_like_flat"), how
        if like_path == 3:
            assert how.startswith("k_like_lean"), how
        scan.close()
    finally:
        cache.close()


# ------------------------------------------------------------------------------------------------------------------
# concurrent callers at the boundary (tests/c_abi/concurrent_callers.c: pthreads, no Python in the measured region)
# ------------------------------------------------------------------------------------------------------------------
def _build_concurrent(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "concurrent_callers")
    libdir = os.path.join(root, "liquid_cache_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "concurrent_callers.c"), "-o", exe,
                    "-L", libdir, "-l:libliquid_cache_amd.so", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_eight_threads_with_their_own_streams_while_a_ninth_stages_and_evicts(product_lib, tmp_path):
    """Eight host threads x distinct streams loop over lc_eval_predicate / lc_eval_predicate_batch / lc_scan_eval /
    lc_get_with_selection on different columns while a ninth thread stages, re-stages and evicts: every answer equals the
    single-threaded one and the eight together take less than twice one alone (no device-wide synchronise in any call;
    reference: liquid_cache_reader.rs:297-391, cache/index.rs:30-34)."""
    import subprocess
    exe = _build_concurrent(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "concurrent callers ok" in r.stdout
