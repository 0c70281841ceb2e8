"""Host logic of the pushdown orchestration (no GPU): conjunct priorities as in row_filter.rs:499-515, the ClickBench
predicate table against the reference's query files, and the fusion plan."""
import os
import re

import pyarrow as pa
import pytest

from liquid_cache_amd import clickbench as cb
from liquid_cache_amd.pushdown import AnyOf, Column, Conjunct, LiquidRowFilter, PushdownExecutor, get_priority


def test_priorities_follow_row_filter_rs():
    assert get_priority(Conjunct("a", "=", 1)) == 0 and get_priority(Conjunct("a", "!=", 1)) == 0
    assert get_priority(Conjunct("a", "like", "%x%")) == 1
    assert get_priority(Conjunct("a", "not like", "%x%")) == 2
    for op in ("<", "<=", ">", ">="):
        assert get_priority(Conjunct("a", op, 1)) == 3
    assert get_priority(AnyOf([Conjunct("a", "=", 1), Conjunct("b", "=", 2)])) == 4     # BinaryExpr(Or)
    assert get_priority(AnyOf([Conjunct("a", "=", 1), Conjunct("a", "=", 2)])) == 5     # InListExpr
    # q22: Title LIKE, URL NOT LIKE, SearchPhrase <> ''  ->  NotEq first, then LIKE, then NOT LIKE
    order = [(p.column, p.op) for p in LiquidRowFilter(cb.QUERIES[22]).predicates]
    assert order == [("SearchPhrase", "!="), ("Title", "like"), ("URL", "not like")]
    order = [p.column for p in LiquidRowFilter(cb.QUERIES[21]).predicates]
    assert order == ["SearchPhrase", "URL"]


REF_QUERIES = "/root/reference/benchmark/clickbench/queries"


@pytest.mark.skipif(not os.path.isdir(REF_QUERIES), reason="reference checkout not present (GPU box)")
def test_query_table_matches_the_reference_query_files():
    """Every q<n>.sql with a WHERE clause has an entry, with one conjunct per top-level AND term naming the same column."""
    for n in range(cb.N_QUERIES):
        sql = open(os.path.join(REF_QUERIES, "q%d.sql" % n)).read()
        m = re.search(r"WHERE(.*?)(GROUP BY|ORDER BY|LIMIT|;|$)", sql, re.S | re.I)
        if not m:
            assert n not in cb.QUERIES, n
            continue
        terms = [t.strip() for t in re.split(r"\bAND\b", m.group(1), flags=re.I) if t.strip()]
        conj = cb.QUERIES[n]
        assert len(conj) == len(terms), (n, terms)
        for term, c in zip(terms, conj):
            col = re.match(r'"(\w+)"', term).group(1)
            got = c.terms[0].column if isinstance(c, AnyOf) else c.column
            assert got == col, (n, term)
            lit = c.terms[0].literal if isinstance(c, AnyOf) else c.literal
            if isinstance(lit, str) and lit:
                assert lit in term, (n, term)
            elif isinstance(lit, int) and "::DATE" not in term and not isinstance(c, AnyOf):
                assert str(lit) in term, (n, term)
    assert sorted(cb.QUERIES) == [n for n in range(43) if "WHERE" in open(os.path.join(REF_QUERIES, "q%d.sql" % n)).read().upper()]


class _FakeScan:
    mask_words = 1
    entries = 1


def test_plan_fuses_adjacent_ranges_and_keeps_or_groups():
    cols = {nm: Column(_FakeScan(), cb.SCHEMA[nm][1], cb.SCHEMA[nm][2], fixed_width=nm not in cb._STRINGS)
            for nm in cb.columns_of()}
    ex = PushdownExecutor(cols)
    steps = ex.plan(LiquidRowFilter(cb.QUERIES[40]))
    kinds = [(s.kind, len(s.exprs)) for s in steps]
    # CounterID =, IsRefresh =, RefererHash = (priority 0, three different columns), EventDate >= AND <= fused, IN list
    assert kinds == [("and", 1), ("and", 1), ("and", 1), ("and", 2), ("or", 2)]
    assert [(s.kind, len(s.exprs)) for s in PushdownExecutor(cols, fuse_ranges=False).plan(LiquidRowFilter(cb.QUERIES[42]))] \
        == [("and", 1)] * 5
    # strings never fuse
    assert all(len(s.exprs) == 1 for s in ex.plan(LiquidRowFilter(cb.QUERIES[22])))
    with pytest.raises(ValueError):
        ex.plan(LiquidRowFilter([Conjunct("URL", "like", 5)]))


def test_synthetic_columns_have_the_schema_types_and_are_deterministic():
    for nm in cb.columns_of():
        a, b = cb.synth_batch(nm, 3, 2, 1000), cb.synth_batch(nm, 3, 2, 1000)
        assert a.type == cb.SCHEMA[nm][1] and len(a) == 1000 and a.equals(b)
    assert cb.synth_batch("URL", 3, 2, 1000) != cb.synth_batch("URL", 3, 3, 1000)
