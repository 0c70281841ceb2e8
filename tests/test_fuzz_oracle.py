"""The CPU oracle against plain Python on the byte-level fuzz data (tests/fuzz_data.py) — no GPU needed.

Modelled on the reference's fuzz target (fuzz/fuzz_targets/fsst_view.rs:48-117): round trip and compare_with == Arrow for
random (needle, operator) pairs; here over many independently trained symbol tables and with the substring operators as
well.  The GPU fuzz (tests/test_gpu_round3.py) checks the HIP path against the oracle on the same cases, so this file is
what ties that comparison to ground truth.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_data as fz  # noqa: E402

OPS = ("eq", "ne", "lt", "le", "gt", "ge")


def _tri(result, n):
    return [None if (result.validity is not None and not result.validity[i]) else bool(result.values[i]) for i in range(n)]


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_python_on_fuzz_cases(oracle, seed):
    lo = oracle
    rows, st, flavour = fz.make_case(lo, seed, n_rows=600, d=200)
    rng = np.random.default_rng(seed)
    utf8 = all(v is None or fz.is_utf8(v) for v in rows)
    for fingerprints in (True, False):
        liquid, _ = lo.encode_byte_view(rows, st=st, fingerprints=fingerprints, arrow_type=lo.BT_BINARY)
        assert lo.filter_byte_view(liquid, st) == rows, flavour  # round trip (fsst_view.rs:66-83)
        for op in OPS:
            for nd in fz.make_needles(rng, rows, st, 5, for_like=False):
                got = _tri(lo.eval_predicate(liquid, lo.OP_NAMES[op], nd, symtab=st), len(rows))
                assert got == fz.python_truth(rows, op, nd), (flavour, op, nd)
        for nd in fz.make_needles(rng, rows, st, 8, for_like=True):
            pat = b"%" + nd + b"%"
            # Without fingerprints the reference runs Arrow's `like` (helpers.rs:86-91), which is defined on strings: it
            # is byte-exact substring search only for valid UTF-8 values and needles (a SQL pattern always is one).
            if not fingerprints and not (utf8 and fz.is_utf8(nd)):
                continue
            got = _tri(lo.eval_predicate(liquid, lo.LIKE, pat, symtab=st), len(rows))
            assert got == fz.python_truth(rows, "like", nd), (flavour, "like", nd)
            got = _tri(lo.eval_predicate(liquid, lo.NOT_LIKE, pat, symtab=st), len(rows))
            want = fz.python_truth(rows, "not_like", nd)
            if fingerprints:
                # comparisons.rs:167-180: NOT LIKE inverts only when some dictionary value passes the fingerprint filter
                nfp = lo.fingerprint(nd)
                if not any(v is not None and (lo.fingerprint(v) & nfp) == nfp for v in rows):
                    want = [None if v is None else False for v in rows]
            assert got == want, (flavour, "not_like", nd)
        if not fingerprints:
            continue
        sel = rng.random(len(rows)) < 0.3
        nd = fz.make_needles(rng, rows, st, 1, for_like=True)[0]
        got = lo.eval_predicate(liquid, lo.LIKE, b"%" + nd + b"%", sel, symtab=st)
        want = [t for t, s in zip(fz.python_truth(rows, "like", nd), sel) if s]
        assert _tri(got, int(sel.sum())) == want
