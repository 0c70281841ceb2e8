"""CPU model of the lane-parallel LIKE walk of k_str_pred (liquid_cache_amd/csrc/lc_kernels.hip).

The kernel cuts a candidate's FSST bytes into 8-byte words, one lane per word.  Every lane first walks its word through the
needle automaton folded over the symbol table from state 0 / "next byte is a code"; then lane i takes lane i-1's end state
(automaton state + "next byte is an escaped literal") as its start state and re-walks, until no start state changes.  A
candidate matches iff some word's walk AT THAT FIXPOINT ends in the (absorbing) matched state.

Round 2 shipped the rule "a match found by ANY of the walks counts" for most of the round.  A word that follows an escape
marker starts with a literal; read as a code it expands to a symbol the value does not contain, so speculative walks can
see needle text that is not there (`%mail%` over the bench column: 3,112 rows too many).  This test restates both rules in
Python over adversarial data (tests/like_adversarial.py) compressed by the ORACLE's FSST encoder and checks that the
fixpoint rule equals a plain substring test while the old rule does not — i.e. that the data is a real detector.  It runs
without a GPU; the device result on the same data is checked in tests/test_gpu_round2.py.
"""
import numpy as np

from like_adversarial import adversarial_strings, adversarial_symtab


def kmp_delta(nd):
    m = len(nd)
    fail, k = [0] * (m + 2), 0
    for i in range(1, m):
        while k > 0 and nd[i] != nd[k]:
            k = fail[k]
        if nd[i] == nd[k]:
            k += 1
        fail[i + 1] = k
    delta = [[0] * 256 for _ in range(m + 1)]
    for s in range(m + 1):
        for b in range(256):
            delta[s][b] = m if s == m else (s + 1 if nd[s] == b else (0 if s == 0 else delta[fail[s]][b]))
    return delta


def walk_model(stream, needle, sym):
    """-> (match by the old rule, match by the fixpoint rule) for one compressed value."""
    m, delta = len(needle), kmp_delta(needle)

    def walk(state, word):  # k_str_automata's image: rows 0..m "next byte is a code", rows m+1.. "next byte is a literal"
        s, lit = state
        for c in word:
            if s == m and not lit:
                continue                       # matched state is absorbing (entry[255] of row m stays in row m)
            if lit:
                s, lit = delta[s][c], 0
            elif c == 255:
                lit = 1
            else:
                for by in sym[c]:
                    s = delta[s][by]
        return (s, lit)

    words = [stream[i:i + 8] for i in range(0, max(len(stream), 1), 8)] or [b""]
    n = len(words)
    s_in = [(0, 0)] * n
    e = [walk((0, 0), w) for w in words]
    any_hit = [x == (m, 0) for x in e]
    while True:
        prev = [(0, 0)] + e[:-1]
        prev = [(0, 0) if (i == 0 or p == (m, 0)) else p for i, p in enumerate(prev)]
        changed = [prev[i] != s_in[i] for i in range(n)]
        if not any(changed):
            break
        for i in range(n):
            if changed[i]:
                s_in[i] = prev[i]
                e[i] = walk(s_in[i], words[i])
                any_hit[i] = any_hit[i] or e[i] == (m, 0)
    return any(any_hit), any(x == (m, 0) for x in e)


def test_fixpoint_rule_is_exact_and_the_old_rule_is_not(oracle):
    lo = oracle
    rng = np.random.default_rng(7)
    st = adversarial_symtab(lo)
    sym = [int(st.sym[c]).to_bytes(8, "little")[: st.len[c]] for c in range(256)]
    values = sorted(set(s.encode() for s in adversarial_strings(rng, 1500)))
    old_false = 0
    for needle in (b"mail", b"email", b"ail.r"):
        for v in values:
            stream = lo.fsst_compress(st, v)
            assert lo.fsst_decompress(st, stream) == v
            old, new = walk_model(stream, needle, sym)
            assert new == (needle in v), (needle, v)
            old_false += old and needle not in v
    assert old_false > 50   # the adversarial data does expose the old rule
