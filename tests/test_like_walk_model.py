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


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the streaming walker (like_walk_stream) replaces the per-byte guard "is this byte still inside the value" by
# overwriting the bytes behind the end of a value with a PAD CODE chosen by k_str_automata: a code (not the escape marker)
# on which no automaton state reaches the matched state, and whose value, read as an escaped literal, is not the needle's
# last byte.  Model of both halves over the oracle's encoder: the rule that picks the code, and a 16-byte-block walk that
# pads instead of guarding — against the guarded walk and against a plain substring test, including streams that end in a
# dangling escape marker (corrupt input: the guarded walker never matches on the marker, nor may the padded one).
# ---------------------------------------------------------------------------------------------------------------------
def pick_pad_code(needle, sym):
    m, delta = len(needle), kmp_delta(needle)
    for code in range(255):
        if code == needle[m - 1]:
            continue
        completes = False
        for s in range(m):
            cur = s
            for by in sym[code]:          # (unused codes have an empty symbol: a self loop, the ideal pad)
                cur = delta[cur][by]
            completes |= cur == m
        if not completes:
            return code
    return None


def _walk_bytes(state, data, needle, sym, delta):
    m = len(needle)
    s, lit = state
    for c in data:
        if s == m and not lit:
            continue
        if lit:
            s, lit = delta[s][c], 0
        elif c == 255:
            lit = 1
        else:
            for by in sym[c]:
                s = delta[s][by]
    return (s, lit)


def stream_walk_model(stream, needle, sym, pad):
    """like_walk_stream for one value: 16-byte blocks, bytes behind the end replaced by the pad code, every block walked in
    full; the state is read after the value's last block."""
    m, delta = len(needle), kmp_delta(needle)
    state = (0, 0)
    nblocks = max(1, (len(stream) + 15) // 16)
    for b in range(nblocks):
        blk = stream[16 * b: 16 * b + 16]
        blk = blk + bytes([pad]) * (16 - len(blk))
        state = _walk_bytes(state, blk, needle, sym, delta)
    return state == (m, 0)


def test_padded_block_walk_equals_guarded_walk(oracle):
    lo = oracle
    rng = np.random.default_rng(11)
    st = adversarial_symtab(lo)
    sym = [int(st.sym[c]).to_bytes(8, "little")[: st.len[c]] if c < st.n else b"" for c in range(256)]
    values = sorted(set(s.encode() for s in adversarial_strings(rng, 1200)))
    checked = dangling = 0
    for needle in (b"mail", b"email", b"ail.r", b"a", b"//", b"ru/", b"google.com/search"):
        pad = pick_pad_code(needle, sym)
        assert pad is not None
        delta, m = kmp_delta(needle), len(needle)
        for v in values:
            stream = lo.fsst_compress(st, v)
            assert stream_walk_model(stream, needle, sym, pad) == (needle in v), (needle, v)
            checked += 1
            # corrupt tail: the value's bytes end in an escape marker without its literal
            bad = stream + b"\xff"
            guarded = _walk_bytes((0, 0), bad, needle, sym, delta) == (m, 0)
            assert stream_walk_model(bad, needle, sym, pad) == guarded, (needle, v)
            dangling += 1
    assert checked > 5000 and dangling > 5000


def test_pad_code_rule_on_full_tables(oracle):
    """Tables trained by the oracle's FSST trainer on the fuzz flavours (255 symbols: no unused code to fall back on): a
    pad code exists for every needle tried, and padding with it never turns a non-match into a match from ANY state."""
    import fuzz_data as fz
    lo = oracle
    n_tables = 0
    for seed in range(6):
        rows, st, _flavour = fz.make_case(lo, seed, n_rows=600, d=300)
        sym = [int(st.sym[c]).to_bytes(8, "little")[: st.len[c]] if c < st.n else b"" for c in range(256)]
        rng = np.random.default_rng(100 + seed)
        needles = [n for n in fz.make_needles(rng, [r for r in rows if r is not None], st, 12, True) if len(n) >= 1]
        for needle in needles:
            needle = bytes(needle)
            pad = pick_pad_code(needle, sym)
            assert pad is not None, (seed, needle)
            m, delta = len(needle), kmp_delta(needle)
            for s in range(m):             # from every unmatched state, code rows and literal rows alike
                for lit in (0, 1):
                    end = _walk_bytes((s, lit), bytes([pad]) * 16, needle, sym, delta)
                    assert end != (m, 0), (seed, needle, s, lit)
            assert _walk_bytes((m, 0), bytes([pad]) * 16, needle, sym, delta) == (m, 0)
        n_tables += 1
    assert n_tables == 6
