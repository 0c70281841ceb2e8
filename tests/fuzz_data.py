"""Seeded byte-level fuzz data for the byte-view predicates, modelled on the reference's fuzz target
(fuzz/fuzz_targets/fsst_view.rs:48-117: arbitrary strings, five (needle, operator) pairs, result must equal Arrow's).

Shared by tests/test_fuzz_oracle.py (CPU: the oracle against plain Python on this data) and tests/test_gpu_round3.py
(GPU: the HIP path against the oracle).  What the round-2 tests could not produce and this does: many independently
trained symbol tables, the full 0..255 alphabet, data that forces FSST escapes (bytes no symbol covers, tables trained on
OTHER data), needles cut at symbol boundaries and out of the symbols themselves.
"""
import numpy as np

FLAVOURS = ("bytes", "small_alphabet", "urls", "foreign_table", "escape_heavy", "adversarial")
LIKE_SPECIAL = frozenset(b"%_\\")  # a needle holding one is not a plain %needle% pattern (liquid_expr.rs / Arrow `like`)


def _zipf_rows(rng, pool, n):
    keys = np.minimum(rng.zipf(1.25, size=n) - 1, len(pool) - 1)
    return [pool[int(k)] for k in keys]


def _pool_bytes(rng, d):
    """Every byte value, short and long values, runs of 0xFF (the escape marker's value as DATA)."""
    out = []
    for _ in range(d):
        ln = int(rng.choice([0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 40, 130, 300], p=[.02, .05, .05, .08, .1, .1, .1, .1, .1, .1, .1, .07, .03]))
        b = bytearray(rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes())
        if ln and rng.random() < 0.3:
            k = int(rng.integers(ln))
            b[k: k + int(rng.integers(1, 4))] = b"\xff" * int(rng.integers(1, 4))
        out.append(bytes(b))
    return out


def _pool_small_alphabet(rng, d):
    """Four letters: the trained table is full of 2..8-byte symbols, values are a handful of codes."""
    return [bytes(rng.choice(list(b"abcd"), size=int(rng.integers(0, 60))).astype(np.uint8).tobytes()) for _ in range(d)]


def _pool_urls(rng, d):
    hosts = ["google", "yandex", "mail", "маркет", "почта", "example", "go", "ogle", "gle.goo", "аб"]
    paths = ["search", "q=%D0%BF%D0%BE", "index.php?id=", "%2F%2F", "поиск", "a", "", "tours", "&page="]
    out = []
    for i in range(d):
        s = ("https://" if rng.random() < 0.5 else "http://") + ".".join(
            hosts[int(rng.integers(len(hosts)))] for _ in range(int(rng.integers(1, 4))))
        for _ in range(int(rng.integers(0, 5))):
            s += "/" + paths[int(rng.integers(len(paths)))] + str(int(rng.integers(0, 50)))
        if rng.random() < 0.04:
            s += "x" * int(rng.integers(250, 600))
        out.append(s.encode())
    return out


def _pool_escape_heavy(rng, d):
    """Text over a few letters interleaved with bytes drawn from a large set that training rarely turns into symbols."""
    rare = list(range(128, 256)) + list(b"#?&=+~^|")
    out = []
    for _ in range(d):
        b = bytearray()
        for _ in range(int(rng.integers(1, 50))):
            if rng.random() < 0.35:
                b.append(rare[int(rng.integers(len(rare)))])
            else:
                b += [b"ma", b"il", b"go", b"og", b"le", b"ai", b"gm", b"x"][int(rng.integers(8))]
        out.append(bytes(b))
    return out


def make_case(lo, seed, n_rows=1500, d=400, nulls=None):
    """One column batch: (values: list of bytes|None, symbol table, flavour).  The symbol table is trained by the oracle's
    FSST trainer on this batch — or, for 'foreign_table', on a different batch (most bytes then travel as escapes), or is
    the hand-built trap table of tests/like_adversarial.py."""
    rng = np.random.default_rng(1000 + seed)
    flavour = FLAVOURS[seed % len(FLAVOURS)]
    st = None
    if flavour == "bytes":
        pool = _pool_bytes(rng, d)
    elif flavour == "small_alphabet":
        pool = _pool_small_alphabet(rng, d)
    elif flavour == "urls":
        pool = _pool_urls(rng, d)
    elif flavour == "foreign_table":
        pool = _pool_urls(rng, d) if seed % 2 else _pool_bytes(rng, d)
        other = _pool_small_alphabet(rng, 200) + _pool_escape_heavy(rng, 100)
        o, dt, _ = lo.strings_to_arrow(other)
        st = lo.fsst_train(o, dt)
    elif flavour == "escape_heavy":
        pool = _pool_escape_heavy(rng, d)
    else:
        from like_adversarial import adversarial_strings, adversarial_symtab
        pool = [s.encode() for s in adversarial_strings(rng, d)]
        st = adversarial_symtab(lo)
    pool = list(dict.fromkeys(pool)) or [b""]
    rows = _zipf_rows(rng, pool, n_rows)
    if nulls is None:
        nulls = seed % 3 == 0
    if nulls:
        for i in rng.choice(n_rows, size=max(1, n_rows // 15), replace=False):
            rows[int(i)] = None
    if st is None:
        nn = [r for r in rows if r is not None] or [b""]
        o, dt, _ = lo.strings_to_arrow(nn)
        st = lo.fsst_train(o, dt)
    return rows, st, flavour


def symbols_of(st):
    return [int(st.sym[c]).to_bytes(8, "little")[: int(st.len[c])] for c in range(int(st.n)) if int(st.len[c])]


def make_needles(rng, rows, st, k, for_like):
    """k needles: substrings of values (random cuts and cuts at multiples of 8 compressed-ish positions), symbols and
    concatenations / halves of symbols, whole values, random bytes, the empty needle (Eq / ordering only)."""
    vals = [r for r in rows if r]
    syms = symbols_of(st) or [b"a"]
    out = []
    while len(out) < k:
        r = rng.random()
        if r < 0.40 and vals:
            v = vals[int(rng.integers(len(vals)))]
            a = int(rng.integers(len(v)))
            nd = v[a: a + int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 15, 16, 17, 31, 40, 70]))]
        elif r < 0.60:
            nd = b"".join(syms[int(rng.integers(len(syms)))] for _ in range(int(rng.integers(1, 4))))
            if rng.random() < 0.5 and len(nd) > 1:
                nd = nd[int(rng.integers(len(nd) - 1)) + (0 if rng.random() < 0.5 else 1):] or nd
        elif r < 0.72 and vals:
            nd = vals[int(rng.integers(len(vals)))]
            if rng.random() < 0.3:
                nd = nd + bytes([int(rng.integers(256))])
        elif r < 0.82:
            nd = rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8).tobytes()
        elif r < 0.90:
            nd = b"\xff" * int(rng.integers(1, 4))
        elif not for_like:
            nd = b""
        else:
            continue
        if for_like:
            nd = bytes(c for c in nd if c not in LIKE_SPECIAL)
            if not nd or len(nd) > 63:
                continue
        out.append(nd)
    return out


def python_truth(rows, op, needle):
    """Arrow semantics on raw bytes: [None | bool] per row."""
    import operator
    f = {"eq": operator.eq, "ne": operator.ne, "lt": operator.lt, "le": operator.le, "gt": operator.gt, "ge": operator.ge,
         "like": lambda v, nd: nd in v, "not_like": lambda v, nd: nd not in v}[op]
    return [None if v is None else bool(f(v, needle)) for v in rows]


def is_utf8(b):
    try:
        b.decode("utf-8")
        return True
    except UnicodeDecodeError:
        return False
