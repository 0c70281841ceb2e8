"""Adversarial data for the LIKE walkers (shared by tests/test_gpu_round2.py and tests/test_like_walk_model.py)."""
import numpy as np  # noqa: F401


def adversarial_symtab(lo):
    """A symbol table in which the codes of bytes that NO symbol covers (so they are always escaped in a stream: marker 255,
    then the literal) hold symbols made of needle pieces: reading such a literal as a code 'decodes' text the value does
    not contain."""
    traps = {ord("#"): b"mail", ord("_"): b"ail.", ord("?"): b"gmai", ord("&"): b"l.ru", ord("="): b"email",
             ord("%"): b"mai", ord("+"): b"ilbox"}
    covered = b"abcdefghijklmnopqrstuvwxyz0123456789/:.-"
    extra = [b"ht", b"tp", b"://", b"ru/", b".com", b"www.", b"ya", b"go", b"le"]
    syms = {}
    free = iter(c for c in range(255) if c not in traps)
    for ch in covered:
        syms[next(free)] = bytes([ch])
    for e in extra:
        syms[next(free)] = e
    syms.update(traps)
    n = max(syms) + 1
    st = lo.SymTab()
    st.n = n
    for c in range(n):
        b = syms.get(c, b"\x01")  # unused codes: a byte the data never holds
        st.len[c] = len(b)
        st.sym[c] = int.from_bytes(b, "little")
    return lo.symtab_load(lo.symtab_bytes(st))


def adversarial_strings(rng, n):
    alpha, traps = b"abcdefghijklmnopqrstuvwxyz0123456789/:.-", b"#_?&=%+"
    pieces = [b"ma", b"ai", b"il", b"gm", b"em", b"l.", b".r", b"ru", b"bo", b"ox", b"lb"]   # every needle bigram, rarely a needle
    rare = [b"mail", b"gmail", b"email", b"mailbox", b"ail.ru"]
    out = []
    for _ in range(n):
        length, s = int(rng.integers(3, 100)), bytearray()
        while len(s) < length:
            r = rng.random()
            if r < 0.18:
                s.append(traps[int(rng.integers(len(traps)))])
            elif r < 0.40:
                s += pieces[int(rng.integers(len(pieces)))]
            elif r < 0.405:
                s += rare[int(rng.integers(len(rare)))]
            else:
                s.append(alpha[int(rng.integers(len(alpha)))])
        out.append(bytes(s).decode())
    return out


