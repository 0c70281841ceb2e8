"""CPU model of the thread <-> row mapping of k_fixed_pred_reg (liquid_cache_amd/csrc/lc_kernels.hip).

The register-resident predicate kernel relies on index arithmetic over the FastLanes block layout
(bit_pack_array.rs:71-169 / fastlanes 0.5.0): which dwords a thread loads, where row r sits in its bit stream, and
which bit of which 64-row mask word a wave ballot of step r produces.  This test restates exactly that arithmetic in
Python (same formulas, same funnel shifts) over blocks packed by the ORACLE's FastLanes packer and checks that every
extracted field is the logical value the mask bit stands for.  It runs without a GPU, so a mapping mistake is caught
before any device time is spent; the device result itself is checked in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

M32 = 0xFFFFFFFF


def alignbit(hi, lo, s):
    """v_alignbit_b32: low 32 bits of {hi, lo} >> (s & 31)."""
    return (((hi << 32) | lo) >> (s & 31)) & M32


def field_top(w, r, W):
    """field_top<W, R>: row r of the thread's stream, left-aligned in 32 bits (junk below)."""
    pos = r * W
    k, off = pos >> 5, pos & 31
    if off + W <= 32:
        return (w[k] << (32 - off - W)) & M32
    return alignbit(w[k + 1], w[k], off + W - 32)


def _check(values, words_of_thread, thread_word_bit, n_rows, W):
    for t, w in words_of_thread.items():
        for r in range(n_rows):
            word, bit = thread_word_bit(t, r)
            got = field_top(w, r, W) >> (32 - W)
            want = int(values[word * 64 + bit])
            assert got == want, (t, r, word, bit, got, want)


@pytest.mark.parametrize("W", [1, 3, 4, 7, 12, 13, 16, 17, 24, 31, 32])
def test_u32_lanes(oracle, W):
    rng = np.random.default_rng(W)
    vals = rng.integers(0, 1 << W, size=1024, dtype=np.uint64).astype(np.uint32)
    packed = oracle.bitpack(vals, W).view(np.uint32)
    threads = {}
    for lane in range(32):  # block A of the pass (block B runs the same code on its own 128*W bytes)
        threads[lane] = [int(packed[k * 32 + lane]) for k in range(W)]
    _check(vals, threads, lambda l, r: (2 * (r & 7) + ((r >> 3) & 1), 32 * (r >> 4) + l), 32, W)


@pytest.mark.parametrize("W", [1, 2, 5, 8, 11, 12, 15, 16])
def test_u16_lanes(oracle, W):
    rng = np.random.default_rng(100 + W)
    vals = rng.integers(0, 1 << W, size=1024, dtype=np.uint64).astype(np.uint16)
    packed = oracle.bitpack(vals, W).view(np.uint16)
    nw = (16 * W + 31) // 32
    threads = {}
    for lane in range(64):
        w = []
        for k in range(nw):
            a = int(packed[(2 * k) * 64 + lane])
            b = int(packed[(2 * k + 1) * 64 + lane]) if 2 * k + 1 < W else 0
            w.append(a | (b << 16))
        threads[lane] = w
    _check(vals, threads, lambda l, r: (2 * (r & 7) + (r >> 3), l), 16, W)


@pytest.mark.parametrize("W", [1, 2, 3, 4, 5, 12, 13, 15, 16, 17, 20, 22, 29, 31, 32])
def test_u64_lanes(oracle, W):
    """u64 lanes in the shape of u32 lanes (load_stream64): thread (h, l) of a block owns rows 32 h .. 32 h + 31 of lane l,
    read as whole u64 words; odd widths select the upper dword for h = 1."""
    rng = np.random.default_rng(200 + W)
    vals = rng.integers(0, 1 << W, size=1024, dtype=np.uint64)
    raw = oracle.bitpack(vals, W)

    def qword_at(byte_off):  # an 8-byte global load
        assert 0 <= byte_off and byte_off + 8 <= raw.size, "load outside the block"
        return int(raw[byte_off:byte_off + 8].view(np.uint64)[0])

    threads = {}
    for lane32 in range(32):  # block A of the pair; block B runs the same code on its own 128*W bytes
        h, l = lane32 >> 4, lane32 & 15
        if W % 2 == 0:
            base = l * 8 + h * (W // 2) * 128
            d = []
            for m in range(W // 2):
                v = qword_at(base + m * 128)
                d += [v & M32, v >> 32]
            w = d
        else:
            nq = (W + 1) // 2
            base = l * 8 + h * ((W - 1) // 2) * 128
            d = []
            for m in range(nq):
                v = qword_at(base + m * 128)
                d += [v & M32, v >> 32]
            w = [d[k + 1] if h else d[k] for k in range(W)]
        threads[lane32] = w
    # the u32-lane rule: step r -> bits [32 (r >> 4), +32) of word 2 (r & 7) + ((r >> 3) & 1); bit = 32 (r >> 4) + 16 h + l
    _check(vals, threads, lambda t, r: (2 * (r & 7) + ((r >> 3) & 1), 32 * (r >> 4) + t), 32, W)


def test_top_aligned_range_compare_is_the_masked_compare():
    """(u - lo) <= span  <=>  (t - (lo << (32-W))) <= (span << (32-W) | ones), t = u << (32-W) | junk (mod 2^32)."""
    rng = np.random.default_rng(5)
    for W in (1, 4, 12, 13, 31, 32):
        umax = (1 << W) - 1
        for _ in range(200):
            lo = int(rng.integers(0, umax + 1))
            span = int(rng.integers(0, umax - lo + 1))
            u = rng.integers(0, umax + 1, size=64, dtype=np.uint64)
            junk = rng.integers(0, 1 << (32 - W), size=64, dtype=np.uint64) if W < 32 else np.zeros(64, np.uint64)
            t = ((u << np.uint64(32 - W)) | junk) & np.uint64(M32)
            lo_t = (lo << (32 - W)) & M32
            span_t = ((span << (32 - W)) | ((1 << (32 - W)) - 1)) & M32
            got = ((t - np.uint64(lo_t)) & np.uint64(M32)) <= np.uint64(span_t)
            want = ((u - np.uint64(lo)) & np.uint64(umax)) <= np.uint64(span)
            assert got.tolist() == want.tolist()
