/*
 * Arrow-semantics helpers used by the reference's generic predicate path (TEST ORACLE — see lo_common.h).
 *
 * The reference bottoms out in arrow-rs 58.1 kernels (third-party, not vendored):
 *   arrow::compute::filter           <- liquid_array/mod.rs:117-121, primitive_array.rs:370-374
 *   arrow cmp (eq/neq/lt/..) vs a scalar via DataFusion BinaryExpr  <- liquid_array/mod.rs:265-280
 *   arrow like                       <- byte_view_array/comparisons.rs:619-651
 *   prep_null_mask_filter            <- datafusion/src/cache/column.rs:137-140
 * and in its own selection-compaction inverse:
 *   boolean_buffer_and_then          <- datafusion/src/utils.rs:17-45 (fallback), :62-83
 * Their published semantics are restated here.  Deviation from pyarrow that matters: arrow-rs compares
 * floats with IEEE-754 totalOrder.
 */
#include "lo_arrowish.h"

/* arrow::compute::filter on a fixed-width values buffer: keep rows whose selection bit is set */
size_t lo_filter_values(int width, const void* values, size_t n, const uint8_t* sel, void* out) {
    const uint8_t* in = (const uint8_t*)values;
    uint8_t* o = (uint8_t*)out;
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (lo_get_bit(sel, i)) {
            memcpy(o + k * (size_t)width, in + i * (size_t)width, (size_t)width);
            k++;
        }
    }
    return k;
}

/* arrow::compute::filter applied to a bitmap (validity / boolean values): bit-compress by selection */
size_t lo_filter_bits(const uint8_t* bits, size_t n, const uint8_t* sel, uint8_t* out) {
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (lo_get_bit(sel, i)) {
            if ((k & 7) == 0) out[k >> 3] = 0;
            if (lo_get_bit(bits, i)) lo_set_bit(out, k);
            k++;
        }
    }
    return k;
}

/* f64/f32 totalOrder keys (arrow-rs ArrowNativeTypeOp::compare for floats == total_cmp) */
static inline int64_t total_key_f64(uint64_t bits) {
    int64_t l = (int64_t)bits;
    l ^= (int64_t)(((uint64_t)(l >> 63)) >> 1);
    return l;
}
static inline int32_t total_key_f32(uint32_t bits) {
    int32_t l = (int32_t)bits;
    l ^= (int32_t)(((uint32_t)(l >> 31)) >> 1);
    return l;
}

static inline int apply_op(int op, int cmp /* -1,0,1 */) {
    switch (op) {
        case LO_EQ: return cmp == 0;
        case LO_NE: return cmp != 0;
        case LO_LT: return cmp < 0;
        case LO_LE: return cmp <= 0;
        case LO_GT: return cmp > 0;
        case LO_GE: return cmp >= 0;
        default: return 0;
    }
}

#define CMP3(a, b) (((a) > (b)) - ((a) < (b)))

/* arrow cmp kernel, array vs scalar.  kind: 0 signed, 1 unsigned, 2 f32, 3 f64, 4 i128.
 * values are `width` bytes each; literal is widened to the comparison domain by the caller. */
int lo_cmp_scalar(int kind, int width, const void* values, size_t k, int op, const void* literal, uint8_t* out_bits) {
    memset(out_bits, 0, lo_bm_bytes(k));
    const uint8_t* p = (const uint8_t*)values;
    if (kind == 0) {
        int64_t lit; memcpy(&lit, literal, 8);
        for (size_t i = 0; i < k; i++) {
            int64_t v;
            switch (width) {
                case 1: v = *(const int8_t*)(p + i); break;
                case 2: { int16_t t; memcpy(&t, p + 2 * i, 2); v = t; break; }
                case 4: { int32_t t; memcpy(&t, p + 4 * i, 4); v = t; break; }
                default: memcpy(&v, p + 8 * i, 8); break;
            }
            if (apply_op(op, CMP3(v, lit))) lo_set_bit(out_bits, i);
        }
    } else if (kind == 1) {
        uint64_t lit; memcpy(&lit, literal, 8);
        for (size_t i = 0; i < k; i++) {
            uint64_t v;
            switch (width) {
                case 1: v = p[i]; break;
                case 2: { uint16_t t; memcpy(&t, p + 2 * i, 2); v = t; break; }
                case 4: { uint32_t t; memcpy(&t, p + 4 * i, 4); v = t; break; }
                default: memcpy(&v, p + 8 * i, 8); break;
            }
            if (apply_op(op, CMP3(v, lit))) lo_set_bit(out_bits, i);
        }
    } else if (kind == 2) {
        uint32_t lb; memcpy(&lb, literal, 4);
        int32_t lk = total_key_f32(lb);
        for (size_t i = 0; i < k; i++) {
            uint32_t vb; memcpy(&vb, p + 4 * i, 4);
            int32_t vk = total_key_f32(vb);
            if (apply_op(op, CMP3(vk, lk))) lo_set_bit(out_bits, i);
        }
    } else if (kind == 3) {
        uint64_t lb; memcpy(&lb, literal, 8);
        int64_t lk = total_key_f64(lb);
        for (size_t i = 0; i < k; i++) {
            uint64_t vb; memcpy(&vb, p + 8 * i, 8);
            int64_t vk = total_key_f64(vb);
            if (apply_op(op, CMP3(vk, lk))) lo_set_bit(out_bits, i);
        }
    } else if (kind == 4) {
        __int128 lit; memcpy(&lit, literal, 16);
        for (size_t i = 0; i < k; i++) {
            __int128 v; memcpy(&v, p + 16 * i, 16);
            if (apply_op(op, CMP3(v, lit))) lo_set_bit(out_bits, i);
        }
    } else {
        return LO_ERR_ARG;
    }
    return LO_OK;
}

/* prep_null_mask_filter: values & validity (nulls become false) */
void lo_prep_null_mask(const uint8_t* values, const uint8_t* validity, size_t nbits, uint8_t* out) {
    size_t nb = lo_bm_bytes(nbits);
    for (size_t i = 0; i < nb; i++) out[i] = validity ? (uint8_t)(values[i] & validity[i]) : values[i];
    if (nbits & 7) out[nb - 1] &= (uint8_t)((1u << (nbits & 7)) - 1);
}

/* datafusion/src/utils.rs:17-45.  left: left_bits bits with k set; right: k bits.
 * Result keeps the set bits of `left` whose corresponding `right` bit is 1.
 * :24-27 — if left_bits == right_bits the result is `right`. Returns the result length in bits. */
size_t lo_and_then(const uint8_t* left, size_t left_bits, const uint8_t* right, size_t right_bits, uint8_t* out) {
    size_t nb = lo_bm_bytes(left_bits);
    if (left_bits == right_bits) {
        memcpy(out, right, nb);
        return left_bits;
    }
    memcpy(out, left, nb);
    if (left_bits & 7) out[nb - 1] &= (uint8_t)((1u << (left_bits & 7)) - 1);
    size_t j = 0;
    for (size_t i = 0; i < left_bits; i++) {
        if (lo_get_bit(left, i)) {
            if (j < right_bits && !lo_get_bit(right, j)) lo_clr_bit(out, i);
            j++;
        }
    }
    return left_bits;
}

/* ---- arrow `like` (SQL LIKE, case sensitive, `\` escape; `_` matches ONE UTF-8 character) ---- */
static inline size_t utf8_len(uint8_t c) {
    if (c < 0x80) return 1;
    if ((c >> 5) == 0x6) return 2;
    if ((c >> 4) == 0xE) return 3;
    if ((c >> 3) == 0x1E) return 4;
    return 1;
}

static int like_rec(const uint8_t* s, size_t sl, const uint8_t* p, size_t pl) {
    size_t si = 0, pi = 0;
    /* iterative with single backtrack point for '%' (classic wildcard matching) */
    size_t star_p = (size_t)-1, star_s = 0;
    while (si < sl) {
        if (pi < pl && p[pi] == '%') {
            star_p = ++pi;
            star_s = si;
            continue;
        }
        int matched = 0;
        if (pi < pl) {
            if (p[pi] == '_') {
                size_t cl = utf8_len(s[si]);
                if (si + cl > sl) cl = sl - si;
                si += cl;
                pi++;
                matched = 1;
            } else {
                size_t lp = pi;
                if (p[pi] == '\\' && pi + 1 < pl) lp = pi + 1; /* escaped literal */
                if (p[lp] == s[si]) {
                    si++;
                    pi = lp + 1;
                    matched = 1;
                }
            }
        }
        if (!matched) {
            if (star_p == (size_t)-1) return 0;
            /* advance the '%' match by one character */
            size_t cl = utf8_len(s[star_s]);
            if (star_s + cl > sl) cl = sl - star_s;
            star_s += cl;
            si = star_s;
            pi = star_p;
        }
    }
    while (pi < pl && p[pi] == '%') pi++;
    return pi == pl;
}

int lo_like_match(const uint8_t* s, size_t sl, const uint8_t* pattern, size_t pl) { return like_rec(s, sl, pattern, pl); }

/* byte-wise substring search (memmem); empty needle matches */
int lo_contains(const uint8_t* s, size_t sl, const uint8_t* needle, size_t nl) {
    if (nl == 0) return 1;
    if (nl > sl) return 0;
    const uint8_t first = needle[0];
    const uint8_t* end = s + (sl - nl) + 1;
    for (const uint8_t* p = s; p < end; p++) {
        p = (const uint8_t*)memchr(p, first, (size_t)(end - p));
        if (!p) return 0;
        if (memcmp(p, needle, nl) == 0) return 1;
    }
    return 0;
}

/* lexicographic byte compare (Rust `<[u8]>::cmp`) */
int lo_bytes_cmp(const uint8_t* a, size_t al, const uint8_t* b, size_t bl) {
    size_t m = al < bl ? al : bl;
    int c = m ? memcmp(a, b, m) : 0;
    if (c != 0) return c < 0 ? -1 : 1;
    return CMP3(al, bl);
}
