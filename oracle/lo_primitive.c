/*
 * LiquidPrimitiveArray / LiquidDecimalArray / LiquidFloatArray (ALP) / date parts — TEST ORACLE (see lo_common.h).
 *
 * Follows:
 *   ints    src/core/src/liquid_array/primitive_array.rs:159-206 (encode), :350-368 (decode),
 *           :370-379 (filter / try_eval_predicate), :599-679 (bytes)
 *   decimal src/core/src/liquid_array/decimal_array.rs:127-177, :185-195, :197-257, :272-289
 *   float   src/core/src/liquid_array/float_array.rs:109-125, :294-316, :397-601, :609-740
 *   dates   src/core/src/liquid_array/squeezed_date32_array.rs:63-221, :289-429
 *   generic predicate  src/core/src/liquid_array/mod.rs:117-130, :265-280
 */
#include "lo_primitive.h"
#include "lo_arrowish.h"
#include <math.h>

static const uint8_t LO_MAGIC[4] = {0x41, 0x44, 0x51, 0x4C}; /* 0x4C51_4441 LE, ipc.rs:23 */

/* ipc.rs:180-199 */
void lo_ipc_header_write(uint8_t* out, int logical, int phys) {
    memset(out, 0, 16);
    memcpy(out, LO_MAGIC, 4);
    lo_wr_u16(out + 4, 1);
    lo_wr_u16(out + 6, (uint16_t)logical);
    lo_wr_u16(out + 8, (uint16_t)phys);
}

/* ipc.rs:201-236 */
int lo_ipc_header_read(const uint8_t* bytes, size_t len, int* logical, int* phys) {
    if (len < 16) return LO_ERR_CORRUPT;
    if (memcmp(bytes, LO_MAGIC, 4) != 0) return LO_ERR_CORRUPT;
    if (lo_rd_u16(bytes + 4) != 1) return LO_ERR_CORRUPT;
    *logical = lo_rd_u16(bytes + 6);
    *phys = lo_rd_u16(bytes + 8);
    return LO_OK;
}

static inline uint64_t load_native_as_u64(const uint8_t* p, int w) {
    switch (w) {
        case 1: return *p;
        case 2: return lo_rd_u16(p);
        case 4: return lo_rd_u32(p);
        default: return lo_rd_u64(p);
    }
}
static inline int64_t sext(uint64_t v, int w) {
    switch (w) {
        case 1: return (int8_t)v;
        case 2: return (int16_t)v;
        case 4: return (int32_t)v;
        default: return (int64_t)v;
    }
}
static inline void store_native(uint8_t* p, int w, uint64_t v) {
    switch (w) {
        case 1: *p = (uint8_t)v; break;
        case 2: lo_wr_u16(p, (uint16_t)v); break;
        case 4: lo_wr_u32(p, (uint32_t)v); break;
        default: lo_wr_u64(p, v); break;
    }
}

size_t lo_prim_encode_bound(int phys, size_t n) {
    size_t w = (size_t)lo_phys_width(phys);
    return 24 + 16 + lo_bm_bytes(n) + 8 + ((n + 1023) / 1024) * 128 * (w * 8) + n * w + 64;
}

/* primitive_array.rs:159-206 + :626-654.  Native values of width `w`; FoR with reference = min (nulls skipped),
 * stored u = (v - min) wrapping, reinterpreted as the unsigned lane type. */
int64_t lo_prim_encode(int phys, const void* values, const uint8_t* validity, size_t n, uint8_t* out, size_t cap) {
    if (lo_phys_is_float(phys)) return LO_ERR_ARG;
    if (cap < lo_prim_encode_bound(phys, n)) return LO_ERR_CAPACITY;
    const int w = lo_phys_width(phys);
    const int uns = lo_phys_is_unsigned(phys);
    const uint8_t* in = (const uint8_t*)values;
    /* arrow aggregate::min / max skip nulls */
    int have = 0;
    uint64_t minb = 0, maxb = 0;
    for (size_t i = 0; i < n; i++) {
        if (validity && !lo_get_bit(validity, i)) continue;
        uint64_t b = load_native_as_u64(in + i * (size_t)w, w);
        if (!have) { minb = maxb = b; have = 1; continue; }
        if (uns) { if (b < minb) minb = b; if (b > maxb) maxb = b; }
        else { if (sext(b, w) < sext(minb, w)) minb = b; if (sext(b, w) > sext(maxb, w)) maxb = b; }
    }
    lo_ipc_header_write(out, LO_LOGICAL_INTEGER, phys);
    memset(out + 16, 0, 8);
    if (!have) { /* :160-170 all null: reference_value = 0, BitPackedArray::new_null_array */
        uint8_t* allnull = (uint8_t*)calloc(lo_bm_bytes(n) + 1, 1);
        size_t s = lo_bitpacked_write(w * 8, 0, NULL, allnull, n, out + 24);
        free(allnull);
        return (int64_t)(24 + s);
    }
    const uint64_t wmask = (w == 8) ? ~(uint64_t)0 : ((((uint64_t)1) << (w * 8)) - 1);
    uint64_t sub = (maxb - minb) & wmask; /* :173-180 sub_wrapping then reinterpret unsigned */
    int W = lo_get_bit_width(sub);
    uint8_t* tmp = (uint8_t*)malloc((n ? n : 1) * (size_t)w);
    for (size_t i = 0; i < n; i++) {
        uint64_t b = load_native_as_u64(in + i * (size_t)w, w);
        store_native(tmp + i * (size_t)w, w, (b - minb) & wmask); /* :183-194 (min == 0 => transmute, same result) */
    }
    store_native(out + 16, w, minb);
    size_t s = lo_bitpacked_write(w * 8, W, tmp, validity, n, out + 24);
    free(tmp);
    return (int64_t)(24 + s);
}

/* ---- generic info ---- */
int lo_array_info_get(const uint8_t* bytes, size_t len, lo_array_info* info) {
    memset(info, 0, sizeof(*info));
    int logical, phys;
    int rc = lo_ipc_header_read(bytes, len, &logical, &phys);
    if (rc) return rc;
    info->logical = logical;
    info->phys = phys;
    size_t bp_off;
    if (logical == LO_LOGICAL_INTEGER) {
        info->value_width = lo_phys_width(phys);
        info->lane_bits = info->value_width * 8;
        bp_off = 24; /* (16 + size_of<Native> + 7) & !7 == 24 for all */
        info->reference = load_native_as_u64(bytes + 16, info->value_width);
    } else if (logical == LO_LOGICAL_DECIMAL) {
        /* decimal_array.rs:68-117, 180-257 */
        if (len < 32) return LO_ERR_CORRUPT;
        info->decimal_is256 = bytes[16];
        info->decimal_precision = bytes[17];
        info->decimal_scale = (int8_t)bytes[18];
        info->value_width = 16;
        info->lane_bits = 64;
        info->reference = lo_rd_u64(bytes + 24);
        bp_off = 32;
    } else if (logical == LO_LOGICAL_FLOAT) {
        int w = lo_phys_width(phys);
        info->value_width = w;
        info->lane_bits = w * 8;
        info->reference = load_native_as_u64(bytes + 16, w);
        size_t next = lo_align8(16 + (size_t)w);
        if (len < next + 16) return LO_ERR_CORRUPT;
        info->alp_e = bytes[next];
        info->alp_f = bytes[next + 1];
        next += 8;
        info->patch_len = lo_rd_u64(bytes + next);
        next += 8;
        info->patch_indices_off = next;
        next += info->patch_len * 8;
        info->patch_values_off = next;
        next += info->patch_len * (size_t)w;
        bp_off = lo_align8(next);
    } else {
        return LO_ERR_UNSUPPORTED;
    }
    if (bp_off > len) return LO_ERR_CORRUPT;
    info->bitpacked_off = bp_off;
    lo_bitpacked_view v;
    rc = lo_bitpacked_parse(bytes + bp_off, len - bp_off, &v);
    if (rc) return rc;
    info->len = v.len;
    info->nullable = v.has_nulls || v.all_null;
    info->all_null = v.all_null;
    info->bit_width = v.all_null ? 0 : v.bit_width;
    return LO_OK;
}

/* ---- ALP constants (float_array.rs:127-224) ---- */
static const float F10_32[11] = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f,
                                 100000000.0f, 1000000000.0f, 10000000000.0f};
static const float IF10_32[11] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f,
                                  0.00000001f, 0.000000001f, 0.0000000001f};
static const double F10_64[24] = {1.0, 10.0, 100.0, 1000.0, 10000.0, 100000.0, 1000000.0, 10000000.0, 100000000.0,
                                  1000000000.0, 10000000000.0, 100000000000.0, 1000000000000.0, 10000000000000.0,
                                  100000000000000.0, 1000000000000000.0, 10000000000000000.0, 100000000000000000.0,
                                  1000000000000000000.0, 10000000000000000000.0, 100000000000000000000.0,
                                  1000000000000000000000.0, 10000000000000000000000.0, 100000000000000000000000.0};
static const double IF10_64[24] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001,
                                   0.000000001, 0.0000000001, 0.00000000001, 0.000000000001, 0.0000000000001,
                                   0.00000000000001, 0.000000000000001, 0.0000000000000001, 0.00000000000000001,
                                   0.000000000000000001, 0.0000000000000000001, 0.00000000000000000001,
                                   0.000000000000000000001, 0.0000000000000000000001, 0.00000000000000000000001};

/* Rust `as` float->int: saturating, NaN -> 0 */
static inline int32_t sat_f32_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static inline int64_t sat_f64_i64(double v) {
    if (v != v) return 0;
    if (v >= 9223372036854775808.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

/* float_array.rs:109-123 — keep evaluation order; compile with -ffp-contract=off */
static inline int32_t alp_enc32(float v, int e, int f) {
    const float SWEET = 8388608.0f + 4194304.0f; /* 2^23 + 2^22 */
    volatile float t = v * F10_32[e];
    t = t * IF10_32[f];
    volatile float r = t + SWEET;
    r = r - SWEET;
    return sat_f32_i32(r);
}
static inline float alp_dec32(int32_t i, int e, int f) {
    volatile float t = (float)i;
    t = t * F10_32[f];
    t = t * IF10_32[e];
    return t;
}
static inline int64_t alp_enc64(double v, int e, int f) {
    const double SWEET = 4503599627370496.0 + 2251799813685248.0; /* 2^52 + 2^51 */
    volatile double t = v * F10_64[e];
    t = t * IF10_64[f];
    volatile double r = t + SWEET;
    r = r - SWEET;
    return sat_f64_i64(r);
}
static inline double alp_dec64(int64_t i, int e, int f) {
    volatile double t = (double)i;
    t = t * F10_64[f];
    t = t * IF10_64[e];
    return t;
}

typedef struct {
    int64_t* enc;      /* encoded ints (widened) */
    uint64_t* pidx;
    uint8_t* pvals;    /* native */
    size_t patch_count;
    int64_t min, max;
} alp_tmp;

/* float_array.rs:609-713 encode_arrow_array (values for ALL slots incl. null slots are encoded) */
static void alp_encode_values(int w, const uint8_t* in, size_t n, int e, int f, alp_tmp* t) {
    t->patch_count = 0;
    for (size_t i = 0; i < n; i++) {
        int neq;
        if (w == 4) {
            float v; memcpy(&v, in + 4 * i, 4);
            int32_t en = alp_enc32(v, e, f);
            float d = alp_dec32(en, e, f);
            neq = !(d == v);
            t->enc[i] = en;
        } else {
            double v; memcpy(&v, in + 8 * i, 8);
            int64_t en = alp_enc64(v, e, f);
            double d = alp_dec64(en, e, f);
            neq = !(d == v);
            t->enc[i] = en;
        }
        if (neq) {
            t->pidx[t->patch_count] = i;
            memcpy(t->pvals + t->patch_count * (size_t)w, in + i * (size_t)w, (size_t)w);
            t->patch_count++;
        }
    }
    /* :652-668 fill value = first successfully encoded value */
    if (t->patch_count > 0 && t->patch_count < n) {
        int have_fill = 0;
        int64_t fill = 0;
        for (size_t i = 0; i < n; i++) {
            if (i >= t->patch_count || t->pidx[i] != (uint64_t)i) { fill = t->enc[i]; have_fill = 1; break; }
        }
        if (have_fill) for (size_t k = 0; k < t->patch_count; k++) t->enc[t->pidx[k]] = fill;
    }
    t->min = INT64_MAX; t->max = INT64_MIN;
    for (size_t i = 0; i < n; i++) { if (t->enc[i] < t->min) t->min = t->enc[i]; if (t->enc[i] > t->max) t->max = t->enc[i]; }
}

size_t lo_float_encode_bound(int phys, size_t n) {
    size_t w = (size_t)lo_phys_width(phys);
    return 64 + n * (8 + w) + 16 + lo_bm_bytes(n) + 8 + ((n + 1023) / 1024) * 128 * (w * 8) + n * w + 64;
}

/* float_array.rs:715-740 get_best_exponents + :609-713 + :430-519 to_bytes.
 * Exponent choice minimises an estimate of the encoded size (packed bytes + patches); the reference ranks by
 * get_array_memory_size() of a trial encode — choice parity is not required for decode/predicate parity. */
int64_t lo_float_encode(int phys, const void* values, const uint8_t* validity, size_t n, uint8_t* out, size_t cap) {
    if (!lo_phys_is_float(phys)) return LO_ERR_ARG;
    if (cap < lo_float_encode_bound(phys, n)) return LO_ERR_CAPACITY;
    const int w = lo_phys_width(phys);
    const uint8_t* in = (const uint8_t*)values;
    lo_ipc_header_write(out, LO_LOGICAL_FLOAT, phys);
    size_t null_count = 0;
    if (validity) null_count = n - lo_popcount_bits(validity, n);
    size_t pos = 16;
    if (n == 0 || (validity && null_count == n)) { /* :620-631 all null (null_count == len, incl. len 0) */
        static const uint8_t empty_validity[1] = {0};
        memset(out + pos, 0, (size_t)w); pos += (size_t)w;
        while (pos & 7) out[pos++] = 0;
        memset(out + pos, 0, 16); pos += 16; /* e,f,pad + patch_len 0 */
        pos += lo_bitpacked_write(w * 8, 0, NULL, validity ? validity : empty_validity, n, out + pos);
        return (int64_t)pos;
    }
    alp_tmp t;
    t.enc = (int64_t*)malloc((n ? n : 1) * 8);
    t.pidx = (uint64_t*)malloc((n ? n : 1) * 8);
    t.pvals = (uint8_t*)malloc((n ? n : 1) * (size_t)w);
    /* sample like :719-727 (every len/1024-th value, nulls dropped) */
    const int max_exp = (w == 4) ? 10 : 18;
    uint8_t* sample = NULL;
    size_t sn = n;
    const uint8_t* sin = in;
    if (n > 1024) {
        size_t step = n / 1024;
        sample = (uint8_t*)malloc(((n / step) + 2) * (size_t)w);
        sn = 0;
        for (size_t i = 0; i < n; i += step) {
            if (validity && !lo_get_bit(validity, i)) continue;
            memcpy(sample + sn * (size_t)w, in + i * (size_t)w, (size_t)w);
            sn++;
        }
        sin = sample;
    }
    int be = 0, bf = 0;
    size_t best = (size_t)-1;
    for (int e = 0; e < max_exp; e++) {
        for (int f = 0; f < e; f++) {
            if (sn == 0) continue;
            alp_encode_values(w, sin, sn, e, f, &t);
            uint64_t sub = (uint64_t)t.max - (uint64_t)t.min;
            if (w == 4) sub &= 0xFFFFFFFFu;
            size_t est = ((sn + 1023) / 1024) * 128 * (size_t)lo_get_bit_width(sub) + t.patch_count * (8 + (size_t)w);
            if (est < best) { best = est; be = e; bf = f; }
        }
    }
    free(sample);
    alp_encode_values(w, in, n, be, bf, &t);
    if (n == 0) { t.min = 0; t.max = 0; }
    /* reference value, exponents, patches (:430-519) */
    if (w == 4) { int32_t r = (int32_t)t.min; memcpy(out + pos, &r, 4); pos += 4; }
    else { memcpy(out + pos, &t.min, 8); pos += 8; }
    while (pos & 7) out[pos++] = 0;
    out[pos] = (uint8_t)be; out[pos + 1] = (uint8_t)bf; memset(out + pos + 2, 0, 6); pos += 8;
    lo_wr_u64(out + pos, (uint64_t)t.patch_count); pos += 8;
    memcpy(out + pos, t.pidx, t.patch_count * 8); pos += t.patch_count * 8;
    memcpy(out + pos, t.pvals, t.patch_count * (size_t)w); pos += t.patch_count * (size_t)w;
    while (pos & 7) out[pos++] = 0;
    uint64_t sub = (uint64_t)t.max - (uint64_t)t.min;
    if (w == 4) sub &= 0xFFFFFFFFu;
    int W = lo_get_bit_width(sub);
    uint8_t* u = (uint8_t*)malloc((n ? n : 1) * (size_t)w);
    for (size_t i = 0; i < n; i++) store_native(u + i * (size_t)w, w, (uint64_t)t.enc[i] - (uint64_t)t.min);
    pos += lo_bitpacked_write(w * 8, W, u, validity, n, out + pos);
    free(u); free(t.enc); free(t.pidx); free(t.pvals);
    return (int64_t)pos;
}

size_t lo_decimal_encode_bound(size_t n) { return 32 + 16 + lo_bm_bytes(n) + 8 + ((n + 1023) / 1024) * 128 * 64 + n * 8 + 64; }

/* decimal_array.rs:127-177 + :197-220.  values: i128 (16 B LE) each; all non-null values must fit u64. */
int64_t lo_decimal_encode(int is256, int precision, int scale, const void* values_i128, const uint8_t* validity,
                          size_t n, uint8_t* out, size_t cap) {
    if (cap < lo_decimal_encode_bound(n)) return LO_ERR_CAPACITY;
    const uint8_t* in = (const uint8_t*)values_i128;
    lo_ipc_header_write(out, LO_LOGICAL_DECIMAL, LO_PHYS_U64);
    memset(out + 16, 0, 16);
    out[16] = (uint8_t)(is256 ? 1 : 0);
    out[17] = (uint8_t)precision;
    out[18] = (uint8_t)(int8_t)scale;
    size_t null_count = validity ? n - lo_popcount_bits(validity, n) : 0;
    if (n == 0 || (validity && null_count == n)) { /* :134-141 null_count == len (incl. len 0) */
        static const uint8_t empty_validity[1] = {0};
        return (int64_t)(32 + lo_bitpacked_write(64, 0, NULL, validity ? validity : empty_validity, n, out + 32));
    }
    uint64_t* vals = (uint64_t*)malloc((n ? n : 1) * 8);
    uint64_t mn = UINT64_MAX, mx = 0;
    for (size_t i = 0; i < n; i++) {
        if (validity && !lo_get_bit(validity, i)) { vals[i] = 0; continue; }
        __int128 v; memcpy(&v, in + 16 * i, 16);
        if (v < 0 || v > (__int128)UINT64_MAX) { free(vals); return LO_ERR_UNSUPPORTED; } /* fits_u64 :120-125 */
        vals[i] = (uint64_t)v;
        if (vals[i] < mn) mn = vals[i];
        if (vals[i] > mx) mx = vals[i];
    }
    if (n == 0) { mn = 0; mx = 0; }
    int W = lo_get_bit_width(mx - mn);
    for (size_t i = 0; i < n; i++) vals[i] = vals[i] >= mn ? vals[i] - mn : 0; /* saturating_sub :164 */
    lo_wr_u64(out + 24, mn);
    size_t s = lo_bitpacked_write(64, W, vals, validity, n, out + 32);
    free(vals);
    return (int64_t)(32 + s);
}

/* to_arrow_array for Integer / Decimal / Float.
 *   primitive_array.rs:350-368; decimal_array.rs:185-195 + :272-289; float_array.rs:294-316
 * out_values: len * value_width bytes (decimal: i128 per row); out_validity: ceil(len/8) bytes or NULL. */
int lo_fixed_to_arrow(const uint8_t* bytes, size_t len, void* out_values, uint8_t* out_validity) {
    lo_array_info info;
    int rc = lo_array_info_get(bytes, len, &info);
    if (rc) return rc;
    lo_bitpacked_view v;
    rc = lo_bitpacked_parse(bytes + info.bitpacked_off, len - info.bitpacked_off, &v);
    if (rc) return rc;
    const size_t n = info.len;
    uint8_t* o = (uint8_t*)out_values;
    if (out_validity) {
        if (info.all_null) memset(out_validity, 0, lo_bm_bytes(n));
        else if (v.has_nulls) { memcpy(out_validity, v.nulls, lo_bm_bytes(n)); }
        else memset(out_validity, 0xFF, lo_bm_bytes(n));
        if (n & 7) out_validity[lo_bm_bytes(n) - 1] &= (uint8_t)((1u << (n & 7)) - 1);
    }
    if (info.all_null) { /* PrimitiveArray::new_null: zeroed values */
        memset(o, 0, n * (size_t)info.value_width);
        return LO_OK;
    }
    const int lw = info.lane_bits / 8;
    uint8_t* u = (uint8_t*)malloc((n ? n : 1) * (size_t)lw);
    lo_bitunpack(info.lane_bits, info.bit_width, v.values, n, u);
    if (info.logical == LO_LOGICAL_INTEGER) {
        const uint64_t wmask = (lw == 8) ? ~(uint64_t)0 : ((((uint64_t)1) << (lw * 8)) - 1);
        for (size_t i = 0; i < n; i++)
            store_native(o + i * (size_t)lw, lw, (load_native_as_u64(u + i * (size_t)lw, lw) + info.reference) & wmask);
    } else if (info.logical == LO_LOGICAL_DECIMAL) {
        for (size_t i = 0; i < n; i++) {
            uint64_t x = lo_rd_u64(u + 8 * i) + info.reference; /* wrapping_add :189 */
            __int128 w128 = (__int128)x;                        /* *v as i128 :285 */
            memcpy(o + 16 * i, &w128, 16);
        }
    } else { /* float */
        if (lw == 4) {
            int32_t ref = (int32_t)(uint32_t)info.reference;
            for (size_t i = 0; i < n; i++) {
                int32_t val = (int32_t)((uint32_t)lo_rd_u32(u + 4 * i) + (uint32_t)ref);
                float d = alp_dec32(val, info.alp_e, info.alp_f);
                memcpy(o + 4 * i, &d, 4);
            }
        } else {
            int64_t ref = (int64_t)info.reference;
            for (size_t i = 0; i < n; i++) {
                int64_t val = (int64_t)(lo_rd_u64(u + 8 * i) + (uint64_t)ref);
                double d = alp_dec64(val, info.alp_e, info.alp_f);
                memcpy(o + 8 * i, &d, 8);
            }
        }
        for (uint64_t k = 0; k < info.patch_len; k++) { /* :306-310 */
            uint64_t idx = lo_rd_u64(bytes + info.patch_indices_off + 8 * k);
            if (idx < n) memcpy(o + idx * (size_t)lw, bytes + info.patch_values_off + k * (size_t)lw, (size_t)lw);
        }
    }
    free(u);
    return LO_OK;
}

/* LiquidArray::filter default (mod.rs:117-121): decode everything, then arrow filter. Returns k or <0. */
int64_t lo_fixed_filter(const uint8_t* bytes, size_t len, const uint8_t* sel, void* out_values, uint8_t* out_validity,
                        int* nullable) {
    lo_array_info info;
    int rc = lo_array_info_get(bytes, len, &info);
    if (rc) return rc;
    const size_t n = info.len;
    uint8_t* vals = (uint8_t*)malloc((n ? n : 1) * (size_t)info.value_width);
    uint8_t* valid = (uint8_t*)malloc(lo_bm_bytes(n) + 1);
    rc = lo_fixed_to_arrow(bytes, len, vals, valid);
    if (rc) { free(vals); free(valid); return rc; }
    size_t k = lo_filter_values(info.value_width, vals, n, sel, out_values);
    if (out_validity) lo_filter_bits(valid, n, sel, out_validity);
    if (nullable) *nullable = info.nullable;
    free(vals); free(valid);
    return (int64_t)k;
}

/* try_eval_predicate default (mod.rs:127-130, primitive_array.rs:376-379): filter(sel) then arrow cmp vs literal.
 * sel == NULL means BooleanBuffer::new_set(len) (cache/core.rs:907-911).
 * Output: out_values (k bits), out_validity (k bits, may be NULL), returns k = popcount(sel). */
int64_t lo_fixed_eval_predicate(const uint8_t* bytes, size_t len, int op, int lit_tag, const void* lit,
                                const uint8_t* sel, uint8_t* out_values, uint8_t* out_validity, int* nullable) {
    if (op > LO_GE) return LO_ERR_UNSUPPORTED;
    lo_array_info info;
    int rc = lo_array_info_get(bytes, len, &info);
    if (rc) return rc;
    const size_t n = info.len;
    uint8_t* allsel = NULL;
    if (!sel) {
        allsel = (uint8_t*)malloc(lo_bm_bytes(n) + 1);
        memset(allsel, 0xFF, lo_bm_bytes(n) + 1);
        sel = allsel;
    }
    uint8_t* fv = (uint8_t*)malloc((n ? n : 1) * (size_t)info.value_width);
    uint8_t* fvalid = (uint8_t*)malloc(lo_bm_bytes(n) + 1);
    int nl = 0;
    int64_t k = lo_fixed_filter(bytes, len, sel, fv, fvalid, &nl);
    free(allsel);
    if (k < 0) { free(fv); free(fvalid); return k; }
    int kind;
    uint8_t litbuf[16];
    memset(litbuf, 0, sizeof(litbuf));
    if (info.logical == LO_LOGICAL_INTEGER) {
        kind = lo_phys_is_unsigned(info.phys) ? 1 : 0;
        if (lit_tag != LO_LIT_I64 && lit_tag != LO_LIT_U64) { free(fv); free(fvalid); return LO_ERR_ARG; }
        memcpy(litbuf, lit, 8);
    } else if (info.logical == LO_LOGICAL_DECIMAL) {
        kind = 4;
        if (lit_tag == LO_LIT_I128) memcpy(litbuf, lit, 16);
        else if (lit_tag == LO_LIT_I64) { int64_t t; memcpy(&t, lit, 8); __int128 w = t; memcpy(litbuf, &w, 16); }
        else { free(fv); free(fvalid); return LO_ERR_ARG; }
    } else {
        kind = info.value_width == 4 ? 2 : 3;
        if (kind == 2 && lit_tag == LO_LIT_F32) memcpy(litbuf, lit, 4);
        else if (kind == 3 && lit_tag == LO_LIT_F64) memcpy(litbuf, lit, 8);
        else { free(fv); free(fvalid); return LO_ERR_ARG; }
    }
    rc = lo_cmp_scalar(kind, info.value_width, fv, (size_t)k, op, litbuf, out_values);
    if (out_validity) memcpy(out_validity, fvalid, lo_bm_bytes((size_t)k));
    if (nullable) *nullable = nl;
    free(fv); free(fvalid);
    return rc ? rc : k;
}

/* ---------------- date parts (squeezed_date32_array.rs:364-429) ---------------- */
static void ymd_from_epoch_days(int32_t days, int32_t* y_out, uint32_t* m_out, uint32_t* d_out) {
    int64_t z = (int64_t)days + 719468;
    int64_t era = z >= 0 ? z / 146097 : (z - 146096) / 146097;
    int64_t doe = z - era * 146097;
    int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    int64_t y = yoe + era * 400;
    int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    int64_t mp = (5 * doy + 2) / 153;
    int64_t d = (doy - (153 * mp + 2) / 5) + 1;
    int64_t m = mp + (mp < 10 ? 3 : -9);
    if (m <= 2) y += 1;
    *y_out = (int32_t)y; *m_out = (uint32_t)m; *d_out = (uint32_t)d;
}

int32_t lo_date_component(int field, int32_t days) {
    int32_t y; uint32_t m, d;
    ymd_from_epoch_days(days, &y, &m, &d);
    switch (field) {
        case LO_DATE_YEAR: return y;
        case LO_DATE_MONTH: return (int32_t)m;
        case LO_DATE_DAY: return (int32_t)d;
        default: { /* (days + 4).rem_euclid(7) — i32 arithmetic in the reference; widen to avoid UB */
            int64_t r = ((int64_t)days + 4) % 7;
            if (r < 0) r += 7;
            return (int32_t)r;
        }
    }
}

int32_t lo_ymd_to_epoch_days(int32_t year, uint32_t month, uint32_t day) {
    int64_t y = (int64_t)year - (month <= 2 ? 1 : 0);
    int64_t era = y >= 0 ? y / 400 : (y - 399) / 400;
    int64_t yoe = y - era * 400;
    int64_t m = month, d = day;
    int64_t mp = m + (m > 2 ? -3 : 9);
    int64_t doy = (153 * mp + 2) / 5 + d - 1;
    int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return (int32_t)(era * 146097 + doe - 719468);
}

/* :406-414 value.div_euclid(ticks_per_day) as i32 ; unit: 0 s, 1 ms, 2 us, 3 ns */
int32_t lo_timestamp_to_days(int64_t value, int unit) {
    static const int64_t TPD[4] = {86400LL, 86400000LL, 86400000000LL, 86400000000000LL};
    int64_t t = TPD[unit & 3];
    int64_t q = value / t;
    if ((value % t) < 0) q -= 1;
    return (int32_t)q;
}

/* :289-359 lossy reconstruction of one component into a Date32 */
int32_t lo_date_lossy_days(int field, int32_t component) {
    switch (field) {
        case LO_DATE_YEAR: return lo_ymd_to_epoch_days(component, 1, 1);
        case LO_DATE_MONTH: return lo_ymd_to_epoch_days(1970, (uint32_t)component, 1);
        case LO_DATE_DAY: return lo_ymd_to_epoch_days(1970, 1, (uint32_t)component);
        default: {
            int64_t r = (int64_t)lo_ymd_to_epoch_days(1970, 1, 4) + component; /* saturating_add */
            if (r > INT32_MAX) r = INT32_MAX;
            if (r < INT32_MIN) r = INT32_MIN;
            return (int32_t)r;
        }
    }
}
