/*
 * FastLanes 1024-value bit-packing + BitPackedArray section (TEST ORACLE — see lo_common.h).
 *
 * Follows: src/core/src/liquid_array/raw/bit_pack_array.rs:71-124 (from_primitive),
 *          :127-169 (to_primitive), :181-333 (to_bytes / from_bytes),
 *          src/core/src/utils/mod.rs:24-32 (get_bit_width).
 * The in-block layout lives in the third-party crate `fastlanes` 0.5.0 (not vendored): restated from
 * the published FastLanes "unified transposed layout".  For a lane type of T bits:
 *   LANES = 1024 / T; logical element for (row in [0,T), lane in [0,LANES)) is
 *   index(row, lane) = FL_ORDER[row / 8] * 16 + (row % 8) * 128 + lane, FL_ORDER = [0,4,2,6,1,5,3,7];
 *   packed word w of `lane` is at packed[LANES * w + lane]; row r occupies bits
 *   [(r*W) mod T, ...) of word (r*W)/T, spilling into the next word of the same lane.
 * Byte-level parity with the crate: UNPINNED (reference tests here are round-trip only).
 */
#include "lo_common.h"

static const int FL_ORDER[8] = {0, 4, 2, 6, 1, 5, 3, 7};

size_t lo_fl_index(int tbits, size_t row, size_t lane) {
    (void)tbits;
    size_t o = row / 8, s = row % 8;
    return (size_t)FL_ORDER[o] * 16 + s * 128 + lane;
}

int lo_get_bit_width(uint64_t max_value) {
    /* utils/mod.rs:24-32: 0 -> 1, else 64 - leading_zeros */
    if (max_value == 0) return 1;
    return 64 - __builtin_clzll(max_value);
}

#define DEFINE_FL(T, TB)                                                                              \
    static void fl_pack_##TB(int W, const T* in, T* out) {                                            \
        const int LANES = 1024 / TB;                                                                  \
        if (W == 0) return;                                                                           \
        if (W == TB) {                                                                                \
            for (int lane = 0; lane < LANES; lane++)                                                  \
                for (int row = 0; row < TB; row++) out[LANES * row + lane] = in[lo_fl_index(TB, row, lane)]; \
            return;                                                                                   \
        }                                                                                             \
        const T mask = (T)((((T)1) << W) - 1);                                                        \
        for (int lane = 0; lane < LANES; lane++) {                                                    \
            T tmp = 0;                                                                                \
            for (int row = 0; row < TB; row++) {                                                      \
                T src = (T)(in[lo_fl_index(TB, row, lane)] & mask);                                   \
                if (row == 0) tmp = src; else tmp |= (T)(src << ((row * W) % TB));                    \
                int curr_word = (row * W) / TB, next_word = ((row + 1) * W) / TB;                     \
                if (next_word > curr_word) {                                                          \
                    out[LANES * curr_word + lane] = tmp;                                              \
                    int remaining = ((row + 1) * W) % TB;                                             \
                    tmp = (remaining == 0) ? 0 : (T)(src >> (W - remaining));                         \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    }                                                                                                 \
    static void fl_unpack_##TB(int W, const T* packed, T* out) {                                      \
        const int LANES = 1024 / TB;                                                                  \
        if (W == 0) { memset(out, 0, 1024 * sizeof(T)); return; }                                     \
        const T mask = (W == TB) ? (T)~(T)0 : (T)((((T)1) << W) - 1);                                 \
        for (int lane = 0; lane < LANES; lane++) {                                                    \
            for (int row = 0; row < TB; row++) {                                                      \
                int start = row * W, wi = start / TB, sh = start % TB;                                \
                T v = (T)(packed[LANES * wi + lane] >> sh);                                           \
                if (sh + W > TB) v |= (T)(packed[LANES * (wi + 1) + lane] << (TB - sh));              \
                out[lo_fl_index(TB, row, lane)] = (T)(v & mask);                                      \
            }                                                                                         \
        }                                                                                             \
    }

DEFINE_FL(uint8_t, 8)
DEFINE_FL(uint16_t, 16)
DEFINE_FL(uint32_t, 32)
DEFINE_FL(uint64_t, 64)

void lo_fl_pack(int tbits, int W, const void* in, void* out) {
    switch (tbits) {
        case 8: fl_pack_8(W, (const uint8_t*)in, (uint8_t*)out); break;
        case 16: fl_pack_16(W, (const uint16_t*)in, (uint16_t*)out); break;
        case 32: fl_pack_32(W, (const uint32_t*)in, (uint32_t*)out); break;
        default: fl_pack_64(W, (const uint64_t*)in, (uint64_t*)out); break;
    }
}

void lo_fl_unpack(int tbits, int W, const void* packed, void* out) {
    switch (tbits) {
        case 8: fl_unpack_8(W, (const uint8_t*)packed, (uint8_t*)out); break;
        case 16: fl_unpack_16(W, (const uint16_t*)packed, (uint16_t*)out); break;
        case 32: fl_unpack_32(W, (const uint32_t*)packed, (uint32_t*)out); break;
        default: fl_unpack_64(W, (const uint64_t*)packed, (uint64_t*)out); break;
    }
}

/* bit_pack_array.rs:76-78: num_chunks = ceil(n/1024); packed_len = ceil(1024*W / T) words = 128*W bytes */
size_t lo_bitpack_size(int tbits, int W, size_t n) {
    (void)tbits;
    return ((n + 1023) / 1024) * (size_t)128 * (size_t)W;
}

/* bit_pack_array.rs:71-124: full chunks, then a zero-padded last chunk */
size_t lo_bitpack(int tbits, int W, const void* values, size_t n, uint8_t* out) {
    size_t tb = (size_t)tbits / 8, chunk_bytes = (size_t)128 * (size_t)W;
    size_t full = n / 1024, chunks = (n + 1023) / 1024;
    const uint8_t* in = (const uint8_t*)values;
    for (size_t c = 0; c < full; c++) lo_fl_pack(tbits, W, in + c * 1024 * tb, out + c * chunk_bytes);
    if (chunks != full) {
        uint64_t last[1024];
        memset(last, 0, sizeof(last));
        memcpy(last, in + full * 1024 * tb, (n % 1024) * tb);
        lo_fl_pack(tbits, W, last, out + full * chunk_bytes);
    }
    return chunks * chunk_bytes;
}

/* bit_pack_array.rs:127-169 */
void lo_bitunpack(int tbits, int W, const uint8_t* packed, size_t n, void* out) {
    size_t tb = (size_t)tbits / 8, chunk_bytes = (size_t)128 * (size_t)W;
    size_t chunks = (n + 1023) / 1024;
    uint8_t* o = (uint8_t*)out;
    uint64_t tmp[1024];
    for (size_t c = 0; c < chunks; c++) {
        size_t take = (c + 1) * 1024 <= n ? 1024 : n - c * 1024;
        if (take == 1024) {
            lo_fl_unpack(tbits, W, packed + c * chunk_bytes, o + c * 1024 * tb);
        } else {
            lo_fl_unpack(tbits, W, packed + c * chunk_bytes, tmp);
            memcpy(o + c * 1024 * tb, tmp, take * tb);
        }
    }
}

/* bit_pack_array.rs:259-333 (from_bytes): 16-byte header, nulls, pad to 8, values */
int lo_bitpacked_parse(const uint8_t* sec, size_t sec_len, lo_bitpacked_view* v) {
    if (sec_len < 16) return LO_ERR_CORRUPT;
    memset(v, 0, sizeof(*v));
    v->len = lo_rd_u32(sec);
    v->bit_width = sec[4];
    v->has_nulls = sec[5] != 0;
    v->nulls_len = lo_rd_u32(sec + 6);
    v->values_len = lo_rd_u32(sec + 10);
    size_t values_off = lo_align8(16 + (v->has_nulls ? v->nulls_len : 0));
    if (v->values_len == 0) { /* :282-285 */
        v->all_null = 1;
        v->section_size = values_off;
        return LO_OK;
    }
    if (v->has_nulls) {
        if (v->nulls_len == 0 || 16 + (size_t)v->nulls_len > sec_len) return LO_ERR_CORRUPT;
        v->nulls = sec + 16;
    }
    if (values_off + v->values_len > sec_len) return LO_ERR_CORRUPT;
    v->values = sec + values_off;
    v->section_size = values_off + v->values_len;
    if (v->has_nulls && lo_popcount_bits(v->nulls, v->len) == 0) v->all_null = 1; /* :323-325 */
    if (!v->all_null && v->bit_width == 0) return LO_ERR_CORRUPT;               /* NonZero::new(0).unwrap() */
    return LO_OK;
}

/* bit_pack_array.rs:212-256 (to_bytes).  W == 0 => new_null_array (:43-50): values = n zero words. */
size_t lo_bitpacked_write(int tbits, int W, const void* values, const uint8_t* validity, size_t n, uint8_t* out) {
    size_t tb = (size_t)tbits / 8;
    int has_nulls = validity != NULL;
    size_t nulls_len = has_nulls ? lo_bm_bytes(n) : 0;
    size_t values_len = (W == 0) ? n * tb : lo_bitpack_size(tbits, W, n);
    size_t values_off = lo_align8(16 + nulls_len);
    memset(out, 0, values_off);
    lo_wr_u32(out, (uint32_t)n);
    out[4] = (uint8_t)W;
    out[5] = (uint8_t)has_nulls;
    lo_wr_u32(out + 6, (uint32_t)nulls_len);
    lo_wr_u32(out + 10, (uint32_t)values_len);
    if (has_nulls) {
        memcpy(out + 16, validity, nulls_len);
        if (n & 7) out[16 + nulls_len - 1] &= (uint8_t)((1u << (n & 7)) - 1);
    }
    if (W == 0) memset(out + values_off, 0, values_len);
    else lo_bitpack(tbits, W, values, n, out + values_off);
    return values_off + values_len;
}
