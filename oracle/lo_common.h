/*
 * liquid_oracle — CPU restatement of LiquidCache's decode + predicate-pushdown hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or executed by the
 * product path (liquid_cache_amd/).  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py — always as the checker / reported baseline, never as the thing
 * that is shipped.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * The reference is Rust and cannot be built in this environment (no cargo/rustc), so this is a
 * plain-C restatement pinned against the reference's own known-answer tests (tests/golden/).
 *
 * Third-party arithmetic that is NOT vendored under /root/reference:
 *   - fastlanes 0.5.0 (Cargo.lock) — 1024-value "unified transposed" bit-packing layout.
 *     Restated from the published FastLanes layout (FL_ORDER = [0,4,2,6,1,5,3,7]); the reference's
 *     tests at this boundary are round-trip only, so BYTE-LEVEL parity with the crate is UNPINNED.
 *   - fsst-rs 0.5.10 (Cargo.lock) — decode is fully determined by the symbol table (format in-tree,
 *     src/core/src/liquid_array/raw/fsst_buffer.rs:848-920) and is pinned by round trips; symbol
 *     TRAINING and greedy matching are the published FSST algorithm restated, parity UNPINNED
 *     (irrelevant to predicate results: equality on compressed bytes == equality on plain bytes).
 */
#ifndef LO_COMMON_H
#define LO_COMMON_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LO_EXPORT __attribute__((visibility("default")))

/* ---- status codes ---- */
#define LO_OK 0
#define LO_ERR_UNSUPPORTED (-1)
#define LO_ERR_CORRUPT (-2)
#define LO_ERR_CAPACITY (-3)
#define LO_ERR_ARG (-4)

/* ---- comparison operators (DataFusion Operator subset, liquid_array/mod.rs:182-204;
 *      byte_view_array/operator.rs:54-105) ---- */
enum lo_op {
    LO_EQ = 0,
    LO_NE = 1,
    LO_LT = 2,
    LO_LE = 3,
    LO_GT = 4,
    LO_GE = 5,
    LO_LIKE = 6,     /* LikeMatch / LikeExpr(negated=false) */
    LO_NOT_LIKE = 7, /* NotLikeMatch / LikeExpr(negated=true) */
};

/* ---- literal tags ---- */
enum lo_lit_tag {
    LO_LIT_I64 = 0,   /* 8 bytes, signed */
    LO_LIT_U64 = 1,   /* 8 bytes, unsigned */
    LO_LIT_F32 = 2,   /* 4 bytes */
    LO_LIT_F64 = 3,   /* 8 bytes */
    LO_LIT_BYTES = 4, /* needle / LIKE pattern */
    LO_LIT_I128 = 5,  /* 16 bytes little endian (decimal unscaled value) */
    LO_LIT_BOOL = 6,  /* 1 byte */
};

/* ---- Liquid IPC ids (liquid_array/mod.rs:50-65, ipc.rs:26-45) ---- */
enum lo_logical {
    LO_LOGICAL_INTEGER = 1,
    LO_LOGICAL_FLOAT = 2,
    LO_LOGICAL_FIXED_LEN = 3,
    LO_LOGICAL_BYTE_VIEW = 4,
    LO_LOGICAL_LINEAR_INT = 5,
    LO_LOGICAL_DECIMAL = 6,
};

enum lo_physical {
    LO_PHYS_I8 = 0,
    LO_PHYS_I16 = 1,
    LO_PHYS_I32 = 2,
    LO_PHYS_I64 = 3,
    LO_PHYS_U8 = 4,
    LO_PHYS_U16 = 5,
    LO_PHYS_U32 = 6,
    LO_PHYS_U64 = 7,
    LO_PHYS_F32 = 8,
    LO_PHYS_F64 = 9,
    LO_PHYS_DATE32 = 10,
    LO_PHYS_DATE64 = 11,
    LO_PHYS_TS_S = 12,
    LO_PHYS_TS_MS = 13,
    LO_PHYS_TS_US = 14,
    LO_PHYS_TS_NS = 15,
};

/* byte_view_array/mod.rs:113-122 */
enum lo_arrow_byte_type {
    LO_BT_UTF8 = 0,
    LO_BT_UTF8VIEW = 1,
    LO_BT_DICT16_BINARY = 2,
    LO_BT_DICT16_UTF8 = 3,
    LO_BT_BINARY = 4,
    LO_BT_BINARYVIEW = 5,
};

/* ---- LSB-first Arrow bitmap helpers ---- */
static inline int lo_get_bit(const uint8_t* bm, size_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
static inline void lo_set_bit(uint8_t* bm, size_t i) { bm[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline void lo_clr_bit(uint8_t* bm, size_t i) { bm[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
static inline size_t lo_bm_bytes(size_t nbits) { return (nbits + 7) >> 3; }

static inline size_t lo_popcount_bits(const uint8_t* bm, size_t nbits) {
    size_t c = 0, full = nbits >> 3;
    for (size_t i = 0; i < full; i++) c += (size_t)__builtin_popcount(bm[i]);
    if (nbits & 7) c += (size_t)__builtin_popcount(bm[full] & ((1u << (nbits & 7)) - 1));
    return c;
}

static inline uint32_t lo_rd_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t lo_rd_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint16_t lo_rd_u16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void lo_wr_u32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
static inline void lo_wr_u64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
static inline void lo_wr_u16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }
static inline size_t lo_align8(size_t x) { return (x + 7) & ~(size_t)7; }

/* physical type -> native width in bytes / signedness / unsigned lane bits
 * (primitive_array.rs:84-99: signed 64-bit types use u64 lanes etc.) */
static inline int lo_phys_width(int phys) {
    switch (phys) {
        case LO_PHYS_I8: case LO_PHYS_U8: return 1;
        case LO_PHYS_I16: case LO_PHYS_U16: return 2;
        case LO_PHYS_I32: case LO_PHYS_U32: case LO_PHYS_DATE32: case LO_PHYS_F32: return 4;
        default: return 8;
    }
}
static inline int lo_phys_is_unsigned(int phys) {
    return phys == LO_PHYS_U8 || phys == LO_PHYS_U16 || phys == LO_PHYS_U32 || phys == LO_PHYS_U64;
}
static inline int lo_phys_is_float(int phys) { return phys == LO_PHYS_F32 || phys == LO_PHYS_F64; }

/* ---------------- fastlanes (lo_fastlanes.c) ---------------- */
LO_EXPORT size_t lo_fl_index(int tbits, size_t row, size_t lane);
LO_EXPORT void lo_fl_pack(int tbits, int W, const void* in1024, void* out);
LO_EXPORT void lo_fl_unpack(int tbits, int W, const void* packed, void* out1024);
LO_EXPORT size_t lo_bitpack_size(int tbits, int W, size_t n);
LO_EXPORT size_t lo_bitpack(int tbits, int W, const void* values, size_t n, uint8_t* out);
LO_EXPORT void lo_bitunpack(int tbits, int W, const uint8_t* packed, size_t n, void* out);
LO_EXPORT int lo_get_bit_width(uint64_t max_value);

/* BitPackedArray serialized section (bit_pack_array.rs:181-333) */
typedef struct {
    uint32_t len;
    int bit_width;        /* 0 => all null */
    int has_nulls;
    const uint8_t* nulls; /* may be NULL */
    uint32_t nulls_len;
    const uint8_t* values;
    uint32_t values_len;
    size_t section_size;  /* bytes consumed from the section start */
    int all_null;         /* reconstructed as new_null_array */
} lo_bitpacked_view;
LO_EXPORT int lo_bitpacked_parse(const uint8_t* sec, size_t sec_len, lo_bitpacked_view* v);
LO_EXPORT size_t lo_bitpacked_write(int tbits, int W /*0 => all null*/, const void* values, const uint8_t* validity,
                                    size_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
