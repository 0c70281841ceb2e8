// Host-side data formats of the LiquidArray encodings: Liquid IPC parsing / writing, FastLanes layout.
//
// Byte formats follow the reference tree (paths relative to the reference repository root):
//   LiquidIPCHeader            src/core/src/liquid_array/ipc.rs:158-236
//   BitPackedArray section     src/core/src/liquid_array/raw/bit_pack_array.rs:181-333
//   LiquidPrimitiveArray       src/core/src/liquid_array/primitive_array.rs:603-679
//   LiquidDecimalArray         src/core/src/liquid_array/decimal_array.rs:68-117, 197-257
//   LiquidFloatArray (ALP)     src/core/src/liquid_array/float_array.rs:397-601
//   LiquidByteViewArray        src/core/src/liquid_array/byte_view_array/serialization.rs:14-326
//   CompactOffsets             src/core/src/liquid_array/raw/fsst_buffer.rs:261-383, 762-846
// FastLanes in-block layout (crate fastlanes 0.5.0, not vendored): element (row, lane) of a 1024-value block
// sits at logical index FL_ORDER[row/8]*16 + (row%8)*128 + lane; word w of a lane is packed[LANES*w + lane].
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/liquid_cache_amd.h"

namespace lc {

// ---- ids ----
enum Logical : int { kInteger = 1, kFloat = 2, kFixedLen = 3, kByteView = 4, kLinearInt = 5, kDecimal = 6 };
enum Phys : int {
    kI8 = 0, kI16, kI32, kI64, kU8, kU16, kU32, kU64, kF32, kF64, kDate32, kDate64, kTsS, kTsMs, kTsUs, kTsNs
};
enum ByteType : int { kUtf8 = 0, kUtf8View = 1, kDict16Binary = 2, kDict16Utf8 = 3, kBinary = 4, kBinaryView = 5 };

inline int phys_width(int p) {
    switch (p) {
        case kI8: case kU8: return 1;
        case kI16: case kU16: return 2;
        case kI32: case kU32: case kDate32: case kF32: return 4;
        default: return 8;
    }
}
inline bool phys_unsigned(int p) { return p == kU8 || p == kU16 || p == kU32 || p == kU64; }
inline bool phys_float(int p) { return p == kF32 || p == kF64; }

// ---- little helpers ----
template <typename T> inline T rd(const uint8_t* p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }
template <typename T> inline void wr(uint8_t* p, T v) { std::memcpy(p, &v, sizeof(T)); }
inline size_t align8(size_t x) { return (x + 7) & ~size_t(7); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline size_t bitmap_bytes(size_t bits) { return (bits + 7) >> 3; }
inline bool get_bit(const uint8_t* bm, size_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
inline void set_bit(uint8_t* bm, size_t i) { bm[i >> 3] |= uint8_t(1u << (i & 7)); }
inline size_t count_bits(const uint8_t* bm, size_t nbits) {
    size_t c = 0, full = nbits >> 3;
    size_t i = 0;
    for (; i + 8 <= full; i += 8) c += size_t(__builtin_popcountll(rd<uint64_t>(bm + i)));
    for (; i < full; i++) c += size_t(__builtin_popcount(bm[i]));
    if (nbits & 7) c += size_t(__builtin_popcount(bm[full] & ((1u << (nbits & 7)) - 1)));
    return c;
}
inline int bit_width_of(uint64_t max_value) {  // utils/mod.rs:24-32
    return max_value == 0 ? 1 : 64 - __builtin_clzll(max_value);
}

// ---- FastLanes ----
constexpr int kFlOrder[8] = {0, 4, 2, 6, 1, 5, 3, 7};
inline size_t fl_index(size_t row, size_t lane) { return size_t(kFlOrder[row >> 3]) * 16 + (row & 7) * 128 + lane; }

template <typename T>
void fl_pack_block(int W, const T* in, T* out) {
    constexpr int TB = int(sizeof(T)) * 8;
    constexpr int LANES = 1024 / TB;
    if (W == TB) {
        for (int row = 0; row < TB; row++)
            for (int lane = 0; lane < LANES; lane++) out[LANES * row + lane] = in[fl_index(size_t(row), size_t(lane))];
        return;
    }
    const T mask = T((T(1) << W) - 1);
    // word-major loops so the store stream is sequential; every lane of a word row shares shift amounts
    for (int lane = 0; lane < LANES; lane++) {
        T acc = 0;
        int bit = 0, word = 0;
        for (int row = 0; row < TB; row++) {
            T v = T(in[fl_index(size_t(row), size_t(lane))] & mask);
            acc = T(acc | T(v << bit));
            int nb = bit + W;
            if (nb >= TB) {
                out[LANES * word + lane] = acc;
                word++;
                nb -= TB;
                acc = nb ? T(v >> (W - nb)) : T(0);
            }
            bit = nb;
        }
    }
}

template <typename T>
void fl_unpack_block(int W, const T* packed, T* out) {
    constexpr int TB = int(sizeof(T)) * 8;
    constexpr int LANES = 1024 / TB;
    const T mask = (W == TB) ? T(~T(0)) : T((T(1) << W) - 1);
    for (int row = 0; row < TB; row++) {
        const int start = row * W, wi = start / TB, sh = start % TB;
        const bool spill = sh + W > TB;
        for (int lane = 0; lane < LANES; lane++) {
            T v = T(packed[LANES * wi + lane] >> sh);
            if (spill) v = T(v | T(packed[LANES * (wi + 1) + lane] << (TB - sh)));
            out[fl_index(size_t(row), size_t(lane))] = T(v & mask);
        }
    }
}

inline size_t packed_bytes(int W, size_t n) { return ((n + 1023) / 1024) * size_t(128) * size_t(W); }

// pack n values (bit_pack_array.rs:71-124: full chunks then a zero padded tail chunk)
template <typename T>
void fl_pack(int W, const T* values, size_t n, uint8_t* out) {
    const size_t full = n / 1024, chunk_words = size_t(128) * size_t(W) / sizeof(T);
    T* o = reinterpret_cast<T*>(out);
    for (size_t c = 0; c < full; c++) fl_pack_block<T>(W, values + c * 1024, o + c * chunk_words);
    if (n % 1024) {
        T tail[1024];
        std::memset(tail, 0, sizeof(tail));
        std::memcpy(tail, values + full * 1024, (n % 1024) * sizeof(T));
        fl_pack_block<T>(W, tail, o + full * chunk_words);
    }
}

template <typename T>
void fl_unpack(int W, const uint8_t* packed, size_t n, T* out) {
    const size_t chunks = (n + 1023) / 1024, chunk_words = size_t(128) * size_t(W) / sizeof(T);
    const T* p = reinterpret_cast<const T*>(packed);
    T tmp[1024];
    for (size_t c = 0; c < chunks; c++) {
        const size_t take = (c + 1) * 1024 <= n ? 1024 : n - c * 1024;
        if (take == 1024) fl_unpack_block<T>(W, p + c * chunk_words, out + c * 1024);
        else {
            fl_unpack_block<T>(W, p + c * chunk_words, tmp);
            std::memcpy(out + c * 1024, tmp, take * sizeof(T));
        }
    }
}

// ---- IPC header ----
constexpr uint8_t kMagic[4] = {0x41, 0x44, 0x51, 0x4C};  // 0x4C51_4441 little endian ("LQDA")

inline void write_ipc_header(uint8_t* out, int logical, int phys) {
    std::memset(out, 0, 16);
    std::memcpy(out, kMagic, 4);
    wr<uint16_t>(out + 4, 1);
    wr<uint16_t>(out + 6, uint16_t(logical));
    wr<uint16_t>(out + 8, uint16_t(phys));
}

// ---- BitPackedArray section ----
struct BitPackedView {
    uint32_t len = 0;
    int bit_width = 0;
    bool has_nulls = false;
    const uint8_t* nulls = nullptr;
    uint32_t nulls_len = 0;
    const uint8_t* values = nullptr;
    uint32_t values_len = 0;
    bool all_null = false;
};

// returns false on malformed input (the reference panics: bit_pack_array.rs:265-301)
// `max_width`: bit width of the lane type the section is unpacked into (a wider field is malformed input)
inline bool parse_bitpacked(const uint8_t* sec, size_t sec_len, BitPackedView* v, int max_width = 64) {
    if (sec_len < 16) return false;
    *v = BitPackedView{};
    v->len = rd<uint32_t>(sec);
    v->bit_width = sec[4];
    v->has_nulls = sec[5] != 0;
    v->nulls_len = rd<uint32_t>(sec + 6);
    v->values_len = rd<uint32_t>(sec + 10);
    const size_t values_off = align8(16 + (v->has_nulls ? v->nulls_len : 0));
    if (v->values_len == 0) {  // :282-285
        v->all_null = true;
        return true;
    }
    if (v->has_nulls) {
        if (v->nulls_len == 0 || 16 + size_t(v->nulls_len) > sec_len) return false;
        if (size_t(v->nulls_len) < bitmap_bytes(v->len)) return false;
        v->nulls = sec + 16;
    }
    if (values_off + v->values_len > sec_len) return false;
    v->values = sec + values_off;
    if (v->has_nulls && count_bits(v->nulls, v->len) == 0) v->all_null = true;  // :323-325
    if (!v->all_null) {
        if (v->bit_width == 0 || v->bit_width > max_width) return false;
        if (size_t(v->values_len) < packed_bytes(v->bit_width, v->len)) return false;
    }
    return true;
}

// Appends a BitPackedArray section; W == 0 writes the all-null form (new_null_array, :43-50).
template <typename T>
void append_bitpacked(std::vector<uint8_t>& out, int W, const T* values, const uint8_t* validity, size_t n) {
    const size_t start = out.size();
    const bool has_nulls = validity != nullptr;
    const size_t nulls_len = has_nulls ? bitmap_bytes(n) : 0;
    const size_t values_len = W == 0 ? n * sizeof(T) : packed_bytes(W, n);
    const size_t values_off = align8(16 + nulls_len);
    out.resize(start + values_off + values_len, 0);
    uint8_t* p = out.data() + start;
    wr<uint32_t>(p, uint32_t(n));
    p[4] = uint8_t(W);
    p[5] = has_nulls ? 1 : 0;
    wr<uint32_t>(p + 6, uint32_t(nulls_len));
    wr<uint32_t>(p + 10, uint32_t(values_len));
    if (has_nulls && nulls_len) {
        std::memcpy(p + 16, validity, nulls_len);
        if (n & 7) p[16 + nulls_len - 1] &= uint8_t((1u << (n & 7)) - 1);
    }
    if (W != 0) fl_pack<T>(W, values, n, p + values_off);
}

// ---- parsed views of whole arrays ----
struct FixedView {  // Integer / Decimal / Float
    int logical = 0, phys = 0;
    int value_width = 0;  // decoded Arrow value bytes (decimal: 16, or 32 for Decimal256)
    int lane_bits = 0;    // FastLanes lane type
    uint64_t reference = 0;
    BitPackedView bp;
    // decimal
    int dec_is256 = 0, dec_precision = 0, dec_scale = 0;
    // ALP
    int alp_e = 0, alp_f = 0;
    uint64_t patch_len = 0;
    const uint8_t* patch_indices = nullptr;
    const uint8_t* patch_values = nullptr;
};

inline uint64_t load_native(const uint8_t* p, int w) {
    switch (w) {
        case 1: return *p;
        case 2: return rd<uint16_t>(p);
        case 4: return rd<uint32_t>(p);
        default: return rd<uint64_t>(p);
    }
}

inline bool read_ipc_header(const uint8_t* b, size_t len, int* logical, int* phys) {
    if (len < 16 || std::memcmp(b, kMagic, 4) != 0 || rd<uint16_t>(b + 4) != 1) return false;
    *logical = rd<uint16_t>(b + 6);
    *phys = rd<uint16_t>(b + 8);
    return true;
}

inline bool parse_fixed(const uint8_t* b, size_t len, FixedView* v) {
    *v = FixedView{};
    if (!read_ipc_header(b, len, &v->logical, &v->phys)) return false;
    size_t bp_off;
    if (v->logical == kInteger) {
        if (v->phys > kTsNs || phys_float(v->phys) || len < 24) return false;
        v->value_width = phys_width(v->phys);
        v->lane_bits = v->value_width * 8;
        v->reference = load_native(b + 16, v->value_width);
        bp_off = 24;
    } else if (v->logical == kDecimal) {
        if (len < 32 || v->phys != kU64) return false;
        v->dec_is256 = b[16];
        v->dec_precision = b[17];
        v->dec_scale = int8_t(b[18]);
        if (v->dec_is256 > 1) return false;
        v->value_width = v->dec_is256 ? 32 : 16;
        v->lane_bits = 64;
        v->reference = rd<uint64_t>(b + 24);
        bp_off = 32;
    } else if (v->logical == kFloat) {
        if (!phys_float(v->phys)) return false;
        const int w = phys_width(v->phys);
        v->value_width = w;
        v->lane_bits = w * 8;
        size_t next = align8(16 + size_t(w));
        if (len < next + 16) return false;
        v->reference = load_native(b + 16, w);
        v->alp_e = b[next];
        v->alp_f = b[next + 1];
        next += 8;
        v->patch_len = rd<uint64_t>(b + next);
        next += 8;
        if (v->patch_len > (len - next) / size_t(8 + w)) return false;
        v->patch_indices = b + next;
        next += size_t(v->patch_len) * 8;
        v->patch_values = b + next;
        next += size_t(v->patch_len) * size_t(w);
        bp_off = align8(next);
    } else {
        return false;
    }
    if (bp_off > len) return false;
    if (!parse_bitpacked(b + bp_off, len - bp_off, &v->bp, v->lane_bits)) return false;
    return true;
}

struct ByteViewParsed {
    int arrow_type = 0;
    uint32_t n = 0, d = 0;
    bool nullable = false, all_null = false;
    std::vector<uint16_t> keys;  // plain row order (the reference's in-memory UInt16Array)
    const uint8_t* key_validity = nullptr;
    const uint8_t* fsst = nullptr;
    uint32_t fsst_len = 0;
    uint64_t uncompressed_bytes = 0;
    int32_t slope = 0, intercept = 0;
    int offset_bytes = 1;
    const uint8_t* residuals = nullptr;  // (d+1) * offset_bytes
    uint32_t residual_count = 0;
    const uint8_t* prefix_keys = nullptr;  // d * 8
    const uint8_t* shared_prefix = nullptr;
    uint32_t shared_prefix_len = 0;
    const uint8_t* fingerprints = nullptr;  // d * 4 or null

    uint32_t offset_at(uint32_t i) const {  // fsst_buffer.rs:365-368
        int32_t r;
        if (offset_bytes == 1) r = int8_t(residuals[i]);
        else if (offset_bytes == 2) r = rd<int16_t>(residuals + 2 * size_t(i));
        else r = rd<int32_t>(residuals + 4 * size_t(i));
        return uint32_t(slope) * i + uint32_t(intercept) + uint32_t(r);
    }
};

inline bool parse_byte_view(const uint8_t* b, size_t len, ByteViewParsed* v) {
    int logical, phys;
    if (!read_ipc_header(b, len, &logical, &phys) || logical != kByteView || len < 40 || phys > kBinaryView) return false;
    v->arrow_type = phys;
    const uint32_t keys_size = rd<uint32_t>(b + 16), co_size = rd<uint32_t>(b + 20), sp_size = rd<uint32_t>(b + 24),
                   fsst_size = rd<uint32_t>(b + 28), fp_size = rd<uint32_t>(b + 32);
    size_t cur = 40;
    if (fsst_size < 12 || cur + fsst_size > len) return false;
    v->uncompressed_bytes = rd<uint64_t>(b + cur);
    v->fsst_len = rd<uint32_t>(b + cur + 8);
    if (12 + size_t(v->fsst_len) > fsst_size) return false;
    v->fsst = b + cur + 12;
    cur = align8(cur + fsst_size);
    if (cur + keys_size > len) return false;
    BitPackedView kv;
    if (!parse_bitpacked(b + cur, keys_size, &kv, 16)) return false;
    v->n = kv.len;
    v->nullable = kv.has_nulls || kv.all_null;
    v->all_null = kv.all_null;
    v->keys.assign(size_t(kv.len), 0);
    if (!kv.all_null) {
        if (kv.bit_width > 16) return false;
        fl_unpack<uint16_t>(kv.bit_width, kv.values, kv.len, v->keys.data());
        v->key_validity = kv.nulls;
    }
    cur = align8(cur + keys_size);
    if (cur + co_size > len) return false;
    if (co_size > 0) {
        if (co_size < 9) return false;
        v->slope = rd<int32_t>(b + cur);
        v->intercept = rd<int32_t>(b + cur + 4);
        v->offset_bytes = b[cur + 8];
        if (v->offset_bytes != 1 && v->offset_bytes != 2 && v->offset_bytes != 4) return false;
        if ((co_size - 9) % uint32_t(v->offset_bytes)) return false;
        v->residual_count = (co_size - 9) / uint32_t(v->offset_bytes);
        v->residuals = b + cur + 9;
    }
    v->d = v->residual_count ? v->residual_count - 1 : 0;
    cur = align8(cur + co_size);
    if (cur + size_t(v->d) * 8 > len) return false;
    v->prefix_keys = b + cur;
    cur = align8(cur + size_t(v->d) * 8);
    if (cur + sp_size > len) return false;
    v->shared_prefix = b + cur;
    v->shared_prefix_len = sp_size;
    cur = align8(cur + sp_size);
    if (fp_size) {
        if (cur + fp_size > len || fp_size != v->d * 4) return false;
        v->fingerprints = b + cur;
    }
    // offsets must be monotone and end inside the raw buffer
    uint32_t prev = 0;
    for (uint32_t i = 0; i <= v->d && v->residual_count; i++) {
        const uint32_t o = v->offset_at(i);
        if (o < prev || o > v->fsst_len) return false;
        prev = o;
    }
    // valid keys must reference the dictionary (reference: debug_assert, comparisons.rs:335)
    for (uint32_t i = 0; i < v->n && !v->all_null; i++) {
        if (v->key_validity && !get_bit(v->key_validity, i)) continue;
        if (v->keys[i] >= v->d) return false;
    }
    return true;
}

// fingerprint.rs:21-28
inline uint32_t fingerprint(const uint8_t* s, size_t l) {
    uint32_t bits = 0;
    for (size_t i = 0; i < l; i++) bits |= 1u << (s[i] & 31);
    return bits;
}

// fingerprint.rs:59-74.  One deliberate narrowing: the reference re-forms "%inner%" and runs Arrow `like`
// (comparisons.rs:629-634), in which a backslash escapes the next character, so an inner part holding a backslash is
// NOT a plain memmem needle.  Such patterns are not treated as substring searches here: entries without fingerprints
// evaluate them with the general Arrow-LIKE matcher (which implements the escape), entries with fingerprints answer
// LC_UNSUPPORTED (the caller runs the reference CPU path).
inline bool substring_pattern(const uint8_t* p, size_t pl, const uint8_t** inner, size_t* il) {
    if (pl < 3 || p[0] != '%' || p[pl - 1] != '%') return false;
    for (size_t i = 1; i + 1 < pl; i++)
        if (p[i] == '%' || p[i] == '_' || p[i] == '\\') return false;
    *inner = p + 1;
    *il = pl - 2;
    return true;
}

// Inverted row lists of a byte-view entry (lc_kernels.hpp): u16 offsets[D + 1], then the VALID rows grouped by key — a
// counting sort.  Keys of null slots may be garbage (the reference allows it) and keys >= D cannot be listed: both are
// skipped, as map_dictionary_results_to_array_results never yields a hit for them.  32 spare entries behind the rows.
inline std::vector<uint16_t> build_row_lists(const uint16_t* keys, const uint8_t* validity, uint32_t n, uint32_t d) {
    std::vector<uint16_t> post(size_t(d) + 1 + size_t(n) + 32, 0);
    uint16_t* off = post.data();
    uint16_t* rows = post.data() + d + 1;
    auto valid = [&](uint32_t r) { return !validity || ((validity[r >> 3] >> (r & 7)) & 1); };
    for (uint32_t r = 0; r < n; r++)
        if (valid(r) && keys[r] < d) off[size_t(keys[r]) + 1]++;
    for (uint32_t k = 0; k < d; k++) off[k + 1] = uint16_t(off[k + 1] + off[k]);
    std::vector<uint16_t> cursor(off, off + d);
    for (uint32_t r = 0; r < n; r++)
        if (valid(r) && keys[r] < d) rows[cursor[keys[r]]++] = uint16_t(r);
    return post;
}


}  // namespace lc
