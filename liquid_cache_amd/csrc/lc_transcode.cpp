// Arrow -> Liquid transcoder (host).  Mirrors transcode_liquid_inner_with_hint
// (reference: src/core/src/cache/transcode.rs:46-290) and the per-encoding `from_arrow_array` constructors:
//   ints     src/core/src/liquid_array/primitive_array.rs:159-206
//   decimal  src/core/src/liquid_array/decimal_array.rs:120-177
//   floats   src/core/src/liquid_array/float_array.rs:109-125, 609-740 (ALP)
//   strings  src/core/src/liquid_array/byte_view_array/conversions.rs:260-373
// and writes the reference's own serialized layouts (see lc_host.hpp).
#include "lc_transcode.hpp"

#include <cmath>
#include <limits>
#include <string_view>
#include <unordered_map>

namespace lc {

namespace {

// Re-slice an Arrow validity bitmap to bit offset 0 (bit_pack_array.rs:215-223 does the same before writing).
std::vector<uint8_t> slice_bitmap(const uint8_t* bm, int64_t offset, size_t n) {
    std::vector<uint8_t> out(bitmap_bytes(n) + 1, 0);
    for (size_t i = 0; i < n; i++)
        if (get_bit(bm, size_t(offset) + i)) set_bit(out.data(), i);
    return out;
}

template <typename U>
void encode_primitive_t(int phys, bool is_signed, const uint8_t* values, const uint8_t* validity, size_t n,
                        std::vector<uint8_t>& out) {
    using S = typename std::make_signed<U>::type;
    const U* v = reinterpret_cast<const U*>(values);
    bool have = false;
    U mn = 0, mx = 0;
    for (size_t i = 0; i < n; i++) {
        if (validity && !get_bit(validity, i)) continue;
        const U x = v[i];
        if (!have) { mn = mx = x; have = true; continue; }
        if (is_signed) {
            if (S(x) < S(mn)) mn = x;
            if (S(x) > S(mx)) mx = x;
        } else {
            if (x < mn) mn = x;
            if (x > mx) mx = x;
        }
    }
    out.assign(24, 0);
    write_ipc_header(out.data(), kInteger, phys);
    if (!have) {  // all null (or empty): reference_value 0 + new_null_array
        std::vector<uint8_t> nulls(bitmap_bytes(n) + 1, 0);
        append_bitpacked<U>(out, 0, nullptr, nulls.data(), n);
        return;
    }
    const int W = bit_width_of(uint64_t(U(mx - mn)));
    std::vector<U> rel(n);
    for (size_t i = 0; i < n; i++) rel[i] = U(v[i] - mn);
    wr<U>(out.data() + 16, mn);
    append_bitpacked<U>(out, W, rel.data(), validity, n);
}

// ---- ALP (float_array.rs:109-224) ----
template <typename F> struct Alp;
template <> struct Alp<float> {
    using I = int32_t;
    using U = uint32_t;
    static constexpr int kMaxExp = 10;
    static constexpr float kSweet = 8388608.0f + 4194304.0f;
    static float f10(int i) {
        static const float t[11] = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f,
                                    100000000.0f, 1000000000.0f, 10000000000.0f};
        return t[i];
    }
    static float if10(int i) {
        static const float t[11] = {1.0f, 0.1f, 0.01f, 0.001f, 0.0001f, 0.00001f, 0.000001f, 0.0000001f,
                                    0.00000001f, 0.000000001f, 0.0000000001f};
        return t[i];
    }
};
template <> struct Alp<double> {
    using I = int64_t;
    using U = uint64_t;
    static constexpr int kMaxExp = 18;
    static constexpr double kSweet = 4503599627370496.0 + 2251799813685248.0;
    static double f10(int i) {
        static const double t[24] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14,
                                     1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23};
        return t[i];
    }
    static double if10(int i) {
        static const double t[24] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001, 0.00000001,
                                     0.000000001, 0.0000000001, 0.00000000001, 0.000000000001, 0.0000000000001,
                                     0.00000000000001, 0.000000000000001, 0.0000000000000001, 0.00000000000000001,
                                     0.000000000000000001, 0.0000000000000000001, 0.00000000000000000001,
                                     0.000000000000000000001, 0.0000000000000000000001, 0.00000000000000000000001};
        return t[i];
    }
};

template <typename F>
inline typename Alp<F>::I alp_encode_one(F v, int e, int f) {
    using I = typename Alp<F>::I;
    volatile F t = v * Alp<F>::f10(e);
    t = t * Alp<F>::if10(f);
    volatile F r = t + Alp<F>::kSweet;
    r = r - Alp<F>::kSweet;
    const F x = r;
    if (x != x) return 0;  // Rust `as`: NaN -> 0, saturating
    if (x >= F(std::numeric_limits<I>::max())) return std::numeric_limits<I>::max();
    if (x <= F(std::numeric_limits<I>::min())) return std::numeric_limits<I>::min();
    return I(x);
}
template <typename F>
inline F alp_decode_one(typename Alp<F>::I i, int e, int f) {
    volatile F t = F(i);
    t = t * Alp<F>::f10(f);
    t = t * Alp<F>::if10(e);
    return t;
}

template <typename F>
struct AlpTrial {
    std::vector<typename Alp<F>::I> enc;
    std::vector<uint64_t> pidx;
    std::vector<F> pval;
    typename Alp<F>::I mn = 0, mx = 0;
};

template <typename F>
void alp_try(const F* v, size_t n, int e, int f, AlpTrial<F>& t) {
    using I = typename Alp<F>::I;
    t.enc.resize(n);
    t.pidx.clear();
    t.pval.clear();
    for (size_t i = 0; i < n; i++) {
        const I en = alp_encode_one<F>(v[i], e, f);
        t.enc[i] = en;
        if (!(alp_decode_one<F>(en, e, f) == v[i])) {
            t.pidx.push_back(i);
            t.pval.push_back(v[i]);
        }
    }
    if (!t.pidx.empty() && t.pidx.size() < n) {  // fill patched slots with the first clean encoding (:652-668)
        size_t first_clean = n;
        for (size_t i = 0; i < n; i++)
            if (i >= t.pidx.size() || t.pidx[i] != i) { first_clean = i; break; }
        if (first_clean < n) {
            const I fill = t.enc[first_clean];
            for (uint64_t p : t.pidx) t.enc[p] = fill;
        }
    }
    t.mn = std::numeric_limits<I>::max();
    t.mx = std::numeric_limits<I>::min();
    for (size_t i = 0; i < n; i++) { t.mn = std::min(t.mn, t.enc[i]); t.mx = std::max(t.mx, t.enc[i]); }
    if (n == 0) t.mn = t.mx = 0;
}

template <typename F>
void encode_float_t(int phys, const uint8_t* values, const uint8_t* validity, size_t n, std::vector<uint8_t>& out) {
    using I = typename Alp<F>::I;
    using U = typename Alp<F>::U;
    const F* v = reinterpret_cast<const F*>(values);
    out.assign(16, 0);
    write_ipc_header(out.data(), kFloat, phys);
    const size_t nulls = validity ? n - count_bits(validity, n) : 0;
    auto pad8 = [&]() { while (out.size() & 7) out.push_back(0); };
    if (n == 0 || (validity && nulls == n)) {  // float_array.rs:620-631
        out.resize(16 + sizeof(I), 0);
        pad8();
        out.resize(out.size() + 16, 0);
        std::vector<uint8_t> allnull(bitmap_bytes(n) + 1, 0);
        append_bitpacked<U>(out, 0, nullptr, validity ? validity : allnull.data(), n);
        return;
    }
    // exponent search on a sample (float_array.rs:715-740)
    std::vector<F> sample;
    const F* sv = v;
    size_t sn = n;
    if (n > 1024) {
        const size_t step = n / 1024;
        for (size_t i = 0; i < n; i += step)
            if (!validity || get_bit(validity, i)) sample.push_back(v[i]);
        sv = sample.data();
        sn = sample.size();
    }
    AlpTrial<F> trial;
    int be = 0, bf = 0;
    size_t best = size_t(-1);
    for (int e = 0; e < Alp<F>::kMaxExp && sn; e++) {
        for (int f = 0; f < e; f++) {
            alp_try<F>(sv, sn, e, f, trial);
            const int W = bit_width_of(uint64_t(U(U(trial.mx) - U(trial.mn))));
            const size_t est = packed_bytes(W, sn) + trial.pidx.size() * (8 + sizeof(F));
            if (est < best) { best = est; be = e; bf = f; }
        }
    }
    alp_try<F>(v, n, be, bf, trial);
    const size_t ref_off = out.size();
    out.resize(ref_off + sizeof(I));
    wr<I>(out.data() + ref_off, trial.mn);
    pad8();
    out.push_back(uint8_t(be));
    out.push_back(uint8_t(bf));
    out.resize(out.size() + 6, 0);
    const uint64_t pl = trial.pidx.size();
    size_t o = out.size();
    out.resize(o + 8 + pl * 8 + pl * sizeof(F));
    wr<uint64_t>(out.data() + o, pl);
    if (pl) {
        std::memcpy(out.data() + o + 8, trial.pidx.data(), pl * 8);
        std::memcpy(out.data() + o + 8 + pl * 8, trial.pval.data(), pl * sizeof(F));
    }
    pad8();
    const int W = bit_width_of(uint64_t(U(U(trial.mx) - U(trial.mn))));
    std::vector<U> rel(n);
    for (size_t i = 0; i < n; i++) rel[i] = U(U(trial.enc[i]) - U(trial.mn));
    append_bitpacked<U>(out, W, rel.data(), validity, n);
}

// 8 bytes at a time, multiply-rotate mixing (the dictionary only needs a well spread 64-bit hash, equality is checked)
inline uint64_t hash_bytes(std::string_view s) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(s.data());
    size_t n = s.size();
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t(n) * 0xC2B2AE3D27D4EB4Full);
    while (n >= 8) {
        uint64_t w;
        std::memcpy(&w, p, 8);
        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
        h = (h << 29) | (h >> 35);
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    if (n) std::memcpy(&w, p, n);
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 32);
}

struct SvHash {
    size_t operator()(std::string_view s) const {
        uint64_t h = 1469598103934665603ull;
        for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
        return size_t(h ^ (h >> 29));
    }
};

}  // namespace

lc_status transcode_primitive(int phys, const void* values, const uint8_t* validity, size_t n,
                              std::vector<uint8_t>& out) {
    const uint8_t* v = static_cast<const uint8_t*>(values);
    const bool sgn = !phys_unsigned(phys);
    switch (phys) {
        case kI8: case kU8: encode_primitive_t<uint8_t>(phys, sgn, v, validity, n, out); return LC_OK;
        case kI16: case kU16: encode_primitive_t<uint16_t>(phys, sgn, v, validity, n, out); return LC_OK;
        case kI32: case kU32: case kDate32: encode_primitive_t<uint32_t>(phys, sgn, v, validity, n, out); return LC_OK;
        case kI64: case kU64: case kDate64: case kTsS: case kTsMs: case kTsUs: case kTsNs:
            encode_primitive_t<uint64_t>(phys, sgn, v, validity, n, out);
            return LC_OK;
        case kF32: encode_float_t<float>(phys, v, validity, n, out); return LC_OK;
        case kF64: encode_float_t<double>(phys, v, validity, n, out); return LC_OK;
        default: return LC_UNSUPPORTED;
    }
}

// decimal_array.rs:127-177; values are `width` (16 or 32) byte little-endian two's complement integers
lc_status transcode_decimal(int width, int precision, int scale, const void* values, const uint8_t* validity, size_t n,
                            std::vector<uint8_t>& out) {
    const uint8_t* in = static_cast<const uint8_t*>(values);
    std::vector<uint64_t> vals(n, 0);
    uint64_t mn = UINT64_MAX, mx = 0;
    size_t nulls = 0;
    for (size_t i = 0; i < n; i++) {
        if (validity && !get_bit(validity, i)) { nulls++; continue; }
        const uint8_t* p = in + i * size_t(width);
        for (int b = 8; b < width; b++)
            if (p[b] != 0) return LC_UNSUPPORTED;  // fits_u64 (:120-125): negative or wider values stay on the CPU path
        vals[i] = rd<uint64_t>(p);
        mn = std::min(mn, vals[i]);
        mx = std::max(mx, vals[i]);
    }
    out.assign(32, 0);
    write_ipc_header(out.data(), kDecimal, kU64);
    out[16] = width == 32 ? 1 : 0;
    out[17] = uint8_t(precision);
    out[18] = uint8_t(int8_t(scale));
    if (n == 0 || nulls == n) {
        std::vector<uint8_t> allnull(bitmap_bytes(n) + 1, 0);
        append_bitpacked<uint64_t>(out, 0, nullptr, validity ? validity : allnull.data(), n);
        return LC_OK;
    }
    const int W = bit_width_of(mx - mn);
    for (size_t i = 0; i < n; i++) vals[i] = vals[i] >= mn ? vals[i] - mn : 0;  // saturating_sub (:164)
    wr<uint64_t>(out.data() + 24, mn);
    append_bitpacked<uint64_t>(out, W, vals.data(), validity, n);
    return LC_OK;
}

// fsst_buffer.rs:267-296
static void fit_line(const std::vector<uint32_t>& offs, int32_t* slope, int32_t* intercept) {
    const size_t n = offs.size();
    if (n <= 1) { *slope = 0; *intercept = n ? int32_t(offs[0]) : 0; return; }
    const double nf = double(n), sx = double(n * (n - 1) / 2), sxx = double(n * (n - 1) * (2 * n - 1) / 6);
    double sy = 0, sxy = 0;
    for (size_t i = 0; i < n; i++) { sy += double(offs[i]); sxy += double(i) * double(offs[i]); }
    const double s = (nf * sxy - sx * sy) / (nf * sxx - sx * sx);
    const double ic = (sy - s * sx) / nf;
    auto sat = [](double x) -> int32_t {
        x = std::round(x);
        if (x != x) return 0;
        if (x > 2147483647.0) return INT32_MAX;
        if (x < -2147483648.0) return INT32_MIN;
        return int32_t(x);
    };
    *slope = sat(s);
    *intercept = sat(ic);
}

// conversions.rs:260-373 + serialization.rs:122-220.  `get(i)` yields the i-th row's bytes.
lc_status transcode_byte_view(int arrow_type, const StringGetter& get, const uint8_t* validity, size_t n,
                              const SymbolTable& st, bool build_fingerprints, std::vector<uint8_t>& out) {
    // dictionary in first-occurrence order (GenericByteDictionaryBuilder::append_option, utils/mod.rs:147-161)
    // open addressing over (hash, dictionary index + 1): the node-based map this replaced was 45 % of the transcode of a
    // URL batch (one allocation per distinct value, a byte-wise hash of 80-byte strings)
    size_t cap = 64;
    while (cap < n * 2 + 16) cap <<= 1;
    std::vector<uint32_t> slot_idx(cap, 0);
    std::vector<uint64_t> slot_hash(cap, 0);
    std::vector<std::string_view> dict;
    std::vector<uint16_t> keys(n, 0);
    for (size_t i = 0; i < n; i++) {
        if (validity && !get_bit(validity, i)) continue;
        const std::string_view s = get(i);
        const uint64_t h = hash_bytes(s);
        size_t p = size_t(h) & (cap - 1);
        uint32_t found = 0;
        for (;;) {
            const uint32_t v = slot_idx[p];
            if (v == 0) break;
            if (slot_hash[p] == h && dict[v - 1] == s) { found = v; break; }
            p = (p + 1) & (cap - 1);
        }
        if (!found) {
            if (dict.size() >= 65536) return LC_UNSUPPORTED;
            dict.push_back(s);
            found = uint32_t(dict.size());
            slot_idx[p] = found;
            slot_hash[p] = h;
        }
        keys[i] = uint16_t(found - 1);
    }
    const size_t d = dict.size();
    // shared prefix (:269-307)
    size_t sp_len = d ? dict[0].size() : 0;
    for (size_t i = 1; i < d && sp_len; i++) {
        size_t c = 0;
        const size_t m = std::min(sp_len, dict[i].size());
        while (c < m && dict[0][c] == dict[i][c]) c++;
        sp_len = c;
    }
    size_t raw = 0;
    for (auto& s : dict) raw += s.size();
    auto pad8 = [&]() { while (out.size() & 7) out.push_back(0); };
    out.assign(40, 0);
    // A) RawFsstBuffer: [uncompressed_bytes u64][values_len u32][compressed values]
    const size_t fsst_start = out.size();
    out.resize(fsst_start + 12 + 2 * raw + 16);
    FsstEncoder enc(st);
    std::vector<uint32_t> offs(d + 1, 0);
    size_t clen = 0;
    uint8_t* cbuf = out.data() + fsst_start + 12;
    for (size_t i = 0; i < d; i++) {
        clen += enc.compress(reinterpret_cast<const uint8_t*>(dict[i].data()), dict[i].size(), cbuf + clen);
        offs[i + 1] = uint32_t(clen);
    }
    wr<uint64_t>(out.data() + fsst_start, uint64_t(raw));
    wr<uint32_t>(out.data() + fsst_start + 8, uint32_t(clen));
    out.resize(fsst_start + 12 + clen);
    const uint32_t fsst_raw_size = uint32_t(out.size() - fsst_start);
    pad8();
    // C) keys: BitPackedArray<u16> at bit width 16 (serialization.rs:141-150)
    const size_t keys_start = out.size();
    append_bitpacked<uint16_t>(out, 16, keys.data(), validity, n);
    const uint32_t keys_size = uint32_t(out.size() - keys_start);
    pad8();
    // E) compact offsets (fsst_buffer.rs:298-359, 762-784)
    const size_t co_start = out.size();
    {
        int32_t slope, intercept;
        fit_line(offs, &slope, &intercept);
        std::vector<int32_t> res(offs.size());
        int32_t mn = INT32_MAX, mx = INT32_MIN;
        for (size_t i = 0; i < offs.size(); i++) {
            const int32_t pred = int32_t(uint32_t(slope) * uint32_t(i) + uint32_t(intercept));
            res[i] = int32_t(offs[i] - uint32_t(pred));
            mn = std::min(mn, res[i]);
            mx = std::max(mx, res[i]);
        }
        const int ob = (mn >= -128 && mx <= 127) ? 1 : (mn >= -32768 && mx <= 32767) ? 2 : 4;
        out.resize(co_start + 9 + offs.size() * size_t(ob));
        uint8_t* p = out.data() + co_start;
        wr<int32_t>(p, slope);
        wr<int32_t>(p + 4, intercept);
        p[8] = uint8_t(ob);
        p += 9;
        for (int32_t r : res) {
            if (ob == 1) *p = uint8_t(int8_t(r));
            else if (ob == 2) wr<int16_t>(p, int16_t(r));
            else wr<int32_t>(p, r);
            p += ob;
        }
    }
    const uint32_t co_size = uint32_t(out.size() - co_start);
    pad8();
    // G) prefix keys (fsst_buffer.rs:175-187)
    for (size_t i = 0; i < d; i++) {
        const std::string_view s = dict[i];
        const size_t rl = sp_len < s.size() ? s.size() - sp_len : 0;
        uint8_t pk[8] = {0};
        if (rl) std::memcpy(pk, s.data() + sp_len, std::min<size_t>(rl, 7));
        pk[7] = rl >= 255 ? 255 : uint8_t(rl);
        out.insert(out.end(), pk, pk + 8);
    }
    pad8();
    // I) shared prefix
    if (sp_len) out.insert(out.end(), dict[0].begin(), dict[0].begin() + long(sp_len));
    pad8();
    // K) fingerprints over the full value (conversions.rs:353-355)
    uint32_t fp_size = 0;
    if (build_fingerprints) {
        for (size_t i = 0; i < d; i++) {
            const uint32_t fp = fingerprint(reinterpret_cast<const uint8_t*>(dict[i].data()), dict[i].size());
            const size_t o = out.size();
            out.resize(o + 4);
            wr<uint32_t>(out.data() + o, fp);
        }
        fp_size = uint32_t(d * 4);
    }
    write_ipc_header(out.data(), kByteView, arrow_type);
    wr<uint32_t>(out.data() + 16, keys_size);
    wr<uint32_t>(out.data() + 20, co_size);
    wr<uint32_t>(out.data() + 24, uint32_t(sp_len));
    wr<uint32_t>(out.data() + 28, fsst_raw_size);
    wr<uint32_t>(out.data() + 32, fp_size);
    return LC_OK;
}

// ---------------------------------------------------------------- Arrow C Data Interface front end
namespace {

struct ArrowStrings {
    const struct ArrowArray* a = nullptr;
    int kind = 0;  // 0: i32 offsets, 1: i64 offsets, 2: view
    std::string_view at(size_t i) const {
        const size_t r = size_t(a->offset) + i;
        if (kind == 0) {
            const int32_t* o = static_cast<const int32_t*>(a->buffers[1]);
            const char* d = static_cast<const char*>(a->buffers[2]);
            return std::string_view(d ? d + o[r] : "", size_t(o[r + 1] - o[r]));
        }
        if (kind == 1) {
            const int64_t* o = static_cast<const int64_t*>(a->buffers[1]);
            const char* d = static_cast<const char*>(a->buffers[2]);
            return std::string_view(d ? d + o[r] : "", size_t(o[r + 1] - o[r]));
        }
        const uint8_t* view = static_cast<const uint8_t*>(a->buffers[1]) + 16 * r;
        const uint32_t len = rd<uint32_t>(view);
        if (len <= 12) return std::string_view(reinterpret_cast<const char*>(view + 4), len);
        const int32_t buf = rd<int32_t>(view + 8), off = rd<int32_t>(view + 12);
        return std::string_view(static_cast<const char*>(a->buffers[2 + buf]) + off, len);
    }
};

}  // namespace

lc_status transcode_arrow(const struct ArrowArray* a, const struct ArrowSchema* s, int32_t hint,
                          SymtabProvider& symtabs, uint64_t path_id, std::vector<uint8_t>& out) {
    if (!a || !s || !s->format) return LC_ERR_INVALID;
    const std::string fmt = s->format;
    const size_t n = size_t(a->length);
    std::vector<uint8_t> validity_store;
    const uint8_t* validity = nullptr;
    if (a->n_buffers >= 1 && a->buffers[0] != nullptr) {
        validity_store = slice_bitmap(static_cast<const uint8_t*>(a->buffers[0]), a->offset, n);
        validity = validity_store.data();
    }
    auto prim = [&](int phys) {
        const uint8_t* v = static_cast<const uint8_t*>(a->buffers[1]);
        static const uint64_t zero = 0;
        if (!v) v = reinterpret_cast<const uint8_t*>(&zero);
        return transcode_primitive(phys, v + size_t(a->offset) * size_t(phys_width(phys)), validity, n, out);
    };
    const bool is_dictionary = s->dictionary != nullptr && a->dictionary != nullptr;
    if (!is_dictionary) {
    if (fmt == "c") return prim(kI8);
    if (fmt == "C") return prim(kU8);
    if (fmt == "s") return prim(kI16);
    if (fmt == "S") return prim(kU16);
    if (fmt == "i") return prim(kI32);
    if (fmt == "I") return prim(kU32);
    if (fmt == "l") return prim(kI64);
    if (fmt == "L") return prim(kU64);
    if (fmt == "f") return prim(kF32);
    if (fmt == "g") return prim(kF64);
    if (fmt == "tdD") return prim(kDate32);
    if (fmt == "tdm") return prim(kDate64);
    if (fmt.rfind("ts", 0) == 0 && fmt.size() >= 4 && fmt[3] == ':') {
        if (fmt.size() > 4) return LC_UNSUPPORTED;  // timezone-aware timestamps stay Arrow (transcode.rs:104-107)
        switch (fmt[2]) {
            case 's': return prim(kTsS);
            case 'm': return prim(kTsMs);
            case 'u': return prim(kTsUs);
            case 'n': return prim(kTsNs);
            default: return LC_UNSUPPORTED;
        }
    }
    if (fmt.rfind("d:", 0) == 0) {
        int precision = 0, scale = 0, bits = 128;
        if (std::sscanf(fmt.c_str(), "d:%d,%d,%d", &precision, &scale, &bits) < 2) return LC_ERR_INVALID;
        if (bits != 128 && bits != 256) return LC_UNSUPPORTED;
        const int width = bits / 8;
        const uint8_t* v = static_cast<const uint8_t*>(a->buffers[1]);
        return transcode_decimal(width, precision, scale, v + size_t(a->offset) * size_t(width), validity, n, out);
    }
    }
    int arrow_type = -1;
    ArrowStrings strs;
    strs.a = a;
    if (fmt == "u") { arrow_type = kUtf8; strs.kind = 0; }
    else if (fmt == "z") { arrow_type = kBinary; strs.kind = 0; }
    else if (fmt == "vu") { arrow_type = kUtf8View; strs.kind = 2; }
    else if (fmt == "vz") { arrow_type = kBinaryView; strs.kind = 2; }
    std::vector<uint16_t> dict_keys;
    ArrowStrings dict_vals;
    bool is_dict = false;
    if (arrow_type < 0 && is_dictionary && fmt == "S") {
        // Dictionary<UInt16, Utf8|Binary> (transcode.rs:262-284)
        const std::string vf = s->dictionary->format ? s->dictionary->format : "";
        if (vf == "u") arrow_type = kDict16Utf8;
        else if (vf == "z") arrow_type = kDict16Binary;
        else return LC_UNSUPPORTED;
        is_dict = true;
        dict_vals.a = a->dictionary;
        dict_vals.kind = 0;
    }
    if (arrow_type < 0) return LC_UNSUPPORTED;
    StringGetter get;
    if (is_dict) {
        const uint16_t* k = static_cast<const uint16_t*>(a->buffers[1]) + a->offset;
        const size_t dlen = size_t(a->dictionary->length);
        get = [k, dict_vals, dlen](size_t i) -> std::string_view {
            return k[i] < dlen ? dict_vals.at(k[i]) : std::string_view();
        };
    } else {
        get = [strs](size_t i) -> std::string_view { return strs.at(i); };
    }
    // train-once-per-path (transcode.rs:16-33): dictionary values / non-null strings of THIS array
    const SymbolTable* st = symtabs.find(path_id);
    SymbolTable trained;
    if (!st) {
        std::vector<std::pair<const uint8_t*, size_t>> train;
        train.reserve(n);
        for (size_t i = 0; i < n; i++) {
            if (validity && !get_bit(validity, i)) continue;
            const std::string_view v = get(i);
            train.emplace_back(reinterpret_cast<const uint8_t*>(v.data()), v.size());
        }
        trained = fsst_train(train);
        st = symtabs.insert(path_id, trained);
    }
    return transcode_byte_view(arrow_type, get, validity, n, *st, hint == LC_HINT_SUBSTRING_SEARCH, out);
}

}  // namespace lc
