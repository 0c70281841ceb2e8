// FSST symbol tables for the host-side transcoder: training, greedy compression, decode.
//
// The reference delegates to the crate fsst-rs 0.5.10 (call sites: src/core/src/liquid_array/raw/fsst_buffer.rs:73,
// :103, :396, :653; byte_view_array/comparisons.rs:541-547).  Decode is fully determined by the exported symbol
// table ([n:u8][len:u8 x n][symbol:u64 LE x n], fsst_buffer.rs:848-883): code c < 255 appends symbol c, code 255
// escapes the next literal byte.  Training follows the published FSST algorithm (5 count+select generations);
// the particular symbols chosen are an encoder-side freedom (the Liquid format lives only inside the cache).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace lc {

constexpr uint8_t kFsstEscape = 255;

struct SymbolTable {
    int n = 0;
    uint8_t len[256] = {};
    uint64_t sym[256] = {};

    bool load(const uint8_t* bytes, size_t size) {
        if (size < 1) return false;
        const int cnt = bytes[0];
        if (cnt > 255 || 1 + size_t(cnt) * 9 > size) return false;
        n = cnt;
        for (int i = 0; i < cnt; i++) {
            len[i] = bytes[1 + i];
            if (len[i] == 0 || len[i] > 8) return false;
            std::memcpy(&sym[i], bytes + 1 + cnt + 8 * size_t(i), 8);
            if (len[i] < 8) sym[i] &= (uint64_t(1) << (8 * len[i])) - 1;
        }
        for (int i = cnt; i < 256; i++) { len[i] = 0; sym[i] = 0; }
        return true;
    }
    std::vector<uint8_t> save() const {
        std::vector<uint8_t> out(1 + size_t(n) * 9);
        out[0] = uint8_t(n);
        for (int i = 0; i < n; i++) out[1 + i] = len[i];
        for (int i = 0; i < n; i++) std::memcpy(out.data() + 1 + n + 8 * size_t(i), &sym[i], 8);
        return out;
    }
};

inline size_t fsst_decode(const SymbolTable& st, const uint8_t* in, size_t in_len, uint8_t* out) {
    size_t o = 0;
    for (size_t i = 0; i < in_len; i++) {
        const uint8_t c = in[i];
        if (c == kFsstEscape) {
            if (++i >= in_len) break;
            out[o++] = in[i];
        } else {
            std::memcpy(out + o, &st.sym[c], 8);  // 8-byte store, advance by the symbol length (caller pads 8 bytes)
            o += st.len[c];
        }
    }
    return o;
}

inline size_t fsst_decoded_len(const SymbolTable& st, const uint8_t* in, size_t in_len) {
    size_t o = 0;
    for (size_t i = 0; i < in_len; i++) {
        if (in[i] == kFsstEscape) { if (++i < in_len) o++; }
        else o += st.len[in[i]];
    }
    return o;
}

// Greedy matcher: long symbols (>= 3 bytes) through a 3-byte-prefix hash with chained candidates (longest first),
// then a 64K-entry table indexed by the next two bytes for 2- and 1-byte symbols.
class FsstEncoder {
public:
    explicit FsstEncoder(const SymbolTable& st) : st_(st) { build(); }

    // out must have room for 2*len bytes (all bytes escaped).
    size_t compress(const uint8_t* in, size_t len, uint8_t* out) const {
        size_t o = 0, pos = 0;
        // eight or more bytes left: one unconditional 8-byte load per position, the matcher inlined (same decisions as
        // match(): longest symbol of >= 3 bytes first, then the 2- / 1-byte table)
        while (pos + 8 <= len) {
            uint64_t w;
            std::memcpy(&w, in + pos, 8);
            const uint32_t h = hash3(uint32_t(w) & 0xFFFFFF);
            int code = -1, l = 0;
            for (uint32_t i = bucket_[h], e = bucket_[h + 1]; i < e; i++) {
                const Long& s = longs_[i];
                if (((w ^ s.sym) & s.mask) == 0) { code = s.code; l = s.len; break; }
            }
            if (code < 0) {
                const uint16_t e2 = short2_[uint16_t(w)];
                if (e2 != kNone) { code = e2 & 0xFF; l = e2 >> 8; }
            }
            if (code >= 0) { out[o++] = uint8_t(code); pos += size_t(l); }
            else { out[o++] = kFsstEscape; out[o++] = in[pos++]; }
        }
        while (pos < len) {
            int l;
            const int c = match(in + pos, len - pos, &l);
            if (c >= 0) { out[o++] = uint8_t(c); pos += size_t(l); }
            else { out[o++] = kFsstEscape; out[o++] = in[pos++]; }
        }
        return o;
    }

    // longest symbol matching at p, or -1
    inline int match(const uint8_t* p, size_t avail, int* out_len) const {
        uint64_t w = 0;
        std::memcpy(&w, p, avail < 8 ? avail : 8);
        if (avail >= 3) {
            const uint32_t h = hash3(uint32_t(w) & 0xFFFFFF);
            for (uint32_t i = bucket_[h]; i < bucket_[h + 1]; i++) {
                const Long& s = longs_[i];
                if (s.len <= avail && ((w ^ s.sym) & s.mask) == 0) { *out_len = s.len; return s.code; }
            }
        }
        const uint16_t e = avail >= 2 ? short2_[uint16_t(w)] : short1_[uint8_t(w)];
        if (e == kNone) return -1;
        *out_len = e >> 8;
        return e & 0xFF;
    }

    // the matcher's tables for the device encoder (DevFsstEncoder, lc_kernels.hpp): same candidates in the same order
    template <class Dev, class Hash2>
    void export_device(Dev* d, uint32_t short2_slots, Hash2 hash2) const {
        std::memset(d, 0, sizeof(Dev));
        for (size_t i = 0; i < longs_.size(); i++) {
            d->long_sym[i] = longs_[i].sym;
            d->long_len[i] = longs_[i].len;
            d->long_code[i] = longs_[i].code;
        }
        for (uint32_t h = 0; h <= kBuckets; h++) d->bucket[h] = uint8_t(bucket_[h]);  // <= 255 symbols
        for (int b = 0; b < 256; b++) d->short1[b] = short1_[b] == kNone ? uint16_t(0xFFFF) : uint16_t(short1_[b] & 0xFF);
        for (int c = 0; c < st_.n; c++) {
            if (st_.len[c] != 2) continue;
            const uint32_t key = uint16_t(st_.sym[c]);
            uint32_t s = hash2(key);
            while (d->short2[s] != 0 && ((d->short2[s] >> 8) & 0xFFFFu) != key) s = (s + 1) & (short2_slots - 1);
            d->short2[s] = (1u << 31) | (key << 8) | uint32_t(c);  // a later code of the same two bytes wins, as in build()
        }
    }

private:
    struct Long { uint64_t sym, mask; uint8_t len; uint8_t code; uint32_t h; };
    static constexpr uint16_t kNone = 0xFFFF;
    static constexpr uint32_t kBuckets = 4096;
    static inline uint32_t hash3(uint32_t x) { return (x * 2654435761u) >> 20; }

    void build() {
        short1_.assign(256, kNone);
        short2_.assign(65536, kNone);
        for (int c = 0; c < st_.n; c++)
            if (st_.len[c] == 1) short1_[uint8_t(st_.sym[c])] = uint16_t((1 << 8) | c);
        for (uint32_t w = 0; w < 65536; w++) short2_[w] = short1_[w & 0xFF];
        for (int c = 0; c < st_.n; c++)
            if (st_.len[c] == 2) short2_[uint16_t(st_.sym[c])] = uint16_t((2 << 8) | c);
        for (int c = 0; c < st_.n; c++) {
            if (st_.len[c] < 3) continue;
            Long l;
            l.len = st_.len[c];
            l.mask = l.len >= 8 ? ~uint64_t(0) : ((uint64_t(1) << (8 * l.len)) - 1);
            l.sym = st_.sym[c] & l.mask;
            l.code = uint8_t(c);
            l.h = hash3(uint32_t(l.sym) & 0xFFFFFF);
            longs_.push_back(l);
        }
        std::sort(longs_.begin(), longs_.end(), [](const Long& a, const Long& b) {
            if (a.h != b.h) return a.h < b.h;
            if (a.len != b.len) return a.len > b.len;
            return a.code < b.code;
        });
        bucket_.assign(kBuckets + 1, 0);
        for (const Long& l : longs_) bucket_[l.h + 1]++;
        for (uint32_t i = 0; i < kBuckets; i++) bucket_[i + 1] += bucket_[i];
    }

    SymbolTable st_;
    std::vector<uint16_t> short1_, short2_;
    std::vector<Long> longs_;
    std::vector<uint32_t> bucket_;
};

// Train a symbol table on a set of byte strings (data + i32/i64-free offsets given as pointers and lengths).
SymbolTable fsst_train(const std::vector<std::pair<const uint8_t*, size_t>>& strings);

}  // namespace lc
