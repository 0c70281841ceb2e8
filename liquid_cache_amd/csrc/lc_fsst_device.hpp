// The FSST encoder as the on-device byte-view transcoder runs it (lc_bv_encode.hip): table layout, matcher and the
// per-value compression loop in ONE place, compiled for the device by hipcc and for the host by any C++17 compiler — the
// CPU model test (tests/c_abi/fsst_device_encoder_model.cpp) runs exactly this code against FsstEncoder::compress
// (lc_fsst.hpp), whose decisions it must reproduce code for code.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define LC_HD __host__ __device__
#else
#define LC_HD
#endif

namespace lc {

// FsstEncoder (lc_fsst.hpp) as the kernels read it: the symbols of >= 3 bytes in bucket order of their 3-byte-prefix hash
// (longest first inside a bucket), the 2-byte symbols in a small open-addressing table, the 1-byte symbols by value.
constexpr uint32_t kDevEncBuckets = 4096;
constexpr uint32_t kDevEncShort2Slots = 1024;
LC_HD inline uint32_t dev_enc_short2_hash(uint32_t key16) { return ((key16 * 40503u) >> 4) & (kDevEncShort2Slots - 1); }
struct DevFsstEncoder {
    uint64_t long_sym[256];                 // masked to the symbol's length
    uint32_t short2[kDevEncShort2Slots];    // 0: free, else 1 << 31 | two bytes << 8 | code
    uint16_t short1[256];                   // 0xFFFF: none, else the code
    uint8_t long_len[256];
    uint8_t long_code[256];
    uint8_t bucket[kDevEncBuckets + 8];     // long symbols of hash h: [bucket[h], bucket[h + 1])
};

// FsstEncoder::match: the longest symbol of >= 3 bytes that fits, else the 2-byte symbol, else the 1-byte one, else -1.
// `w` holds the next min(avail, 8) bytes, zero extended.  E: DevFsstEncoder or a copy of it with the same members (LDS).
template <class E>
LC_HD inline int dev_enc_match(const E& e, uint64_t w, uint32_t avail, uint32_t* out_len) {
    if (avail >= 3) {
        const uint32_t h = ((uint32_t(w) & 0xFFFFFFu) * 2654435761u) >> 20;
        for (uint32_t i = e.bucket[h], end = e.bucket[h + 1]; i < end; i++) {
            const uint32_t l = e.long_len[i];
            const uint64_t mask = l >= 8 ? ~uint64_t(0) : ((uint64_t(1) << (8u * l)) - 1);
            if (l <= avail && ((w ^ e.long_sym[i]) & mask) == 0) { *out_len = l; return e.long_code[i]; }
        }
    }
    if (avail >= 2) {
        const uint32_t key = uint32_t(w) & 0xFFFFu;
        for (uint32_t s = dev_enc_short2_hash(key);; s = (s + 1) & (kDevEncShort2Slots - 1)) {
            const uint32_t v = e.short2[s];
            if (v == 0) break;
            if (((v >> 8) & 0xFFFFu) == key) { *out_len = 2; return int(v & 0xFFu); }
        }
    }
    const uint16_t s1 = e.short1[uint32_t(w) & 0xFFu];
    if (s1 == 0xFFFFu) return -1;
    *out_len = 1;
    return int(s1 & 0xFFu);
}

// One value: greedy longest match per position, escape (255, byte) where no symbol starts; also the 32-bucket byte
// fingerprint of the value (fingerprint.rs:33-35), which falls out of the same pass.  `load(pos, avail)` returns the next
// min(avail, 8) bytes at `pos`, zero extended.  `out` needs room for 2 * len bytes.  Returns the compressed length.
template <class E, class Load>
LC_HD inline uint32_t dev_enc_compress(const E& e, Load load, uint32_t len, uint8_t* out, uint32_t* fingerprint) {
    uint32_t o = 0, pos = 0, fp = 0;
    while (pos < len) {
        const uint32_t avail = len - pos;
        const uint64_t w = load(pos, avail);
        uint32_t l = 1;
        const int code = dev_enc_match(e, w, avail < 8u ? avail : 8u, &l);
        if (code >= 0) {
            out[o++] = uint8_t(code);
        } else {
            out[o++] = 255;
            out[o++] = uint8_t(w);
            l = 1;
        }
        for (uint32_t b = 0; b < l; b++) fp |= 1u << (uint32_t(w >> (8u * b)) & 31u);
        pos += l;
    }
    *fingerprint = fp;
    return o;
}

}  // namespace lc
