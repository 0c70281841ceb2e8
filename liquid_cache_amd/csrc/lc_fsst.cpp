// FSST symbol-table training (host).  See lc_fsst.hpp for the format and the reference call sites.
#include "lc_fsst.hpp"

#include <unordered_map>

namespace lc {

namespace {

struct Cand {
    uint64_t sym;
    uint8_t len;
    uint64_t gain;
};

struct KeyHash {
    size_t operator()(const std::pair<uint64_t, uint8_t>& k) const {
        return size_t((k.first ^ (uint64_t(k.second) << 57)) * 0x9E3779B97F4A7C15ull);
    }
};

inline uint64_t low_mask(int l) { return l >= 8 ? ~uint64_t(0) : ((uint64_t(1) << (8 * l)) - 1); }

}  // namespace

SymbolTable fsst_train(const std::vector<std::pair<const uint8_t*, size_t>>& strings) {
    SymbolTable st;
    if (strings.empty()) return st;
    // sample ~128 KiB, evenly spread over the input
    size_t total = 0;
    for (auto& s : strings) total += s.second;
    const size_t step = total / (128 * 1024) + 1;
    std::vector<std::pair<const uint8_t*, size_t>> sample;
    for (size_t i = 0; i < strings.size(); i += step)
        if (strings[i].second) sample.push_back(strings[i]);
    if (sample.empty()) return st;

    // code space while training: 0..255 raw bytes, 256+i = symbol i of the current table
    std::vector<uint32_t> count1(512), count2(size_t(512) * 512);
    for (int gen = 0; gen < 5; gen++) {
        std::fill(count1.begin(), count1.end(), 0u);
        std::fill(count2.begin(), count2.end(), 0u);
        FsstEncoder enc(st);
        for (auto& s : sample) {
            const uint8_t* p = s.first;
            const size_t sl = s.second;
            size_t pos = 0;
            int prev = -1;
            while (pos < sl) {
                int l = 1;
                const int c = enc.match(p + pos, sl - pos, &l);
                const int code = c >= 0 ? 256 + c : p[pos];
                if (c < 0) l = 1;
                count1[size_t(code)]++;
                if (prev >= 0) count2[size_t(prev) * 512 + size_t(code)]++;
                if (l > 1) {  // let the leading byte keep competing so single bytes can bootstrap longer symbols
                    count1[p[pos]]++;
                    if (prev >= 0) count2[size_t(prev) * 512 + p[pos]]++;
                }
                prev = code;
                pos += size_t(l);
            }
        }
        uint64_t csym[512];
        uint8_t clen[512];
        for (int c = 0; c < 512; c++) {
            if (c < 256) { csym[c] = uint64_t(c); clen[c] = 1; }
            else if (c - 256 < st.n) { csym[c] = st.sym[c - 256] & low_mask(st.len[c - 256]); clen[c] = st.len[c - 256]; }
            else { csym[c] = 0; clen[c] = 0; }
        }
        // every candidate with its gain; equal (symbol, length) candidates are merged after a sort by key (the sums and the
        // final order do not depend on how they are collected: a hash map here cost a third of the training time)
        std::vector<Cand> raw;
        raw.reserve(16384);
        for (int c = 0; c < 512; c++)
            if (count1[size_t(c)] && clen[c]) raw.push_back(Cand{csym[c], clen[c], uint64_t(count1[size_t(c)]) * clen[c]});
        for (int a = 0; a < 512; a++) {
            if (!clen[a] || clen[a] >= 8 || !count1[size_t(a)]) continue;  // (a code that never occurred starts no pair)
            const uint32_t* row = &count2[size_t(a) * 512];
            for (int b = 0; b < 512; b++) {
                const uint32_t cnt = row[b];
                if (cnt < 2 || !clen[b]) continue;
                const int lc = std::min(8, int(clen[a]) + int(clen[b]));
                const uint64_t s = (csym[a] | (csym[b] << (8 * clen[a]))) & low_mask(lc);
                raw.push_back(Cand{s, uint8_t(lc), uint64_t(cnt) * uint64_t(lc)});
            }
        }
        std::sort(raw.begin(), raw.end(), [](const Cand& x, const Cand& y) { return x.sym != y.sym ? x.sym < y.sym : x.len < y.len; });
        std::vector<Cand> cands;
        cands.reserve(raw.size());
        for (const Cand& c : raw) {
            if (!cands.empty() && cands.back().sym == c.sym && cands.back().len == c.len) cands.back().gain += c.gain;
            else cands.push_back(c);
        }
        std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) {
            if (x.gain != y.gain) return x.gain > y.gain;
            if (x.len != y.len) return x.len > y.len;
            return x.sym < y.sym;
        });
        SymbolTable next;
        for (const Cand& c : cands) {
            if (next.n >= 255) break;
            if (c.len == 1 && c.gain < 2) continue;  // an escape costs the same two bytes
            next.sym[next.n] = c.sym;
            next.len[next.n] = c.len;
            next.n++;
        }
        st = next;
    }
    return st;
}

}  // namespace lc
