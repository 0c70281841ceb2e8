// Synthetic ClickBench-shaped data (see include/liquid_cache_amd_bench.h).  Benchmark tooling, not product path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/liquid_cache_amd_bench.h"

namespace {

struct Rng {  // splitmix64 / xoshiro-style mixing: cheap and reproducible across platforms
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return uint32_t((next() >> 32) * uint64_t(n) >> 32); }
    double unit() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
};

const char* const kHosts[] = {
    "yandex.ru", "mail.ru", "auto.ru", "kinopoisk.ru", "images.yandex.ru", "market.yandex.ru", "news.yandex.ru",
    "afisha.mail.ru", "pogoda.yandex.ru", "maps.yandex.ru", "video.yandex.ru", "rabota.yandex.ru", "irr.ru",
    "avito.ru", "drom.ru", "tut.by", "rambler.ru", "liveinternet.ru", "vk.com", "ok.ru", "wildberries.ru",
    "hh.ru", "cian.ru", "komok.com", "smeshariki.ru", "sprashivai.ru", "tv.yandex.ru", "realty.yandex.ru",
    "music.yandex.ru", "pikabu.ru", "bigcinema.tv", "kp.ru", "sport.mail.ru", "my.mail.ru", "otvet.mail.ru",
    "lady.mail.ru", "hi-tech.mail.ru", "deti.mail.ru", "torg.mail.ru", "travel.mail.ru"};
const char* const kWords[] = {
    "search", "catalog", "item", "view", "page", "product", "category", "user", "profile", "photo", "album", "video",
    "watch", "news", "article", "story", "forum", "topic", "thread", "message", "inbox", "compose", "cart", "order",
    "checkout", "price", "model", "brand", "city", "region", "moskva", "spb", "kazan", "samara", "omsk", "ufa",
    "rostov", "volgograd", "perm", "voronezh", "auto", "moto", "realty", "flat", "house", "rent", "sale", "job",
    "vacancy", "resume", "film", "serial", "season", "episode", "music", "track", "artist", "playlist", "weather",
    "forecast", "map", "route", "taxi", "hotel", "tour", "ticket", "bank", "credit", "card", "phone", "tablet",
    "laptop", "tv", "fridge", "sofa", "dress", "shoes", "kids", "toys", "sport", "fitness", "health", "recipe",
    "index", "main", "list", "detail", "filter", "sort", "popular", "new", "top", "best", "free", "online", "mobile",
    "wap", "api", "ajax", "json", "static", "img", "css", "js", "upload", "download", "file", "doc", "pdf"};
const char* const kParams[] = {"id", "page", "q", "text", "sid", "from", "utm_source", "ref", "lr", "p", "cat",
                               "sort", "lang", "rid", "uid", "clid", "msid", "type", "mode", "from_serp"};
constexpr uint32_t kNHosts = sizeof(kHosts) / sizeof(kHosts[0]);
constexpr uint32_t kNWords = sizeof(kWords) / sizeof(kWords[0]);
constexpr uint32_t kNParams = sizeof(kParams) / sizeof(kParams[0]);

void append_num(std::string& s, Rng& r, uint32_t digits) {
    for (uint32_t i = 0; i < digits; i++) s.push_back(char('0' + r.below(10)));
}

void make_url(std::string& s, Rng& r, bool with_needle) {
    s.clear();
    s += r.below(10) == 0 ? "https://" : "http://";
    if (r.below(4) == 0) s += "www.";
    // Zipf-ish host choice
    const uint32_t h = uint32_t(double(kNHosts) * std::pow(r.unit(), 2.2));
    s += kHosts[h < kNHosts ? h : kNHosts - 1];
    // path: length drawn so that the total length is roughly log-normal around ~70 bytes
    const double target = std::exp(3.9 + 0.55 * (r.unit() + r.unit() + r.unit() + r.unit() - 2.0) * 1.7);
    const size_t want = size_t(target < 18 ? 18 : (target > 500 ? 500 : target));
    uint32_t seg = 0;
    const uint32_t needle_at = with_needle ? r.below(3) : 99;
    while (s.size() < want && seg < 12) {
        s.push_back('/');
        if (seg == needle_at) s += "google";
        else if (r.below(5) == 0) append_num(s, r, 3 + r.below(7));
        else s += kWords[r.below(kNWords)];
        if (r.below(6) == 0) { s.push_back('_'); s += kWords[r.below(kNWords)]; }
        seg++;
    }
    if (with_needle && needle_at >= seg) s += "/google";
    if (r.below(3) != 0) {
        char sep = '?';
        const uint32_t np = 1 + r.below(4);
        for (uint32_t i = 0; i < np && s.size() < 500; i++) {
            s.push_back(sep);
            sep = '&';
            s += kParams[r.below(kNParams)];
            s.push_back('=');
            if (r.below(3) == 0) s += kWords[r.below(kNWords)];
            else append_num(s, r, 1 + r.below(9));
        }
    } else if (r.below(2) == 0) {
        s += ".html";
    }
    if (s.size() > 500) s.resize(500);
}

}  // namespace

extern "C" {

size_t lc_synth_url_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique, uint32_t needle_ppm,
                          int32_t* offsets, uint8_t* data, size_t data_cap) {
    if (n_unique == 0) n_unique = 1;
    Rng r(seed * 0x100000001B3ull + batch_index * 0xD6E8FEB86659FD93ull + 1);
    std::vector<std::string> pool(n_unique);
    std::string tmp;
    for (uint32_t i = 0; i < n_unique; i++) {
        const bool needle = needle_ppm && (r.next() % 1000000ull) < needle_ppm;
        make_url(tmp, r, needle);
        // make values distinct within the batch
        tmp += '#';
        uint32_t v = i;
        do { tmp.push_back(char('a' + v % 26)); v /= 26; } while (v);
        pool[i] = tmp;
    }
    size_t pos = 0;
    offsets[0] = 0;
    for (uint32_t row = 0; row < rows; row++) {
        // Zipf-like repetition: most rows reuse a small head of the pool, every pool entry appears at least once
        uint32_t k;
        if (row < n_unique) k = row;
        else {
            const double u = r.unit();
            k = uint32_t(double(n_unique) * u * u * u);
            if (k >= n_unique) k = n_unique - 1;
        }
        const std::string& s = pool[k];
        if (pos + s.size() > data_cap) return 0;
        std::memcpy(data + pos, s.data(), s.size());
        pos += s.size();
        offsets[row + 1] = int32_t(pos);
    }
    return pos;
}

size_t lc_synth_phrase_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique,
                             uint32_t empty_permille, int32_t* offsets, uint8_t* data, size_t data_cap) {
    if (n_unique == 0) n_unique = 1;
    Rng r(seed * 0x100000001B3ull + batch_index * 0xA24BAED4963EE407ull + 3);
    std::vector<std::string> pool(n_unique);
    for (uint32_t i = 0; i < n_unique; i++) {
        std::string& p = pool[i];
        const uint32_t nw = 1 + r.below(5);
        for (uint32_t w = 0; w < nw; w++) {
            if (w) p.push_back(' ');
            p += kWords[r.below(kNWords)];
        }
        if (r.below(4) == 0) { p.push_back(' '); append_num(p, r, 1 + r.below(4)); }
    }
    size_t pos = 0;
    offsets[0] = 0;
    for (uint32_t row = 0; row < rows; row++) {
        if (r.below(1000) >= empty_permille) {
            const double u = r.unit();
            uint32_t k = uint32_t(double(n_unique) * u * u);
            if (k >= n_unique) k = n_unique - 1;
            const std::string& p = pool[k];
            if (pos + p.size() > data_cap) return 0;
            std::memcpy(data + pos, p.data(), p.size());
            pos += p.size();
        }
        offsets[row + 1] = int32_t(pos);
    }
    return pos;
}

size_t lc_synth_title_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique, uint32_t needle_ppm,
                            int32_t* offsets, uint8_t* data, size_t data_cap) {
    if (n_unique == 0) n_unique = 1;
    Rng r(seed * 0x100000001B3ull + batch_index * 0xC2B2AE3D27D4EB4Full + 5);
    std::vector<std::string> pool(n_unique);
    for (uint32_t i = 0; i < n_unique; i++) {
        std::string& p = pool[i];
        const bool needle = needle_ppm && (r.next() % 1000000ull) < needle_ppm;
        const uint32_t nw = 3 + r.below(7);
        const uint32_t at = needle ? r.below(nw) : 99;
        for (uint32_t w = 0; w < nw; w++) {
            if (w) p.push_back(' ');
            if (w == at) { p += "Google"; continue; }
            const char* word = kWords[r.below(kNWords)];
            p.push_back(char(word[0] - 32));  // capitalised (the vocabulary is lower-case ASCII)
            p += word + 1;
        }
        p += " - ";
        p += kHosts[r.below(kNHosts)];
        p += '#';
        uint32_t v = i;
        do { p.push_back(char('a' + v % 26)); v /= 26; } while (v);
    }
    size_t pos = 0;
    offsets[0] = 0;
    for (uint32_t row = 0; row < rows; row++) {
        uint32_t k;
        if (row < n_unique) k = row;
        else {
            const double u = r.unit();
            k = uint32_t(double(n_unique) * u * u * u);
            if (k >= n_unique) k = n_unique - 1;
        }
        const std::string& t = pool[k];
        if (pos + t.size() > data_cap) return 0;
        std::memcpy(data + pos, t.data(), t.size());
        pos += t.size();
        offsets[row + 1] = int32_t(pos);
    }
    return pos;
}

void lc_synth_int64_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, int32_t bit_width, int64_t base,
                          int64_t* out) {
    Rng r(seed * 0x100000001B3ull + batch_index * 0x9E3779B97F4A7C15ull + 7);
    const uint64_t mask = bit_width >= 64 ? ~uint64_t(0) : ((uint64_t(1) << bit_width) - 1);
    for (uint32_t i = 0; i < rows; i++) out[i] = int64_t(uint64_t(base) + (r.next() & mask));
    if (rows >= 2 && bit_width < 64) {  // pin the batch range so every batch has exactly this bit width
        out[0] = base;
        out[1] = int64_t(uint64_t(base) + mask);
    }
}

}  // extern "C"
