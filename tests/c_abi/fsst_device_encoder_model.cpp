// CPU run of the FSST encoder the on-device byte-view transcoder uses (liquid_cache_amd/csrc/lc_fsst_device.hpp: the very
// functions k_bv_build calls) against the host transcoder's FsstEncoder::compress (lc_fsst.hpp): same code stream, byte for
// byte, for tables trained on several kinds of data and for values the tables were not trained on; the fingerprint that falls
// out of the pass is checked against the plain definition.  Built and run by tests/test_c_abi_program.py with g++ (no GPU).
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../liquid_cache_amd/csrc/lc_fsst.hpp"
#include "../../liquid_cache_amd/csrc/lc_fsst_device.hpp"

using namespace lc;

static std::vector<std::string> make_values(std::mt19937_64& rng, int flavour, int n) {
    std::vector<std::string> out;
    static const char* hosts[] = {"google", "yandex", "mail", "example", "go", "ogle", "gle.goo"};
    static const char* paths[] = {"search", "q=%D0%BF%D0%BE", "index.php?id=", "%2F%2F", "a", "", "tours", "&page="};
    for (int i = 0; i < n; i++) {
        std::string s;
        if (flavour == 0) {  // URLs
            s = (rng() & 1) ? "https://" : "http://";
            const int nh = 1 + int(rng() % 3);
            for (int k = 0; k < nh; k++) { if (k) s += "."; s += hosts[rng() % 7]; }
            const int np = int(rng() % 5);
            for (int k = 0; k < np; k++) { s += "/"; s += paths[rng() % 8]; s += std::to_string(rng() % 50); }
            if (rng() % 25 == 0) s += std::string(250 + rng() % 350, 'x');
        } else if (flavour == 1) {  // every byte value, short and long, runs of 0xFF
            static const int lens[] = {0, 1, 2, 3, 7, 8, 9, 15, 16, 17, 40, 130, 300};
            const int ln = lens[rng() % 13];
            for (int k = 0; k < ln; k++) s.push_back(char(rng() & 0xFF));
            if (ln && rng() % 3 == 0) for (int k = 0; k < 3 && k < ln; k++) s[(rng() % ln)] = char(0xFF);
        } else if (flavour == 2) {  // four letters: the table is full of long symbols
            const int ln = int(rng() % 60);
            for (int k = 0; k < ln; k++) s.push_back("abcd"[rng() % 4]);
        } else {  // text with rare bytes in between (escapes)
            const int ln = 1 + int(rng() % 50);
            static const char* bits[] = {"ma", "il", "go", "og", "le", "ai", "gm", "x"};
            for (int k = 0; k < ln; k++) {
                if (rng() % 3 == 0) s.push_back(char(128 + rng() % 128));
                else s += bits[rng() % 8];
            }
        }
        out.push_back(s);
    }
    return out;
}

int main() {
    std::mt19937_64 rng(20260925);
    size_t checked = 0, bytes = 0, escapes = 0;
    for (int table_flavour = 0; table_flavour < 4; table_flavour++) {
        for (int rep = 0; rep < 3; rep++) {
            const std::vector<std::string> train = make_values(rng, table_flavour, 1500);
            std::vector<std::pair<const uint8_t*, size_t>> tv;
            for (const std::string& s : train) tv.emplace_back(reinterpret_cast<const uint8_t*>(s.data()), s.size());
            const SymbolTable st = fsst_train(tv);
            const FsstEncoder enc(st);
            static DevFsstEncoder dev;
            enc.export_device(&dev, kDevEncShort2Slots, dev_enc_short2_hash);
            for (int data_flavour = 0; data_flavour < 4; data_flavour++) {  // its own data and foreign data
                for (const std::string& s : make_values(rng, data_flavour, 400)) {
                    std::vector<uint8_t> want(2 * s.size() + 16), got(2 * s.size() + 16);
                    std::vector<uint8_t> padded(s.begin(), s.end());
                    padded.resize(s.size() + 16, 0xA5);  // the device reads 8 bytes at a time past the end of a value
                    const size_t wl = enc.compress(reinterpret_cast<const uint8_t*>(s.data()), s.size(), want.data());
                    uint32_t fp = 0;
                    const uint8_t* p = padded.data();
                    const uint32_t gl = dev_enc_compress(dev, [p](uint32_t pos, uint32_t avail) {
                        uint64_t w;
                        std::memcpy(&w, p + pos, 8);
                        return avail >= 8 ? w : (w & ((uint64_t(1) << (8u * avail)) - 1));
                    }, uint32_t(s.size()), got.data(), &fp);
                    if (gl != wl || std::memcmp(want.data(), got.data(), wl) != 0) {
                        std::printf("MISMATCH table %d/%d data %d len %zu: host %zu device %u bytes\n", table_flavour, rep, data_flavour,
                                    s.size(), wl, gl);
                        return 1;
                    }
                    uint32_t fp_want = 0;
                    for (unsigned char c : s) fp_want |= 1u << (c & 31);
                    if (fp != fp_want) { std::printf("FINGERPRINT mismatch\n"); return 1; }
                    // and the stream decodes to the value
                    std::vector<uint8_t> back(s.size() + 16);
                    if (fsst_decode(st, got.data(), gl, back.data()) != s.size() || std::memcmp(back.data(), s.data(), s.size()) != 0) {
                        std::printf("ROUND TRIP mismatch\n");
                        return 1;
                    }
                    for (size_t i = 0; i < wl; i++) if (want[i] == 255) { escapes++; i++; }
                    checked++;
                    bytes += s.size();
                }
            }
        }
    }
    std::printf("device encoder model ok: %zu values, %zu bytes, %zu escapes\n", checked, bytes, escapes);
    return checked > 15000 && escapes > 1000 ? 0 : 2;
}
