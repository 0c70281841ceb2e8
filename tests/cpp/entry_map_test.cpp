// EntryMap (csrc/lc_internal.hpp) against std::unordered_map under a random stream of emplace / erase / find / find_many —
// the table behind lc_ctx::entries.  Built and run by tests/test_host_logic.py (CPU tier).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_map>

#include "../../liquid_cache_amd/csrc/lc_internal.hpp"

int main() {
    lc::EntryMap m;
    std::unordered_map<uint64_t, uint32_t> ref;
    std::mt19937_64 rng(7);
    auto id_of = [&](uint64_t k) { return (uint64_t(1) << 48) | ((k % 97) << 32) | ((k % 13) << 16) | (k % 5003); };  // ParquetArrayID-like
    for (int round = 0; round < 400000; round++) {
        const uint64_t id = id_of(rng());
        const int op = int(rng() % 10);
        if (op < 5) {
            lc::Entry e;
            e.len = uint32_t(rng());
            const uint32_t len = e.len;
            const bool ins = m.emplace(id, std::move(e)).second;
            const bool ins_ref = ref.emplace(id, len).second;
            if (ins != ins_ref) { std::printf("emplace mismatch\n"); return 1; }
        } else if (op < 8) {
            auto it = m.find(id);
            const bool have = it != m.end(), have_ref = ref.count(id) != 0;
            if (have != have_ref) { std::printf("find mismatch\n"); return 1; }
            if (have) {
                if (it->second.len != ref[id] || it->first != id) { std::printf("value mismatch\n"); return 1; }
                m.erase(it);
                ref.erase(id);
            }
        } else {
            if ((m.count(id) != 0) != (ref.count(id) != 0)) { std::printf("count mismatch\n"); return 1; }
        }
        if (m.size() != ref.size()) { std::printf("size mismatch %zu %zu\n", m.size(), ref.size()); return 1; }
        if (round % 50000 == 49999) {
            std::vector<uint64_t> ids;
            for (int k = 0; k < 5000; k++) ids.push_back(id_of(rng()));
            std::vector<lc::EntryMap::value_type*> out(ids.size());
            m.find_many(ids.data(), ids.size(), out.data());
            size_t visited = 0;
            const bool all = m.visit_many(ids.data(), ids.size(), [&](size_t k, lc::EntryMap::value_type* node) {
                if (k != visited++ || node != out[k]) { std::printf("visit_many mismatch at %zu\n", k); std::exit(1); }
                return true;
            });
            if (!all || visited != ids.size()) { std::printf("visit_many stopped early\n"); return 1; }
            size_t stop_at = 0;
            (void)m.visit_many(ids.data(), ids.size(), [&](size_t k, lc::EntryMap::value_type*) { stop_at = k; return k < 100; });
            if (stop_at != 100) { std::printf("visit_many did not stop where told (%zu)\n", stop_at); return 1; }
            for (size_t k = 0; k < ids.size(); k++) {
                const bool have_ref = ref.count(ids[k]) != 0;
                if ((out[k] != nullptr) != have_ref || (out[k] && (out[k]->first != ids[k] || out[k]->second.len != ref[ids[k]]))) {
                    std::printf("find_many mismatch\n");
                    return 1;
                }
            }
        }
    }
    std::printf("entry map ok (%zu entries at the end)\n", m.size());
    return 0;
}
