// Row-group-granular driver (include/liquid_cache_amd_bench.h: lc_bench_rowgroup_run, lc_bench_entry_calls): walks a staged
// column the way the reference's reader does — one evaluation per ROW GROUP (the ~54 batches that share a ColumnAccessPath,
// src/datafusion/src/reader/runtime/liquid_stream.rs:358-430, liquid_cache_reader.rs:264-294), T host threads at once, each
// on a stream of its own, the row-group scans created once and kept — and times the per-entry drop-in call.  Bench
// infrastructure: plain C++ over the PUBLIC C ABI only, built into libliquid_cache_amd_bench.so.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/liquid_cache_amd.h"
#include "../../include/liquid_cache_amd_bench.h"

namespace {
using Clock = std::chrono::steady_clock;
double seconds(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }
}  // namespace

extern "C" {

int32_t lc_bench_rowgroup_run(void* ctx_, uint64_t n_groups, const uint64_t* group_begin, const uint64_t* entry_ids,
                              const void* pred_, int32_t threads, int32_t passes, int32_t with_mask, int32_t groups_per_scan,
                              lc_rowgroup_stats* out) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    const lc_predicate* pred = static_cast<const lc_predicate*>(pred_);
    if (!ctx || !group_begin || !entry_ids || !pred || !out || n_groups == 0 || threads <= 0 || passes <= 0 || groups_per_scan <= 0)
        return LC_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    // a "unit" = groups_per_scan consecutive row groups evaluated by one call (1: the reference's granularity; more: what a
    // reader that batches the row groups of its partition into one scan gets)
    const uint64_t n_units = (n_groups + uint64_t(groups_per_scan) - 1) / uint64_t(groups_per_scan);
    std::vector<lc_scan*> scans(n_units, nullptr);
    void* d_totals = nullptr;
    if (lc_device_alloc(ctx, n_units * 8, &d_totals) != LC_OK) return LC_ERR_OOM;
    std::atomic<int32_t> rc{LC_OK};
    std::atomic<uint64_t> call_ns{0}, calls{0};
    std::vector<double> first_s(size_t(threads), 0.0), wall_s(size_t(threads), 0.0);
    auto worker = [&](int t) {
        void* stream = nullptr;
        if (lc_stream_create(ctx, &stream) != LC_OK) { rc = LC_ERR_DEVICE; return; }
        uint64_t max_words = 1;
        const auto t_first = Clock::now();
        for (uint64_t u = uint64_t(t); u < n_units && rc == LC_OK; u += uint64_t(threads)) {
            const uint64_t g0 = u * uint64_t(groups_per_scan), g1 = std::min<uint64_t>(n_groups, g0 + uint64_t(groups_per_scan));
            const lc_status st = lc_scan_create(ctx, group_begin[g1] - group_begin[g0], entry_ids + group_begin[g0], &scans[u]);
            if (st != LC_OK) { rc = st; break; }
            max_words = std::max<uint64_t>(max_words, lc_scan_mask_words(scans[u]));
        }
        void* d_mask = nullptr;
        if (rc == LC_OK && with_mask && lc_device_alloc(ctx, max_words * 8, &d_mask) != LC_OK) rc = LC_ERR_OOM;
        auto pass = [&](bool timed) {
            for (uint64_t u = uint64_t(t); u < n_units && rc == LC_OK; u += uint64_t(threads)) {
                const auto a = Clock::now();
                const lc_status st = lc_scan_eval_count(ctx, scans[u], pred, 1, nullptr, d_mask, nullptr,
                                                        static_cast<uint8_t*>(d_totals) + u * 8, stream);
                if (st != LC_OK) { rc = st; break; }
                if (timed) {
                    call_ns += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - a).count());
                    calls++;
                }
            }
            if (lc_stream_synchronize(ctx, stream) != LC_OK) rc = LC_ERR_DEVICE;
        };
        pass(false);  // (automata, plans; the scan-level indexes are built off the path ...)
        for (uint64_t u = uint64_t(t); u < n_units && rc == LC_OK; u += uint64_t(threads)) (void)lc_scan_index_wait(scans[u]);
        pass(false);  // (... and the timed passes are the steady state: planned on the index that is in place)
        first_s[size_t(t)] = seconds(t_first, Clock::now());
        const auto t0 = Clock::now();
        for (int p = 0; p < passes && rc == LC_OK; p++) pass(true);
        wall_s[size_t(t)] = seconds(t0, Clock::now());
        if (d_mask) (void)lc_device_free(ctx, d_mask);
        // (a stream must outlive the scans used on it: scans first, then the stream)
        (void)lc_stream_synchronize(ctx, stream);
        for (uint64_t u = uint64_t(t); u < n_units; u += uint64_t(threads)) {
            if (scans[u]) lc_scan_destroy(scans[u]);
            scans[u] = nullptr;
        }
        (void)lc_stream_destroy(ctx, stream);
    };
    const auto t_all = Clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
    const double all_s = seconds(t_all, Clock::now());
    std::vector<uint64_t> totals(n_units, 0);
    if (rc == LC_OK && lc_device_to_host(ctx, totals.data(), d_totals, n_units * 8, nullptr) != LC_OK) rc = LC_ERR_DEVICE;
    (void)lc_device_free(ctx, d_totals);
    if (rc != LC_OK) return rc;
    double wall = 0, first = 0;
    for (int t = 0; t < threads; t++) { wall = std::max(wall, wall_s[size_t(t)]); first = std::max(first, first_s[size_t(t)]); }
    uint64_t rows = 0;
    for (uint64_t h : totals) out->hits += h;
    (void)rows;
    out->wall_s = wall;
    out->first_pass_s = first;
    out->total_s = all_s;
    out->calls = calls.load();
    out->call_us_mean = out->calls ? double(call_ns.load()) / 1e3 / double(out->calls) : 0.0;
    out->units = n_units;
    out->passes = uint32_t(passes);
    out->threads = uint32_t(threads);
    return LC_OK;
}

// MANY row groups per call (round 6): thread t owns the contiguous slice [t n / T, (t + 1) n / T) of the row groups — a
// reader's partition — and evaluates ALL of them with one call per pass, per-row-group counts out:
//   mode 0  lc_eval_predicate_row_groups: entry ids in, counts on the host out, no scan object in the caller's hands (the
//           context's scan cache finds the scan of the previous pass) — the reference's call shape at a device's granularity
//   mode 1  lc_scan_eval_count_groups on a scan the thread keeps, counts left on the device, one stream wait per pass
int32_t lc_bench_rowgroup_many(void* ctx_, uint64_t n_groups, const uint64_t* group_begin, const uint64_t* entry_ids,
                               const void* pred_, int32_t threads, int32_t passes, int32_t mode, lc_rowgroup_stats* out) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    const lc_predicate* pred = static_cast<const lc_predicate*>(pred_);
    if (!ctx || !group_begin || !entry_ids || !pred || !out || n_groups == 0 || threads <= 0 || passes <= 0 || mode < 0 || mode > 1)
        return LC_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    threads = int32_t(std::min<uint64_t>(uint64_t(threads), n_groups));
    std::atomic<int32_t> rc{LC_OK};
    std::atomic<uint64_t> hits{0}, call_ns{0}, calls{0};
    std::vector<double> first_s(size_t(threads), 0.0), wall_s(size_t(threads), 0.0);
    auto worker = [&](int t) {
        const uint64_t g0 = n_groups * uint64_t(t) / uint64_t(threads), g1 = n_groups * uint64_t(t + 1) / uint64_t(threads);
        const uint64_t e0 = group_begin[g0], n_e = group_begin[g1] - e0;
        std::vector<uint32_t> ends(g1 - g0);
        for (uint64_t g = g0; g < g1; g++) ends[g - g0] = uint32_t(group_begin[g + 1] - e0);
        std::vector<uint64_t> counts(g1 - g0, 0);
        void* stream = nullptr;
        lc_scan* scan = nullptr;
        void* d_counts = nullptr;
        if (mode == 1) {
            if (lc_stream_create(ctx, &stream) != LC_OK) { rc = LC_ERR_DEVICE; return; }
            if (lc_scan_create(ctx, n_e, entry_ids + e0, &scan) != LC_OK || lc_device_alloc(ctx, (g1 - g0) * 8, &d_counts) != LC_OK) {
                rc = LC_ERR_DEVICE;
                return;
            }
        }
        auto pass = [&](bool timed) -> uint64_t {
            const auto a = Clock::now();
            lc_status st;
            uint64_t total = 0;
            if (mode == 0) {
                st = lc_eval_predicate_row_groups(ctx, n_e, entry_ids + e0, uint32_t(ends.size()), ends.data(), pred, 1, counts.data(),
                                                  nullptr, 0, &total);
            } else {
                st = lc_scan_eval_count_groups(ctx, scan, pred, 1, nullptr, uint32_t(ends.size()), ends.data(), d_counts, nullptr, nullptr,
                                               nullptr, stream);
                if (st == LC_OK) st = lc_stream_synchronize(ctx, stream);
            }
            if (st != LC_OK) { rc = st; return 0; }
            if (timed) {
                call_ns += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - a).count());
                calls++;
            }
            return total;
        };
        const auto t_first = Clock::now();
        (void)pass(false);
        if (mode == 1 && rc == LC_OK) (void)lc_scan_index_wait(scan);
        if (mode == 0 && rc == LC_OK) {  // (steady state: the scan-level index is in place before the clock starts)
            lc_scan* sc = nullptr;
            if (lc_scan_create(ctx, n_e, entry_ids + e0, &sc) == LC_OK) {
                (void)lc_scan_index_wait(sc);
                lc_scan_destroy(sc);
            }
        }
        (void)pass(false);  // (re-planned on the scan-level index)
        first_s[size_t(t)] = seconds(t_first, Clock::now());
        const auto t0 = Clock::now();
        for (int p = 0; p < passes && rc == LC_OK; p++) (void)pass(true);
        wall_s[size_t(t)] = seconds(t0, Clock::now());
        if (rc == LC_OK) {
            if (mode == 1 && lc_device_to_host(ctx, counts.data(), d_counts, counts.size() * 8, stream) != LC_OK) rc = LC_ERR_DEVICE;
            uint64_t h = 0;
            for (uint64_t c : counts) h += c;
            hits += h;
        }
        if (d_counts) (void)lc_device_free(ctx, d_counts);
        if (scan) lc_scan_destroy(scan);
        if (stream) (void)lc_stream_destroy(ctx, stream);
    };
    const auto t_all = Clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
    if (rc != LC_OK) return rc;
    double wall = 0, first = 0;
    for (int t = 0; t < threads; t++) { wall = std::max(wall, wall_s[size_t(t)]); first = std::max(first, first_s[size_t(t)]); }
    out->wall_s = wall;
    out->first_pass_s = first;
    out->total_s = seconds(t_all, Clock::now());
    out->calls = calls.load();
    out->call_us_mean = out->calls ? double(call_ns.load()) / 1e3 / double(out->calls) : 0.0;
    out->hits = hits.load();
    out->units = uint64_t(threads);
    out->passes = uint32_t(passes);
    out->threads = uint32_t(threads);
    return LC_OK;
}

int32_t lc_bench_entry_calls(void* ctx_, uint64_t n, const uint64_t* entry_ids, const void* pred_, int32_t threads, int32_t rounds,
                             uint32_t rows_per_entry, double* out_call_us, uint64_t* out_hits) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    const lc_predicate* pred = static_cast<const lc_predicate*>(pred_);
    if (!ctx || !entry_ids || !pred || !out_call_us || n == 0 || threads <= 0 || rounds <= 0 || rows_per_entry == 0) return LC_ERR_INVALID;
    std::atomic<int32_t> rc{LC_OK};
    std::atomic<uint64_t> hits{0};
    std::vector<double> wall(size_t(threads), 0.0);
    auto worker = [&](int t) {
        const size_t bytes = (size_t(rows_per_entry) + 7) / 8 + 8;
        std::vector<uint8_t> values(bytes), validity(bytes);
        uint64_t h = 0;
        for (int r = -1; r < rounds && rc == LC_OK; r++) {  // (round -1 warms the cached one-entry scans)
            const auto a = Clock::now();
            for (uint64_t i = uint64_t(t); i < n; i += uint64_t(threads)) {
                uint32_t len = 0;
                int32_t nullable = 0;
                const lc_status st = lc_eval_predicate(ctx, entry_ids[i], pred, nullptr, values.data(), validity.data(), &len, &nullable);
                if (st != LC_OK) { rc = st; break; }
                if (r == 0)
                    for (uint32_t w = 0; w < (len + 7) / 8; w++) {
                        uint32_t bits = values[w] & (nullable ? validity[w] : 0xFFu);
                        if (w == len / 8) bits &= (1u << (len & 7u)) - 1u;  // (bits behind the result's length are unspecified)
                        h += uint64_t(__builtin_popcount(bits));
                    }
            }
            if (r >= 0) wall[size_t(t)] += seconds(a, Clock::now());
        }
        hits += h;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto& th : pool) th.join();
    if (rc != LC_OK) return rc;
    double w = 0;
    for (double x : wall) w = std::max(w, x);
    const double per_thread_calls = double((n + uint64_t(threads) - 1) / uint64_t(threads)) * double(rounds);
    *out_call_us = w / per_thread_calls * 1e6;  // what ONE caller waits per call while `threads` callers run
    if (out_hits) *out_hits = hits.load();
    return LC_OK;
}

}  // extern "C"
