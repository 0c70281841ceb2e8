/*
 * Bench / test aids, built into libliquid_cache_amd_bench.so (never into the product library).
 * Synthetic ClickBench-shaped column generators used by bench.py and the full-size tests.
 * They only produce Arrow-layout host buffers; everything downstream goes through the public API in
 * liquid_cache_amd.h (lc_insert_arrow / lc_scan_*).  There is no dataset access on the benchmark machines
 * (hits.parquet is download-only), hence generators that mimic the statistics measured on the
 * reference's examples/nano_hits.parquet: ~2,200 distinct URLs per 8192-row batch, mean length ~76 bytes,
 * Zipf-distributed repetition, ~40 % of distinct values passing the 32-bucket fingerprint of "google".
 */
#ifndef LIQUID_CACHE_AMD_BENCH_H
#define LIQUID_CACHE_AMD_BENCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_BENCH_API __attribute__((visibility("default")))

/* Fill one batch of a URL-like Utf8 column.  Deterministic in (seed, batch_index).
 *   offsets: rows + 1 int32; data: capacity data_cap bytes; returns the number of data bytes written,
 *   or 0 if data_cap is too small (rows * 512 is always enough).
 *   n_unique: distinct values to draw from in this batch; needle_ppm: parts-per-million of DISTINCT values that
 *   contain the token "google" (true matches of LIKE '%google%'). */
LC_BENCH_API size_t lc_synth_url_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique,
                                       uint32_t needle_ppm, int32_t* offsets, uint8_t* data, size_t data_cap);

/* Fill one batch of a ClickBench-"SearchPhrase"-shaped column: `empty_permille` of the rows are the empty string
 * (hits: ~87 %), the others draw a 1-5 word phrase from `n_unique` distinct ones.  Same output convention as
 * lc_synth_url_batch. */
LC_BENCH_API size_t lc_synth_phrase_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique,
                                          uint32_t empty_permille, int32_t* offsets, uint8_t* data, size_t data_cap);

/* Fill one batch of a ClickBench-"Title"-shaped column: 3-9 capitalised words plus " - <host>", Zipf repetition over
 * `n_unique` distinct titles; needle_ppm parts-per-million of the DISTINCT titles contain the token "Google". */
LC_BENCH_API size_t lc_synth_title_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, uint32_t n_unique,
                                         uint32_t needle_ppm, int32_t* offsets, uint8_t* data, size_t data_cap);

/* Fill `rows` 64-bit integers uniform in [base, base + 2^bit_width) (bit_width 1..64).
 * Deterministic in (seed, batch_index). */
LC_BENCH_API void lc_synth_int64_batch(uint64_t seed, uint64_t batch_index, uint32_t rows, int32_t bit_width,
                                       int64_t base, int64_t* out);

/* Profiling aid (scripts/pmc_calibrate.py): launches `iters` kernels that read exactly `bytes` bytes of a scratch buffer
 * with a given access shape — 4 / 8 / 16 = coalesced bytes per lane, 1008 = 8 unaligned bytes out of every 64-byte
 * sector — so that rocprofv3's FETCH_SIZE can be calibrated on the access patterns of the scan kernels (the counter's
 * unit is only documented for 16-byte coalesced reads).  `ctx` is an lc_ctx*. */
LC_BENCH_API int32_t lc_calibrate_read(void* ctx, uint64_t bytes, int32_t shape, int32_t iters);

/* Streaming-read probe: the average time (HIP events, microseconds) of a kernel that only reads `bytes` bytes once with
 * 16-byte loads on `grid_blocks` workgroups of 256 threads — back to back (hot: sizes below the 256 MiB Infinity Cache stay
 * resident) and with a 1 GiB flush read before every launch (L3-cold).  The measured ceiling the scan kernels' times are
 * read against. */
LC_BENCH_API int32_t lc_probe_stream_read(void* ctx, uint64_t bytes, int32_t iters, int32_t grid_blocks, double* out_hot_us,
                                          double* out_cold_us);

/* Kernel time of lc_scan_eval over a scan (`scan`: lc_scan*, `pred`: const lc_predicate*), HIP events recorded on `stream`
 * (the stream the kernels run on); the average milliseconds per evaluation over `iters` launches.  flush_bytes == 0: back
 * to back (hot: a column below the 256 MiB Infinity Cache stays resident); > 0: that many bytes of scratch are streamed
 * through the memory-side cache by a read-only kernel before every launch (>= 512 MiB defeats the Infinity Cache) — the
 * L3-cold kernel time of one evaluation.  Uses the public scan API only; the roofline figures of bench.py come from here. */
/* `iters` back-to-back calls of the hit-list byte gather between two events on `stream`: average milliseconds per call as
 * the device sees them.  Same arguments as the public gather entry point.  flags = 0: every call zeroes its byte counter (a
 * one-wave kernel in front of the gather: the stand-alone call); LC_HITS_COUNTERS_ZEROED: the counter is zeroed once in front
 * of the loop and the calls append behind each other in d_data (capacity_bytes must hold iters outputs): the gather kernel
 * alone, as it runs inside a pipeline that zeroed its counters up front. */
LC_BENCH_API int32_t lc_bench_gather_bytes_hits_timed(void* ctx, void* scan, const void* d_hits, const void* d_n_hits,
                                                      uint64_t capacity_rows, void* d_views, void* d_data, uint64_t capacity_bytes,
                                                      void* d_n_bytes, uint32_t flags, void* stream, int32_t iters,
                                                      float* out_avg_ms);
LC_BENCH_API int32_t lc_bench_eval_timed(void* ctx, void* scan, const void* pred, const void* d_selection, void* d_mask_out,
                                         void* d_counts_out, void* stream, int32_t iters, uint64_t flush_bytes,
                                         float* out_avg_ms);

/* Row-group-granular driver: the column is walked the way the reference's reader walks it — one evaluation per ROW GROUP
 * (the batches that share a ColumnAccessPath: liquid_stream.rs:358-430, liquid_cache_reader.rs:264-294), `threads` host
 * threads at once, each on a stream of its own (one lc_stream_create each), the row-group scans created once and kept.
 *   group_begin: n_groups + 1 indices into entry_ids (row group g = entry_ids[group_begin[g] .. group_begin[g + 1]))
 *   groups_per_scan: consecutive row groups one scan (one call) covers — 1 is the reference's granularity
 *   with_mask: 0 = COUNT(*) only (d_mask_out = NULL), 1 = the hit mask of every call is written as well
 * Every call is the public lc_scan_eval_count; a pass = every unit once; `passes` passes are timed after one untimed pass
 * (scan creation, scan-level indexes, plans: first_pass_s). */
typedef struct {
    double wall_s;        /* the timed passes: slowest thread */
    double first_pass_s;  /* scan creation + first evaluation of every unit: slowest thread */
    double total_s;       /* everything, threads started to joined */
    double call_us_mean;  /* host time inside one lc_scan_eval_count call (launch side), mean over the timed calls */
    uint64_t calls;       /* timed calls */
    uint64_t hits;        /* COUNT(*) of ONE pass summed over the units (== the whole-column scan's) */
    uint64_t units;
    uint32_t passes, threads;
} lc_rowgroup_stats;
LC_BENCH_API int32_t lc_bench_rowgroup_run(void* ctx, uint64_t n_groups, const uint64_t* group_begin, const uint64_t* entry_ids,
                                           const void* pred, int32_t threads, int32_t passes, int32_t with_mask,
                                           int32_t groups_per_scan, lc_rowgroup_stats* out);
/* MANY row groups per call (round 6): thread t owns a contiguous slice of the row groups (a reader's partition) and evaluates
 * all of them with ONE call per pass and per-row-group counts: mode 0 = the id-list call of the product ABI (ids in, host
 * counts out, the scan comes from the context's scan cache), mode 1 = the scan call (a kept scan + one stream wait).
 * call_us_mean is the whole call here (results on the host / stream drained); units = threads. */
LC_BENCH_API int32_t lc_bench_rowgroup_many(void* ctx, uint64_t n_groups, const uint64_t* group_begin, const uint64_t* entry_ids,
                                            const void* pred, int32_t threads, int32_t passes, int32_t mode, lc_rowgroup_stats* out);
/* The per-entry drop-in call (lc_eval_predicate, host buffers out, no selection) over `n` entries, `threads` callers at
 * once, `rounds` rounds after one warming round: *out_call_us = what one caller waits per call; *out_hits = set bits of
 * round 0 (== the scan's COUNT(*)). */
LC_BENCH_API int32_t lc_bench_entry_calls(void* ctx, uint64_t n, const uint64_t* entry_ids, const void* pred, int32_t threads,
                                          int32_t rounds, uint32_t rows_per_entry, double* out_call_us, uint64_t* out_hits);

/* Test aid (host only, no context): the inverted row lists lc_stage attaches to byte-view entries of substring-search
 * columns — u16 offsets[d + 1], then the valid rows grouped by dictionary key — for `n` (<= 65535) keys, an optional
 * LSB-first validity bitmap and a dictionary of `d` values.  Returns the number of u16 written to `out` (d + 1 + n + 32;
 * 0: bad arguments or `cap` too small). */
LC_BENCH_API size_t lc_debug_row_lists(const uint16_t* keys, const uint8_t* validity, uint32_t n, uint32_t d, uint16_t* out,
                                       size_t cap);

#ifdef __cplusplus
}
#endif
#endif
