/* CPU-baseline driver (test / benchmark infrastructure, like the rest of oracle/): evaluates one predicate over many
 * staged Liquid blobs with a native loop — optionally one batch per OpenMP task — so that the baseline bench.py reports
 * is the restated algorithm's speed, not the Python wrapper's.  Mirrors how DataFusion runs one partition per core
 * over batches (target_partitions), SURVEY.md §8(d) "CPU baseline beside it". */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "lo_byteview.h"
#include "lo_common.h"
#include "lo_primitive.h"

static int64_t popcount_and(const uint8_t* a, const uint8_t* b, int64_t bits, int use_b) {
    int64_t c = 0;
    for (int64_t i = 0; i < bits; i++) {
        int v = (a[i >> 3] >> (i & 7)) & 1;
        if (use_b) v &= (b[i >> 3] >> (i & 7)) & 1;
        c += v;
    }
    return c;
}

/* blobs[i]: Liquid bytes of batch i; symtabs[i]: its FSST symbol table (NULL for fixed-width batches).
 * threads <= 1: plain loop.  Returns the number of rows whose predicate value is true (and valid), or <0 on error. */
static int64_t eval_batches(size_t n, const uint8_t* const* blobs, const size_t* lens, const lo_symtab* const* symtabs, int op,
                            int lit_tag, const void* lit, size_t lit_len, int threads, const uint64_t* seg_offsets,
                            uint64_t* mask_out, uint32_t* counts_out);

LO_EXPORT int64_t lo_bench_eval_batches(size_t n, const uint8_t* const* blobs, const size_t* lens,
                                        const lo_symtab* const* symtabs, int op, int lit_tag, const void* lit,
                                        size_t lit_len, int threads) {
    return eval_batches(n, blobs, lens, symtabs, op, lit_tag, lit, lit_len, threads, NULL, NULL, NULL);
}

/* The same, also returning what a device scan returns: the hit mask (pred AND valid) in scan layout — batch i's bits start
 * at u64 word seg_offsets[i], bits past the batch's last row are zero — and the per-batch hit counts.  bench.py compares
 * both with the GPU's on the whole column (not only the COUNT(*) total). */
LO_EXPORT int64_t lo_bench_eval_batches_masks(size_t n, const uint8_t* const* blobs, const size_t* lens,
                                              const lo_symtab* const* symtabs, int op, int lit_tag, const void* lit,
                                              size_t lit_len, int threads, const uint64_t* seg_offsets, uint64_t* mask_out,
                                              uint32_t* counts_out) {
    return eval_batches(n, blobs, lens, symtabs, op, lit_tag, lit, lit_len, threads, seg_offsets, mask_out, counts_out);
}

static int64_t eval_batches(size_t n, const uint8_t* const* blobs, const size_t* lens, const lo_symtab* const* symtabs, int op,
                            int lit_tag, const void* lit, size_t lit_len, int threads, const uint64_t* seg_offsets,
                            uint64_t* mask_out, uint32_t* counts_out) {
    int64_t total = 0;
    int failed = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : total) if (threads > 1)
    {
        uint8_t* ov = (uint8_t*)malloc(65536 / 8 + 16);
        uint8_t* ovalid = (uint8_t*)malloc(65536 / 8 + 16);
#pragma omp for schedule(dynamic, 4)
        for (long i = 0; i < (long)n; i++) {
            int nullable = 0;
            int64_t k;
            memset(ov, 0, 65536 / 8 + 16);
            memset(ovalid, 0, 65536 / 8 + 16);
            if (symtabs && symtabs[i])
                k = lo_bv_eval_predicate(blobs[i], lens[i], symtabs[i], op, lit_tag, (const uint8_t*)lit, lit_len, NULL, ov,
                                         ovalid, &nullable);
            else
                k = lo_fixed_eval_predicate(blobs[i], lens[i], op, lit_tag, lit, NULL, ov, ovalid, &nullable);
            if (k < 0) {
#pragma omp atomic write
                failed = 1;
                continue;
            }
            const int64_t c = popcount_and(ov, ovalid, k, nullable);
            total += c;
            if (counts_out) counts_out[i] = (uint32_t)c;
            if (mask_out && seg_offsets) {
                /* no selection: k == rows of the batch; ov / ovalid are zero beyond bit k */
                const size_t words = (size_t)(seg_offsets[i + 1] - seg_offsets[i]);
                for (size_t w = 0; w < words; w++) {
                    uint64_t a, b = ~(uint64_t)0;
                    memcpy(&a, ov + 8 * w, 8);
                    if (nullable) memcpy(&b, ovalid + 8 * w, 8);
                    const int64_t left = k - (int64_t)(64 * w);
                    const uint64_t tail = left >= 64 ? ~(uint64_t)0 : (left <= 0 ? 0 : (((uint64_t)1 << left) - 1));
                    mask_out[seg_offsets[i] + w] = a & b & tail;
                }
            }
        }
        free(ov);
        free(ovalid);
    }
    return failed ? -1 : total;
}
