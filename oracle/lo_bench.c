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
LO_EXPORT int64_t lo_bench_eval_batches(size_t n, const uint8_t* const* blobs, const size_t* lens,
                                        const lo_symtab* const* symtabs, int op, int lit_tag, const void* lit,
                                        size_t lit_len, int threads) {
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
            total += popcount_and(ov, ovalid, k, nullable);
        }
        free(ov);
        free(ovalid);
    }
    return failed ? -1 : total;
}
