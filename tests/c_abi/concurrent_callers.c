/* Concurrent callers at the drop-in boundary ("thread-safe and re-entrant; per-thread streams; no global
 * device sync inside calls").  The reference's read path is driven from `target_partitions` tokio workers at once
 * (datafusion/src/reader/runtime/liquid_cache_reader.rs:297-391; the cache index is lock-free, core/src/cache/index.rs:30-34).
 *
 * Eight worker threads, each with its own stream (lc_stream_create) and its own column of the table, loop over
 *   lc_eval_predicate (per entry, with a selection)        -- the per-batch call of the reference's reader
 *   lc_eval_predicate_batch (all entries of the column)
 *   lc_scan_eval on the thread's stream + read-back
 *   lc_get_with_selection (per entry)
 *   whole-column queries in the reference's call shape (round 6): lc_eval_predicate_row_groups (ids in, per-row-group counts
 *   out: the scan behind it comes from the context's scan cache) and a scan created per query — lc_scan_create over the id
 *   list, COUNT(*) without a mask, lc_scan_destroy — while other threads do the same on their columns
 * while a ninth thread keeps staging, re-staging and evicting entries of a scratch column.  Every answer must equal the
 * answer the same call gave single-threaded before the threads started, and the wall time of the eight threads together
 * must stay below twice the time one of them needs alone.
 * Exit code 0 + "concurrent callers ok".  Built and run by tests/test_gpu_round4.py. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "liquid_cache_amd.h"

#define N_COLS 8
#define N_BATCHES 12
#define ROWS 8192
#define REPS 6

static void release_noop_array(struct ArrowArray* a) { a->release = NULL; }
static void release_noop_schema(struct ArrowSchema* s) { s->release = NULL; }
static uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static uint64_t entry_id(int col, int batch) { return ((uint64_t)1 << 48) | ((uint64_t)col << 16) | (uint64_t)batch; }
static uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

static lc_ctx* ctx;
static int is_string_col(int col) { return col & 1; }
/* diagnosis knobs (environment): CC_THREADS (default 8), CC_NO_CHURN, CC_MODE bit mask of the call kinds a pass makes
 * (1 lc_eval_predicate, 2 lc_get_with_selection, 4 lc_eval_predicate_batch, 8 lc_scan_eval, 16 whole-column queries; default all) */
static int g_threads = N_COLS, g_mode = 31, g_no_churn = 0;

/* int64 column: pseudo-random values below 2^20; string column: URL-like values, some with "google" */
static int stage_batch(int col, int batch, uint64_t id, uint64_t salt) {
    struct ArrowArray arr;
    struct ArrowSchema schema;
    memset(&arr, 0, sizeof(arr));
    memset(&schema, 0, sizeof(schema));
    arr.length = ROWS;
    arr.release = release_noop_array;
    schema.name = "";
    schema.flags = 2;
    schema.release = release_noop_schema;
    int rc;
    if (!is_string_col(col)) {
        int64_t* v = (int64_t*)malloc(sizeof(int64_t) * ROWS);
        for (int i = 0; i < ROWS; i++) v[i] = (int64_t)(mix(salt + (uint64_t)col * 1000003u + (uint64_t)batch * 8209u + (uint64_t)i) & 0xFFFFF);
        const void* buffers[2] = {NULL, v};
        arr.n_buffers = 2;
        arr.buffers = buffers;
        schema.format = "l";
        rc = lc_insert_arrow(ctx, id, &arr, &schema, LC_HINT_NONE, 0);
        free(v);
    } else {
        int32_t* offs = (int32_t*)malloc(sizeof(int32_t) * (ROWS + 1));
        char* data = (char*)malloc((size_t)ROWS * 64);
        int32_t o = 0;
        for (int i = 0; i < ROWS; i++) {
            const uint64_t h = mix(salt + (uint64_t)col * 7919u + (uint64_t)batch * 104729u + (uint64_t)(i % 1500));
            offs[i] = o;
            o += snprintf(data + o, 64, "http://%s.example%u.ru/%s/%u", (h & 63) == 0 ? "google" : ((h & 7) == 1 ? "mail" : "site"),
                          (unsigned)(h >> 8) % 97u, (h & 0x300) ? "search" : "catalog", (unsigned)(h >> 20) % 1000u);
        }
        offs[ROWS] = o;
        const void* buffers[3] = {NULL, offs, data};
        arr.n_buffers = 3;
        arr.buffers = buffers;
        schema.format = "u";
        rc = lc_insert_arrow(ctx, id, &arr, &schema, LC_HINT_SUBSTRING_SEARCH, ((uint64_t)9 << 32) | (uint64_t)col);
        free(offs);
        free(data);
    }
    return rc;
}

static void predicate_for(int col, lc_predicate* p, int64_t* lit_store) {
    if (!is_string_col(col)) {
        *lit_store = 0x7FFFF;
        p->op = LC_OP_GT; p->lit_tag = LC_LIT_I64; p->lit = lit_store; p->lit_len = 8;
    } else {
        p->op = LC_OP_LIKE; p->lit_tag = LC_LIT_BYTES; p->lit = "%google%"; p->lit_len = 8;
    }
}

/* one pass of a worker over its column; returns a digest of every answer */
static int column_pass(int col, void* stream, lc_scan* scan, void* d_mask, uint64_t* digest) {
    lc_predicate pred;
    int64_t lit;
    predicate_for(col, &pred, &lit);
    uint64_t h = 1469598103934665603ULL;
    uint8_t sel[ROWS / 8], out_v[ROWS / 8 + 8], out_n[ROWS / 8 + 8];
    for (int b = 0; b < N_BATCHES; b++) {
        for (int i = 0; i < ROWS / 8; i++) sel[i] = (uint8_t)(0x5A ^ (i * 37 + b));
        uint32_t len = 0;
        int32_t nullable = 0;
        memset(out_v, 0, sizeof(out_v));
        if (g_mode & 1) {
            if (lc_eval_predicate(ctx, entry_id(col, b), &pred, (b & 1) ? sel : NULL, out_v, out_n, &len, &nullable) != LC_OK) return 1;
            h = fnv(h, &len, 4);
            h = fnv(h, out_v, (len + 7) / 8);
        }
        if (!(g_mode & 2)) continue;
        struct ArrowArray got;
        struct ArrowSchema gs;
        if (lc_get_with_selection(ctx, entry_id(col, b), sel, &got, &gs) != LC_OK) return 2;
        h = fnv(h, &got.length, 8);
        if (!is_string_col(col)) h = fnv(h, got.buffers[1], (size_t)got.length * 8);
        else {
            const int32_t* o = (const int32_t*)got.buffers[1];
            h = fnv(h, o, (size_t)(got.length + 1) * 4);
            h = fnv(h, got.buffers[2], (size_t)o[got.length]);
        }
        got.release(&got);
        gs.release(&gs);
    }
    if (g_mode & 4) {   /* the batch call over the whole column */
        uint64_t ids[N_BATCHES];
        uint8_t* ov[N_BATCHES];
        uint32_t lens[N_BATCHES];
        lc_status sts[N_BATCHES];
        static __thread uint8_t bufs[N_BATCHES][ROWS / 8 + 8];
        for (int b = 0; b < N_BATCHES; b++) { ids[b] = entry_id(col, b); ov[b] = bufs[b]; memset(bufs[b], 0, sizeof(bufs[b])); }
        if (lc_eval_predicate_batch(ctx, N_BATCHES, ids, &pred, NULL, ov, NULL, lens, NULL, sts) != LC_OK) return 3;
        for (int b = 0; b < N_BATCHES; b++) { h = fnv(h, &lens[b], 4); h = fnv(h, bufs[b], (lens[b] + 7) / 8); }
    }
    if (g_mode & 8) {   /* the scan-level call on this thread's stream */
        const uint64_t words = lc_scan_mask_words(scan);
        static __thread uint64_t host_mask[N_BATCHES * ROWS / 64];
        if (lc_scan_eval(ctx, scan, &pred, NULL, d_mask, NULL, stream) != LC_OK) return 4;
        if (lc_device_to_host(ctx, host_mask, d_mask, words * 8, stream) != LC_OK) return 5;
        if (lc_stream_synchronize(ctx, stream) != LC_OK) return 6;
        h = fnv(h, host_mask, words * 8);
    }
    if (g_mode & 16) {  /* whole-column queries the way a reader that holds no scan objects asks them */
        uint64_t ids[N_BATCHES], counts[4], total = 0, total2 = 0;
        const uint32_t ends[4] = {N_BATCHES / 4, N_BATCHES / 2, 3 * N_BATCHES / 4, N_BATCHES};
        for (int b = 0; b < N_BATCHES; b++) ids[b] = entry_id(col, b);
        if (lc_eval_predicate_row_groups(ctx, N_BATCHES, ids, 4, ends, &pred, 1, counts, NULL, 0, &total) != LC_OK) return 7;
        if (counts[0] + counts[1] + counts[2] + counts[3] != total) return 8;
        h = fnv(h, counts, sizeof(counts));
        lc_scan* q = NULL;  /* a scan per query: the context hands the kept one back */
        if (lc_scan_create(ctx, N_BATCHES, ids, &q) != LC_OK) return 9;
        if (lc_scan_eval_count(ctx, q, &pred, 1, NULL, NULL, NULL, d_mask, stream) != LC_OK) return 10;
        if (lc_device_to_host(ctx, &total2, d_mask, 8, stream) != LC_OK || lc_stream_synchronize(ctx, stream) != LC_OK) return 11;
        lc_scan_destroy(q);
        if (total2 != total) return 12;
        h = fnv(h, &total2, 8);
    }
    *digest = h;
    return 0;
}

static pthread_barrier_t start_line;  /* the workers' timed loops start together, after their streams / scans exist */
static int use_barrier = 0;
struct worker {
    int col, reps, rc;
    uint64_t want;
    double seconds;
};
static void* worker_main(void* arg) {
    struct worker* w = (struct worker*)arg;
    void* stream = NULL;
    lc_scan* scan = NULL;
    void* d_mask = NULL;
    uint64_t ids[N_BATCHES];
    for (int b = 0; b < N_BATCHES; b++) ids[b] = entry_id(w->col, b);
    if (lc_stream_create(ctx, &stream) != LC_OK || lc_scan_create(ctx, N_BATCHES, ids, &scan) != LC_OK ||
        lc_device_alloc(ctx, lc_scan_mask_words(scan) * 8, &d_mask) != LC_OK) w->rc = 100;
    {
        /* warm-up of the concurrent configuration with this thread's own stream and scan (the scratch pools grow to eight
         * callers' worth, the stream gets its hardware queue: first-use costs of milliseconds that a running server's
         * long-lived workers do not see), then everybody starts the timed loop together */
        for (int r = 0; r < 2 && w->rc == 0; r++) {
            uint64_t got = 0;
            const int rc = column_pass(w->col, stream, scan, d_mask, &got);
            if (rc) w->rc = rc;
            else if (got != w->want) w->rc = 50;
        }
        if (use_barrier) pthread_barrier_wait(&start_line);
    }
    const double t0 = now_s();
    for (int r = 0; r < w->reps && w->rc == 0; r++) {
        uint64_t got = 0;
        const int rc = column_pass(w->col, stream, scan, d_mask, &got);
        if (rc) w->rc = rc;
        else if (got != w->want) w->rc = 50;
    }
    w->seconds = now_s() - t0;
    lc_stream_synchronize(ctx, stream);
    lc_scan_destroy(scan);
    lc_device_free(ctx, d_mask);
    lc_stream_destroy(ctx, stream);
    return NULL;
}

static volatile int churn_stop = 0;
static int churn_rc = 0, churn_rounds = 0;
static void* churn_main(void* arg) {
    (void)arg;
    /* a scratch column (ids of column N_COLS and N_COLS + 1): stage, re-stage under the same id, evict, again */
    for (uint64_t round = 0; !churn_stop; round++) {
        for (int b = 0; b < 4 && churn_rc == 0; b++) {
            if (stage_batch((int)(round & 1), b, entry_id(N_COLS + (int)(round & 1), b), round) != LC_OK) churn_rc = 1;
        }
        uint64_t ids[4];
        for (int b = 0; b < 4; b++) ids[b] = entry_id(N_COLS + (int)(round & 1), b);
        if ((round % 3) == 2 && lc_evict(ctx, 4, ids) != LC_OK) churn_rc = 2;
        churn_rounds++;
    }
    return NULL;
}

int main(void) {
    if (getenv("CC_THREADS")) g_threads = atoi(getenv("CC_THREADS"));
    if (getenv("CC_MODE")) g_mode = atoi(getenv("CC_MODE"));
    if (getenv("CC_NO_CHURN")) g_no_churn = 1;
    if (g_threads < 1 || g_threads > N_COLS) g_threads = N_COLS;
    if (lc_ctx_create(NULL, 1, 0, &ctx) != LC_OK) {
        fprintf(stderr, "lc_ctx_create: %s\n", lc_last_error(NULL));
        return 2;
    }
    for (int c = 0; c < N_COLS; c++)
        for (int b = 0; b < N_BATCHES; b++)
            if (stage_batch(c, b, entry_id(c, b), 0) != LC_OK) {
                fprintf(stderr, "staging failed: %s\n", lc_last_error(ctx));
                return 1;
            }
    /* single-threaded answers (twice: the second pass runs from cached scans and plans and must agree) */
    struct worker w[N_COLS];
    for (int c = 0; c < N_COLS; c++) {
        memset(&w[c], 0, sizeof(w[c]));
        w[c].col = c;
    }
    for (int c = 0; c < N_COLS; c++) {
        void* stream = NULL;
        lc_scan* scan = NULL;
        void* d_mask = NULL;
        uint64_t ids[N_BATCHES], d1 = 0, d2 = 0;
        for (int b = 0; b < N_BATCHES; b++) ids[b] = entry_id(c, b);
        if (lc_stream_create(ctx, &stream) != LC_OK || lc_scan_create(ctx, N_BATCHES, ids, &scan) != LC_OK ||
            lc_device_alloc(ctx, lc_scan_mask_words(scan) * 8, &d_mask) != LC_OK) return 1;
        if (column_pass(c, stream, scan, d_mask, &d1) || column_pass(c, stream, scan, d_mask, &d2) || d1 != d2) {
            fprintf(stderr, "single-threaded passes of column %d disagree (%s)\n", c, lc_last_error(ctx));
            return 1;
        }
        w[c].want = d1;
        lc_scan_destroy(scan);
        lc_device_free(ctx, d_mask);
        lc_stream_destroy(ctx, stream);
    }
    /* one thread alone: the slowest column's time for REPS passes */
    double alone = 0;
    for (int c = 0; c < 2; c++) {  /* one int column, one string column */
        w[c].reps = REPS; w[c].rc = 0;
        worker_main(&w[c]);
        if (w[c].rc) { fprintf(stderr, "alone: column %d rc %d (%s)\n", c, w[c].rc, lc_last_error(ctx)); return 1; }
        if (w[c].seconds > alone) alone = w[c].seconds;
    }
    /* eight threads at once + the churn thread */
    pthread_t th[N_COLS], churn;
    if (!g_no_churn) pthread_create(&churn, NULL, churn_main, NULL);
    pthread_barrier_init(&start_line, NULL, (unsigned)g_threads);
    use_barrier = 1;
    for (int c = 0; c < g_threads; c++) {
        w[c].reps = REPS; w[c].rc = 0;
        pthread_create(&th[c], NULL, worker_main, &w[c]);
    }
    for (int c = 0; c < g_threads; c++) pthread_join(th[c], NULL);
    double together = 0;  /* the slowest worker's timed loop (they start together; set-up and warm-up are outside) */
    for (int c = 0; c < g_threads; c++)
        if (w[c].seconds > together) together = w[c].seconds;
    churn_stop = 1;
    if (!g_no_churn) pthread_join(churn, NULL);
    for (int c = 0; c < N_COLS; c++)
        if (w[c].rc) { fprintf(stderr, "concurrent: column %d rc %d (50 = answer differs)\n", c, w[c].rc); return 1; }
    if (churn_rc) { fprintf(stderr, "churn thread failed: %d (%s)\n", churn_rc, lc_last_error(ctx)); return 1; }
    printf("one thread alone %.1f ms, %d threads together %.1f ms (x%.2f), %d stage/evict rounds beside them\n", alone * 1e3,
           g_threads, together * 1e3, together / alone, churn_rounds);
    if (together >= 2.0 * alone) { fprintf(stderr, "eight concurrent callers took more than twice one caller\n"); return 3; }
    lc_ctx_destroy(ctx);
    printf("concurrent callers ok\n");
    return 0;
}
