/* The reader's loop at the REFERENCE's granularity, from a plain C program (round 5): the reference evaluates a pushed-down
 * filter and gathers the projection per row group / per batch (src/datafusion/src/reader/runtime/liquid_stream.rs:358-430,
 * liquid_cache_reader.rs:264-391), from `target_partitions` workers at once.  Here: a table of two columns (a URL-like Utf8
 * column with the SubstringSearch hint, one FSST symbol table per row group; an Int64 column), N_RG row groups of N_BATCHES
 * batches.  Four worker threads, each on its own stream, take the row groups round robin; per row group
 *     WHERE url LIKE '%google%' AND num > LIT        (the filter)         SELECT url, num   (the projection)
 * runs in the SPARSE form — lc_scan_eval_hits (the selective conjunct first) -> lc_scan_filter_hits (the next conjunct on
 * the listed rows) -> lc_scan_gather_bytes_hits / lc_scan_gather_fixed_hits (one launch per projected column), all counters
 * zeroed by one memset — and in the MASK form (lc_scan_eval_filter + lc_scan_gather_fixed), and both must return exactly
 * the rows a plain C loop over the generated data selects, with their values.
 * Exit code 0 + "rowgroup reader ok".  Built and run by tests/test_gpu_round5.py. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "liquid_cache_amd.h"

#define N_RG 6
#define N_BATCHES 5
#define ROWS 8192
#define N_THREADS 4
#define CAP 4096
#define LIT 400000

static void release_noop_array(struct ArrowArray* a) { a->release = NULL; }
static void release_noop_schema(struct ArrowSchema* s) { s->release = NULL; }
static uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static lc_ctx* ctx;
static uint64_t url_id(int rg, int b) { return ((uint64_t)7 << 48) | ((uint64_t)rg << 32) | ((uint64_t)13 << 16) | (uint64_t)b; }
static uint64_t num_id(int rg, int b) { return ((uint64_t)7 << 48) | ((uint64_t)rg << 32) | ((uint64_t)14 << 16) | (uint64_t)b; }

/* row (rg, b, i): its url and its number, from a hash — the truth needs no copy of the table */
static int make_url(int rg, int b, int i, char* out) {
    const uint64_t h = mix((uint64_t)rg * 1000003u + (uint64_t)b * 8209u + (uint64_t)(i % 1700));
    if ((h & 0xFF) == 0xFF) return snprintf(out, 16, "%s", (h & 0x100) ? "" : "g");            /* empty and 1-byte values */
    return snprintf(out, 96, "http://%s.example%u.ru/%s/%u%s", (h & 127) == 0 ? "google" : ((h & 7) == 1 ? "mail" : "site"),
                    (unsigned)(h >> 8) % 97u, (h & 0x300) ? "search" : "catalog/items/list", (unsigned)(h >> 20) % 1000u,
                    (h & 0x7C00) == 0x7C00 ? "?ref=www.google.com" : "");
}
static int64_t make_num(int rg, int b, int i) { return (int64_t)(mix(77 + (uint64_t)rg * 31u + (uint64_t)b * 131u + (uint64_t)i) % 1000000u); }
static int url_null(int rg, int b, int i) { return (mix(5 + (uint64_t)rg * 3u + (uint64_t)b * 7u + (uint64_t)i) % 53u) == 0; }

static int stage_row_group(int rg) {
    uint64_t ids[N_BATCHES], paths[N_BATCHES];
    int32_t hints[N_BATCHES];
    struct ArrowArray arr[N_BATCHES];
    struct ArrowSchema sch[N_BATCHES];
    const struct ArrowArray* ap[N_BATCHES];
    const struct ArrowSchema* sp[N_BATCHES];
    const void* bufs[N_BATCHES][3];
    int32_t* offs[N_BATCHES];
    char* data[N_BATCHES];
    uint8_t* valid[N_BATCHES];
    int64_t* nums[N_BATCHES];
    int rc = 0;
    for (int b = 0; b < N_BATCHES; b++) {
        offs[b] = (int32_t*)malloc(sizeof(int32_t) * (ROWS + 1));
        data[b] = (char*)malloc((size_t)ROWS * 128);
        valid[b] = (uint8_t*)calloc(ROWS / 8, 1);
        int32_t o = 0, nulls = 0;
        for (int i = 0; i < ROWS; i++) {
            offs[b][i] = o;
            if (url_null(rg, b, i)) { nulls++; continue; }
            valid[b][i >> 3] |= (uint8_t)(1u << (i & 7));
            o += make_url(rg, b, i, data[b] + o);
        }
        offs[b][ROWS] = o;
        memset(&arr[b], 0, sizeof(arr[b]));
        memset(&sch[b], 0, sizeof(sch[b]));
        bufs[b][0] = valid[b]; bufs[b][1] = offs[b]; bufs[b][2] = data[b];
        arr[b].length = ROWS; arr[b].null_count = nulls; arr[b].n_buffers = 3; arr[b].buffers = bufs[b]; arr[b].release = release_noop_array;
        sch[b].format = "u"; sch[b].name = ""; sch[b].flags = 2; sch[b].release = release_noop_schema;
        ap[b] = &arr[b]; sp[b] = &sch[b];
        ids[b] = url_id(rg, b); paths[b] = ((uint64_t)7 << 48) | ((uint64_t)rg << 32) | ((uint64_t)13 << 16); hints[b] = LC_HINT_SUBSTRING_SEARCH;
    }
    rc |= lc_insert_arrow_batch(ctx, N_BATCHES, ids, ap, sp, hints, paths);
    for (int b = 0; b < N_BATCHES; b++) {
        nums[b] = (int64_t*)malloc(sizeof(int64_t) * ROWS);
        for (int i = 0; i < ROWS; i++) nums[b][i] = make_num(rg, b, i);
        bufs[b][0] = NULL; bufs[b][1] = nums[b];
        arr[b].null_count = 0; arr[b].n_buffers = 2;
        sch[b].format = "l";
        ids[b] = num_id(rg, b); hints[b] = LC_HINT_NONE; paths[b] = 0;
    }
    rc |= lc_insert_arrow_batch(ctx, N_BATCHES, ids, ap, sp, hints, paths);
    for (int b = 0; b < N_BATCHES; b++) { free(offs[b]); free(data[b]); free(valid[b]); free(nums[b]); }
    return rc;
}

typedef struct { int t; int rc; uint64_t rows_out; } worker_t;

static int cmp_u64(const void* a, const void* b) { const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

static void* worker(void* arg) {
    worker_t* w = (worker_t*)arg;
    void* stream = NULL;
    void *d_hits1, *d_hits2, *d_ctr, *d_views, *d_data, *d_vals, *d_valid, *d_ma, *d_mb, *d_total, *d_ro, *d_mvals;
    w->rc = 1;
    if (lc_stream_create(ctx, &stream) != LC_OK) return NULL;
    if (lc_device_alloc(ctx, CAP * 8, &d_hits1) || lc_device_alloc(ctx, CAP * 8, &d_hits2) || lc_device_alloc(ctx, 32, &d_ctr) ||
        lc_device_alloc(ctx, CAP * 16, &d_views) || lc_device_alloc(ctx, CAP * 160, &d_data) || lc_device_alloc(ctx, CAP * 8, &d_vals) ||
        lc_device_alloc(ctx, CAP, &d_valid) || lc_device_alloc(ctx, N_BATCHES * ROWS / 8, &d_ma) || lc_device_alloc(ctx, N_BATCHES * ROWS / 8, &d_mb) ||
        lc_device_alloc(ctx, 8, &d_total) || lc_device_alloc(ctx, (N_BATCHES + 1) * 8, &d_ro) || lc_device_alloc(ctx, CAP * 8, &d_mvals))
        return NULL;
    int64_t lit = LIT;
    lc_predicate like = {LC_OP_LIKE, LC_LIT_BYTES, "%google%", 8};
    lc_predicate gt = {LC_OP_GT, LC_LIT_I64, &lit, 8};
    uint64_t* hits = (uint64_t*)malloc(CAP * 8);
    uint8_t* views = (uint8_t*)malloc(CAP * 16);
    uint8_t* data = (uint8_t*)malloc(CAP * 160);
    int64_t* vals = (int64_t*)malloc(CAP * 8);
    int64_t* mvals = (int64_t*)malloc(CAP * 8);
    uint8_t* valid = (uint8_t*)malloc(CAP);
    uint64_t want[CAP];
    for (int rg = w->t; rg < N_RG; rg += N_THREADS) {
        uint64_t uids[N_BATCHES], nids[N_BATCHES];
        for (int b = 0; b < N_BATCHES; b++) { uids[b] = url_id(rg, b); nids[b] = num_id(rg, b); }
        lc_scan *us = NULL, *ns = NULL;
        if (lc_scan_create(ctx, N_BATCHES, uids, &us) != LC_OK || lc_scan_create(ctx, N_BATCHES, nids, &ns) != LC_OK) { w->rc = 2; return NULL; }
        /* truth: the rows a plain loop selects */
        uint64_t n_want = 0;
        char tmp[128];
        for (int b = 0; b < N_BATCHES; b++)
            for (int i = 0; i < ROWS; i++) {
                if (url_null(rg, b, i)) continue;
                make_url(rg, b, i, tmp);
                if (strstr(tmp, "google") && make_num(rg, b, i) > LIT && n_want < CAP) want[n_want++] = ((uint64_t)b << 32) | (uint64_t)i;
            }
        for (int rep = 0; rep < 3; rep++) {
            /* ---- sparse form: four counters, one memset.  The second repetition takes the URL projection with a SLOTTED data
             * buffer (row r's bytes at r * LC_GATHER_SLOT_BYTES, longer values behind CAP slots): same views, checked the same way */
            const int slotted = rep == 1;
            uint8_t* c = (uint8_t*)d_ctr;
            if (lc_device_memset(ctx, d_ctr, 0, 32, stream) != LC_OK) { w->rc = 3; return NULL; }
            if (lc_scan_eval_hits(ctx, us, &like, 1, NULL, d_hits1, CAP, c, NULL, NULL, NULL, LC_HITS_COUNTERS_ZEROED, stream) != LC_OK ||
                lc_scan_filter_hits(ctx, ns, &gt, d_hits1, c, CAP, d_hits2, CAP, c + 8, LC_HITS_COUNTERS_ZEROED, stream) != LC_OK ||
                lc_scan_gather_bytes_hits(ctx, us, d_hits2, c + 8, CAP, d_views, d_valid, d_data, CAP * 160, c + 16,
                                          LC_HITS_COUNTERS_ZEROED | (slotted ? LC_GATHER_SLOTTED : 0u), stream) != LC_OK ||
                lc_scan_gather_fixed_hits(ctx, ns, d_hits2, c + 8, CAP, d_vals, NULL, 0, stream) != LC_OK) { w->rc = 4; return NULL; }
            uint64_t ctr[4];
            if (lc_device_to_host(ctx, ctr, d_ctr, 32, stream) != LC_OK) { w->rc = 5; return NULL; }
            const uint64_t k = ctr[1];
            const uint64_t data_used = slotted ? (uint64_t)CAP * LC_GATHER_SLOT_BYTES + ctr[2] : ctr[2];
            if (k != n_want || ctr[0] < k || ctr[0] > CAP || data_used > CAP * 160) { fprintf(stderr, "rg %d: %llu rows, want %llu (like %llu)\n", rg, (unsigned long long)k, (unsigned long long)n_want, (unsigned long long)ctr[0]); w->rc = 6; return NULL; }
            if (lc_device_to_host(ctx, hits, d_hits2, k * 8, stream) || lc_device_to_host(ctx, views, d_views, k * 16, stream) ||
                lc_device_to_host(ctx, data, d_data, data_used ? data_used : 1, stream) || lc_device_to_host(ctx, vals, d_vals, k * 8, stream) ||
                lc_device_to_host(ctx, valid, d_valid, k ? k : 1, stream)) { w->rc = 7; return NULL; }
            uint64_t sorted[CAP];
            memcpy(sorted, hits, k * 8);
            qsort(sorted, k, 8, cmp_u64);
            if (memcmp(sorted, want, k * 8) != 0) { w->rc = 8; return NULL; }
            for (uint64_t r = 0; r < k; r++) {  /* row r of both projections is record r of the list */
                const int b = (int)(hits[r] >> 32), i = (int)(uint32_t)hits[r];
                const int n = make_url(rg, b, i, tmp);
                int32_t vl, vo;
                memcpy(&vl, views + r * 16, 4);
                memcpy(&vo, views + r * 16 + 12, 4);
                const uint8_t* p = vl <= 12 ? views + r * 16 + 4 : data + vo;
                if (slotted && vl > 12 && (vl <= (int32_t)LC_GATHER_SLOT_BYTES ? (uint64_t)vo != r * LC_GATHER_SLOT_BYTES
                                                                                  : (uint64_t)vo < (uint64_t)CAP * LC_GATHER_SLOT_BYTES)) {
                    fprintf(stderr, "rg %d row %llu: slotted value outside its place\n", rg, (unsigned long long)r); w->rc = 15; return NULL; }
                if (!valid[r] || vl != n || memcmp(p, tmp, (size_t)n) != 0 || (vl > 12 && memcmp(views + r * 16 + 4, tmp, 4) != 0) ||
                    vals[r] != make_num(rg, b, i)) { fprintf(stderr, "rg %d row %llu: value mismatch\n", rg, (unsigned long long)r); w->rc = 9; return NULL; }
            }
            /* ---- mask form of the same filter: the reference's order of calls, one lc_scan_eval_filter */
            lc_scan* s1[1] = {us};
            lc_scan* s2[1] = {ns};
            lc_filter_step steps[2] = {{LC_STEP_AND, 1, s1, &like}, {LC_STEP_AND, 1, s2, &gt}};
            void* final_mask = NULL;
            if (lc_scan_eval_filter(ctx, 2, steps, NULL, d_ma, d_mb, NULL, d_total, &final_mask, stream) != LC_OK ||
                lc_scan_gather_fixed(ctx, ns, final_mask, d_mvals, CAP * 8, d_ro, stream) != LC_OK) { w->rc = 10; return NULL; }
            uint64_t total = 0;
            if (lc_device_to_host(ctx, &total, d_total, 8, stream) || lc_device_to_host(ctx, mvals, d_mvals, k * 8, stream)) { w->rc = 11; return NULL; }
            if (total != k) { w->rc = 12; return NULL; }
            for (uint64_t r = 0; r < k; r++)  /* the mask form returns the rows in row order */
                if (mvals[r] != make_num(rg, (int)(want[r] >> 32), (int)(uint32_t)want[r])) { w->rc = 13; return NULL; }
            w->rows_out += k;
        }
        if (lc_stream_synchronize(ctx, stream) != LC_OK) { w->rc = 14; return NULL; }
        lc_scan_destroy(us);
        lc_scan_destroy(ns);
    }
    free(hits); free(views); free(data); free(vals); free(mvals); free(valid);
    lc_device_free(ctx, d_hits1); lc_device_free(ctx, d_hits2); lc_device_free(ctx, d_ctr); lc_device_free(ctx, d_views);
    lc_device_free(ctx, d_data); lc_device_free(ctx, d_vals); lc_device_free(ctx, d_valid); lc_device_free(ctx, d_ma);
    lc_device_free(ctx, d_mb); lc_device_free(ctx, d_total); lc_device_free(ctx, d_ro); lc_device_free(ctx, d_mvals);
    lc_stream_destroy(ctx, stream);
    w->rc = 0;
    return NULL;
}

int main(void) {
    if (lc_ctx_create(NULL, 1, 0, &ctx) != LC_OK) { fprintf(stderr, "no device: %s\n", lc_last_error(NULL)); return 1; }
    lc_ctx_set_option(ctx, LC_OPT_LIKE_PIPELINE_MIN_ENTRIES, 1);  /* five-entry scans take the planned LIKE path as well */
    for (int rg = 0; rg < N_RG; rg++)
        if (stage_row_group(rg) != LC_OK) { fprintf(stderr, "staging failed: %s\n", lc_last_error(ctx)); return 1; }
    pthread_t th[N_THREADS];
    worker_t ws[N_THREADS];
    for (int t = 0; t < N_THREADS; t++) { ws[t].t = t; ws[t].rc = -1; ws[t].rows_out = 0; pthread_create(&th[t], NULL, worker, &ws[t]); }
    uint64_t rows = 0;
    int bad = 0;
    for (int t = 0; t < N_THREADS; t++) {
        pthread_join(th[t], NULL);
        if (ws[t].rc != 0) { fprintf(stderr, "worker %d failed at step %d: %s\n", t, ws[t].rc, lc_last_error(ctx)); bad = 1; }
        rows += ws[t].rows_out;
    }
    lc_ctx_destroy(ctx);
    if (bad || rows == 0) return 1;
    printf("rowgroup reader ok: %llu rows returned (both forms, every value checked)\n", (unsigned long long)rows);
    return 0;
}
