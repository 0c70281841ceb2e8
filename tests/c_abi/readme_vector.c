/* A plain C99 program (no Python, no ctypes) that links libliquid_cache_amd.so and runs the reference's README example
 * (README.md:43-88 of XiangpengHao/liquid-cache) through the C ABI:
 *     cache.insert(id, UInt64 [10,11,12,13,14,15])
 *     cache.eval_predicate(id, col > 12)                              -> [F,F,F,T,T,T]
 *     cache.eval_predicate(id, col > 12).with_selection([T,F,T,F,T,F]) -> [F,F,T]
 *     cache.get(id).with_selection([T,F,T,F,T,F])                      -> [10,12,14]
 *     cache.get(other id)                                              -> None (LC_NOT_STAGED)
 * Exit code 0 = every answer matched.  Built and run by tests/test_c_abi_program.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "liquid_cache_amd.h"

static void release_noop_array(struct ArrowArray* a) { a->release = NULL; }
static void release_noop_schema(struct ArrowSchema* s) { s->release = NULL; }

#define CHECK(cond, what)                                                          \
    do {                                                                           \
        if (!(cond)) {                                                             \
            fprintf(stderr, "FAILED: %s (%s)\n", what, lc_last_error(ctx));        \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(void) {
    lc_ctx* ctx = NULL;
    lc_status st = lc_ctx_create(NULL, 1, 0, &ctx);
    if (st != LC_OK) {
        fprintf(stderr, "lc_ctx_create: %d (%s)\n", st, lc_last_error(NULL));
        return 2;
    }
    uint64_t values[6] = {10, 11, 12, 13, 14, 15};
    const void* buffers[2] = {NULL, values};
    struct ArrowArray arr;
    struct ArrowSchema schema;
    memset(&arr, 0, sizeof(arr));
    memset(&schema, 0, sizeof(schema));
    arr.length = 6;
    arr.n_buffers = 2;
    arr.buffers = buffers;
    arr.release = release_noop_array;
    schema.format = "L"; /* uint64 */
    schema.name = "";
    schema.flags = 2;
    schema.release = release_noop_schema;
    CHECK(lc_insert_arrow(ctx, 42, &arr, &schema, LC_HINT_NONE, 0) == LC_OK, "lc_insert_arrow");

    uint64_t twelve = 12;
    lc_predicate gt12 = {LC_OP_GT, LC_LIT_U64, &twelve, 8};
    uint8_t out_values[16] = {0}, out_validity[16] = {0};
    uint32_t out_len = 0;
    int32_t nullable = -1;
    CHECK(lc_eval_predicate(ctx, 42, &gt12, NULL, out_values, out_validity, &out_len, &nullable) == LC_OK, "eval");
    CHECK(out_len == 6 && (out_values[0] & 0x3F) == 0x38 && nullable == 0, "col > 12 -> [F,F,F,T,T,T]");

    const uint8_t selection[1] = {0x15}; /* T,F,T,F,T,F (LSB first) */
    memset(out_values, 0, sizeof(out_values));
    CHECK(lc_eval_predicate(ctx, 42, &gt12, selection, out_values, out_validity, &out_len, &nullable) == LC_OK, "eval+sel");
    CHECK(out_len == 3 && (out_values[0] & 0x7) == 0x4, "with_selection -> [F,F,T]");

    struct ArrowArray got;
    struct ArrowSchema got_schema;
    CHECK(lc_get_with_selection(ctx, 42, selection, &got, &got_schema) == LC_OK, "get");
    CHECK(got.length == 3 && got.n_buffers == 2 && strcmp(got_schema.format, "L") == 0, "get shape");
    const uint64_t* v = (const uint64_t*)got.buffers[1];
    CHECK(v[0] == 10 && v[1] == 12 && v[2] == 14, "get().with_selection() -> [10,12,14]");
    got.release(&got);
    got_schema.release(&got_schema);

    CHECK(lc_get_with_selection(ctx, 43, NULL, &got, &got_schema) == LC_NOT_STAGED, "uncached id -> None");

    /* boolean_buffer_and_then doc example (datafusion/src/utils.rs:54-57): NNYYYNNYYNYN , YNYNYN -> NNYNYNNNYNNN */
    const uint8_t left[2] = {0x9C, 0x05}, right[1] = {0x15};
    uint8_t and_out[2] = {0, 0};
    CHECK(lc_mask_and_then(ctx, left, 12, right, 6, and_out) == LC_OK, "and_then");
    CHECK(and_out[0] == 0x14 && (and_out[1] & 0x0F) == 0x01, "and_then doc example");

    lc_ctx_destroy(ctx);
    printf("c abi ok\n");
    return 0;
}
