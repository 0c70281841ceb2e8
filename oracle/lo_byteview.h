/* LiquidByteViewArray<FsstArray> (TEST ORACLE — see lo_common.h). */
#ifndef LO_BYTEVIEW_H
#define LO_BYTEVIEW_H
#include "lo_common.h"
#include "lo_fsst.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t arrow_type;       /* lo_arrow_byte_type */
    uint32_t n;               /* rows */
    uint32_t d;               /* dictionary entries */
    int32_t nullable;         /* keys carry a validity buffer */
    int32_t all_null;
    uint16_t* keys;           /* n keys, plain order (owned) */
    uint8_t* key_validity;    /* ceil(n/8) (owned) or NULL */
    const uint8_t* fsst;      /* compressed bytes */
    uint32_t fsst_len;
    uint64_t uncompressed_bytes;
    uint32_t* offsets;        /* d+1 (owned) */
    const uint8_t* prefix_keys; /* d * 8 */
    const uint8_t* shared_prefix;
    uint32_t shared_prefix_len;
    const uint8_t* fingerprints; /* d * 4 LE or NULL */
    /* section sizes for algorithmic byte accounting */
    uint32_t compact_offsets_size;
    int32_t offset_bytes;
} lo_bv;

LO_EXPORT int lo_bv_parse(const uint8_t* bytes, size_t len, lo_bv* out);
LO_EXPORT void lo_bv_free(lo_bv* bv);

LO_EXPORT size_t lo_bv_encode_bound(size_t n, size_t data_len);
/* from Arrow Utf8/Binary (i32 offsets); dictionary built in first-occurrence order */
LO_EXPORT int64_t lo_bv_encode(int arrow_type, const int32_t* offsets, const uint8_t* data, const uint8_t* validity,
                               size_t n, const lo_symtab* st, int build_fingerprints, uint8_t* out, size_t cap);
/* from a Dictionary<UInt16, Utf8|Binary> with UNIQUE values (keys in null slots may be garbage) */
LO_EXPORT int64_t lo_bv_encode_dict(int arrow_type, const uint16_t* keys, const uint8_t* key_validity, size_t n,
                                    const int32_t* dict_offsets, const uint8_t* dict_data, size_t d,
                                    const lo_symtab* st, int build_fingerprints, uint8_t* out, size_t cap);

/* try_eval_predicate: returns k = popcount(sel) (sel NULL => all rows) or <0.
 * lit_tag LO_LIT_BYTES (needle / LIKE pattern) or LO_LIT_BOOL (constant). */
LO_EXPORT int64_t lo_bv_eval_predicate(const uint8_t* bytes, size_t len, const lo_symtab* st, int op, int lit_tag,
                                       const uint8_t* lit, size_t lit_len, const uint8_t* sel, uint8_t* out_values,
                                       uint8_t* out_validity, int* nullable);
/* per-dictionary-entry results of compare_with on the UNFILTERED array (for kernel unit tests); out: d bytes 0/1 */
LO_EXPORT int lo_bv_dict_results(const uint8_t* bytes, size_t len, const lo_symtab* st, int op, const uint8_t* needle,
                                 size_t nlen, uint8_t* out_dict, uint32_t* out_candidates);

/* filter (sel NULL => to_arrow_array): out_offsets i32[k+1], out_data, out_validity (k bits).
 * returns k; *data_len receives the bytes written (call with out_data NULL to size). */
LO_EXPORT int64_t lo_bv_filter_to_arrow(const uint8_t* bytes, size_t len, const lo_symtab* st, const uint8_t* sel,
                                        int32_t* out_offsets, uint8_t* out_data, size_t data_cap, uint8_t* out_validity,
                                        size_t* data_len, int* nullable);

LO_EXPORT uint32_t lo_fingerprint(const uint8_t* s, size_t l);
LO_EXPORT int lo_substring_pattern(const uint8_t* pattern, size_t pl, const uint8_t** inner, size_t* il);

#ifdef __cplusplus
}
#endif
#endif
