/*
 * liquid_cache_amd — C ABI of the MI355X-native LiquidCache decode + predicate-pushdown path.
 *
 * This header is the drop-in boundary: plain C, pointers and sizes only.  It replaces, for entries
 * whose state is `CacheEntry::MemoryLiquid`, the reference calls (paths relative to the reference
 * repository root, crate `liquid-cache` v0.1.12):
 *
 *   LiquidCache::insert / transcode          src/core/src/cache/core.rs:122-128, cache/transcode.rs:46-290
 *   LiquidCache::get().with_selection()      src/core/src/cache/builders.rs:218-276, cache/core.rs:595-634
 *   LiquidCache::eval_predicate()            src/core/src/cache/builders.rs:314-356, cache/core.rs:862-930
 *   LiquidArray::{filter,try_eval_predicate} src/core/src/liquid_array/mod.rs:117-130
 *   boolean_buffer_and_then                  src/datafusion/src/utils.rs:62-83
 *   read_from_bytes (Liquid IPC)             src/core/src/liquid_array/ipc.rs:250-283
 *
 * Conventions (same as the reference / Arrow): little endian, LSB-first bitmaps, i32 offsets for
 * Utf8/Binary.  Cached arrays are immutable once staged.  All functions are thread-safe and
 * re-entrant (calls on ONE lc_scan must be serialised by the caller, see lc_scan_eval); none of them aborts or
 * throws across the ABI — every entry point catches C++ exceptions and returns an lc_status.
 * `LC_NOT_STAGED` corresponds to the reference's `None` ("not cached"), `LC_UNSUPPORTED` means
 * "run the reference CPU path for this call".
 *
 * One lc_ctx drives ONE HIP device (one process per GPU); multi-GPU = one ctx per rank, entries
 * sharded by row range (file,row-group,batch), see DESIGN.md.
 */
#ifndef LIQUID_CACHE_AMD_H
#define LIQUID_CACHE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_API __attribute__((visibility("default")))

typedef int32_t lc_status;
#define LC_OK 0
#define LC_NOT_STAGED 1      /* == Option::None from the cache (core.rs:595, :862) */
#define LC_UNSUPPORTED 2     /* caller must fall back to the reference CPU path */
#define LC_NEEDS_BACKING 3   /* the entry is squeezed: this read needs the full array from the caller's disk tier
                              * (the reference's squeezed arrays `read_backing()` in these cases,
                              * squeezed_date32_array.rs:231-241, :440-486; hybrid_primitive_array.rs "NeedsBacking") */
#define LC_ERR_INVALID (-1)  /* bad argument */
#define LC_ERR_CORRUPT (-2)  /* malformed Liquid IPC bytes (the reference panics: ipc.rs:221-226) */
#define LC_ERR_DEVICE (-3)   /* HIP runtime error */
#define LC_ERR_OOM (-4)      /* HBM arena exhausted (reference: CacheFull on insert) */
#define LC_ERR_NO_SYMTAB (-5)/* ByteView entry staged without its FSST symbol table */

/* Comparison operators: datafusion Operator subset accepted by LiquidExpr::try_new
 * (src/core/src/cache/liquid_expr.rs:85-125) and ByteViewOperator (byte_view_array/operator.rs:54-105). */
#define LC_OP_EQ 0
#define LC_OP_NE 1
#define LC_OP_LT 2
#define LC_OP_LE 3
#define LC_OP_GT 4
#define LC_OP_GE 5
#define LC_OP_LIKE 6      /* LikeExpr / LikeMatch, case sensitive */
#define LC_OP_NOT_LIKE 7  /* negated */

/* Literal tags (ScalarValue subset). */
#define LC_LIT_I64 0    /* 8 bytes: Int8..Int64, Date32/64, Timestamp(_, None) */
#define LC_LIT_U64 1    /* 8 bytes: UInt8..UInt64 */
#define LC_LIT_F32 2    /* 4 bytes */
#define LC_LIT_F64 3    /* 8 bytes */
#define LC_LIT_BYTES 4  /* Utf8/Binary needle or LIKE pattern */
#define LC_LIT_I128 5   /* 16 bytes LE: Decimal128 unscaled value (same scale as the column) */
#define LC_LIT_BOOL 6   /* 1 byte: Literal(Boolean) on byte-like columns (liquid_expr.rs:78-80) */

/* CacheExpression hints (src/core/src/cache/expressions.rs:36-51) that change the encoding. */
#define LC_HINT_NONE 0
#define LC_HINT_SUBSTRING_SEARCH 1 /* build 32-bucket string fingerprints (byte_view_array/fingerprint.rs) */
#define LC_HINT_PREDICATE_COLUMN 2

typedef struct {
    int32_t op;       /* LC_OP_* */
    int32_t lit_tag;  /* LC_LIT_* */
    const void* lit;  /* literal bytes */
    uint64_t lit_len; /* byte length of `lit` */
} lc_predicate;

typedef struct lc_ctx lc_ctx;
typedef struct lc_scan lc_scan;

typedef struct {
    int32_t device_id;
    int32_t compute_units;
    uint64_t hbm_total_bytes;
    uint64_t hbm_staged_bytes;   /* bytes of Liquid data resident in HBM */
    uint64_t staged_entries;
    char name[64];
    char gcn_arch[32];
} lc_device_info;

typedef struct {
    int32_t logical_type;   /* LiquidDataType (liquid_array/mod.rs:50-65): 1 int, 2 float, 4 byte-view, 6 decimal */
    int32_t physical_type;  /* ipc.rs:28-45, or ArrowByteType for byte views (byte_view_array/mod.rs:113-122) */
    uint32_t len;           /* rows */
    int32_t nullable;
    int32_t all_null;
    int32_t bit_width;      /* W, 0 if all-null */
    uint32_t dict_len;      /* D for byte views */
    int32_t has_fingerprints;
    uint64_t device_bytes;  /* HBM bytes held by the entry */
    uint64_t algorithmic_pred_bytes; /* SURVEY §8(d) bytes one predicate evaluation reads+writes */
    int32_t squeezed_date_field;     /* LC_DATE_* if the entry is a squeezed date component (lc_squeeze_date), else -1 */
    int32_t clamped_from_bit_width;  /* original W if the entry is clamp-squeezed (lc_squeeze_clamp), else 0 */
    int32_t quantized_from_bit_width; /* original W if the entry is quantize-squeezed (lc_squeeze_quantize), else 0 */
    int32_t reserved0;
    uint64_t quantized_bucket_width; /* offsets per bucket of a quantize-squeezed entry, else 0 */
} lc_entry_info;

/* Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html), declared here so the
 * header is self-contained; layout-identical to arrow's own structs. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif

/* ------------------------------------------------------------------ context */

/* LiquidCacheBuilder::new().build() (builders.rs:50-157).  `n_devices` must be 1 (one process per GPU);
 * device_ids == NULL selects the current HIP device.  n_devices == 0 with device_ids == NULL creates a HOST-ONLY
 * context that can transcode Arrow -> Liquid bytes and hold symbol tables but fails every staging / evaluation
 * call with LC_ERR_DEVICE (there is no CPU fallback for the compute path).  `max_hbm_bytes` == 0 means "as much as is free"
 * (reference default max_memory_bytes is 1 GiB; HBM has 288 GB, the arena grows in slabs).  A device context also makes the
 * HIP runtime load the library's kernel code objects now (a few milliseconds) rather than in front of a query's first launch. */
LC_API lc_status lc_ctx_create(const int32_t* device_ids, int32_t n_devices, uint64_t max_hbm_bytes, lc_ctx** out);
LC_API void lc_ctx_destroy(lc_ctx* ctx);
/* Staging options of a context (set them before staging; entries already staged keep what they were staged with).  The
 * library reads NO environment variable: nothing outside these calls changes what is staged or how it is evaluated.
 * Results are identical under every combination (tested); only HBM use and speed differ.
 *   LC_OPT_SIGNATURE_INDEX  (default 1) byte views of substring-search columns get the bit-sliced bigram signature index
 *                           (+ ~25 % HBM per entry); 0 = the reference layout only: LIKE runs the reference's
 *                           fingerprint filter (byte_view_array/fingerprint.rs) and walks its candidates
 *   LC_OPT_ROW_LISTS        (default 1) such entries also get inverted row lists (+ ~12 %)
 *   LC_OPT_HOST_BUILT_INDEX (default 0) 1 = the signature index is built by the host while staging instead of by the
 *                           device kernel (bit-identical; kept as the device builder's cross-check)
 * Evaluation options (may be changed at any time):
 *   LC_OPT_LIKE_PIPELINE_MIN_ENTRIES (default 32) on scans of at least this many entries a LIKE '%needle%' is PLANNED once
 *                           per scan and needle (one trial evaluation counts the hit rows; the one host round trip) and
 *                           selective needles then run the lean kernel (k_like_lean) instead of the general one
 *                           (k_str_pred); negative = never
 *   LC_OPT_LIKE_PATH        tuning / A-B aid: 0 automatic (default: selective needles, 1-byte needles and string = / <>
 *                           through the scan-level bigram / unigram index where the scan has one), 1 k_str_pred only,
 *                           2 automatic without the scan-level signature index, 3 / 4 k_like_lean / k_like_flat for
 *                           every needle, 5 k_like_scanall (the walker) for every needle.  Results never depend on it. */
#define LC_OPT_SIGNATURE_INDEX 1
#define LC_OPT_ROW_LISTS 2
#define LC_OPT_HOST_BUILT_INDEX 3
#define LC_OPT_LIKE_PIPELINE_MIN_ENTRIES 4
#define LC_OPT_LIKE_PATH 5
#define LC_OPT_LIKE_MANY_HINT 6 /* A/B aid (default 1): needles the plan found unselective run k_str_pred's sequential walker */
/* Residency of the scan-level LIKE indexes (lc_scan_info: 64 + 32 bytes per dictionary value of a scan, built on its first
 * LIKE).  LC_OPT_LIKE_INDEX_BUDGET_BYTES (default 0 = none of its own): indexes alive in the context may not exceed this —
 * when a scan's first LIKE would, the indexes cached for future scans are dropped first, and if live scans still hold the
 * budget the scan evaluates with the entry-level index (k_like_lean, same results).  max_hbm_bytes (slabs + indexes) and
 * "half of the free device memory" apply as well.  LC_OPT_LIKE_INDEX_CACHE (default 4): how many indexes of DESTROYED scans
 * are kept for the next scan over the same publications of the same entries (oldest first out; 0: none). */
#define LC_OPT_LIKE_INDEX_BUDGET_BYTES 7
#define LC_OPT_LIKE_INDEX_CACHE 8
/* LC_OPT_LIKE_INDEX_ASYNC (default 1, round 6): the scan-level LIKE indexes are built OFF the query path — a scan's first LIKE
 * is answered at once from the entry-level index (which exists since staging) while the context's builder thread builds the
 * scan-level one on a stream of its own; evaluations switch when it is in place.  This is where the reference puts its
 * prefilter construction as well: outside the read path (at insert time under the SubstringSearch hint,
 * byte_view_array/conversions.rs:353-355).  A build that would have to EVICT another scan's cached index is only started once
 * the asking scan has served 8 LIKE evaluations (an index costs ~4 ms of device time per 100 M rows and repays ~13 us per
 * evaluation).  Results are identical before, during and after the build (tested).  0: the first LIKE of a scan waits for
 * the build, as before round 6.  lc_scan_index_wait blocks until the builds in flight for a scan are in place
 * (lc_scan_explain does the same, so that it describes the steady state; lc_scan_info_get does not wait and says whether a
 * build is pending).
 * LC_OPT_SCAN_CACHE (default 32, round 6): scans given back with lc_scan_destroy are kept, and lc_scan_create over an entry-id
 * list seen before returns the kept scan — no entry look-ups, no descriptor upload, no records / automata / plans to rebuild —
 * as long as none of its entries has been replaced or evicted since (those scans are destroyed when that happens, and
 * their pins with them; the scan-level LIKE index of a kept scan stays evictable by the index budget).  The reference's reader names the entries of a row group per query and holds no scan objects
 * (liquid_cache_reader.rs:264-339); a host that follows it creates a scan per query.  0: every lc_scan_destroy frees the scan. */
#define LC_OPT_LIKE_INDEX_ASYNC 9
#define LC_OPT_SCAN_CACHE 10
/* LC_OPT_COMM_SHARED_MEMORY (default 0): 1 = the lc_comm_* calls of this DEVICE context run over the shared-memory test backend
 * of the host-only contexts (device pointers travel through host copies) instead of RCCL — for dry runs of a multi-rank job
 * whose ranks share one GPU, which RCCL refuses.  Never a measurement. */
#define LC_OPT_COMM_SHARED_MEMORY 11
LC_API lc_status lc_ctx_set_option(lc_ctx* ctx, int32_t option, int64_t value);
LC_API const char* lc_last_error(lc_ctx* ctx); /* thread-local message of the last failing call */
LC_API lc_status lc_device_info_get(lc_ctx* ctx, lc_device_info* out);
LC_API const char* lc_version(void);

/* ------------------------------------------------------------------ staging */

/* FSST symbol table of one ColumnAccessPath (file,row-group,column): src/datafusion/src/cache/id.rs:141-146.
 * `bytes` in the reference's own save_symbol_table format (raw/fsst_buffer.rs:848-883):
 * [n:u8][len:u8 x n][symbol:u64 LE x n].  Must be set before ByteView entries of that path are staged
 * (the table is NOT part of LiquidArray::to_bytes(), ipc.rs:238-264). */
LC_API lc_status lc_symtab_set(lc_ctx* ctx, uint64_t path_id, const uint8_t* bytes, size_t len);

/* Stage `n` entries given as the reference's serialized LiquidArray bytes (LiquidArray::to_bytes(),
 * ipc.rs:158-236, bit_pack_array.rs:181-256, primitive_array.rs:603-679, decimal_array.rs:197-220,
 * float_array.rs:397-519, byte_view_array/serialization.rs:87-220).  `path_ids` may be NULL when no entry
 * is a byte view.  Replaces an entry that is already staged.  Data is copied; the caller keeps `bytes`. */
LC_API lc_status lc_stage(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes,
                          const size_t* lens, const uint64_t* path_ids);
/* Drop entries (CacheEntry eviction / squeeze to disk in the reference); unknown ids are ignored.  A scan pins the
 * entries it was created over (the reference's scans hold `Arc<dyn LiquidArray>` clones): evicting or re-staging an
 * entry under a live scan is safe — the scan keeps evaluating the data it captured, and the HBM is returned once the
 * last scan holding it is destroyed. */
LC_API lc_status lc_evict(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids);
LC_API lc_status lc_entry_info_get(lc_ctx* ctx, uint64_t entry_id, lc_entry_info* out);

/* Arrow -> Liquid transcoder (transcode_liquid_inner_with_hint, cache/transcode.rs:46-290).  Produces the same
 * Liquid IPC bytes a Rust `to_bytes()` would stage.  For byte-like arrays the symbol table of `path_id`
 * is trained from this array if none is registered yet (transcode.rs:16-33) and registered on `ctx`.
 * `*out_bytes` is malloc'ed; release with lc_free.  Returns LC_UNSUPPORTED for types the reference does not
 * transcode (they stay Arrow in the cache). */
LC_API lc_status lc_transcode_arrow(lc_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema,
                                    int32_t hint, uint64_t path_id, uint8_t** out_bytes, size_t* out_len);
/* cache.insert(entry_id, array) with eager transcoding (benchmark/README.md:42 `liquid_eager_transcode`). */
LC_API lc_status lc_insert_arrow(lc_ctx* ctx, uint64_t entry_id, const struct ArrowArray* array,
                                 const struct ArrowSchema* schema, int32_t hint, uint64_t path_id);
/* The same for `n` arrays in one call (typically the batches of one row group): the arrays are transcoded one after the
 * other on the calling thread, then staged together — one upload, ONE launch of the signature builder for all of them, one
 * publication.  A column inserted entry by entry pays a ~80 us builder launch per entry (12,207 launches for the 100 M-row
 * URL column); callers that parallelise staging give every thread whole row groups.  hints / path_ids may be NULL. */
LC_API lc_status lc_insert_arrow_batch(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids,
                                       const struct ArrowArray* const* arrays, const struct ArrowSchema* const* schemas,
                                       const int32_t* hints, const uint64_t* path_ids);
/* Arrow -> Liquid transcoding ON THE DEVICE for integer-like arrays (Int8..UInt64, Date32/64, Timestamp without zone),
 * Decimal128 / Decimal256 arrays and Float32 / Float64 arrays: the raw values cross PCIe once; min / max (frame of
 * reference, bit width), the ALP exponent search on the reference's sample, encoding, exception (patch) extraction and
 * the FastLanes packing run as kernels (LiquidPrimitiveArray::from_arrow_array, primitive_array.rs:159-206;
 * LiquidDecimalArray::from_decimal_array, decimal_array.rs:127-177; LiquidFloatArray::from_arrow_array,
 * float_array.rs:590-740; BitPackedArray::from_primitive, bit_pack_array.rs:71-124).  The staged entries are
 * byte-identical to what lc_insert_arrow stages (checked through lc_entry_to_liquid_bytes).
 * Utf8 / Binary arrays (i32 offsets) and Utf8View / BinaryView arrays (what DataFusion's Parquet reader produces; their rows
 * are laid out as offsets + bytes by the same host pass that copies them into pinned memory), up to 65536 rows, become
 * LiquidByteViewArrays on the device as well
 * (LiquidByteViewArray::from_string_array, byte_view_array/conversions.rs:260-373): dictionary in first-occurrence order,
 * FSST compression of the dictionary values with the path's symbol table (trained on the host from the first array of a
 * path, transcode.rs:16-33 — training is once per column chunk, encoding is per batch), shared prefix, prefix keys,
 * fingerprints, compact offsets, and the acceleration index (signatures, row lists).  Without hints / path ids
 * (lc_insert_arrow_device) byte views get no fingerprints and path 0.
 * LC_UNSUPPORTED for other array types (64-bit offsets, dictionaries), for decimal arrays with a value that does not
 * fit a u64 (the reference's fits_u64, decimal_array.rs:120-125), and for a byte-view array whose offset line fit would
 * round in f64 (sums above 2^53: gigabyte-sized batches): use lc_insert_arrow.  All-or-nothing per array class. */
LC_API lc_status lc_insert_arrow_device(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids,
                                        const struct ArrowArray* const* arrays, const struct ArrowSchema* const* schemas);
LC_API lc_status lc_insert_arrow_batch_device(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids,
                                              const struct ArrowArray* const* arrays, const struct ArrowSchema* const* schemas,
                                              const int32_t* hints, const uint64_t* path_ids);
/* LiquidArray::to_bytes() of a staged entry, rebuilt from HBM (malloc'ed; release with lc_free): what the reference
 * writes to its disk tier when an entry is squeezed or evicted (core.rs:246, :314).  Fixed-width entries
 * (primitive_array.rs:603-679, decimal_array.rs:197-220, float_array.rs:397-519) and byte views
 * (byte_view_array/serialization.rs:122-220: keys at 16 bits, raw FSST buffer, compact offsets, prefix keys, shared
 * prefix, fingerprints).  LC_NEEDS_BACKING for squeezed entries (they no longer hold the full array). */
LC_API lc_status lc_entry_to_liquid_bytes(lc_ctx* ctx, uint64_t entry_id, uint8_t** out_bytes, size_t* out_len);
/* The device-side acceleration index of a byte-view entry (bigram signature slices + inverted row lists: DESIGN.md §2;
 * not part of the reference format) as one opaque, versioned blob (malloc'ed; release with lc_free; *out_len == 0 when
 * the entry carries none).  A disk tier that keeps it beside the Liquid bytes hands it back to lc_stage_indexed and the
 * entry becomes visible without the index being rebuilt (80 us of device time and a counting sort per entry). */
LC_API lc_status lc_entry_index_to_bytes(lc_ctx* ctx, uint64_t entry_id, uint8_t** out_bytes, size_t* out_len);
/* lc_stage with prebuilt indexes: index_bytes[i] (may be NULL) is what lc_entry_index_to_bytes returned for the same
 * Liquid bytes.  A blob that does not describe exactly this entry (dictionary size, rows, signature width, section sizes,
 * list bounds) is ignored and the index is rebuilt: a stale or foreign blob can cost time, never a result. */
LC_API lc_status lc_stage_indexed(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, const uint8_t* const* bytes,
                                  const size_t* lens, const uint64_t* path_ids, const uint8_t* const* index_bytes,
                                  const size_t* index_lens);
LC_API void lc_free(void* p);
/* Export the registered symbol table of `path_id` in save_symbol_table format (malloc'ed). */
LC_API lc_status lc_symtab_get(lc_ctx* ctx, uint64_t path_id, uint8_t** out_bytes, size_t* out_len);

/* ------------------------------------------------------------------ per-entry API (drop-in) */

/* cache.eval_predicate(&entry_id, &expr).with_selection(&sel).read()  (builders.rs:336-346).
 *   selection: `len` bits, LSB first, or NULL == BooleanBuffer::new_set(len) (core.rs:907-911)
 *   out_values / out_validity: caller-allocated, ceil(len/8) bytes each is always enough;
 *   the result BooleanArray has *out_len = popcount(selection) bits (Appendix B.1 of SURVEY.md);
 *   *out_nullable != 0 iff the result carries a validity bitmap (written to out_validity).
 * Value bits under null slots are unspecified (as in Arrow). */
LC_API lc_status lc_eval_predicate(lc_ctx* ctx, uint64_t entry_id, const lc_predicate* pred,
                                   const uint8_t* selection, uint8_t* out_values, uint8_t* out_validity,
                                   uint32_t* out_len, int32_t* out_nullable);

/* Same for `n` entries in one device pass (launch amortisation: 8192 rows x 1-8 B per entry is far too little
 * for one launch).  selections[i] may be NULL; statuses[i] receives the per-entry status. */
LC_API lc_status lc_eval_predicate_batch(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids,
                                         const lc_predicate* pred, const uint8_t* const* selections,
                                         uint8_t* const* out_values, uint8_t* const* out_validity,
                                         uint32_t* out_lens, int32_t* out_nullable, lc_status* statuses);

/* A predicate that is an OR over two or more COLUMNS of one batch (`a = 1 OR b LIKE '%x%'`): what
 * CachedRowGroup::evaluate_selection_with_predicate does for such predicates (src/datafusion/src/cache/mod.rs:111-150,
 * extract_multi_column_or): every (entry, predicate) pair is evaluated on the encoded data over the SAME selection and
 * the BooleanArrays are combined with arrow's or_kleene (true if any side is true; else null if any side is null; else
 * false).  entry_ids[i] is column i's entry of the batch.  Outputs as lc_eval_predicate; LC_NOT_STAGED if any entry is
 * absent (the reference returns None and materialises instead). */
LC_API lc_status lc_eval_predicate_or(lc_ctx* ctx, uint32_t n, const uint64_t* entry_ids, const lc_predicate* preds,
                                      const uint8_t* selection, uint8_t* out_values, uint8_t* out_validity,
                                      uint32_t* out_len, int32_t* out_nullable);

/* cache.get(&entry_id).with_selection(&sel).read()  (builders.rs:236-266): rows whose selection bit is set, in
 * order, in the entry's original Arrow type, exported through the Arrow C Data Interface (caller releases).
 * selection NULL == no selection (to_arrow_array). */
LC_API lc_status lc_get_with_selection(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection,
                                       struct ArrowArray* out_array, struct ArrowSchema* out_schema);

/* cache.get(&entry_id).with_expression_hint(CacheExpression::extract_date32(field)).with_selection(&sel).read()
 * (builders.rs:242-266 -> core.rs:725-745 try_read_squeezed_date32_array): for Date32 / Timestamp entries, the
 * selected rows with only `field` preserved — SqueezedDate32Array's lossy reconstruction
 * (squeezed_date32_array.rs:267-359: Year -> (y,1,1), Month -> (1970,m,1), Day -> (1970,1,d), DayOfWeek -> 1970-01-04
 * + dow; timestamps at midnight UTC of that date, days = value.div_euclid(ticks_per_day) :406-414), in the entry's
 * original Arrow type.  LC_UNSUPPORTED for other types. */
#define LC_DATE_YEAR 0
#define LC_DATE_MONTH 1
#define LC_DATE_DAY 2
#define LC_DATE_DAY_OF_WEEK 3
LC_API lc_status lc_get_date_part_with_selection(lc_ctx* ctx, uint64_t entry_id, const uint8_t* selection,
                                                 int32_t field, struct ArrowArray* out_array,
                                                 struct ArrowSchema* out_schema);

/* A whole pushed-down filter in ONE call: the steps of the reference's LiquidRowFilter in evaluation order
 * (build_row_filter / get_priority, src/datafusion/src/reader/plantime/row_filter.rs:428-515), as LiquidCacheReader runs
 * them per batch (liquid_cache_reader.rs:297-339) — here over whole scans: every step's hit mask is the selection of
 * the next (boolean_buffer_and_then), nothing returns to the host in between.  A step is one predicate or a fusable pair
 * of range predicates on one column (LC_STEP_AND; an unfusable pair is chained as two passes), or a Kleene OR over
 * n_terms (scan, predicate) pairs (LC_STEP_OR, cache/mod.rs:111-150).  d_mask_a / d_mask_b: two device buffers of
 * lc_scan_mask_words u64 used alternately; *d_final_mask receives the one that holds the filter's result (d_selection
 * itself when n_steps == 0).  d_counts_out (optional): per-entry hits of the last pass; d_total_out (optional): COUNT(*)
 * of the filter, written by the last predicate kernel (the last step must be LC_STEP_AND).  Asynchronous on `stream`. */
#define LC_STEP_AND 0
#define LC_STEP_OR 1
typedef struct {
    int32_t kind;               /* LC_STEP_AND / LC_STEP_OR */
    uint32_t n_terms;           /* AND: 1 or 2 predicates on scans[0]; OR: number of disjuncts */
    lc_scan* const* scans;      /* AND: scans[0]; OR: n_terms scans (the same scan may repeat: IN lists) */
    const lc_predicate* preds;  /* n_terms predicates */
} lc_filter_step;
LC_API lc_status lc_scan_eval_filter(lc_ctx* ctx, uint32_t n_steps, const lc_filter_step* steps, const void* d_selection,
                                     void* d_mask_a, void* d_mask_b, void* d_counts_out, void* d_total_out,
                                     void** d_final_mask, void* stream);

/* Partial aggregation under a selection — the step after the path: what DataFusion's AggregateExec
 * (mode: Partial, the url_prefix_filtering snapshot under datafusion-local/src/tests/snapshots) computes from the rows that
 * get().with_selection() returns, without returning them.  COUNT, SUM, MIN and MAX of the valid rows of a fixed-width
 * scan (integers, dates, timestamps, decimals as their unscaled integers) that d_selection selects (null: every row;
 * usually the hit mask of the last conjunct), one pass over the packed data, result left on the device: the caller
 * combines the partials of its row ranges / ranks (sum of counts and sums, min of mins, ...).  The sum is exact
 * (128-bit two's complement: lo, hi); min / max are the value's 64-bit pattern (signed or unsigned as the column) and
 * only meaningful when count > 0.  Asynchronous on `stream`.  LC_UNSUPPORTED for floats and byte views,
 * LC_NEEDS_BACKING when a selected row of a squeezed entry has no value in HBM. */
typedef struct {
    uint64_t count;
    uint64_t sum_lo, sum_hi;
    uint64_t min, max;
    uint64_t reserved;
} lc_aggregate;
LC_API lc_status lc_scan_aggregate(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_out /* lc_aggregate */,
                                   void* stream);
/* SUM(a * b) — TPC-H Q6's sum(l_extendedprice * l_discount) — over the rows that d_selection selects and that are valid
 * in BOTH columns: two scans over the same row ranges with the same lane width (two decimal columns, two Int64 columns,
 * ...).  count and the exact product sum (two's-complement i128, valid while the true sum stays below 2^127) are
 * written to *d_out; min / max are 0.  Decimals multiply as their unscaled integers: the result has scale sa + sb. */
LC_API lc_status lc_scan_sum_product(lc_ctx* ctx, lc_scan* scan_a, lc_scan* scan_b, const void* d_selection,
                                     void* d_out /* lc_aggregate */, void* stream);

/* Partial GROUP BY over a byte-view column with COUNT(*) and MIN / MAX of a second byte-view column — the step after the
 * path for ClickBench q21.sql (`SELECT "SearchPhrase", MIN("URL"), COUNT(*) ... WHERE <pushed-down filter> GROUP BY
 * "SearchPhrase"`): what DataFusion's AggregateExec(mode = Partial) computes from the rows get().with_selection() returns,
 * without returning them.  group_scan and value_scan (NULL: COUNT only) cover the same row ranges.  For the rows that
 * d_selection selects (NULL: all) every entry is grouped by the group column's dictionary key (inside a batch equal keys
 * are equal strings) and one lc_group_partial per (entry, group) is appended to d_partials:
 *   entry       scan index of the entry
 *   group_row   the first selected row of the group in that entry (its group value — NULL for the group of null rows —
 *               is what lc_scan_gather_bytes decodes for the row reference (entry << 32 | group_row))
 *   count       selected rows of the group, null values of the value column included (COUNT(*))
 *   best_row    the row that holds the smallest (want_max: largest) NON-NULL value of the value column among them, byte-wise
 *               order as Arrow's min / max of a string array; 0xFFFFFFFF if every value is null (or no value_scan)
 * *d_n_partials (u64, device; zeroed by the call) receives how many partials the selection produces; records beyond
 * `capacity` are dropped — compare and retry with a larger buffer.  An entry may emit several partials for one group
 * (more than 704 distinct groups among its selected rows): partials merge by value like those of different entries.
 * The final aggregate (merging partials whose group strings are equal) is the caller's, as it is DataFusion's final
 * AggregateExec's.  Order of the records is unspecified.  Asynchronous on `stream`. */
typedef struct {
    uint32_t entry, group_row, count, best_row;
} lc_group_partial;
LC_API lc_status lc_scan_group_partials(lc_ctx* ctx, lc_scan* group_scan, lc_scan* value_scan, int32_t want_max,
                                        const void* d_selection, void* d_partials, uint64_t capacity, void* d_n_partials,
                                        void* stream);

/* Squeeze Date32 / Timestamp entries to ONE calendar component (LiquidPrimitiveArray::squeeze with the hint
 * CacheExpression::extract_date32(field), primitive_array.rs:389-420 -> SqueezedDate32Array, squeezed_date32_array.rs:
 * 46-221): the entry is replaced in HBM by the component, frame-of-reference + bit-packed on u32 lanes (TPC-H ship dates
 * squeezed to YEAR: 3 bits per row).  Afterwards lc_get_date_part_with_selection(field) is served from the component
 * (same values as before the squeeze); every other read or predicate on the entry answers LC_NEEDS_BACKING — the caller
 * keeps the full bytes on its disk tier (lc_entry_to_liquid_bytes BEFORE squeezing gives them) exactly like the
 * reference's `read_backing()`.  Decode, component extraction, min / max and packing run on the device. */
LC_API lc_status lc_squeeze_date(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, int32_t field);

/* Squeeze integer entries to half their bit width with the Clamp policy (LiquidPrimitiveArray::squeeze,
 * primitive_array.rs:589-660 -> LiquidPrimitiveClampedArray, hybrid_primitive_array.rs:73-160): offsets at or above the
 * sentinel 2^(W/2) - 1 are stored as the sentinel.  Entries under 8 bits, all-null entries and other encodings are left
 * alone; *out_squeezed (optional) counts the entries that were squeezed.  Afterwards a predicate is answered from HBM
 * whenever the reference's try_eval_predicate_inner can decide it — always, unless a valid selected row holds the
 * sentinel AND the literal lies at or above reference + sentinel (:199-222) — and a read whenever no selected valid row
 * holds the sentinel (to_arrow_known_only); otherwise the call answers LC_NEEDS_BACKING (per entry in
 * lc_eval_predicate_batch's `statuses`) and the caller reads the full array from its disk tier.  Calls on scans that
 * contain clamped entries synchronise the stream (the sentinel check). */
LC_API lc_status lc_squeeze_clamp(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, uint64_t* out_squeezed);

/* The same with the Quantize policy (IntegerSqueezePolicy::Quantize, the reference's default; primitive_array.rs:455-498
 * -> LiquidPrimitiveQuantizedArray, hybrid_primitive_array.rs:427-665): a row keeps the bucket
 * (value - reference) / bucket_width at half the bit width, bucket_width = ceil((max offset + 1) / 2^(W/2)).  A
 * comparison with literal k is answered from HBM unless a valid selected row lies in k's own bucket and that bucket
 * does not decide it: `< k` and `>= k` are decided there when k is the first value of its bucket, `<= k` and `> k` when
 * it is the last, `=` / `<>` never (:575-618); a literal below the reference decides everything (:535-547).  Otherwise
 * LC_NEEDS_BACKING.  Every read of a quantized entry answers LC_NEEDS_BACKING (its to_arrow_array hydrates from disk,
 * :688-690).  The division, clamp to the last bucket and packing run on the device.
 * Float32 / Float64 entries take the float form (FloatSqueezePolicy::Quantize, the only float policy; float_array.rs:338-395
 * -> LiquidFloatQuantizedArray :742-953): a row keeps (encoded >> shift) - (reference >> shift) of its ALP-encoded value at
 * half the bit width, shift = W - W / 2; validity and the ALP exceptions stay.  A comparison is decided per row from the
 * decoded bounds of its bucket exactly as the reference computes them (plain IEEE operators; exception rows by their
 * values); any valid selected unpatched row left undecided -> LC_NEEDS_BACKING.  LC_UNSUPPORTED for a selection over an
 * entry WITH exceptions (the reference filters the buckets but not the patch indices, :772-792: there is no defined
 * result to be identical to).  Entries in which a bucket would need one bit more than the halved width (the reference
 * packs such a value with whatever its packer does) are left unsqueezed. */
LC_API lc_status lc_squeeze_quantize(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, uint64_t* out_squeezed);

/* boolean_buffer_and_then(left, right) (src/datafusion/src/utils.rs:62-83): `left` has left_bits bits of which
 * right_bits are set; out (ceil(left_bits/8) bytes) keeps the set bits of `left` whose `right` bit is 1. */
LC_API lc_status lc_mask_and_then(lc_ctx* ctx, const uint8_t* left, uint64_t left_bits, const uint8_t* right,
                                  uint64_t right_bits, uint8_t* out);

/* ------------------------------------------------------------------ device-resident column scans */

/* A scan is an ordered list of staged entries of ONE column (consecutive 8192-row batches of a row range).
 * Its hit mask lives in HBM as per-entry segments of ceil(len/64) 64-bit words, concatenated in entry order
 * (row-range shards concatenate across ranks without bit shifting).  Conjunctions chain on the device:
 * the mask produced by one predicate is the selection of the next (what LiquidCacheReader::build_predicate_filter
 * does with boolean_buffer_and_then, src/datafusion/src/reader/runtime/liquid_cache_reader.rs:297-339). */
LC_API lc_status lc_scan_create(lc_ctx* ctx, uint64_t n, const uint64_t* entry_ids, lc_scan** out);
LC_API void lc_scan_destroy(lc_scan* scan);
LC_API uint64_t lc_scan_mask_words(const lc_scan* scan);  /* total u64 words of the mask */
LC_API uint64_t lc_scan_rows(const lc_scan* scan);
LC_API uint64_t lc_scan_entries(const lc_scan* scan);
/* What a scan holds.  index_bytes / unigram_index_bytes: the scan-level LIKE indexes built on this scan's first LIKE (first
 * 1-byte LIKE) — derived data outside the entries' blobs, 64 + 32 bytes per dictionary value of a group of up to four entries,
 * every group rounded up to whole 128-byte lines per slice (at least 1,024 values' worth: 64 KB + 32 KB per group — a scan of
 * many small dictionaries pays up to 8x the per-value figure); they count against the
 * context's max_hbm_bytes when they are built (a scan whose index does not fit the budget evaluates with the entry-level
 * index instead), stay with the scan, and after lc_scan_destroy are kept (bounded, oldest first) for the next scan over the
 * same publications of the same entries.  ctx_index_bytes: all such indexes alive in the context, cached ones included. */
typedef struct {
    uint64_t entries, rows, mask_words;
    uint64_t entry_bytes;          /* HBM of the scan's entries (sum of lc_entry_info.device_bytes) */
    uint64_t index_bytes;          /* scan-level bigram signature index, 0 if not built (yet) */
    uint64_t unigram_index_bytes;  /* scan-level unigram index, 0 if not built */
    uint64_t ctx_index_bytes;
    uint64_t ctx_slab_bytes;       /* slab capacity the context has reserved for entries (what max_hbm_bytes bounds, with the indexes) */
    double index_build_ms;         /* device time of the builds */
    uint32_t like_plans;           /* needles planned on this scan */
    int32_t is_byte_view;
    int32_t max_bit_width;
    int32_t index_build_pending;   /* 1 while the builder thread is at work for this scan (LC_OPT_LIKE_INDEX_ASYNC); the call does not wait */
    int32_t last_like_kernel;      /* which kernel answered the scan's last [NOT] LIKE / indexed string `=`: LC_LIKE_KERNEL_* */
    int32_t reserved;
} lc_scan_info;
#define LC_LIKE_KERNEL_NONE 0
#define LC_LIKE_KERNEL_STR_PRED 1
#define LC_LIKE_KERNEL_LEAN 2
#define LC_LIKE_KERNEL_FLAT 3
#define LC_LIKE_KERNEL_SCANALL 4
#define LC_LIKE_KERNEL_UNIGRAM 5
LC_API lc_status lc_scan_info_get(lc_scan* scan, lc_scan_info* out);
/* Blocks until the index builds in flight for this scan (LC_OPT_LIKE_INDEX_ASYNC) have finished and their results are in place:
 * the next evaluation runs on the scan-level index if the scan got one.  Returns at once when nothing is being built. */
LC_API lc_status lc_scan_index_wait(lc_scan* scan);

/* Byte accounting of ONE evaluation of `pred` over the scan, for roofline reports:
 *   *out_algorithmic  the reference algorithm's bytes: packed values / keys + selection + validity + output, and for
 *                     byte views the prefilter (4D fingerprints or 8D prefix keys), the offsets and the compressed
 *                     bytes of the prefilter's candidates;
 *   *out_kernel_bytes the bytes this library's kernel itself has to move for the same result (it skips the packed
 *                     data of entries whose FoR range decides the predicate, reads bigram-signature slices instead of
 *                     fingerprints, walks only the candidates that survive them and reads keys only for entries in
 *                     which some dictionary value matched) — the numerator of an honest HBM-roofline fraction.
 * For byte views both are data dependent and are measured by one instrumented device pass on the default stream
 * (synchronises; same serialisation rule as lc_scan_eval).  lc_scan_algorithmic_bytes returns the first figure.
 * `with_selection` is a set of flags: LC_TRAFFIC_WITH_SELECTION (1; any odd value of earlier versions), LC_TRAFFIC_NO_MASK
 * (the evaluation is a COUNT(*) / hit-list call with d_mask_out == NULL: kernels that skip the mask words are not charged
 * for them), LC_TRAFFIC_HIT_LIST (8 bytes per hit row written). */
#define LC_TRAFFIC_WITH_SELECTION 1
#define LC_TRAFFIC_NO_MASK 2
#define LC_TRAFFIC_HIT_LIST 4
LC_API lc_status lc_scan_traffic_model(lc_scan* scan, const lc_predicate* pred, int32_t with_selection,
                                       uint64_t* out_algorithmic, uint64_t* out_kernel_bytes);
LC_API uint64_t lc_scan_algorithmic_bytes(lc_scan* scan, const lc_predicate* pred, int32_t with_selection);
/* One line on how `pred` is evaluated over this scan (which kernels; for LIKE whether the scan-level pipeline planned the
 * needle and with how many candidates) — the EXPLAIN of this library, for logs and benchmarks.  `out` receives a
 * NUL-terminated string of at most cap - 1 characters. */
LC_API lc_status lc_scan_explain(lc_scan* scan, const lc_predicate* pred, char* out, size_t cap);
/* word offset of entry i's segment inside the mask (n+1 values, host memory owned by the scan) */
LC_API const uint64_t* lc_scan_segment_offsets(const lc_scan* scan);

/* Evaluate `pred` over every entry of the scan.
 *   d_selection: device pointer to a mask in scan layout or NULL (all rows); rows outside it are not evaluated.
 *   d_mask_out : device pointer, lc_scan_mask_words() words: hit = pred(row) AND valid(row) AND selected(row)
 *                (i.e. prep_null_mask_filter + boolean_buffer_and_then already applied).
 *   d_counts_out: device pointer to one u32 per entry (popcount of its segment) or NULL.
 *   stream: hipStream_t (NULL = default stream).  Asynchronous; the caller synchronises the stream.
 * A scan owns device scratch (folded automata, work counters, gather buffers): calls on ONE scan must be ordered on
 * one stream (or otherwise serialised); different scans are independent. */
LC_API lc_status lc_scan_eval(lc_ctx* ctx, lc_scan* scan, const lc_predicate* pred, const void* d_selection,
                              void* d_mask_out, void* d_counts_out, void* stream);

/* The same for a conjunction of `n_preds` (1 or 2) predicates on THIS column evaluated in one pass — the common
 * `col >= a AND col < b` pair that the reference's conjunct split (src/datafusion/src/reader/plantime/row_filter.rs:
 * 428-515) turns into two passes over the same array: hit = preds[0] AND preds[1] AND valid AND selected, identical to
 * chaining the two evaluations.  Fixed-width columns, operators Eq / Lt / LtEq / Gt / GtEq; LC_UNSUPPORTED otherwise
 * (the caller chains two lc_scan_eval calls). */
LC_API lc_status lc_scan_eval_and(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                                  const void* d_selection, void* d_mask_out, void* d_counts_out, void* stream);

/* Multi-column OR over whole scans (the device-resident form of lc_eval_predicate_or): scans[i] / preds[i] are the
 * columns of the OR, all covering the same row ranges (same entry lengths, hence the same mask layout); every pair is
 * evaluated over d_selection and combined with Kleene OR.  d_mask_out: hit = (any side true) AND selected;
 * d_valid_out (optional): the validity of the Kleene result restricted to the selection; d_counts_out (optional): hits
 * per entry.  The same scan may appear several times (`col IN (a, b)` == `col = a OR col = b`).  Calls that share
 * scans[0] must be serialised by the caller (its scratch holds the intermediate masks). */
LC_API lc_status lc_scan_eval_or(lc_ctx* ctx, uint32_t n, lc_scan* const* scans, const lc_predicate* preds,
                                 const void* d_selection, void* d_mask_out, void* d_valid_out, void* d_counts_out,
                                 void* stream);

/* lc_scan_eval_and plus the COUNT(*) of the launch: *d_total_out (one u64, device) receives the number of hits of the
 * whole scan, produced by the predicate kernel itself (no reduction pass, no memset between launches) — what a
 * `SELECT COUNT(*) ... WHERE <pushed-down predicate>` consumer (ClickBench q20) or the 8-byte count all-reduce of a
 * sharded scan reads.  d_counts_out (per entry) stays optional.  d_mask_out may be NULL (round 5): a COUNT(*) consumer
 * needs no mask, and for a selective predicate the mask of zeros is a third of the kernel's traffic — k_like_flat then
 * writes none; evaluation paths that cannot skip it write to scan-owned scratch. */
LC_API lc_status lc_scan_eval_count(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                                    const void* d_selection, void* d_mask_out, void* d_counts_out, void* d_total_out,
                                    void* stream);

/* MANY ROW GROUPS PER CALL (round 6).  The reference evaluates a pushed-down predicate per row group and batch
 * (LiquidStream / LiquidCacheReader: liquid_stream.rs:358-430, liquid_cache_reader.rs:297-339); a device wants the launch to
 * cover many of them.  lc_scan_eval_count_groups is lc_scan_eval_count with PER-ROW-GROUP results: the scan's entries are cut
 * into n_groups consecutive groups — group g holds entries [group_ends[g - 1], group_ends[g]) with group_ends[-1] = 0 and
 * group_ends[n_groups - 1] == lc_scan_entries(scan) (host array) — and d_group_counts_out (n_groups u64, device) receives each
 * group's hit count from ONE evaluation launch plus one small reduction.  d_mask_out, d_counts_out (per entry) and
 * d_total_out stay optional.  Asynchronous on `stream`. */
LC_API lc_status lc_scan_eval_count_groups(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                                           const void* d_selection, uint32_t n_groups, const uint32_t* group_ends,
                                           void* d_group_counts_out, void* d_mask_out, void* d_counts_out, void* d_total_out,
                                           void* stream);
/* The same in the reference's call shape, host in / host out: the entries of MANY row groups of one column named by id
 * (row group g = entry_ids[group_ends[g - 1] .. group_ends[g])), no scan object in the caller's hands — the context's scan cache
 * (LC_OPT_SCAN_CACHE) makes the second call over a list O(1).  out_group_counts: n_groups u64 (host).  out_mask (optional,
 * host): the hit mask in scan layout — per entry ceil(len / 64) u64 words, entries concatenated — of out_mask_words words
 * (the sum of ceil(len / 64) over the entries; LC_ERR_INVALID, with the number in the message, if too small; rows past an
 * entry's length are zero).  out_total (optional): the COUNT(*) of the whole list.  Runs on the calling thread's stream
 * and returns when the results are in the caller's buffers.  LC_NOT_STAGED when an entry is absent (the reference's `None`). */
LC_API lc_status lc_eval_predicate_row_groups(lc_ctx* ctx, uint64_t n_entries, const uint64_t* entry_ids, uint32_t n_groups,
                                              const uint32_t* group_ends, const lc_predicate* preds, uint32_t n_preds,
                                              uint64_t* out_group_counts, uint64_t* out_mask, uint64_t out_mask_words,
                                              uint64_t* out_total);

/* get-with-selection over a whole scan for fixed-width columns: compacts the selected rows' decoded values
 * (original Arrow value width) into d_values_out in row order.  d_row_offsets (n+1 u64, device) receives the
 * exclusive prefix sum of per-entry selected counts.  Asynchronous on `stream`. */
LC_API lc_status lc_scan_gather_fixed(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_values_out,
                                      uint64_t values_capacity_bytes, void* d_row_offsets, void* stream);

/* get-with-selection over a whole byte-view scan, device resident (LiquidByteViewArray::filter + to_arrow,
 * byte_view_array/mod.rs:421-424, 266-290, for every entry of the scan): the selected rows' decoded values in row order
 * as (value offsets, bytes), i.e. the buffers of an Arrow LargeBinary / LargeUtf8 array.
 *   plan: d_row_offsets (n+1 u64) receives the exclusive prefix sum of per-entry selected counts; d_row_refs
 *         (capacity_rows u64: entry index << 32 | row) and d_value_offsets (capacity_rows+1 u64) describe the k selected
 *         rows; d_row_valid (capacity_rows bytes, optional) their validity; nulls have length 0.  *out_rows = k,
 *         *out_bytes = total decoded bytes.  Synchronises `stream`.  LC_ERR_INVALID (with *out_rows set) when k exceeds
 *         capacity_rows.
 *   fill: decodes the k values into d_data (>= *out_bytes).  Asynchronous on `stream`. */
LC_API lc_status lc_scan_gather_bytes_plan(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_row_offsets,
                                           void* d_row_refs, void* d_value_offsets, void* d_row_valid,
                                           uint64_t capacity_rows, uint64_t* out_rows, uint64_t* out_bytes, void* stream);
LC_API lc_status lc_scan_gather_bytes(lc_ctx* ctx, lc_scan* scan, const void* d_row_refs, const void* d_value_offsets,
                                      uint64_t rows, void* d_data, void* stream);

/* The same in ONE asynchronous call, for callers that pre-size the outputs (the projection after a selective filter):
 * no host round trip.  Afterwards (stream order) d_row_offsets[n] holds k, d_value_offsets[min(k, capacity_rows)] the
 * total bytes; rows beyond capacity_rows are dropped, values ending beyond capacity_bytes are not written — the caller
 * compares both with its capacities and retries with larger buffers if needed. */
LC_API lc_status lc_scan_gather_bytes_async(lc_ctx* ctx, lc_scan* scan, const void* d_selection, void* d_row_offsets,
                                            void* d_row_refs, void* d_value_offsets, void* d_row_valid,
                                            uint64_t capacity_rows, void* d_data, uint64_t capacity_bytes, void* stream);

/* ---- sparse results: hit lists (round 5) -------------------------------------------------------------------------------
 * A selective filter leaves a handful of rows per batch (ClickBench q21: 16,635 of 99,997,497).  The reference hands such a
 * result on as a BooleanBuffer per batch (liquid_cache_reader.rs:297-339) and gathers with it (:342-391,
 * byte_view_array/helpers.rs:44-64); over whole scans the mask of zeros is most of the traffic and turning it back into
 * rows costs more launches than the filter.  A HIT LIST is the same result as u64 records (scan index of the entry << 32 |
 * row): the records of one entry are contiguous and in ascending row order; the order of the entries is unspecified (the
 * waves append with one atomic each).  The set of records is exactly the set bits of the mask lc_scan_eval would write.
 *
 * lc_scan_eval_hits: evaluate preds (1 or 2, as lc_scan_eval_count) over the scan and append the hit rows to d_hits_out
 *   (`capacity` records).  *d_n_hits (u64, device; zeroed by the call) receives the number of hits, which may EXCEED capacity:
 *   records beyond it are dropped — compare and retry with a larger buffer (or fall back to the mask form).
 *   d_hit_first (optional, u32 per entry): for every entry WITH hits the index of its first record (entries without: untouched);
 *   d_counts_out (optional): hits per entry; d_total_out (optional): COUNT(*).  No mask is written.  Kernels that can emit the
 *   list themselves (k_like_flat: selective [NOT] LIKE and string = / <>) do; for every other evaluation path the mask goes to
 *   scan-owned scratch and one more kernel lists it — always correct, fastest where it matters.  Asynchronous on `stream`.
 * lc_scan_mask_to_hits: the same list from a mask in scan layout (e.g. the result of lc_scan_eval_filter); flags:
 *   LC_HITS_COUNTERS_ZEROED, LC_HITS_PARTITIONED.
 * flags: LC_HITS_COUNTERS_ZEROED — the caller has zeroed the u64 device counters this call writes (*d_n_hits; for the
 *   gathers *d_n_bytes; for the filter *d_n_hits_out), typically all counters of a query with ONE lc_device_memset: the call
 *   then issues no memset of its own (each is a small kernel in front of the real one). */
#define LC_HITS_COUNTERS_ZEROED 1u
/* LC_HITS_PARTITIONED (round 6): the hit list in LC_HITS_PARTITIONS partitions.  A contiguous list is allocated by returning
 * atomics on ONE address, which complete ~10 ns apart however many workgroups wait — 1,100 of them were 11 of the 23.7 us of a
 * selective LIKE with a list.  Partitioned, workgroup b appends to partition b % 16: partition p holds the records
 * d_hits[p * S .. p * S + min(n_p, S)) with S = capacity / LC_HITS_PARTITIONS, and its count n_p is the u64 at
 * d_n_hits[p * LC_HITS_COUNTER_STRIDE] — d_n_hits points at LC_HITS_PARTITIONS * LC_HITS_COUNTER_STRIDE u64 (2 KB; every
 * counter on a 128-byte line of its own).  n_p may exceed S (records beyond are dropped: compare and retry, as with
 * `capacity`); d_hit_first holds positions in the buffer.  The calls that CONSUME a list (lc_scan_filter_hits,
 * lc_scan_gather_fixed_hits, lc_scan_gather_bytes_hits) take the same flag and read the partitions as one list in partition
 * order: row i of a gather's output is record i of that order, the outputs are as dense as with a contiguous list.
 * lc_scan_filter_hits writes its output list in the form it reads.  lc_hits_compact copies the partitions, in that order,
 * into a contiguous list (+ *d_n_hits_out) for a caller that wants to read one. */
#define LC_HITS_PARTITIONED 4u
#define LC_HITS_PARTITIONS 16u
#define LC_HITS_COUNTER_STRIDE 16u
/* lc_scan_gather_bytes_hits only: SLOTTED data buffer.  Record i's bytes start at i * LC_GATHER_SLOT_BYTES of d_data when the
 * value fits a slot; longer values are appended behind the slots, from capacity_rows * LC_GATHER_SLOT_BYTES on, and only they
 * are counted in *d_n_bytes.  A BinaryView's offset may point anywhere in its buffer, so the views are the same Arrow array;
 * what changes is that no space has to be claimed for the common value — the claim (one returning atomic per batch of rows on
 * one counter, served one after the other) is a third of the dense form's time on a short list: 32 -> 22 us per call for the 16,635 URLs a
 * selective LIKE leaves.  Costs: the buffer is sparse (slot bytes per row instead of the value's length) and
 * capacity_bytes must hold capacity_rows slots (LC_ERR_INVALID otherwise) plus room for the long values. */
#define LC_GATHER_SLOTTED 2u
#define LC_GATHER_SLOT_BYTES 128u
LC_API lc_status lc_scan_eval_hits(lc_ctx* ctx, lc_scan* scan, const lc_predicate* preds, uint32_t n_preds,
                                   const void* d_selection, void* d_hits_out, uint64_t capacity, void* d_n_hits,
                                   void* d_hit_first, void* d_counts_out, void* d_total_out, uint32_t flags, void* stream);
/* Selection chaining in sparse form (boolean_buffer_and_then for the next conjunct, src/datafusion/src/utils.rs:62-83, when
 * the selection is a handful of rows): the records of d_hits_in whose row is valid in THIS scan's column and satisfies
 * `pred` are appended to d_hits_out (a different buffer); *d_n_hits_out receives their number (may exceed capacity_out).
 * `scan` covers the same row ranges as the scan that produced the list.  The predicate is evaluated on the row's own
 * value — byte views: compare / match on the compressed value; fixed width: decode and compare, Arrow totalOrder for
 * floats — so the cost is per record, not per row of the column: after a selective first conjunct the remaining
 * conjuncts cost microseconds whatever their columns' sizes.  Conjunctions commute: a host that knows (from
 * lc_scan_explain's plan, or from statistics) which conjunct is selective runs that one first with lc_scan_eval_hits and
 * the others through this call.  The records of a 64-record batch keep their relative order; batches are appended in no
 * particular order (an entry's records may end up apart — gathers do not care).  LC_UNSUPPORTED for squeezed entries. */
LC_API lc_status lc_scan_filter_hits(lc_ctx* ctx, lc_scan* scan, const lc_predicate* pred, const void* d_hits_in,
                                     const void* d_n_hits_in, uint64_t capacity_in, void* d_hits_out, uint64_t capacity_out,
                                     void* d_n_hits_out, uint32_t flags, void* stream);
LC_API lc_status lc_scan_mask_to_hits(lc_ctx* ctx, lc_scan* scan, const void* d_mask, void* d_hits_out, uint64_t capacity,
                                      void* d_n_hits, void* d_hit_first, uint32_t flags, void* stream);
LC_API lc_status lc_hits_compact(lc_ctx* ctx, const void* d_hits, const void* d_n_hits, uint64_t capacity, void* d_hits_out,
                                 uint64_t capacity_out, void* d_n_hits_out, void* stream);
/* get().with_selection() for the rows of a hit list, ONE launch, no host round trip.  `scan` is any scan over the same row
 * ranges as the scan that produced the list (the projection columns of the filtered batches).  Row i of the output is
 * record i of the list; k = min(*d_n_hits, capacity_rows) rows are produced.
 *   fixed width: d_values_out receives k decoded values (the column's Arrow value width); d_row_valid (optional) k bytes.
 *   byte views : d_views receives k Arrow BinaryView / Utf8View records (16 bytes: i32 length; up to 12 bytes the value
 *                itself, zero padded; else 4-byte prefix, buffer index 0, i32 offset into d_data); nulls have length 0 and
 *                d_row_valid[i] == 0.  Space in d_data is claimed with one atomic per batch of rows: *d_n_bytes (u64,
 *                device; zeroed by the call) receives the bytes claimed — every non-empty value claims its length, also
 *                the ones of 12 bytes and less that live in their view (a batch's values are then ONE range of the
 *                buffer, which a wave stores coalesced).  A value that would end beyond capacity_bytes is not written
 *                (its view carries length and offset only): compare and retry.  capacity_bytes < 2 GiB.
 *                flags: LC_HITS_COUNTERS_ZEROED, LC_GATHER_SLOTTED, LC_HITS_PARTITIONED (above; lc_scan_gather_fixed_hits:
 *                LC_HITS_PARTITIONED).
 * LC_UNSUPPORTED for scans that hold squeezed entries (lc_scan_gather_fixed decides their reads). */
LC_API lc_status lc_scan_gather_fixed_hits(lc_ctx* ctx, lc_scan* scan, const void* d_hits, const void* d_n_hits,
                                           uint64_t capacity_rows, void* d_values_out, void* d_row_valid, uint32_t flags,
                                           void* stream);
LC_API lc_status lc_scan_gather_bytes_hits(lc_ctx* ctx, lc_scan* scan, const void* d_hits, const void* d_n_hits,
                                           uint64_t capacity_rows, void* d_views, void* d_row_valid, void* d_data,
                                           uint64_t capacity_bytes, void* d_n_bytes, uint32_t flags, void* stream);

/* ExtractDate32 over gathered values of a Date32 / Timestamp scan: replaces `n_values` decoded values in d_values
 * (as written by lc_scan_gather_fixed) in place by their lossy date-part reconstruction (see
 * lc_get_date_part_with_selection).  Asynchronous on `stream`. */
LC_API lc_status lc_scan_date_part(lc_ctx* ctx, lc_scan* scan, void* d_values, uint64_t n_values, int32_t field,
                                   void* stream);

/* ------------------------------------------------------------------ multi-GPU exchange (one process per GPU)
 *
 * Entries are sharded by ROW RANGE (file, row group, batch): every column of a batch lives on the same rank, so
 * conjunctions chain their selections on one device and the data path needs no collective.  What is
 * left are the two exchange steps of a sharded scan, here directly on RCCL over xGMI (librccl.so is opened lazily: a
 * single-GPU process never loads it).  One lc_comm per lc_ctx; rank 0 creates the unique id and the host distributes
 * its LC_COMM_ID_BYTES bytes to the other ranks by whatever means it has (a file, a socket, MPI ...).
 * For a HOST-ONLY context (lc_ctx_create with n_devices == 0) the same calls run over a shared-memory file of the node and
 * the "device" pointers are host pointers: this exists so that the multi-rank logic runs in CPU test suites. */
#define LC_COMM_ID_BYTES 128
typedef struct lc_comm lc_comm;
LC_API lc_status lc_comm_unique_id(lc_ctx* ctx, uint8_t* out_id /* LC_COMM_ID_BYTES */);
LC_API lc_status lc_comm_init(lc_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id, lc_comm** out);
LC_API void lc_comm_destroy(lc_comm* comm);
LC_API int32_t lc_comm_rank(const lc_comm* comm);
LC_API int32_t lc_comm_world(const lc_comm* comm);
/* COUNT(*) of a sharded scan: *d_total (one u64 on the device, e.g. what lc_scan_eval_count wrote) becomes the sum over
 * all ranks, in place.  Asynchronous on `stream` (ordered after the kernel that produced the partial count). */
LC_API lc_status lc_comm_allreduce_count(lc_comm* comm, void* d_total, void* stream);
/* The hit mask of the whole table from the per-rank masks: rank r contributes words_per_rank[r] u64 words (its
 * lc_scan_mask_words; host array of `world` values, the same on every rank) and every rank receives the concatenation
 * in rank order in d_mask_all — per-entry segments are word aligned, so row-range shards concatenate without bit
 * shifting into the single Arrow BooleanArray.  Asynchronous on `stream`. */
LC_API lc_status lc_comm_allgather_mask(lc_comm* comm, const void* d_mask_local, uint64_t local_words, void* d_mask_all,
                                        const uint64_t* words_per_rank, void* stream);

/* Convenience for hosts without their own HIP runtime binding. */
LC_API lc_status lc_device_alloc(lc_ctx* ctx, uint64_t bytes, void** out_dptr);
LC_API lc_status lc_device_free(lc_ctx* ctx, void* dptr);
LC_API lc_status lc_device_memset(lc_ctx* ctx, void* dptr, int value, uint64_t bytes, void* stream);
LC_API lc_status lc_device_to_host(lc_ctx* ctx, void* host_dst, const void* dptr, uint64_t bytes, void* stream);
LC_API lc_status lc_host_to_device(lc_ctx* ctx, void* dptr, const void* host_src, uint64_t bytes, void* stream);
LC_API lc_status lc_stream_synchronize(lc_ctx* ctx, void* stream);
/* A stream of the caller's own (non-blocking: it does not synchronise with the default stream) — one per worker thread
 * is the intended use: the reference's read path runs on `target_partitions` tokio workers concurrently
 * (datafusion/src/reader/runtime/liquid_cache_reader.rs:297-391) and no call of this library synchronises the device.
 * A stream must OUTLIVE the scans it was used with: lc_scan_destroy drains the streams its launches went to before it
 * recycles the scan's descriptors — destroy the scans first, then the stream. */
LC_API lc_status lc_stream_create(lc_ctx* ctx, void** out_stream);
LC_API lc_status lc_stream_destroy(lc_ctx* ctx, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* LIQUID_CACHE_AMD_H */
