// Device-side descriptors and kernel launchers (implemented in lc_kernels.hip).
//
// HBM layout: every staged entry owns one blob inside a context slab; sections are 128-byte aligned.
//   fixed-width entry : [FastLanes packed values: ceil(len/1024) blocks of 128*W bytes][validity as u64 words]
//                       [ALP patch indices u64][ALP patch values]
//   byte-view entry   : [keys u16 x n, plain row order][validity u64 words][prefix keys 8 x D][fingerprints u32 x D]
//                       [offset residuals (1|2|4) x (D+1)][FSST bytes + 16 pad][shared prefix]
// A scan owns a device array of descriptors (one per entry, in scan order).  Masks are per-entry segments of
// ceil(len/64) u64 words (LSB first == Arrow bitmap bytes on little endian).
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "../../include/liquid_cache_amd.h"
#include "lc_fsst_device.hpp"

// Ablation hooks (skip a kernel phase, replace counts by timestamps) are compiled ONLY into profiling builds
// (`make ABLATION=1`): in the shipped library no environment variable can change a result.
#ifdef LC_ABLATION
#define LC_ABL(expr) (expr)
#else
#define LC_ABL(expr) (false)
#endif

namespace lc {

constexpr int kMaxNeedleAutomaton = 63;  // KMP automaton states must fit a u8 table
constexpr uint32_t kMaxLdsNeedle = 63;   // every automaton needle gets an LDS image: (m + 1) KB at LDS address 0, u16 row
                                         // addresses (the last row of a 63-byte needle starts at 65,024); above 47 bytes a
                                         // workgroup's LDS passes the default 64 KB (launches raise the limit: gfx950 gives
                                         // a workgroup up to 160 KB)
// Per symbol table, k_str_automata emits: the u8 next-state table ((m+1) x 512 bytes) and, for short needles, the
// image the scan kernel copies verbatim to LDS address 0: 2 (m+1) rows x 256 u16 entries holding the LDS byte address
// of the next state's row (rows 0..m: next byte is a code; rows m+1..2m+1: next byte is an escaped literal).
__host__ __device__ inline uint32_t automaton_u8_bytes(uint32_t m) { return (m + 1u) * 512u; }
__host__ __device__ inline uint32_t automaton_image_bytes(uint32_t m) {
    return m <= kMaxLdsNeedle ? (m + 1u) * 1024u : 0u;
}
__host__ __device__ inline uint32_t automaton_stride(uint32_t m) { return automaton_u8_bytes(m) + automaton_image_bytes(m); }
constexpr int kMaxNeedleBytes = 4096;

enum FixedKind : uint8_t { kKindInt = 0, kKindDecimal = 1, kKindF32 = 2, kKindF64 = 3 };

struct alignas(16) FixedDesc {
    const uint8_t* packed;     // 128-byte aligned
    const uint64_t* validity;  // nullptr when the entry has no validity buffer
    const uint64_t* patch_idx; // ALP
    const uint8_t* patch_val;  // ALP
    uint64_t reference;        // FoR reference, sign-extended to 64 bits for signed logical types
    uint64_t mask_word_off;    // first u64 word of this entry's mask segment inside the scan mask
    uint32_t len;
    uint32_t patch_len;
    uint8_t W;                 // 0 => all null
    uint8_t lane_log2;         // 3..6
    uint8_t is_signed;
    uint8_t kind;              // FixedKind
    uint8_t alp_e, alp_f;
    uint8_t value_width;       // bytes of a decoded Arrow value
    uint8_t quantized;         // 1 / 2: the packed values are bucket indices (LiquidPrimitiveQuantizedArray / decimal); the bucket
                               // width (u64) lives in the bits of `patch_idx`, which integer entries do not use.
                               // 0x80 | shift: LiquidFloatQuantizedArray (float_array.rs:742-953), bucket = encoded >> shift
};
static_assert(sizeof(FixedDesc) == 64, "FixedDesc layout");
__host__ __device__ inline uint64_t quant_bucket_width(const FixedDesc& d) { return uint64_t(reinterpret_cast<uintptr_t>(d.patch_idx)); }

struct alignas(16) StrDesc {
    const uint16_t* keys;
    const uint64_t* validity;
    const uint8_t* prefix_keys;
    const uint32_t* fingerprints;  // nullptr when absent
    const uint8_t* residuals;
    const uint8_t* fsst;
    const uint8_t* shared_prefix;
    const uint64_t* signatures;    // bit-sliced bigram signatures: 128 slices x ceil(D/64) u64 words, or nullptr
    uint64_t mask_word_off;
    int32_t slope, intercept;
    uint32_t n, d;
    uint32_t fsst_len, shared_prefix_len;
    uint32_t symtab_slot;
    uint8_t offset_bytes;
    uint8_t multi_empty;           // the dictionary holds MORE than one empty value (values are distinct in the reference's
                                   // dictionaries, so at most one; staged bytes from elsewhere may differ): k_like_scanall's
                                   // end-rank -> value arithmetic needs that, such scans keep k_str_pred
    uint8_t pad[2];
    const uint16_t* postings;      // inverted row lists (see below), or nullptr
};
static_assert(sizeof(StrDesc) == 112, "StrDesc layout");

// Inverted row lists of a byte-view entry (device-side acceleration index like the signatures; entries of up to 65,535 rows
// on substring-search columns): u16 offsets[D + 1], then the VALID rows grouped by dictionary key (u16 row numbers,
// rows[offsets[k] .. offsets[k + 1]) reference key k).  A selective predicate matches one or two dictionary values per
// entry; their rows are then read from these lists (a few bytes) instead of mapping all keys (2 bytes per row) to results —
// what map_dictionary_results_to_array_results (comparisons.rs:325-347) computes, for the matching values only.
constexpr uint32_t kPostMaxRows = 65535;    // entries with more rows carry no lists (u16 offsets count the valid rows)
constexpr uint32_t kPostLdsRows = 8192;     // k_str_pred / k_like_lean keep an entry's mask words in 1 KB of LDS: larger
                                            // entries take the lists only in k_like_flat (mask size per scan); the
                                            // device transcoder sorts an entry's (key, row) pairs in LDS up to this size
constexpr uint32_t kPostMaxMatches = 32;    // more matching dictionary values than this: the keys are mapped instead
constexpr uint32_t kPostLdsBytes = kPostLdsRows / 8 + kPostMaxMatches * 2;  // per wave: mask words + matched keys

// What one workgroup of k_str_pred works on: a run of at most four consecutive entries that share a symbol table (one
// per wave), with a COPY of their descriptors.  The record's address follows from blockIdx alone, so a wave fetches the
// range header and its own descriptor in one round trip instead of range -> descriptor one after the other.
struct alignas(16) StrWgRecord {
    uint32_t begin, end;   // entry range [begin, end), end - begin <= 4
    uint32_t symtab_slot;  // of all its entries
    uint32_t pad;
    StrDesc d[4];          // descriptors of entries begin .. end-1 (zero filled beyond)
};
static_assert(sizeof(StrWgRecord) == 464, "StrWgRecord layout");

// Bigram Bloom signature (device-side acceleration index, built at staging for entries that carry fingerprints):
// bit h(a,b) of a 128-bit set for every pair of adjacent bytes of the dictionary value.  A value can only contain
// `needle` if it has every needle bigram — a necessary condition exactly like the reference's 32-bucket byte
// fingerprint (byte_view_array/fingerprint.rs:33-35), only far more selective.  The signatures are stored BIT-SLICED:
// slice b is a bitmap over the dictionary entries (ceil(D/64) u64 words) telling which values have bit b, so a query
// ANDs only the slices of the needle's bigrams (<= 8 x D/8 bytes) instead of reading 16 bytes per entry.
// Measured with 256 / 512 bits (-DLC_SIG_BITS): 9 -> ~4 / ~1.5 candidates per entry and 5-7 % fewer bytes to move, but the
// headline scan got SLOWER (33 -> 43 us): the index grows from 36 to 72 / 144 KB per entry, the column from 1.5 to 2.0 /
// 3.3 GB, and the five slices a query reads sit up to 140 KB apart — the kernel's dependent loads pay for the larger
// footprint (TLB reach, Infinity Cache residency) more than the walk saves.
#ifndef LC_SIG_BITS
#define LC_SIG_BITS 128
#endif
constexpr int kSigBits = LC_SIG_BITS;
constexpr int kMaxSigProbe = 8;      // slices k_str_pred ANDs (always this many: short needles repeat theirs)
constexpr int kMaxSigProbeWide = 16; // slices k_like_lean ANDs at most (long needles: every further bigram is another factor
                                     // fewer candidates to walk, and a slice is a few hundred bytes per entry)
__host__ __device__ inline uint32_t bigram_bit(uint32_t a, uint32_t b) {
    return ((((a << 8) | b) * 40503u) >> 7) & uint32_t(kSigBits - 1);
}

// Symbol table as the kernels see it.
struct DevSymtab {
    uint64_t sym[256];
    uint8_t len[256];
};

// On-device transcoder (k_col_minmax / k_fl_pack): one array to encode.
struct EncodeDesc {
    const uint8_t* values;     // native-width values on the device
    const uint64_t* validity;  // u64 words, or null
    uint8_t* packed;           // output of the pack phase (ceil(n/1024) blocks of 128*W bytes)
    uint64_t* validity_out;    // pack phase: the validity words are copied here (or null)
    uint64_t reference;        // pack phase: frame of reference (sign-extended for signed types)
    uint64_t clamp_max;        // pack phase: offsets are clamped to this value first (clamp squeeze); 0: no clamp
    uint64_t quant_width;      // pack phase: offsets are divided by this bucket width first (quantize squeeze); 0/1: no
    uint32_t n;
    uint8_t W;                 // pack phase: bit width (0: nothing to pack)
    uint8_t value_log2;        // 0..3: bytes per value = 1 << value_log2
    uint8_t is_signed;
    uint8_t stride_log2;       // 0: values are dense; 4 / 5: Decimal128 / Decimal256 values, the low u64 of every 16 / 32
                               // bytes is the value (fits_u64, decimal_array.rs:120-125), null slots pack 0
    uint8_t fq_shift;          // pack phase, float Quantize squeeze (float_array.rs:357-369): the values are packed-domain offsets
                               // of an ALP array; what is packed is ((fq_ref + v) >> fq_shift) - (fq_ref >> fq_shift), signed
                               // arithmetic in the lane width; 0: no such transform
    uint8_t pad[3];
    uint64_t fq_ref;           // the array's reference (sign-extended)
};
struct EncodeMinMax {
    uint64_t mn, mx;  // as int64 bits for signed types
    uint32_t n_valid;
    uint32_t n_wide;  // decimals: valid values that do not fit a u64 (negative or wider): the array stays on the CPU path
};

// Internal operator: "packed value == all ones" — the sentinel rows of a clamp-squeezed entry
// (LiquidPrimitiveClampedArray, hybrid_primitive_array.rs:129-146, :194-196).
#define LC_OP_INTERNAL_SENTINEL 100

// Integer-domain predicate after host normalisation of the literal.
struct FixedPred {
    int32_t op;           // LC_OP_EQ..LC_OP_GE
    int32_t lit_class;    // -1: literal below every representable value, +1: above, 0: `lit` is exact
    uint64_t lit;         // int64 bits (signed logical types) or uint64
    uint32_t lit_f32;     // float predicates: raw IEEE bits of the literal
    int32_t inner_op;     // op == LC_OP_INTERNAL_SENTINEL over quantized entries: the comparison whose undecidable
                          // rows (bucket of the literal) are looked for
    uint64_t lit_f64;
    uint32_t debug_flags; // profiling builds (-DLC_ABLATION) only: LC_DEBUG_FLAGS
    uint32_t pad;
};

// a conjunction over several fixed-width columns in one launch (k_fixed_chain)
constexpr int kMaxChainSteps = 6;
struct FixedChainStep {
    const FixedDesc* descs;  // the column's descriptors, same entry order and row ranges for every step
    FixedPred pred, pred2;   // pred2.op < 0: absent
    int32_t lane_log2;       // 4..6
    int32_t pad;
};
struct FixedChainArgs {
    FixedChainStep step[kMaxChainSteps];
    uint32_t n_steps;
    uint32_t pad;
};

constexpr int kInlineNeedle = 64;

struct StrPred {
    int32_t op;            // LC_OP_*
    int32_t mode;          // 0: Eq/Ne/ordering on `needle`, 1: substring automaton (LIKE %needle%), 2: constant,
                           // 3: general LIKE pattern in `needle` (entries without fingerprints)
    uint32_t needle_len;
    int32_t use_fingerprints;  // LIKE: prune with fingerprints (and apply the reference's candidate quirk)
    const uint8_t* needle;     // device copy when needle_len > kInlineNeedle (padded by 8 bytes)
    const uint8_t* automata;   // mode 1: per symbol-table slot, automaton_stride(m) bytes (see above)
    uint32_t automaton_stride;
    int32_t const_value;       // mode 2: Literal(Boolean)
    int32_t debug_flags;       // -DLC_ABLATION builds only (env LC_DEBUG_FLAGS): 1 skip phase B, 2 skip phase C, 8 no signatures
    uint32_t needle_fp;        // LIKE: 32-bucket fingerprint of the needle (fingerprint.rs:33-35)
    uint32_t n_sig_bits;       // LIKE: distinct bigram-signature bits of the needle that are probed (<= kMaxSigProbe)
    uint16_t sig_bits[kMaxSigProbe];
    // the same list continued (k_like_lean): n_sig_wide >= n_sig_bits distinct bits, the first n_sig_bits are sig_bits.
    // The bigrams are taken in an order that spreads them over the whole needle (ends first, then midpoints), so that
    // whatever prefix of the list a kernel uses covers the needle evenly ('%yandex.ru/search%': its first 8 bigrams only
    // say "yandex.ru", which 45 % of the values of a URL column contain).
    uint32_t n_sig_wide;
    uint16_t sig_wide[kMaxSigProbeWide];
    // mode 1, needles over kMaxNeedleAutomaton bytes: the automaton runs over the needle's FIRST kMaxNeedleAutomaton bytes
    // (needle_len says so) — a necessary condition like the prefilters, whose bits come from the whole needle — and the
    // dictionary values it accepts are then matched exactly against the pattern (`needle`: the literal '%...%', verify_len
    // bytes, device copy).  0: the automaton's answer is exact.
    uint32_t verify_len;
    // mode 1 through k_like_flat only: != 0 makes the evaluation `=` / `<>` on the needle (its length) instead of [NOT] LIKE —
    // a value equals the needle exactly when it contains it and has its length (lc_runtime.cpp: eq_via_index)
    uint32_t eq_len;
    uint8_t needle_inline[kInlineNeedle];
};

constexpr uint32_t kWorkGroupsMax = 16384;

struct ScanLaunch {
    uint32_t n_entries;
    uint32_t blocks_per_entry;  // ceil(max entry len / 1024)
    const uint64_t* d_selection;
    uint64_t* d_hit;      // pred & valid & selected
    uint64_t* d_valid;    // optional: valid & selected
    uint32_t* d_counts;   // optional, must be zeroed by the launcher
    uint32_t* d_cand_bytes;  // optional (byte views): per entry, compressed bytes of the reference's fingerprint candidates
    uint32_t* d_own_bytes;   // optional (byte views): per entry, bytes this kernel had to read + write for the entry
    uint32_t max_dict_len;   // byte views: largest dictionary in the scan (sizes the LDS result bitmap)
    int32_t uniform_slot;    // byte views: symbol-table slot shared by every entry of the scan, or -1
    uint32_t* d_work;        // byte views: kWorkGroupsMax x {next, finished waves} at a 64-byte stride, zero between
                             // launches (self-resetting)
    uint32_t work_groups;    // byte views: number of counter groups used by this launch (set by the launcher)
    uint32_t n_wg_ranges;    // byte views: entries of d_wg_ranges (0: entries are split evenly over the groups)
    const StrWgRecord* d_wg_ranges;  // byte views: one record per workgroup; a range never mixes symbol tables
    uint32_t many_candidates;     // byte views: some entry has no bigram signature index (LIKE walks whole dictionaries)
    uint32_t acct_postings;       // byte-accounting pass: the launch it accounts for reads rows through the inverted lists
    // Fused COUNT(*) of the launch (optional): every wave adds the hits of its entries to a sharded accumulator and the
    // wave that arrives last writes the total to *d_total_out — no separate reduction kernel, no memset between launches.
    unsigned long long* d_total_acc;  // kTotalWords u64 owned by the scan, zero between launches (self-resetting)
    uint64_t* d_total_out;            // null: no total wanted
    // Sparse result (optional; lc_scan_eval_hits): every hit row as (entry << 32 | row) appended to d_hits — the rows of one
    // entry contiguous and ascending, entries in no particular order.  *d_n_hits (zero before the launch) receives the number
    // of hits, which may exceed hits_cap (records beyond it are dropped).  d_hit_first (optional): per entry WITH hits, the
    // index of its first record.  Kernels that do not emit the list themselves leave it to k_mask_to_hits.
    uint64_t* d_hits;
    uint64_t hits_cap;
    unsigned long long* d_n_hits;
    uint32_t* d_hit_first;
    uint32_t mask_optional;           // d_hit is the scan's own scratch (the caller wants no mask): a kernel that can answer
                                      // d_counts / d_total_out / d_hits without storing mask words may skip them
    uint32_t entry_split_log2;        // fixed width, register-resident kernels: an entry is worked on by 2^this waves, each taking
                                      // a run of its 1024-row blocks (set by the launcher; 0 when per-entry counts are asked for)
    uint32_t hits_parts;              // 0 / 1: d_hits is ONE list with ONE counter; kHitParts: the PARTITIONED form (below)
    uint32_t uniform_w;               // fixed width: every entry of the scan with packed data has this width (0: mixed widths)
};
// The partitioned hit list (LC_HITS_PARTITIONED, round 6).  A contiguous list is allocated by returning atomics on ONE address,
// which complete ~10 ns apart however many workgroups wait: 1,100 of them were 11 of the 23.7 us of a selective LIKE with a
// list.  Partitioned, workgroup b appends to partition b % kHitParts: region p = records [p S, (p + 1) S) with S = capacity /
// kHitParts, its counter the u64 at n_hits[p * kHitCounterStride] (a 128-byte line of its own).  Consumers see the partitions
// as ONE list in partition order (hitlist_prefix / hitlist_at in lc_device.hpp): their outputs are as dense as before.
constexpr uint32_t kHitParts = 16;
constexpr uint32_t kHitCounterStride = 16;  // u64 words between two partitions' counters
// accumulator layout: word 0 = top level, words 8, 16, ... = shards (one 64-byte line each).  A word packs
// {arrivals : 24 | hits : 40}, so ONE returning atomic both adds a count and tells the caller whether it was the last.
constexpr uint32_t kTotalShards = 64;
constexpr uint32_t kTotalWords = 8 * (kTotalShards + 1);

// pred2 (optional): a second conjunct on the same column, fused into the same pass (both must be Eq/Lt/LtEq/Gt/GtEq);
// max_width: largest bit width among the scan's entries (<= 32 selects the register-resident kernel)
hipError_t launch_fixed_pred(const FixedDesc* d_descs, int lane_log2, const FixedPred& pred, const FixedPred* pred2,
                             uint32_t max_width, const ScanLaunch& L, hipStream_t stream);
hipError_t launch_alp_patch_fix(const FixedDesc* d_descs, int lane_log2, const FixedPred& pred, const FixedPred* pred2,
                                const ScanLaunch& L, hipStream_t stream);
// float-quantized entries of a scan (FixedDesc::quantized & 0x80; the other kernels leave their mask words zero): the
// reference's bucket-bound decision per row, patches by value; d_undecided[entry] != 0: some valid selected unpatched row
// could not be decided (Err(NeedsBacking)).  Adds the entries' hits to d_counts / *d_total_out.
hipError_t launch_float_quant_pred(const FixedDesc* d_descs, int lane_log2, const FixedPred& pred, const ScanLaunch& L,
                                   uint32_t* d_undecided, hipStream_t stream);
hipError_t launch_str_entry_offsets(const StrDesc* d_descs, const ScanLaunch& L, uint32_t* d_entry_counts, uint64_t* d_tiles,
                                    uint64_t* d_entry_row_offsets, hipStream_t stream);
hipError_t launch_str_sel_rows(const StrDesc* d_descs, const DevSymtab* d_symtabs, const ScanLaunch& L,
                               const uint64_t* d_entry_row_offsets, uint64_t capacity, uint64_t k, uint64_t* d_row_refs,
                               uint32_t* d_row_len, uint8_t* d_row_valid, uint64_t* d_tiles, uint64_t* d_value_offsets,
                               hipStream_t stream);
hipError_t launch_str_decode_sel(const StrDesc* d_descs, const DevSymtab* d_symtabs, const uint64_t* d_row_refs,
                                 const uint64_t* d_value_offsets, uint64_t k, const uint64_t* d_k, uint64_t capacity_rows,
                                 uint64_t capacity_bytes, uint8_t* d_data, hipStream_t stream);
hipError_t launch_date_lossy(void* d_values, uint64_t n, int value_width, int field, int64_t ticks_per_day,
                             hipStream_t stream);
hipError_t launch_str_automata(const DevSymtab* d_symtabs, uint32_t n_symtabs, const uint8_t* needle,
                               uint32_t needle_len, uint8_t* d_automata, hipStream_t stream);
hipError_t launch_str_pred(const StrDesc* d_descs, const DevSymtab* d_symtabs, const StrPred& pred,
                           const ScanLaunch& L, hipStream_t stream);
// [NOT] LIKE '%needle%' with many candidates: the whole FSST buffer of every entry streamed once, lane per 8-byte word
// (lc_like_scanall.hip); d_recs: the scan's workgroup records (<= 4 entries of one symbol table each)
// the workgroup records of a byte-view scan (StrWgRecord) from its descriptors: record r covers entries [begins[r], begins[r + 1])
hipError_t launch_str_wg_records(const StrDesc* d_descs, const uint32_t* d_begins, uint32_t n_recs, StrWgRecord* d_recs, hipStream_t stream);
// (uni_slice / uni_word: a 1-byte needle answered from the scan-level unigram index instead of a walk, see ScanAllArgs)
hipError_t launch_like_scanall(const StrWgRecord* d_recs, uint32_t n_recs, const StrPred& pred, const ScanLaunch& L,
                               unsigned long long* d_total_acc, hipStream_t stream, const uint64_t* uni_slice = nullptr,
                               const uint32_t* uni_word = nullptr);
// per-block selected-row counts -> exclusive offsets, then compaction of decoded values
// d_block_counts: n_entries*blocks_per_entry u32; d_block_offsets: that + 1 u64; d_entry_row_offsets: n_entries + 1 u64
// u64 elements the caller provides for d_block_offsets: n_blocks + 1 offsets followed by the scan's tile sums
inline size_t fixed_gather_offsets_len(size_t n_blocks) { return n_blocks + 1 + (n_blocks + 1023) / 1024 + 1; }
// COUNT / SUM / MIN / MAX of the valid selected rows of a fixed-width scan; d_partials: 48 bytes of scratch per workgroup
// (fixed_agg_workgroups); d_out: six u64 {count, sum lo, sum hi (two's complement i128), min, max, 0}
constexpr size_t kAggPartialBytes = 48;
uint32_t fixed_agg_workgroups(uint32_t n_entries, int lane_log2);
// SUM(a * b) over rows selected and valid in both columns (same lane type); d_out as launch_fixed_agg (min / max are 0)
hipError_t launch_fixed_sum_product(const FixedDesc* d_descs_a, const FixedDesc* d_descs_b, int lane_log2, const ScanLaunch& L,
                                    void* d_partials, uint64_t* d_out, hipStream_t stream);
hipError_t launch_fixed_agg(const FixedDesc* d_descs, int lane_log2, int is_signed, const ScanLaunch& L, void* d_partials,
                            uint64_t* d_out, hipStream_t stream);
hipError_t launch_fixed_chain(const FixedChainArgs& chain, uint32_t max_width, const uint32_t* col_max_w, const ScanLaunch& L,
                              hipStream_t stream);
hipError_t launch_fixed_gather(const FixedDesc* d_descs, int lane_log2, const ScanLaunch& L, uint32_t* d_block_counts,
                               uint64_t* d_block_offsets, uint64_t* d_entry_row_offsets, uint8_t* d_values_out,
                               uint64_t capacity_rows, hipStream_t stream);
// bit compress (PEXT) / deposit (PDEP) per entry segment
// (hit, valid) <- Kleene OR with (hit_b, valid_b); d_valid may be null (hit only)
hipError_t launch_mask_or_kleene(uint64_t* d_hit, uint64_t* d_valid, const uint64_t* d_hit_b, const uint64_t* d_valid_b,
                                 uint64_t n_words, hipStream_t stream);
// per-entry popcounts of the mask passed as L.d_selection
hipError_t launch_mask_entry_counts(const void* d_descs, bool is_str, const ScanLaunch& L, uint32_t* d_entry_counts,
                                    hipStream_t stream);
// bigram signature slices of freshly staged byte-view entries (descs[i].signatures: zeroed kSigBits x ceil(D/64) words)
hipError_t launch_str_build_signatures(const StrDesc* d_descs, uint32_t n_entries, uint32_t max_dict_len,
                                       const DevSymtab* d_symtabs, hipStream_t stream);
// On-device ALP encoder (floats): exponent search, encode + exceptions, patch placement.  d_stats: one AlpStatsHost per array.
struct AlpStatsHost {
    uint32_t e, f, n_exc, pad;
    int64_t mn, mx;
};
hipError_t launch_alp_search(const EncodeDesc* d_descs, uint32_t n_arrays, int value_log2, void* d_stats, hipStream_t stream);
hipError_t launch_alp_encode(const EncodeDesc* d_descs, uint32_t n_arrays, int value_log2, void* d_stats, uint32_t stride,
                             void* d_enc, uint64_t* d_exc_idx, void* d_exc_val, hipStream_t stream);
hipError_t launch_alp_copy_patches(const void* d_stats, uint32_t n_arrays, int value_log2, uint32_t stride,
                                   const uint64_t* d_exc_idx, const void* d_exc_val, void* const* d_dst_idx,
                                   void* const* d_dst_val, uint32_t max_exc, hipStream_t stream);
hipError_t launch_col_minmax(const EncodeDesc* d_descs, uint32_t n_entries, EncodeMinMax* d_out, hipStream_t stream);
hipError_t launch_fl_pack(const EncodeDesc* d_descs, uint32_t n_entries, uint32_t max_rows, int lane_log2, hipStream_t stream);
// date / timestamp values -> one calendar component (i32), and its lossy reconstruction in the original Arrow type
hipError_t launch_date_component(const void* d_values, uint64_t n, int value_width, int field, int64_t ticks_per_day,
                                 int32_t* d_out, hipStream_t stream);
hipError_t launch_component_lossy(const int32_t* d_comps, uint64_t n, int value_width, int field, int64_t ticks_per_day,
                                  void* d_out, hipStream_t stream);
// dst[i] = src[i] for n_words u64 words; either side may be pinned host memory (results of the per-entry calls)
hipError_t launch_copy_words(void* dst, const void* src, uint64_t n_words, hipStream_t stream);
hipError_t launch_mask_compress(const uint64_t* d_src, const uint64_t* d_sel, const uint64_t* d_seg_offsets,
                                uint32_t n_entries, uint64_t* d_out, uint32_t* d_out_bits, hipStream_t stream);
hipError_t launch_mask_and_then(const uint64_t* d_left, uint64_t left_bits, const uint64_t* d_right, uint64_t* d_out,
                                hipStream_t stream);
// byte-view get-with-selection of ONE entry: d_dict_len (D u32 scratch), d_offsets (n+1 i32), d_rows (n u32 scratch),
// d_totals (2 u64: selected rows, data bytes), d_data (>= uncompressed bytes of the referenced values + 8)
// partial GROUP BY over dictionary keys (lc_groupby.hip)
hipError_t launch_group_partials(const StrDesc* g_descs, const StrDesc* v_descs, const DevSymtab* symtabs, const uint64_t* selection,
                                 uint32_t n_entries, int want_max, lc_group_partial* out, uint64_t capacity,
                                 unsigned long long* n_out, hipStream_t stream);
// hit lists (sparse results): mask -> list, and get-with-selection for the rows of a list in ONE launch each
// one per translation unit with kernels: makes the runtime load the unit's code object now instead of at a query's first launch
hipError_t warm_code_object_kernels();
hipError_t warm_code_object_like_pipeline();
hipError_t warm_code_object_like_scanall();
hipError_t warm_code_object_groupby();
hipError_t warm_code_object_bv_encode();
// `bytes` (a multiple of 4, <= 4096) at p <- 0 by one wave: the counters of a sparse query.  (hipMemsetAsync's fill kernel takes
// 4-5 us for 32 bytes in front of a 10 us kernel.)
hipError_t launch_zero_small(void* p, uint32_t bytes, hipStream_t stream);
hipError_t launch_group_counts(const uint32_t* d_entry_counts, const uint32_t* d_group_ends, uint32_t n_groups, uint64_t* d_out,
                               hipStream_t stream);
// (`parts`: 1 = contiguous list, kHitParts = partitioned; the list's capacity is `cap` records either way)
hipError_t launch_mask_to_hits(const void* d_descs, bool is_str, uint32_t n_entries, const uint64_t* d_mask, uint64_t* d_hits,
                               uint64_t cap, unsigned long long* d_n_hits, uint32_t* d_hit_first, uint32_t parts, hipStream_t stream);
hipError_t launch_fixed_gather_hits(const FixedDesc* d_descs, int lane_log2, const uint64_t* d_hits,
                                    const unsigned long long* d_n_hits, uint64_t cap, uint8_t* d_values_out, uint8_t* d_row_valid,
                                    uint32_t parts, hipStream_t stream);
hipError_t launch_str_gather_hits(const StrDesc* d_descs, const DevSymtab* d_symtabs, const uint64_t* d_hits,
                                  const unsigned long long* d_n_hits, uint64_t cap_rows, uint32_t* d_views, uint8_t* d_row_valid,
                                  uint8_t* d_data, uint64_t cap_bytes, unsigned long long* d_n_bytes, bool slotted, uint32_t parts,
                                  hipStream_t stream);
// the partitions of a list in partition order -> one contiguous list (+ its count)
hipError_t launch_hits_compact(const uint64_t* d_hits, const unsigned long long* d_n_hits, uint64_t cap, uint64_t* d_out, uint64_t cap_out,
                               unsigned long long* d_n_out, hipStream_t stream);
// a predicate over the rows of a hit list (k_pred_hits): lane_log2 0 = byte views (op on the literal / pattern bytes; lit_len <=
// kInlineNeedle travels in the kernel arguments from h_lit, longer literals from the device copy d_lit), 3..6 = fixed width
struct HitsPredLaunch {
    const void* descs;
    const DevSymtab* symtabs;
    const uint64_t* hits_in;
    const unsigned long long* n_in;
    uint64_t cap_in;
    uint64_t* hits_out;
    uint64_t cap_out;
    unsigned long long* n_out;
    uint32_t parts;      // 1 / kHitParts: layout of BOTH lists
    int32_t lane_log2;
    int32_t op;
    int32_t const_value;
    int32_t substring;   // [NOT] LIKE with a plain '%needle%' pattern: the literal passed is the needle (byte-wise contains)
    uint32_t lit_len;
    const uint8_t* h_lit;
    const uint8_t* d_lit;
    FixedPred fp;
};
hipError_t launch_pred_hits(const HitsPredLaunch& h, hipStream_t stream);
hipError_t launch_str_gather(const StrDesc* d_descs, const DevSymtab* d_symtabs, uint32_t entry, uint32_t dict_len,
                             uint32_t n_rows, const uint64_t* d_selection, uint32_t* d_dict_len, int32_t* d_offsets,
                             uint32_t* d_rows, uint64_t* d_totals, uint8_t* d_data, hipStream_t stream);

// ---- on-device byte-view transcoder (lc_bv_encode.hip) ----
// (DevFsstEncoder, its matcher and the per-value compression loop: lc_fsst_device.hpp, shared with the CPU model test)
struct BvEncodeStats {
    uint64_t raw_bytes;       // sum of the dictionary values' lengths
    uint32_t d;               // dictionary values
    uint32_t fsst_len;        // compressed bytes
    uint32_t shared_prefix_len;
    int32_t slope, intercept; // compact offsets
    uint32_t offset_bytes;    // 1 / 2 / 4
    uint32_t inexact;         // the f64 sums of the reference's line fit would round (> 2^53): host path
    uint32_t pad;
};
struct BvEncodeDesc {  // one Utf8 / Binary array (i32 offsets)
    const int32_t* offsets;    // n + 1, as in the Arrow buffer (not rebased)
    const uint8_t* data;       // bytes from offsets[0] on, 16 readable bytes behind the end
    const uint64_t* validity;  // u64 words, or null
    uint32_t* table;           // open addressing, table_mask + 1 slots preset to 0xFFFFFFFF
    uint32_t* row_slot;        // n
    uint32_t* dict_row;        // n: first row of dictionary value k
    uint32_t* dict_index;      // n: dictionary index of a first-occurrence row
    uint32_t* clen;            // n: compressed length of value k
    uint32_t* offsets_out;     // n + 1: compressed offsets
    uint32_t* fingerprints;    // n
    uint16_t* keys;            // n
    uint8_t* comp;             // 2 * data bytes + 16: value k compressed at twice its input position
    BvEncodeStats* stats;
    uint32_t n, table_mask, encoder, pad;
};
struct BvPackDesc {  // where k_bv_pack writes the entry (the sections of StrDesc, writable)
    uint16_t* keys;
    uint64_t* validity;        // null: not nullable
    uint8_t* prefix_keys;
    uint32_t* fingerprints;    // null: none
    uint8_t* residuals;
    uint8_t* fsst;
    uint8_t* shared_prefix;
    uint16_t* postings;        // null: no row lists
    uint32_t d, shared_prefix_len, offset_bytes;
    int32_t slope, intercept;
    uint32_t pad;
};
hipError_t launch_bv_build(const BvEncodeDesc* d_descs, uint32_t n_arrays, const DevFsstEncoder* d_encoders, hipStream_t stream);
hipError_t launch_bv_pack(const BvEncodeDesc* d_descs, const BvPackDesc* d_packs, uint32_t n_arrays, hipStream_t stream);

}  // namespace lc
