// Partial GROUP BY over dictionary keys with MIN / MAX over byte views — the step after the path for ClickBench q21.sql
//   SELECT "SearchPhrase", MIN("URL"), COUNT(*) FROM hits WHERE "URL" LIKE '%google%' AND "SearchPhrase" <> '' GROUP BY ...
// What DataFusion's AggregateExec(mode = Partial) computes from the rows `get().with_selection()` returns (the plan shape of
// src/datafusion-local/src/tests/snapshots/*url_prefix_filtering.snap), here without returning them: per ENTRY (one 8192-
// row batch = one dictionary per column) the selected rows are grouped by the group column's DICTIONARY KEY — equal keys
// are equal strings inside a batch (LiquidByteViewArray keeps a unique dictionary, byte_view_array/conversions.rs:260-373) —
// and every group keeps COUNT(*) and the row that holds the smallest (or largest) value of the value column.  A partial is
//   { entry, a row of the group (its group value), count, the row holding the MIN / MAX value | none }
// Rows, not keys: the caller decodes both strings with the gather it already has (lc_scan_gather_bytes) and the final
// aggregate merges partials of different entries by value, exactly as DataFusion's final AggregateExec merges partitions.
//
// Values are compared WITHOUT decoding them to memory: first the 7-byte prefix keys the entry carries for ordering
// predicates (comparisons.rs:361-404: shared prefix stripped, byte-wise order == numeric order of the byte-swapped word),
// and only on a tie a lock-step walk of the two FSST code streams (raw/fsst_buffer.rs:642-663).
//
// One wave per entry.  The groups of an entry live in an LDS hash table keyed by the group key (claimed with LDS
// compare-and-swap by the lanes of a 64-row batch in parallel); an entry whose selected rows hold more distinct groups
// than the table emits what it has and goes on — duplicate (entry, group) partials are still partials.
#include <algorithm>
#include <cstdio>
#include <atomic>

#include "lc_device.hpp"
#include "lc_internal.hpp"

namespace lc {
namespace {

constexpr uint32_t kGroupSlots = 1024;            // hash table slots per wave (power of two)
constexpr uint32_t kGroupFlushAt = 704;           // distinct groups after which the table is emitted before the next batch
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;       // slot is free
constexpr uint32_t kNullGroup = 0xFFFFFFFEu;      // the group of rows whose group value is NULL
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

struct GroupArgs {
    const StrDesc* g_descs;
    const StrDesc* v_descs;  // null: COUNT only
    const DevSymtab* symtabs;
    const uint64_t* selection;
    uint32_t n_entries;
    int want_max;
    lc_group_partial* out;
    uint64_t capacity;
    unsigned long long* n_out;  // partials appended (may exceed capacity: the caller retries with a larger buffer)
};

// three-way compare of dictionary values ka, kb of ONE entry: < 0, 0, > 0
__device__ __noinline__ int dict_compare(const StrDesc& d, const DevSymtab& st, uint32_t ka, uint32_t kb) {
    if (ka == kb) return 0;
    const uint64_t pa = reinterpret_cast<const uint64_t*>(d.prefix_keys)[ka], pb = reinterpret_cast<const uint64_t*>(d.prefix_keys)[kb];
    // prefix key: bytes 0..6 = the first 7 bytes after the entry's shared prefix (zero padded), byte 7 = that length (255:
    // longer).  The first m = min(both lengths, 7) bytes decide if they differ; if they agree and one value ends there it
    // is the smaller one; only two values that agree on 7 bytes need the code streams.
    const uint32_t la = uint32_t(pa >> 56), lb = uint32_t(pb >> 56);
    const uint32_t m = min(min(la, lb), 7u);
    if (m > 0) {
        const uint64_t low56 = 0x00FFFFFFFFFFFFFFull;
        const uint64_t am = __builtin_bswap64(pa & low56) >> (8u * (8u - m)), bm = __builtin_bswap64(pb & low56) >> (8u * (8u - m));
        if (am != bm) return am < bm ? -1 : 1;
    }
    if (m < 7u) return la == lb ? 0 : (la < lb ? -1 : 1);
    uint32_t sa, ea, sb, eb;
    str_offset_pair(d, ka, sa, ea);
    str_offset_pair(d, kb, sb, eb);
    // Equal compressed bytes decode to equal bytes: the common compressed prefix is skipped eight codes per load, and only
    // what follows is decoded (the values of a group are URLs of one site: the decoding iterator costs three dependent
    // loads per code, 20-40 us for two 76-byte values — the slowest wave's two such compares were most of this kernel).
    // The skip never stops inside an escape pair (marker 255 + literal): an odd run of 255s before the stop gives one back.
    {
        const uint32_t mc = min(ea - sa, eb - sb);
        uint32_t common = 0;
        while (common + 8u <= mc) {
            const uint64_t wa = load_unaligned<uint64_t>(d.fsst + sa + common), wb = load_unaligned<uint64_t>(d.fsst + sb + common);
            if (wa != wb) { common += uint32_t(__builtin_ctzll(wa ^ wb)) >> 3; break; }
            common += 8u;
        }
        uint32_t run = 0;
        while (run < common && d.fsst[sa + common - 1u - run] == 255u) run++;
        common -= run & 1u;
        sa += common;
        sb += common;
    }
    FsstIter ia{sa, ea, 0, 0, 0, false}, ib{sb, eb, 0, 0, 0, false};
    fsst_iter_load(ia, st, d.fsst);
    fsst_iter_load(ib, st, d.fsst);
    while (!ia.at_end && !ib.at_end) {
        const uint32_t ca = fsst_iter_cur(ia), cb = fsst_iter_cur(ib);
        if (ca != cb) return ca < cb ? -1 : 1;
        fsst_iter_next(ia, st, d.fsst);
        fsst_iter_next(ib, st, d.fsst);
    }
    return ia.at_end ? (ib.at_end ? 0 : -1) : 1;
}

// One wave per entry, two per workgroup (2 x 16 KB of tables), a persistent grid: the waves stride over the entries.  A
// selective filter leaves five entries in six without a row (a wave reads their 128 selection words and moves on); an entry
// that has one is a chain of ~8 dependent round trips (descriptors, selection words, validity and keys of both columns, the
// returning append to the output), which is what the kernel's time is made of (q21.sql over 100 M rows: 2,153 rows in
// ~2,000 entries, 80 us; until the end of round 4 one 64-thread workgroup per entry and one load per selection WORD: 98 us).
constexpr uint32_t kGroupWaves = 2;
__global__ __launch_bounds__(kGroupWaves * 64) void k_group_partials(GroupArgs a) {
    __shared__ uint32_t s_key[kGroupWaves][kGroupSlots];
    __shared__ uint32_t s_row[kGroupWaves][kGroupSlots];    // a row of the group
    __shared__ uint32_t s_cnt[kGroupWaves][kGroupSlots];
    __shared__ uint32_t s_best[kGroupWaves][kGroupSlots];   // row holding the MIN / MAX value so far, kNoRow: none yet
    __shared__ uint32_t s_distinct[kGroupWaves];
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    // (the wave's tables by name, not through pointers: a generic pointer to LDS makes every access a flat one, and a flat
    // access waits for every outstanding global load as well)
#define t_key s_key[wave]
#define t_row s_row[wave]
#define t_cnt s_cnt[wave]
#define t_best s_best[wave]
#define n_distinct s_distinct[wave]
    for (uint32_t entry = blockIdx.x * kGroupWaves + wave; entry < a.n_entries; entry += gridDim.x * kGroupWaves) {
    // (by value: through a reference every use of a pointer field is another global load in front of the load it is for —
    // the stores and atomics in between keep the compiler from holding it in a register)
    const StrDesc g = a.g_descs[entry];
    const uint32_t nwords = (g.n + 63u) >> 6;
    {   // anything selected in this entry?
        uint32_t any = 0;
        for (uint32_t w = uint32_t(lane); w < nwords; w += kWave) {
            uint64_t sw = a.selection ? a.selection[g.mask_word_off + w] : ~uint64_t(0);
            if (w == nwords - 1u && (g.n & 63u)) sw &= (uint64_t(1) << (g.n & 63u)) - 1;
            any |= sw != 0;
        }
        if (__ballot(any != 0) == 0) continue;
    }
    const bool has_v = a.v_descs != nullptr;
    StrDesc vd{};
    if (has_v) vd = a.v_descs[entry];
    const StrDesc* v = has_v ? &vd : nullptr;
    const DevSymtab* vst = has_v ? a.symtabs + vd.symtab_slot : nullptr;
    auto clear_table = [&]() {
        for (uint32_t i = uint32_t(lane); i < kGroupSlots; i += kWave) t_key[i] = kEmptyKey;
        if (lane == 0) n_distinct = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto emit_table = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i0 = 0; i0 < kGroupSlots; i0 += kWave) {
            const uint32_t i = i0 + uint32_t(lane);
            const bool used = t_key[i] != kEmptyKey;
            const uint64_t um = __ballot(used);
            if (um == 0) continue;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(a.n_out, (unsigned long long)__popcll(um));
            base = uniform_u64(base);
            const uint64_t pos = base + lanes_below(um);
            if (used && pos < a.capacity) a.out[pos] = lc_group_partial{entry, t_row[i], t_cnt[i], t_best[i]};
        }
    };
    clear_table();
    // 64 selection words at a time, one per lane, and only the non-empty ones are visited: a selective filter leaves one
    // or two words of an entry with a row (one dependent load per WORD of the entry, 128 of them, made this kernel 98 us
    // for the 2,153 rows of q21.sql over 100 M)
    for (uint32_t wb = 0; wb < nwords; wb += kWave) {
      const uint32_t wl = wb + uint32_t(lane);
      uint64_t sw_l = 0;
      if (wl < nwords) {
          sw_l = a.selection ? a.selection[g.mask_word_off + wl] : ~uint64_t(0);
          if (wl == nwords - 1u && (g.n & 63u)) sw_l &= (uint64_t(1) << (g.n & 63u)) - 1;
      }
      uint64_t nz = __ballot(sw_l != 0);
      while (nz) {
        const int src = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)nz)) - 1);
        nz &= nz - 1;
        const uint32_t w = wb + uint32_t(src);
        const uint64_t sw = uniform_u64(uint64_t(uint32_t(__shfl(int(uint32_t(sw_l)), src, kWave))) |
                                        (uint64_t(uint32_t(__shfl(int(uint32_t(sw_l >> 32)), src, kWave))) << 32));
        if (n_distinct >= kGroupFlushAt) {  // (read after the fence of the previous batch: wave uniform)
            emit_table();
            clear_table();
        }
        const bool active = (sw >> lane) & 1u;
        const uint32_t row = w * 64u + uint32_t(lane);
        uint32_t gk = kNullGroup, vk = kEmptyKey;
        {
            // the four loads of a row are requested together (validity word and key of both columns): one round trip
            const uint32_t rc = min(row, g.n - 1u);
            const uint64_t gvw = g.validity ? g.validity[w] : ~uint64_t(0);
            const uint32_t gkr = g.keys[rc];
            const uint64_t vvw = (has_v && vd.validity) ? vd.validity[w] : ~uint64_t(0);
            const uint32_t vkr = has_v ? uint32_t(vd.keys[rc]) : 0u;
            if (active && ((gvw >> lane) & 1u)) gk = gkr;
            if (active && has_v && ((vvw >> lane) & 1u)) vk = vkr;
        }
        if (active) {
            // claim / find the group's slot
            uint32_t h = (gk * 2654435761u) >> 22;  // 10 bits
            for (;;) {
                const uint32_t old = atomicCAS(&t_key[h], kEmptyKey, gk);
                if (old == kEmptyKey) {
                    t_row[h] = row;
                    t_cnt[h] = 0;
                    t_best[h] = kNoRow;
                    atomicAdd(&n_distinct, 1u);
                    break;
                }
                if (old == gk) break;
                h = (h + 1u) & (kGroupSlots - 1u);
            }
            // (the lane that claimed the slot initialises it before anybody adds to it: same wave, program order per
            // lane, and the adds below come after a wave-level fence)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            atomicAdd(&t_cnt[h], 1u);
            atomicMin(&t_row[h], row);
            if (vk != kEmptyKey) {
                for (;;) {
                    const uint32_t cur = t_best[h];
                    if (cur != kNoRow) {
                        const int c = dict_compare(*v, *vst, vk, uint32_t(v->keys[cur]));
                        // ties keep the earlier row (deterministic partials)
                        const bool better = a.want_max ? c > 0 : c < 0;
                        if (!better && !(c == 0 && row < cur)) break;
                    }
                    if (atomicCAS(&t_best[h], cur, row) == cur) break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    emit_table();
    }
#undef t_key
#undef t_row
#undef t_cnt
#undef t_best
#undef n_distinct
}

}  // namespace

hipError_t launch_group_partials(const StrDesc* g_descs, const StrDesc* v_descs, const DevSymtab* symtabs, const uint64_t* selection,
                                 uint32_t n_entries, int want_max, lc_group_partial* out, uint64_t capacity,
                                 unsigned long long* n_out, hipStream_t stream) {
    if (n_entries == 0) return hipSuccess;
    GroupArgs a{g_descs, v_descs, symtabs, selection, n_entries, want_max, out, capacity, n_out};
    // (queried once PER DEVICE: hipGetDeviceProperties on every launch cost more host time than the launch; an atomic per device
    // slot, so that concurrent callers and contexts on different devices each get their own answer)
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cus_of[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    const uint32_t wgs = std::min<uint32_t>((n_entries + kGroupWaves - 1u) / kGroupWaves, uint32_t(cus) * 5u);  // 5 x 32 KB of LDS per CU
    hipLaunchKernelGGL(k_group_partials, dim3(wgs), dim3(kGroupWaves * 64), 0, stream, a);
    return hipGetLastError();
}

hipError_t warm_code_object_groupby() {  // (see warm_code_object_kernels)
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_group_partials));
}

}  // namespace lc
