// k_like_lean — selective `LIKE '%needle%'` over byte-view columns that carry the bigram signature index and the inverted
// row lists (the headline scan: ClickBench q20 / "Q21", URL LIKE '%google%'), ONE kernel, one wave per entry.
//
// What it replaces: the kSigOnly variant of k_str_pred, which is bound by instruction issue (SQ counters, round 3: ~2,900
// instructions per entry, 12,207 waves; 27.5 us per 100 M rows hot, 35.8 us L3-cold).  This kernel keeps only what a
// LIKE over indexed entries needs: ~780 instructions per entry, 20.1 us hot / 26.1 us cold.  Chain of a wave:
//   record (scalar loads; its address follows from blockIdx) -> the needle's signature slices, ANDed -> candidate keys
//   listed in LDS -> their offset pairs -> their compressed bytes cut into 8-byte words, one lane per word, walked through
//   the LDS copy of the needle's automaton folded over the FSST symbols (k_str_automata; exact at the fixpoint of the
//   neighbour-state correction, see lc_kernels.hip) -> list bounds and rows of the matching values from the entry's
//   inverted row list, OR-ed into an LDS copy of the entry's mask words -> AND selection -> store.
// It is correct for every needle (candidates beyond the LDS list are walked in further rounds) and fastest for selective
// ones; a needle that hits many rows is better served by k_str_pred's key mapping (measured: '%mail%', 19 % of the rows,
// 805 vs 437 us), so the dispatcher PLANS every (scan, needle) once: one trial evaluation into scratch counts the hit rows
// (the one host round trip; entries are immutable while a scan pins them, so the count is a property of the pair) and the
// choice is cached.
//
// Measured on the way and rejected (commit 7f0e9d1..): a two-kernel scan-level pipeline — probe over a flat
// index of all dictionaries + a walk over 64-word chunks of one candidate list — 26.8 us hot: two launch ramps and eight
// dependent round trips instead of six cost what the leaner code saves.
//
// Reference counterpart: LiquidByteViewArray::compare_like_substring — fingerprint filter, decode + memmem of the
// candidates, map_dictionary_results_to_array_results (byte_view_array/comparisons.rs:159-183, 325-347, 598-651).
// Results are identical (the candidates of the signature AND are a superset of the matches, the walk is exact).
#include <chrono>
#include <cstring>

#include "lc_device.hpp"
#include "lc_internal.hpp"

namespace lc {

// what a workgroup needs for its (at most four) entries of ONE symbol table, fetched with scalar loads
struct alignas(16) LeanEntry {
    const uint64_t* sig;
    const uint8_t* residuals;
    const uint8_t* fsst;
    const uint16_t* postings;
    uint64_t mask_word_off;
    int32_t slope, intercept;
    uint32_t d, n;
    uint32_t offset_bytes, nw;
    const uint64_t* validity;      // NOT LIKE only (LIKE hits come from the row lists, which hold valid rows)
    const uint32_t* fingerprints;  // NOT LIKE only: the reference's candidate rule (comparisons.rs:167-180)
};
static_assert(sizeof(LeanEntry) == 80, "LeanEntry layout");
#ifndef LC_LEAN_E
#define LC_LEAN_E 1
#endif
// entries a wave takes AT ONCE (their candidates share the wave's lanes).  Two were tried because 12,207 one-entry waves need
// 1.5 generations of the 8,192 wave slots and 6,104 two-entry waves one: parity green, but 24.9 us against 20.5 — the second
// probe round and the second walk pass lengthen every wave's dependent chain by more than the saved generation is worth.
// A second attempt requested the slices of both entries in ONE round and the words of the second walk pass before the first
// pass is walked (79 VGPRs, 6,104 waves on 6,144 slots: one generation): 24.3 us — the chain of a two-entry wave is ~21 us
// against ~9 us for one entry (the walks' fixpoint iterations and the row phases add up, they do not overlap).
constexpr uint32_t kLeanE = LC_LEAN_E;
static_assert(kLeanE == 1 || kLeanE == 2, "a wave takes one or two entries");
struct alignas(16) LeanRec {
    uint32_t begin, end;  // entries [begin, end) of the scan, end - begin <= kLeanWaves * kLeanE: wave w takes entries
                          // begin + kLeanE * w (+ 1)
    uint32_t slot;        // their symbol table
    uint32_t pad;
    LeanEntry e[8 + 1];   // (+ 1: the second-entry pointer of the last wave stays inside the record)
};
static_assert(sizeof(LeanRec) == 16 + 9 * 80, "LeanRec layout");
constexpr uint32_t kLeanCap = 512;  // candidate keys a wave lists in LDS before it walks them
#ifndef LC_LEAN_WAVES
#define LC_LEAN_WAVES 4
#endif
constexpr uint32_t kLeanWaves = LC_LEAN_WAVES;  // waves (= entries) per workgroup: they share the LDS automaton (measured:
                                                // 2 -> same time, 1 -> 24.5 us)
static_assert(kLeanWaves >= 1 && kLeanWaves <= 4, "a LeanRec holds eight entries");
// -DLC_LEAN_STOP=n (variant builds only, results are WRONG): leave the kernel after phase n — -1 at once, -3 record, -2 record +
// automaton image + barrier, 1 probe, 2 offset pairs, 3 compressed words — to measure where the time goes
#ifndef LC_LEAN_STOP
#define LC_LEAN_STOP 0
#endif
#ifndef LC_LEAN_XCD
#define LC_LEAN_XCD 1
#endif
constexpr uint32_t kMaxPlans = 8;
// statistics of a planning run: [COUNT(*) | pad] then kStatShards x [candidates | candidate bytes | matching values | pad], a
// 128-byte line per shard
constexpr uint32_t kStatShards = 64;
constexpr uint32_t kStatStride = 16;  // u64 words
constexpr uint32_t kStatWords = kStatStride + kStatShards * kStatStride;
// a needle is "selective" (worth this kernel) up to this many hit rows per 1024 rows of the scan
constexpr uint32_t kMaxHitsPer1024 = 16;

struct LikePlan {
    std::vector<uint8_t> needle;
    bool use_lean = false;
    uint64_t hits = 0, n_cand = 0, cand_bytes = 0, matches = 0;  // of the trial run (byte accounting, EXPLAIN)
    uint64_t last_use = 0;
    uint32_t n_probe = 0;   // k_like_flat: slices probed (a long needle has more bigrams than are worth reading, see make_plan)
    // A plan made ON THE FLY (round 6): the first evaluation of a plain LIKE is launched for real with the statistics
    // counters attached — no trial run, no host round trip — and the figures travel to pinned memory behind it; the plan is
    // settled by the first later evaluation that finds the event complete (until then the kernel that ran keeps running).
    bool pending = false;
    int32_t slot = -1;                    // the context's plan slot (plan_slot_take) that holds the three below
    hipEvent_t ev = nullptr;
    unsigned long long* d_res = nullptr;  // device: kStatWords words (COUNT(*) + the sharded counters, see sum_stats)
    uint64_t* h_res = nullptr;            // pinned: the same
    bool flat_planned = false;  // the trial ran on the scan-level index (a plan made while the index was still being built is
                                // made again, once, when the index is in place: slices to probe, candidate statistics)
    float plan_ms = 0;      // what planning cost (trial launches + the host round trip)
};

struct FlatGroup;
// An index build asked for by an evaluation starts when that evaluation's kernels have run: the job waits for the gate, which
// the asking call opens on its way out with an event recorded behind its launches (the 2 GB hipMalloc and the first waves of
// k_flat_build used to land in the middle of the 40 us kernel that answers the scan's first LIKE: 180-320 us instead of 80).
struct BuildGate {
    std::atomic<int> open{0};
    hipEvent_t ev = nullptr;
};
static void build_gate_pass(const std::shared_ptr<BuildGate>& g) {
    if (!g) return;
    while (!g->open.load(std::memory_order_acquire)) std::this_thread::yield();  // (microseconds: the asking call is on its way out)
    if (g->ev) {
        (void)hipEventSynchronize(g->ev);
        (void)hipEventDestroy(g->ev);
        g->ev = nullptr;
    }
}
struct LikePipeline {
    bool built = false, eligible = false;
    std::shared_ptr<BuildGate> gate;  // of the builds this evaluation submitted (opened by like_pipeline_eval before it returns)
    LeanRec* d_lean = nullptr;  // one record per workgroup
    uint32_t n_lean = 0;
    uint32_t* d_lean_begins = nullptr;  // first entry of every record (+ the end): k_lean_records builds the records from the
    void* h_lean_begins = nullptr;      // scan's descriptors on the device; the pinned source of the upload lives as long
    // the scan-level wide signature index of k_like_flat (see below): kFlatBits slices over the dictionary words of ALL
    // entries of the scan, slice-major; null when the scan is not eligible for it (or it did not fit)
    bool lean_ok = false;                      // every entry fits k_like_lean's 1 KB of mask words (<= kPostLdsRows rows)
    uint32_t flat_mask_bytes = 0;              // k_like_flat: LDS bytes of one entry's mask words (the scan's largest entry)
    bool flat = false, flat_tried = false;
    bool eq_ok = false;                        // k_like_flat can evaluate `=` / `<>` (prefix keys everywhere)
    const DevSymtab* d_symtabs = nullptr;      // the scan's
    uint64_t* d_slices = nullptr;
    FlatGroup* d_groups = nullptr;
    uint32_t n_groups = 0, n_group_slots = 0;  // groups with entries / records incl. the padding of workgroup batches
    uint32_t group_words = 0;                  // signature words of a group inside a slice: whole 128-byte lines (<= kFlatGroupWords)
    uint32_t probe_words = 0;                  // ... of them in use by the largest group, even: what the probe of a group reads
    uint64_t slices_bytes = 0;
    double flat_build_ms = 0;
    // the unigram index (1-byte needles): 256 slices in the layout of the bigram slices — bit i of entry e's words in slice
    // b: dictionary value i holds byte b.  Built the first time a 1-byte needle runs on the scan.
    uint32_t* d_dst_word = nullptr;            // per entry: its first word inside a slice
    uint64_t slice_words = 0;
    uint64_t* d_uni = nullptr;
    bool uni_tried = false;
    double uni_build_ms = 0;
    unsigned long long* d_total_acc = nullptr;
    std::vector<LikePlan> plans;
    uint64_t tick = 0;
    std::vector<uint64_t> uids;  // the publications (Entry::uid) of the scan's entries, in scan order: what the index describes
    // Builds off the query path (round 6).  The scan-level index is built by the context's builder thread into `flat_pending`
    // (a LikePipeline that only carries the flat fields) while evaluations go on over the entry-level index; the evaluating
    // thread moves the finished fields in (promote_flat) the next time it holds the scan's lock.  States: 0 nothing started,
    // 1 job in flight, 2 job finished (fields waiting in flat_pending), 3 settled (index in place, or this scan keeps the
    // entry-level index).  The unigram index follows the same protocol (its job needs the bigram index in place).
    std::atomic<int> flat_state{0}, uni_state{0};
    LikePipeline* flat_pending = nullptr;
    uint64_t* uni_pending = nullptr;
    double uni_pending_ms = 0;
    std::future<void> flat_job, uni_job;
    bool flat_needs_evict = false;  // the last attempt found the budget full of cached indexes of other scans: one more attempt,
                                    // with eviction, once this scan has proven hot (kEvictAfterEvals evaluations)
    bool flat_no_room = false;      // the last attempt found the budget held by LIVE scans (or the device out of memory): tried
    uint64_t flat_retry_events = 0; // again once ctx->index_events has moved past this value
    int flat_why = 0;               // outcome of the last build_flat: 0 built / not eligible, 1 needs eviction, 2 no room
    uint32_t like_evals = 0;        // LIKE evaluations this pipeline has served (its heat)
};
// LIKE evaluations a scan must have served from the entry-level index before its scan-level index may EVICT the cached
// index of another scan (round 5's budget-of-3 stream rebuilt 4 ms of index per query; an index pays for itself after ~300)
constexpr uint32_t kEvictAfterEvals = 8;

namespace {

struct LeanArgs {
    const LeanRec* recs;
    uint32_t n_recs;
    const uint8_t* automata;
    uint32_t automaton_stride;
    uint32_t nl;
    uint32_t n_extra;                       // signature bits beyond the N the kernel is instantiated for
    uint32_t needle_fp;                     // NOT LIKE: 32-bucket fingerprint of the needle (fingerprint.rs:33-35)
    uint16_t sig_bits[kMaxSigProbeWide];
    const uint64_t* selection;
    uint64_t* mask;
    uint32_t* counts;
    unsigned long long* stats;  // trial run only: {candidates, their compressed bytes, matching dictionary values}
    ScanLaunch total;  // d_total_acc / d_total_out only
};
using ConstLeanPtr = const __attribute__((address_space(4))) LeanEntry*;

// a per-lane choice between the fields of the wave's two entries (compiles to nothing when a wave holds one entry)
template <typename T>
__device__ __forceinline__ T pick(bool second, T a, T b) {
    return kLeanE == 2 && second ? b : a;
}

// kNot: NOT LIKE.  The reference inverts the dictionary results only when at least one dictionary value passes the 32-bucket
// fingerprint filter of the needle (comparisons.rs:167-180, :644-648) — an entry without such a value answers all false.
// A value that matches passes the filter, so the fingerprints are only consulted for entries without a match, and there the
// first 64 values almost always hold a candidate (one 256-byte load, requested together with the signature slices).
template <int N, bool kNot>
__global__ __launch_bounds__(kLeanWaves * 64, 24 / kLeanWaves) void k_like_lean(LeanArgs a) {
    // dynamic LDS: [automaton image][per wave: kLeanE x 128 mask words | kLeanCap candidates (slot << 16 | key) |
    //                                          64 hit flags + head mask]
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kMaskBytes = kPostLdsRows / 8u;
    constexpr uint32_t kPerWave = kLeanE * kMaskBytes + kLeanCap * 4u + 80u;
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    if (LC_LEAN_STOP == -1) return;  // launch floor
    const uint32_t nl = a.nl;
    const uint32_t tbl_bytes = automaton_image_bytes(nl);
    // XCD-aware record order: the hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own L2), and
    // ~13 consecutive records share a symbol table — dealt by blockIdx every XCD would fetch every table's automaton image
    // (PMC: 15 MB per launch).  Workgroup b takes record (b % 8) * ceil(G / 8) + b / 8: an XCD works through one contiguous
    // eighth of the scan, a table's image is fetched by one or two L2s.
#if LC_LEAN_XCD
    const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
    const uint32_t rec_idx = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (rec_idx >= a.n_recs) {  // (the grid is rounded up to a multiple of 8; every wave of it reports to the COUNT(*))
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, blockIdx.x * kLeanWaves + wave, gridDim.x * kLeanWaves, 0);
        return;
    }
#else
    const uint32_t rec_idx = blockIdx.x;
#endif
    const LeanRec* rec = a.recs + rec_idx;
    // the wave's entries: begin + kLeanE * wave (+ 1); their fields come with the record header in one round of scalar loads
    ConstLeanPtr EA = reinterpret_cast<ConstLeanPtr>(reinterpret_cast<uintptr_t>(&rec->e[kLeanE * wave]));
    ConstLeanPtr EB = reinterpret_cast<ConstLeanPtr>(reinterpret_cast<uintptr_t>(&rec->e[kLeanE * wave + (kLeanE - 1u)]));
    const uint32_t begin = rec->begin, end = rec->end;
    const uint32_t nwA_raw = EA->nw, nwB_raw = EB->nw;
    {
        const uint8_t* src = a.automata + size_t(rec->slot) * a.automaton_stride + automaton_u8_bytes(nl);
        for (uint32_t c = wave * 1024u; c < tbl_bytes; c += kLeanWaves * 1024u) async_copy16(src + c + uint32_t(lane) * 16u, smem + c);
    }
    const uint32_t row0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    if (row0 != 0u) __builtin_trap();  // the image holds absolute LDS addresses computed for address 0
    const uint32_t hitrow = row0 + nl * 512u;
    uint8_t* wbase = smem + tbl_bytes + wave * kPerWave;
    uint64_t* pmask = reinterpret_cast<uint64_t*>(wbase);  // entry A's words, then entry B's
    uint32_t* list = reinterpret_cast<uint32_t*>(wbase + kLeanE * kMaskBytes);
    uint8_t* hitflag = wbase + kLeanE * kMaskBytes + kLeanCap * 4u;
    uint64_t* headmask = reinterpret_cast<uint64_t*>(hitflag + 64);
    const uint32_t entryA = begin + kLeanE * wave;
    const bool hasA = entryA < end, hasB = kLeanE == 2 && entryA + 1u < end;
    if (!hasA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, blockIdx.x * kLeanWaves + wave, gridDim.x * kLeanWaves, 0);
        return;
    }
    if (LC_LEAN_STOP == -2) {  // record + automaton image + barrier only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nwA_raw == 0x7FFFFFFFu) list[0] = 1;
        return;
    }
    if (LC_LEAN_STOP == -3) {  // record only
        if (nwA_raw == 0x7FFFFFFFu) list[0] = 1;
        return;
    }
    const uint32_t nwA = nwA_raw, nwB = hasB ? nwB_raw : 0u;
    // mask words of the entries start clear in LDS (16 bytes per lane and entry)
#pragma unroll
    for (uint32_t q = 0; q < kLeanE; q++) reinterpret_cast<uint4*>(pmask)[q * 64u + uint32_t(lane)] = make_uint4(0, 0, 0, 0);
    bool synced = false;
    uint32_t n_list = 0;
    uint32_t fp0[kLeanE];  // NOT LIKE: fingerprints of the first 64 values of each entry, in flight with the slices
    if (kNot) {
#pragma unroll
        for (uint32_t q = 0; q < kLeanE; q++) {
            ConstLeanPtr E = q ? EB : EA;
            fp0[q] = 0;
            if ((q == 0 || hasB) && uint32_t(lane) < E->d) fp0[q] = as_global(E->fingerprints)[lane];
        }
    }
    uint64_t any_match[kLeanE] = {};  // NOT LIKE: the entry has a matching dictionary value (wave uniform)

    // walk candidates list[0 .. count): 64 per batch, one lane per 8-byte word; rows of the matches go into pmask
    auto walk_list = [&](uint32_t count) {
        for (uint32_t b0 = 0; b0 < count; b0 += kWave) {
            const uint32_t j = b0 + uint32_t(lane);
            const bool cl = j < count;
            const uint32_t c32 = cl ? list[j] : 0u;
            const uint32_t key = c32 & 0xFFFFu;
            const bool sb = (c32 >> 16) != 0u;  // the candidate belongs to the wave's second entry
            uint64_t abs_start = 0;
            uint32_t len = 0;
            if (cl) {
                const uint32_t ob = pick(sb, EA->offset_bytes, EB->offset_bytes);
                const uint8_t* residuals = pick(sb, EA->residuals, EB->residuals);
                const uint32_t slope = uint32_t(pick(sb, EA->slope, EB->slope)), intercept = uint32_t(pick(sb, EA->intercept, EB->intercept));
                const uint64_t v = load_unaligned<uint64_t>(residuals + size_t(key) * ob);
                const uint32_t sh = 32u - 8u * ob;
                const int32_t q0 = int32_t(uint32_t(v) << sh) >> sh;
                const int32_t q1 = int32_t(uint32_t(v >> (8u * ob)) << sh) >> sh;
                const uint32_t start = slope * key + intercept + uint32_t(q0);
                len = slope * (key + 1u) + intercept + uint32_t(q1) - start;
                abs_start = uint64_t(reinterpret_cast<uintptr_t>(pick(sb, EA->fsst, EB->fsst))) + start;
            }
            const uint32_t words = cl ? max(1u, (len + 7u) >> 3) : 0u;
            const uint32_t incl = wave_inclusive_sum(words);
            const uint32_t off = incl - words;
            const uint32_t total = read_lane(incl, kWave - 1);
            hitflag[lane] = 0;
            if (LC_LEAN_STOP == 2) { if (total == 0x7FFFFFFFu) list[0] = 1; continue; }
            uint32_t carry_state = row0;
            for (uint32_t t0 = 0; t0 < total; t0 += kWave) {
                if (lane == 0) *headmask = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const bool head = cl && off >= t0 && off < t0 + kWave;
                if (head) atomicOr(reinterpret_cast<unsigned long long*>(headmask), 1ull << (off - t0));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint64_t hm = *headmask;
                const uint32_t before = uint32_t(__popcll(__ballot(cl && off < t0)));
                const uint64_t upto = lane == 63 ? ~uint64_t(0) : ((uint64_t(2) << lane) - 1);
                const uint32_t r = before + uint32_t(__popcll(hm & upto)) - 1u;  // owner lane of task t0 + lane
                const bool live = t0 + uint32_t(lane) < total;
                const uint32_t o_off = uint32_t(__shfl(int(off), int(r), kWave));
                const uint32_t o_len = uint32_t(__shfl(int(len), int(r), kWave));
                const uint32_t o_lo = uint32_t(__shfl(int(uint32_t(abs_start)), int(r), kWave));
                const uint32_t o_hi = uint32_t(__shfl(int(uint32_t(abs_start >> 32)), int(r), kWave));
                const uint32_t k = t0 + uint32_t(lane) - o_off;
                const uint32_t p = 8u * k;
                const uint32_t rem = live && p < o_len ? o_len - p : 0u;
                uint64_t wd = 0;
                if (rem) wd = load_unaligned<uint64_t>(reinterpret_cast<const uint8_t*>((uint64_t(o_hi) << 32 | o_lo) + p));
                if (LC_LEAN_STOP == 3) { if (wd == 0x123456789ull) list[0] = 1; continue; }
                const bool first = k == 0;
                auto walk_task = [&](uint32_t st) {
                    uint32_t x[8];
                    const uint32_t lo = uint32_t(wd), hi = uint32_t(wd >> 32);
#pragma unroll
                    for (int q = 0; q < 8; q++) x[q] = (((q < 4 ? lo : hi) >> (8 * (q & 3))) & 0xFFu) << 1;
                    return walk8(st, x, rem);
                };
                if (!synced) {  // the LDS automaton: every wave of the workgroup passes this barrier exactly once
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    synced = true;
                }
                uint32_t s_in = row0;
                uint32_t en = walk_task(s_in);
                for (;;) {
                    uint32_t prev = lane_shift_up1(en, carry_state);
                    if (first || prev == hitrow) prev = row0;
                    // (idle lanes behind the last task would hand a state from lane to lane for up to 63 more rounds)
                    const bool changed = live && prev != s_in;
                    if (__ballot(changed) == 0) break;
                    if (changed) {
                        s_in = prev;
                        en = walk_task(s_in);
                    }
                }
                const bool hit = en == hitrow;  // a match counts only at the fixpoint (see k_str_pred)
                carry_state = read_lane(en, kWave - 1);
                if (carry_state == hitrow) carry_state = row0;
                if (hit && live) hitflag[r] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool res = cl && hitflag[lane] != 0;
            uint64_t matched = __ballot(res);
            if (a.stats) {
                const uint64_t lb = wave_sum_u64(uint64_t(len));
                if (lane == 0) {
                    // (kStatShards copies of the three counters, by workgroup: 36,000 atomics of a 12,207-entry scan on ONE
                    // address took the kernel from 27 to 453 us — what round 5's 0.37 ms "plan" was made of)
                    unsigned long long* st = a.stats + (blockIdx.x & (kStatShards - 1u)) * kStatStride;
                    atomicAdd(st, (unsigned long long)min(count - b0, uint32_t(kWave)));
                    atomicAdd(st + 1, (unsigned long long)lb);
                    atomicAdd(st + 2, (unsigned long long)__popcll(matched));
                }
            }
            if (kNot) {
                const uint64_t msb = kLeanE == 2 ? __ballot(res && sb) : 0;
                any_match[0] |= matched & ~msb;
                if (kLeanE == 2) any_match[kLeanE - 1] |= msb;
            }
            if (matched) {
                // rows of the matching dictionary values from the entry's inverted row lists, into the LDS mask words
                uint32_t o0 = 0, o1 = 0;
                if (res) {
                    const uint16_t* post = pick(sb, EA->postings, EB->postings);
                    const uint32_t v = load_unaligned<uint32_t>(reinterpret_cast<const uint8_t*>(post) + 2u * size_t(key));
                    o0 = v & 0xFFFFu;
                    o1 = v >> 16;
                }
                const uint64_t second = __ballot(res && sb);
                while (matched) {
                    const int ml = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)matched)) - 1);
                    matched &= matched - 1;
                    const uint32_t b = read_lane(o0, ml), e1 = read_lane(o1, ml);
                    const bool usb = kLeanE == 2 && ((second >> ml) & 1u) != 0;  // wave uniform
                    const uint16_t* prow = usb ? EB->postings + EB->d + 1u : EA->postings + EA->d + 1u;
                    uint64_t* pm = pmask + (usb ? kMaskBytes / 8u : 0u);
                    for (uint32_t rr = b + uint32_t(lane); rr < e1; rr += kWave) {
                        const uint32_t row = as_global(prow)[rr];
                        atomicOr(reinterpret_cast<unsigned long long*>(&pm[row >> 6]), 1ull << (row & 63u));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    // ---- probe: AND of the needle's signature slices over the words of both entries, 64 words per round
    const uint32_t nw_all = nwA + nwB;
    for (uint32_t f0 = 0; f0 < nw_all; f0 += kWave) {
        const uint32_t f = f0 + uint32_t(lane);
        const bool sb = kLeanE == 2 && f >= nwA;
        const uint32_t w = f - (sb ? nwA : 0u);
        uint64_t m = 0;
        if (f < nw_all) {
            const uint64_t* sig = pick(sb, EA->sig, EB->sig);
            const uint32_t nw = pick(sb, nwA, nwB);
            uint64_t sv[N];
#pragma unroll
            for (int k = 0; k < N; k++) sv[k] = as_global(sig)[size_t(a.sig_bits[k]) * nw + w];
            m = sv[0];
#pragma unroll
            for (int k = 1; k < N; k++) m &= sv[k];
            if (N == kMaxSigProbe && a.n_extra) {
                // long needles: the further bigrams, four slices in flight at a time
                for (uint32_t k0 = 0; k0 < a.n_extra; k0 += 4) {
                    uint64_t xv[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        xv[k] = as_global(sig)[size_t(a.sig_bits[kMaxSigProbe + min(k0 + k, a.n_extra - 1u)]) * nw + w];
                    m &= xv[0] & xv[1] & xv[2] & xv[3];
                }
            }
        }
        const uint32_t tag = sb ? 0x10000u : 0u;
        const uint32_t cnt = uint32_t(__popcll(m));
        const uint32_t incl = wave_inclusive_sum(cnt);
        const uint32_t tot = read_lane(incl, kWave - 1);
        if (tot > kLeanCap) {
            // a round with more candidates than the list holds (the needle is not selective here): the lanes' words are
            // taken one after the other, each word's values (<= 64) as one batch
            walk_list(n_list);
            n_list = 0;
            for (int sl = 0; sl < kWave; sl++) {
                const uint64_t ms = uniform_u64(uint64_t(uint32_t(__shfl(int(uint32_t(m)), sl, kWave))) |
                                                (uint64_t(uint32_t(__shfl(int(uint32_t(m >> 32)), sl, kWave))) << 32));
                if (ms == 0) continue;
                const uint32_t fs = f0 + uint32_t(sl);
                const bool ssb = kLeanE == 2 && fs >= nwA;
                const uint32_t ws = fs - (ssb ? nwA : 0u);
                if ((ms >> lane) & 1u) list[lanes_below(ms)] = (ssb ? 0x10000u : 0u) | (ws * 64u + uint32_t(lane));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                walk_list(uint32_t(__popcll(ms)));
            }
            continue;
        }
        if (n_list + tot > kLeanCap) {
            walk_list(n_list);
            n_list = 0;
        }
        uint32_t o = n_list + incl - cnt;
        while (m) {
            const uint32_t bit = uint32_t(__ffsll((long long)m)) - 1u;
            m &= m - 1;
            list[o++] = tag | (w * 64u + bit);
        }
        n_list += tot;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (LC_LEAN_STOP == 1) n_list = n_list == 0x7FFFFFFFu ? 1u : 0u;
    walk_list(n_list);
    if (!synced) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- the entries' mask words: rows of the lists are valid rows, the selection is applied here
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint64_t wave_hits = 0;
#pragma unroll
    for (uint32_t q = 0; q < kLeanE; q++) {
        if (q == 1 && !hasB) break;
        ConstLeanPtr E = q ? EB : EA;
        const uint64_t moff = E->mask_word_off;
        const uint32_t nwords = (E->n + 63u) >> 6;
        uint32_t c = 0;
        bool invert = false;  // wave uniform
        if (kNot) {
            invert = any_match[q] != 0 || __ballot((fp0[q] & a.needle_fp) == a.needle_fp && uint32_t(lane) < E->d) != 0;
            for (uint32_t i0 = kWave; !invert && i0 < E->d; i0 += kWave) {  // rare: no candidate among the first 64 values
                const uint32_t i = i0 + uint32_t(lane);
                const uint32_t fp = i < E->d ? as_global(E->fingerprints)[i] : 0u;
                invert = __ballot((fp & a.needle_fp) == a.needle_fp && i < E->d) != 0;
            }
        }
        for (uint32_t w = uint32_t(lane); w < nwords; w += kWave) {
            uint64_t hitw = pmask[q * (kMaskBytes / 8u) + w];
            if (kNot) {
                const uint32_t rows_left = E->n - (w << 6);
                uint64_t keep = rows_left >= 64 ? ~uint64_t(0) : ((uint64_t(1) << rows_left) - 1);
                if (E->validity) keep &= as_global(E->validity)[w];
                hitw = invert ? ~hitw & keep : 0;
                if (a.selection) hitw &= as_global(a.selection)[moff + w];
            } else if (a.selection && hitw) {
                hitw &= as_global(a.selection)[moff + w];
            }
            as_global_mut(a.mask)[moff + w] = hitw;
            c += uint32_t(__popcll(hitw));
        }
        if (a.counts || a.total.d_total_out) {
            const uint32_t ct = read_lane(wave_inclusive_sum(c), kWave - 1);
            if (lane == 0 && a.counts) as_global_mut(a.counts)[entryA + q] = ct;
            wave_hits += ct;
        }
    }
    if (a.total.d_total_out && lane == 0) total_contribute(a.total, blockIdx.x * kLeanWaves + wave, gridDim.x * kLeanWaves, wave_hits);
}

// The records of k_like_lean, built on the device from the scan's descriptors (round 6: the host used to fill 736 bytes per
// record and copy 2.2 MB from pageable memory — 0.5 ms of a 12,207-entry scan's first LIKE).  One thread per (record, slot).
__global__ __launch_bounds__(256) void k_lean_records(const StrDesc* __restrict__ descs, const uint32_t* __restrict__ begins,
                                                      uint32_t n_recs, LeanRec* __restrict__ recs) {
    constexpr uint32_t kSlots = 9;
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t r = uint32_t(t / kSlots), k = uint32_t(t % kSlots);
    if (r >= n_recs) return;
    const uint32_t begin = begins[r], end = begins[r + 1];
    if (k == 0) {
        recs[r].begin = begin;
        recs[r].end = end;
        recs[r].slot = descs[begin].symtab_slot;
        recs[r].pad = 0;
    }
    LeanEntry e;
    __builtin_memset(&e, 0, sizeof(e));
    if (begin + k < end) {
        const StrDesc& d = descs[begin + k];
        e.sig = d.signatures;
        e.residuals = d.residuals;
        e.fsst = d.fsst;
        e.postings = d.postings;
        e.mask_word_off = d.mask_word_off;
        e.slope = d.slope;
        e.intercept = d.intercept;
        e.d = d.d;
        e.n = d.n;
        e.offset_bytes = d.offset_bytes;
        e.nw = (d.d + 63u) / 64u;
        e.validity = d.validity;
        e.fingerprints = d.fingerprints;
    }
    recs[r].e[k] = e;
}

hipError_t launch_lean(int n_sig, bool negated, const LeanArgs& a, uint32_t n_recs, hipStream_t stream) {
    if (n_recs == 0) return hipSuccess;
    typedef void (*Kern)(LeanArgs);
    static const Kern table[2][kMaxSigProbe] = {
        {k_like_lean<1, false>, k_like_lean<2, false>, k_like_lean<3, false>, k_like_lean<4, false>, k_like_lean<5, false>,
         k_like_lean<6, false>, k_like_lean<7, false>, k_like_lean<8, false>},
        {k_like_lean<1, true>, k_like_lean<2, true>, k_like_lean<3, true>, k_like_lean<4, true>, k_like_lean<5, true>,
         k_like_lean<6, true>, k_like_lean<7, true>, k_like_lean<8, true>}};
    const size_t lds = automaton_image_bytes(a.nl) + kLeanWaves * (kLeanE * (kPostLdsRows / 8u) + kLeanCap * 4u + 80u);
    const uint32_t grid = LC_LEAN_XCD ? (n_recs + 7u) / 8u * 8u : n_recs;
    if (lds > 64 * 1024) {  // needles of 48-63 bytes: the image alone is 49-64 KB
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(table[negated ? 1 : 0][n_sig - 1]),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(table[negated ? 1 : 0][n_sig - 1], dim3(grid), dim3(kLeanWaves * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

// ================================================================================================================
// k_like_flat — the same evaluation over a SCAN-LEVEL signature index: 512-bit bigram signatures, stored slice-major
// across all entries of the scan.
//
// Why (round 4): k_like_lean walks 9.4 candidates per entry for 0.34 matching values — everything behind its probe is
// paid per candidate (offset pair, compressed words, walk: half its time and 37 of its 80 MB).  512 signature bits leave
// the matches and next to nothing else (URL LIKE '%google%': 0.40 candidates per entry = the matching values), but
// per-entry storage of that many slices lost in round 2: the five slices a query reads sat up to 140 KB apart.  Here
// slice b of ALL entries is one dense array, so a query streams N arrays (N = distinct needle bigrams, <= 16) of 8 bytes
// per 64 dictionary values whatever the number of bits, fully coalesced: a wave owns a GROUP of up to four consecutive
// entries whose dictionaries fit 128 signature words (URL batches: three, 35 words each) and reads 16 bytes per lane and
// slice at an address that follows from its index alone — no descriptor round trip in front of the probe.  The group's
// record (one 64-byte line of scalar loads + 64 bytes per entry copied to LDS) is in flight beside the slices.
// Behind the probe the chain is the one of k_like_lean with the list bounds of the inverted row lists requested together
// with the offset pairs: [slices | record | automaton image] -> [offset pair | list bounds] -> words -> walk -> rows -> store.
//
// The index is derived data like the bigram slices of the entries (a necessary condition of the same kind as the
// reference's fingerprint, byte_view_array/fingerprint.rs:33-35; comparisons.rs:598-615), built on the device from the
// dictionary values the first time a LIKE runs on the scan (k_flat_build), 64 bytes per dictionary value.
constexpr int kFlatBits = 512;
constexpr uint32_t kFlatGroupWords = 128;  // signature words of a group: 16 bytes per lane and slice
constexpr uint32_t kFlatMaxE = 4;          // entries of a group (LDS: 1 KB of mask words each)
constexpr uint32_t kFlatCap = 256;         // candidates a wave lists before it walks them
#ifndef LC_FLAT_WAVES
#define LC_FLAT_WAVES 4
#endif
// -DLC_FLAT_STOP=n (variant builds only, results are WRONG): -1 leave at once, 1 probe + zero stores only, 2 up to the offset
// pairs (12: without loading them), 3 up to the walk (no rows)
#ifndef LC_FLAT_STOP
#define LC_FLAT_STOP 0
#endif
constexpr uint32_t kFlatWaves = LC_FLAT_WAVES;  // waves (= groups) per workgroup; > 1: they share the LDS automaton image
static_assert(kFlatWaves == 1 || kFlatWaves == 2 || kFlatWaves == 4, "waves per workgroup");
__host__ __device__ inline uint32_t flat_bigram_bit(uint32_t a, uint32_t b) {
    return ((((a << 8) | b) * 40503u) >> 7) & uint32_t(kFlatBits - 1);
}

struct alignas(16) FlatEntry {  // what the walk needs of an entry, copied to LDS
    const uint8_t* residuals;
    const uint8_t* fsst;
    const uint16_t* postings;
    const uint64_t* validity;      // NOT LIKE only
    const uint32_t* fingerprints;  // NOT LIKE only
    int32_t slope, intercept;
    uint32_t d;
    uint32_t ob_sp;                // offset_bytes | shared_prefix_len << 8
    const uint8_t* prefix_keys;    // `=` / `<>` only: byte 7 of a key = length of the value behind the shared prefix (255: >= 255)
};
static_assert(sizeof(FlatEntry) == 64, "FlatEntry layout");
struct alignas(64) FlatGroup {
    // one line, scalar loads
    uint32_t first_entry, n_entries;  // entries [first_entry, first_entry + n_entries) of the scan; 0 entries: padding
    uint32_t slot;                    // their symbol table
    uint32_t n_words;                 // signature words in use (<= kFlatGroupWords)
    uint32_t word_off[kFlatMaxE];     // first signature word of entry j (0xFFFFFFFF beyond n_entries)
    uint64_t mask_word_off;           // of entry 0; the segments of a scan's consecutive entries are consecutive
    uint32_t n_rows[kFlatMaxE];
    uint32_t pad[2];
    FlatEntry e[kFlatMaxE];
};
constexpr uint32_t kFlatHotBytes = 64;
// Layout of the scan-level slices.  Round 4 stored them slice major ([slice][group][word]: a slice is one array over all
// groups); since round 5 they are GROUP major ([group][slice][word]): what the probe of a group reads (its words of 5-8
// slices) and what a builder workgroup writes (its words of all 512 slices) lie inside the group's 0.3 MB instead of being
// spread over 512 regions 3.8 MB apart — the builder's stores were bound by address translation (1.96 GB in 3.5 ms).
#ifndef LC_FLAT_GROUP_MAJOR
#define LC_FLAT_GROUP_MAJOR 1
#endif
inline uint64_t flat_slice_stride(uint32_t n_slots, uint32_t group_words) {
    return LC_FLAT_GROUP_MAJOR ? uint64_t(group_words) : uint64_t(n_slots) * group_words;
}
inline uint64_t flat_group_stride(uint32_t n_bits, uint32_t group_words) {
    return LC_FLAT_GROUP_MAJOR ? uint64_t(n_bits) * group_words : uint64_t(group_words);
}
static_assert(sizeof(FlatGroup) == kFlatHotBytes + 64 * kFlatMaxE, "FlatGroup layout");

void drop_flat_index(lc_ctx* ctx, LikePipeline* lp);  // (defined with the index cache below)

namespace {

struct FlatArgs {
    const FlatGroup* groups;
    uint32_t n_slots;                       // records (groups + padding)
    const uint64_t* slices;
    uint64_t slice_stride;                  // u64 words from a group's words of slice s to its words of slice s + 1
    uint64_t group_stride;                  // ... from group g's words of a slice to group g + 1's
    uint32_t group_words;                   // words of a group the probe reads: the largest group of the scan, even (the
                                            // strides are rounded up to whole lines for the builder; the padding is not read)
    uint32_t mask_bytes;                    // kBig: LDS bytes of one entry's mask words (a multiple of 1 KB)
    uint32_t eq_len;                        // != 0: `=` / `<>` (kNot) on the needle: a match must also have this length
    const DevSymtab* symtabs;               // eq_len: lengths of values of 255 bytes and more are counted from their codes
    const uint8_t* eq_lit;                  // eq_len > kMaxNeedleAutomaton: the whole literal (the automaton ran over its first
                                            // 63 bytes): values of its length that contain those are compared byte by byte
    const uint8_t* automata;
    uint32_t automaton_stride;
    uint32_t nl;
    uint32_t n_extra;                       // signature bits beyond the N the kernel is instantiated for
    uint32_t needle_fp;
    uint16_t sig_bits[kMaxSigProbeWide];
    const uint64_t* selection;
    uint64_t* mask;                         // null: no mask wanted (COUNT(*) / hit-list consumers): no zero words are written
    uint32_t* counts;
    unsigned long long* stats;
    uint64_t* hits;                         // sparse result (ScanLaunch::d_hits), or null
    uint64_t hits_cap;
    unsigned long long* n_hits;
    uint32_t* hit_first;
    uint32_t hits_parts;   // 1: one list; kHitParts: partitioned (lc_kernels.hpp)
    uint32_t pad_parts;
    ScanLaunch total;
};
using ConstFlatPtr = const __attribute__((address_space(4))) FlatGroup*;

// kBig: scans with entries of more than 8,192 rows (batch sizes of 16,384 .. 65,535): the LDS mask words of an entry take
// a.mask_bytes instead of 1 KB, and groups hold as many entries as 16 KB of them allow (build_flat).
template <int N, bool kNot, bool kBig>
__global__ __launch_bounds__(kFlatWaves * 64) void k_like_flat(FlatArgs a) {
    // dynamic LDS: [automaton image][per wave: kFlatMaxE x mask words | kFlatMaxE x FlatEntry | kFlatCap candidates
    //                                          (entry << 16 | key) | 64 hit flags + head mask]
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t kMaskBytes = kBig ? a.mask_bytes : kPostLdsRows / 8u;
    const uint32_t kPerWave = kFlatMaxE * kMaskBytes + kFlatMaxE * 64u + kFlatCap * 4u + 80u;
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    if (LC_FLAT_STOP == -1) return;
    const uint32_t nl = a.nl;
    const uint32_t tbl_bytes = automaton_image_bytes(nl);
    // XCD-aware order (see k_like_lean): workgroup b takes batch (b % 8) * ceil(G / 8) + b / 8
    const uint32_t per_xcd = (gridDim.x + 7u) / 8u;
    const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t gi = wg * kFlatWaves + wave;
    const uint32_t unit = blockIdx.x * kFlatWaves + wave, n_units = gridDim.x * kFlatWaves;
    if (wg * kFlatWaves >= a.n_slots) {  // (the grid is rounded up to a multiple of 8: a whole workgroup without records)
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, unit, n_units, 0);
        return;
    }
    const bool live = gi < a.n_slots;
    // ---- the probe's loads: 16 bytes per lane and slice, addresses from the wave's index alone
    u32x4 sv[N];
    u32x4 xv[kMaxSigProbeWide - kMaxSigProbe];
#pragma unroll
    for (int k = 0; k < N; k++) sv[k] = u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kMaxSigProbeWide - kMaxSigProbe; k++) xv[k] = u32x4{~0u, ~0u, ~0u, ~0u};
    if (live && 2u * uint32_t(lane) < a.group_words) {
        const uint64_t* base = a.slices + size_t(gi) * a.group_stride + size_t(lane) * 2u;
#pragma unroll
        for (int k = 0; k < N; k++)
            sv[k] = *reinterpret_cast<GlobalPtr<u32x4>>(reinterpret_cast<uintptr_t>(base + size_t(a.sig_bits[k]) * a.slice_stride));
        if (N == kMaxSigProbe && a.n_extra) {
#pragma unroll
            for (int k = 0; k < kMaxSigProbeWide - kMaxSigProbe; k++)
                xv[k] = *reinterpret_cast<GlobalPtr<u32x4>>(reinterpret_cast<uintptr_t>(
                    base + size_t(a.sig_bits[kMaxSigProbe + min(uint32_t(k), a.n_extra - 1u)]) * a.slice_stride));
        }
    }
    ConstFlatPtr G = reinterpret_cast<ConstFlatPtr>(reinterpret_cast<uintptr_t>(a.groups + (live ? gi : wg * kFlatWaves)));
    const uint32_t n_entries = live ? G->n_entries : 0u;
    const uint32_t first_entry = G->first_entry;
    const uint32_t wo1 = G->word_off[1], wo2 = G->word_off[2], wo3 = G->word_off[3];
    const uint64_t moff0 = G->mask_word_off;
    const uint32_t nr0 = G->n_rows[0], nr1 = G->n_rows[1], nr2 = G->n_rows[2], nr3 = G->n_rows[3];
    uint8_t* wbase = smem + tbl_bytes + wave * kPerWave;
    uint64_t* pmask = reinterpret_cast<uint64_t*>(wbase);
    const FlatEntry* cold = reinterpret_cast<const FlatEntry*>(wbase + kFlatMaxE * kMaskBytes);
    uint32_t* list = reinterpret_cast<uint32_t*>(wbase + kFlatMaxE * kMaskBytes + kFlatMaxE * 64u);
    uint8_t* hitflag = reinterpret_cast<uint8_t*>(list) + kFlatCap * 4u;
    uint64_t* headmask = reinterpret_cast<uint64_t*>(hitflag + 64);
    // the entries' walk fields -> LDS (16 lanes x 16 bytes), in flight with the slices
    if (lane < int(kFlatMaxE * 4u))
        async_copy16(reinterpret_cast<const uint8_t*>(a.groups + (live ? gi : wg * kFlatWaves)) + kFlatHotBytes + uint32_t(lane) * 16u,
                     wbase + kFlatMaxE * kMaskBytes);
    auto load_image = [&]() {
        const uint8_t* src = a.automata + size_t(G->slot) * a.automaton_stride + automaton_u8_bytes(nl);
        for (uint32_t c = wave * 1024u; c < tbl_bytes; c += kFlatWaves * 1024u) async_copy16(src + c + uint32_t(lane) * 16u, smem + c);
    };
    if (kFlatWaves > 1) load_image();  // shared by the workgroup: every wave brings its share, one barrier below
    const uint32_t row0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    if (row0 != 0u) __builtin_trap();  // the image holds absolute LDS addresses computed for address 0
    const uint32_t hitrow = row0 + nl * 512u;
    // mask words of the entries start clear in LDS (16 bytes per lane and entry)
    if (kBig) {
        for (uint32_t i = uint32_t(lane); i < kFlatMaxE * kMaskBytes / 16u; i += kWave) reinterpret_cast<uint4*>(pmask)[i] = make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
        for (uint32_t q = 0; q < kFlatMaxE; q++) reinterpret_cast<uint4*>(pmask)[q * 64u + uint32_t(lane)] = make_uint4(0, 0, 0, 0);
    }

    // ---- AND of the needle's slices: 128 dictionary values per lane
    uint64_t m0, m1;
    {
        u32x4 m = sv[0];
#pragma unroll
        for (int k = 1; k < N; k++) m &= sv[k];
        if (N == kMaxSigProbe && a.n_extra) {
#pragma unroll
            for (int k = 0; k < kMaxSigProbeWide - kMaxSigProbe; k++) m &= xv[k];
        }
        m0 = uint64_t(m.x) | (uint64_t(m.y) << 32);
        m1 = uint64_t(m.z) | (uint64_t(m.w) << 32);
    }
    const uint32_t cnt = uint32_t(__popcll(m0)) + uint32_t(__popcll(m1));
    const uint32_t incl = wave_inclusive_sum(cnt);
    const uint32_t tot = read_lane(incl, kWave - 1);
    bool image_ready = false;
    auto sync_image = [&]() {  // every wave of the workgroup passes this exactly once
        if (image_ready) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kFlatWaves > 1) __syncthreads();
        image_ready = true;
    };
    uint32_t any_match = 0;  // NOT LIKE: bit j: entry j has a matching dictionary value (wave uniform)
    uint64_t wave_hits = 0;
    // The end of every wave that has records: the sparse result (its hit rows as (entry << 32 | row) in (entry, row) order, read
    // back from the final mask words in LDS) and the fused COUNT(*).  The list position comes from ONE returning atomic per
    // WORKGROUP: returning atomics on one address complete a few nanoseconds apart however many waves wait for them (2,800
    // groups with a hit: ~15 us when every wave asked for itself), so the waves add up in LDS and share a base.  Called exactly
    // once by every wave of a workgroup that has records, after its sync_image() (the barriers are workgroup wide).
    auto finish = [&](uint64_t wh) {
        if (a.hits) {
            // (partitioned list: this workgroup's partition — its counter, its region of the buffer; one list: partition 0 of 1)
            const uint32_t part = a.hits_parts > 1u ? (blockIdx.x & (kHitParts - 1u)) : 0u;
            const uint64_t plim = a.hits_parts > 1u ? a.hits_cap / kHitParts : a.hits_cap;
            const uint64_t pbase = uint64_t(part) * plim;
            unsigned long long* const ctr = a.n_hits + part * kHitCounterStride;
            unsigned long long b = 0;
            if (kFlatWaves > 1) {
                auto slot_of = [&](uint32_t w) {
                    return reinterpret_cast<unsigned long long*>(smem + tbl_bytes + w * kPerWave + kFlatMaxE * kMaskBytes +
                                                                 kFlatMaxE * 64u + kFlatCap * 4u + 64u);  // (its headmask)
                };
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (lane == 0) *slot_of(wave) = wh;
                __syncthreads();
                unsigned long long all = 0, before = 0;
                for (uint32_t w = 0; w < kFlatWaves; w++) {
                    const unsigned long long v = *slot_of(w);
                    all += v;
                    if (w < wave) before += v;
                }
                unsigned long long* bslot = reinterpret_cast<unsigned long long*>(smem + tbl_bytes + kFlatMaxE * kMaskBytes +
                                                                                  kFlatMaxE * 64u + kFlatCap * 4u);  // wave 0's hit flags
                __syncthreads();
#if defined(LC_FLAT_HITS_ABL) && (LC_FLAT_HITS_ABL & 1)  // timing aid (wrong positions): no list allocation
                if (wave == 0 && lane == 0) *bslot = 0ull;
#else
                if (wave == 0 && lane == 0) *bslot = all ? atomicAdd(ctr, all) : 0ull;
#endif
                __syncthreads();
                b = uniform_u64(*bslot + before);
            } else if (wh) {
                if (lane == 0) b = atomicAdd(ctr, (unsigned long long)wh);
                b = uniform_u64(b);
            }
#if defined(LC_FLAT_HITS_ABL) && (LC_FLAT_HITS_ABL & 2)  // timing aid: no records written
            if (wh == ~0ull) {
#else
            if (wh != 0) {
#endif
                for (uint32_t j = 0; j < n_entries; j++) {
                    const uint32_t nr = j == 0 ? nr0 : j == 1 ? nr1 : j == 2 ? nr2 : nr3;
                    const uint32_t nwords = (nr + 63u) >> 6;
                    const unsigned long long eb = b;
                    for (uint32_t w0 = 0; w0 < nwords; w0 += kWave) {
                        const uint32_t w = w0 + uint32_t(lane);
                        uint64_t m = w < nwords ? pmask[j * (kMaskBytes / 8u) + w] : 0;
                        const uint32_t cnt = uint32_t(__popcll(m));
                        const uint32_t incl = wave_inclusive_sum(cnt);
                        unsigned long long pos = b + incl - cnt;
                        while (m) {
                            const uint32_t bit = uint32_t(__ffsll((long long)m)) - 1u;
                            m &= m - 1;
                            if (pos < plim) as_global_mut(a.hits)[pbase + pos] = (uint64_t(first_entry + j) << 32) | (w * 64u + bit);
                            pos++;
                        }
                        b += read_lane(incl, kWave - 1);
                    }
                    if (a.hit_first && b != eb && lane == 0) as_global_mut(a.hit_first)[first_entry + j] = uint32_t(pbase + eb);
                }
            }
        }
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, unit, n_units, wh);
    };

    if ((!kNot && tot == 0) || LC_FLAT_STOP == 1) {
        // no dictionary value of the group can contain the needle: zeros, straight from registers
        uint64_t moff = moff0;
        for (uint32_t j = 0; j < n_entries; j++) {
            const uint32_t nr = j == 0 ? nr0 : j == 1 ? nr1 : j == 2 ? nr2 : nr3;
            const uint32_t nwords = (nr + 63u) >> 6;
            if (a.mask)
                for (uint32_t w = uint32_t(lane); w < nwords; w += kWave) as_global_mut(a.mask)[moff + w] = 0;
            if (a.counts && lane == 0) as_global_mut(a.counts)[first_entry + j] = 0;
            moff += nwords;
        }
        if (kFlatWaves > 1) sync_image();
        finish(0);
        return;
    }
    if (kFlatWaves == 1 && tot != 0) load_image();

    // walk candidates list[0 .. count): 64 per batch, one lane per 8-byte word; rows of the matches go into pmask
    auto walk_list = [&](uint32_t count) {
        for (uint32_t b0 = 0; b0 < count; b0 += kWave) {
            const uint32_t j = b0 + uint32_t(lane);
            const bool cl = j < count;
            const uint32_t c32 = cl ? list[j] : 0u;
            const uint32_t key = c32 & 0xFFFFu;
            const uint32_t ej = c32 >> 16;
            uint64_t abs_start = 0;
            uint32_t len = 0, o0 = 0, o1 = 0;
            const uint16_t* prow = nullptr;
            if (cl && LC_FLAT_STOP != 12) {
                const FlatEntry& E = cold[ej];
                const uint32_t ob = E.ob_sp & 0xFFu;
                const uint64_t v = load_unaligned<uint64_t>(E.residuals + size_t(key) * ob);
                const uint32_t pb = load_unaligned<uint32_t>(reinterpret_cast<const uint8_t*>(E.postings) + 2u * size_t(key));
                const uint32_t slope = uint32_t(E.slope), intercept = uint32_t(E.intercept);
                const uint32_t sh = 32u - 8u * ob;
                const int32_t q0 = int32_t(uint32_t(v) << sh) >> sh;
                const int32_t q1 = int32_t(uint32_t(v >> (8u * ob)) << sh) >> sh;
                const uint32_t start = slope * key + intercept + uint32_t(q0);
                len = slope * (key + 1u) + intercept + uint32_t(q1) - start;
                abs_start = uint64_t(reinterpret_cast<uintptr_t>(E.fsst)) + start;
                o0 = pb & 0xFFFFu;
                o1 = pb >> 16;
                prow = E.postings + E.d + 1u;
            }
            // the first four rows of every candidate's list, requested with its compressed words (at 512 signature bits
            // nearly every candidate is a match, and a URL value occurs in ~4 rows of its batch): no round trip behind the walk
            uint64_t rows4 = 0;
            if (cl && o1 > o0) rows4 = load_unaligned<uint64_t>(reinterpret_cast<const uint8_t*>(prow + o0));
            const uint32_t words = cl ? max(1u, (len + 7u) >> 3) : 0u;
            const uint32_t wincl = wave_inclusive_sum(words);
            const uint32_t off = wincl - words;
            const uint32_t total = read_lane(wincl, kWave - 1);
            hitflag[lane] = 0;
            if (LC_FLAT_STOP == 2 || LC_FLAT_STOP == 12) { if (total == 0x7FFFFFFFu) list[0] = 1; continue; }
            uint32_t carry_state = row0;
            for (uint32_t t0 = 0; t0 < total; t0 += kWave) {
                if (lane == 0) *headmask = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const bool head = cl && off >= t0 && off < t0 + kWave;
                if (head) atomicOr(reinterpret_cast<unsigned long long*>(headmask), 1ull << (off - t0));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint64_t hm = *headmask;
                const uint32_t before = uint32_t(__popcll(__ballot(cl && off < t0)));
                const uint64_t upto = lane == 63 ? ~uint64_t(0) : ((uint64_t(2) << lane) - 1);
                const uint32_t r = before + uint32_t(__popcll(hm & upto)) - 1u;  // owner lane of task t0 + lane
                const bool tlive = t0 + uint32_t(lane) < total;
                const uint32_t o_off = uint32_t(__shfl(int(off), int(r), kWave));
                const uint32_t o_len = uint32_t(__shfl(int(len), int(r), kWave));
                const uint32_t o_lo = uint32_t(__shfl(int(uint32_t(abs_start)), int(r), kWave));
                const uint32_t o_hi = uint32_t(__shfl(int(uint32_t(abs_start >> 32)), int(r), kWave));
                const uint32_t k = t0 + uint32_t(lane) - o_off;
                const uint32_t p = 8u * k;
                const uint32_t rem = tlive && p < o_len ? o_len - p : 0u;
                uint64_t wd = 0;
                if (rem) wd = load_unaligned<uint64_t>(reinterpret_cast<const uint8_t*>((uint64_t(o_hi) << 32 | o_lo) + p));
                const bool first = k == 0;
                auto walk_task = [&](uint32_t st) {
                    uint32_t x[8];
                    const uint32_t lo = uint32_t(wd), hi = uint32_t(wd >> 32);
#pragma unroll
                    for (int q = 0; q < 8; q++) x[q] = (((q < 4 ? lo : hi) >> (8 * (q & 3))) & 0xFFu) << 1;
                    return walk8(st, x, rem);
                };
                sync_image();
                uint32_t s_in = row0;
                uint32_t en = walk_task(s_in);
                for (;;) {
                    uint32_t prev = lane_shift_up1(en, carry_state);
                    if (first || prev == hitrow) prev = row0;
                    const bool changed = tlive && prev != s_in;
                    if (__ballot(changed) == 0) break;
                    if (changed) {
                        s_in = prev;
                        en = walk_task(s_in);
                    }
                }
                const bool hit = en == hitrow;  // a match counts only at the fixpoint (see k_str_pred)
                carry_state = read_lane(en, kWave - 1);
                if (carry_state == hitrow) carry_state = row0;
                if (hit && tlive) hitflag[r] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            bool res = cl && hitflag[lane] != 0;
            if (a.eq_len != 0 && res) {
                // `=`: the value contains the needle; it IS the needle when it is as long.  The prefix key knows the length
                // behind the entry's shared prefix (raw/fsst_buffer.rs:162-188; 255: that many or more, or unknown — then
                // the symbol lengths of the value's codes are added up)
                const FlatEntry& E = cold[ej];
                const uint32_t rl = uint32_t(as_global(E.prefix_keys)[size_t(key) * 8u + 7u]);
                uint32_t vlen = (E.ob_sp >> 8) + rl;
                if (rl == 255u) {
                    const DevSymtab& st = a.symtabs[G->slot];
                    const uint8_t* p = reinterpret_cast<const uint8_t*>(abs_start);
                    vlen = 0;
                    for (uint32_t i = 0; i < len; i++) {
                        const uint32_t code = as_global(p)[i];
                        if (code == 255u) { vlen += 1u; i++; }
                        else vlen += st.len[code];
                    }
                }
                res = vlen == a.eq_len;
                if (res && a.eq_len > uint32_t(kMaxNeedleAutomaton)) {
                    // a literal the automaton could not hold: decode and compare (same length, so every byte is one of both)
                    const DevSymtab& st = a.symtabs[G->slot];
                    const uint8_t* p = reinterpret_cast<const uint8_t*>(abs_start);
                    uint32_t j = 0;
                    for (uint32_t i = 0; i < len && res; i++) {
                        const uint32_t code = as_global(p)[i];
                        uint64_t sym;
                        uint32_t sl;
                        if (code == 255u) { i++; sym = i < len ? as_global(p)[i] : 0u; sl = 1; }
                        else { sym = st.sym[code]; sl = st.len[code]; }
                        for (uint32_t k = 0; k < sl && res; k++, j++)
                            res = j < a.eq_len && uint32_t((sym >> (8u * k)) & 0xFFu) == uint32_t(as_global(a.eq_lit)[j]);
                    }
                }
            }
            uint64_t matched = __ballot(res);
            if (a.stats) {
                const uint64_t lb = wave_sum_u64(uint64_t(len));
                if (lane == 0) {
                    // (kStatShards copies of the three counters, by workgroup: 36,000 atomics of a 12,207-entry scan on ONE
                    // address took the kernel from 27 to 453 us — what round 5's 0.37 ms "plan" was made of)
                    unsigned long long* st = a.stats + (blockIdx.x & (kStatShards - 1u)) * kStatStride;
                    atomicAdd(st, (unsigned long long)min(count - b0, uint32_t(kWave)));
                    atomicAdd(st + 1, (unsigned long long)lb);
                    atomicAdd(st + 2, (unsigned long long)__popcll(matched));
                }
            }
            // rows of the matching dictionary values from the entries' inverted row lists, into the LDS mask words
            if (LC_FLAT_STOP == 3) matched = matched == 0x123456789ull ? 1 : 0;
            if (kNot && matched) {
#pragma unroll
                for (uint32_t q = 0; q < kFlatMaxE; q++)
                    if (__ballot(res && ej == q)) any_match |= 1u << q;
            }
            if (LC_FLAT_STOP != 3 && res) {
                uint64_t* pm = pmask + ej * (kMaskBytes / 8u);
                const uint32_t nr4 = min(o1 - o0, 4u);
                for (uint32_t q = 0; q < nr4; q++) {
                    const uint32_t row = uint32_t(rows4 >> (16u * q)) & 0xFFFFu;
                    atomicOr(reinterpret_cast<unsigned long long*>(&pm[row >> 6]), 1ull << (row & 63u));
                }
            }
            matched = LC_FLAT_STOP == 3 ? 0 : __ballot(res && o1 - o0 > 4u);  // longer lists: the rest, a value at a time
            while (matched) {
                const int ml = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)matched)) - 1);
                matched &= matched - 1;
                const uint32_t b = read_lane(o0, ml) + 4u, e1 = read_lane(o1, ml);
                const uint32_t uj = read_lane(ej, ml);  // wave uniform
                const FlatEntry& E = cold[uj];
                const uint16_t* pr = E.postings + E.d + 1u;
                uint64_t* pm = pmask + uj * (kMaskBytes / 8u);
                for (uint32_t rr = b + uint32_t(lane); rr < e1; rr += kWave) {
                    const uint32_t row = as_global(pr)[rr];
                    atomicOr(reinterpret_cast<unsigned long long*>(&pm[row >> 6]), 1ull << (row & 63u));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    // ---- the candidates: (entry of the group << 16) | dictionary key
    auto entry_of = [&](uint32_t w) { return uint32_t(w >= wo1) + uint32_t(w >= wo2) + uint32_t(w >= wo3); };
    auto word_base = [&](uint32_t ej) { return ej == 0 ? 0u : ej == 1 ? wo1 : ej == 2 ? wo2 : wo3; };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the entries' fields are in LDS)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (tot <= kFlatCap) {
        uint32_t o = incl - cnt;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint64_t m = h ? m1 : m0;
            const uint32_t w = 2u * uint32_t(lane) + uint32_t(h);
            const uint32_t ej = entry_of(w);
            const uint32_t kb = (w - word_base(ej)) * 64u;
            while (m) {
                const uint32_t bit = uint32_t(__ffsll((long long)m)) - 1u;
                m &= m - 1;
                list[o++] = (ej << 16) | (kb + bit);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        walk_list(tot);
    } else {
        // more candidates than the list holds (the needle is not selective here): the signature words are taken one after
        // the other, each word's values (<= 64) as one batch
        for (int sl = 0; sl < kWave; sl++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint64_t mm = h ? m1 : m0;
                const uint64_t ms = uniform_u64(uint64_t(uint32_t(__shfl(int(uint32_t(mm)), sl, kWave))) |
                                                (uint64_t(uint32_t(__shfl(int(uint32_t(mm >> 32)), sl, kWave))) << 32));
                if (ms == 0) continue;
                const uint32_t w = 2u * uint32_t(sl) + uint32_t(h);
                const uint32_t ej = entry_of(w);
                if ((ms >> lane) & 1u) list[lanes_below(ms)] = (ej << 16) | ((w - word_base(ej)) * 64u + uint32_t(lane));
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                walk_list(uint32_t(__popcll(ms)));
            }
        }
    }
    sync_image();
    // ---- the entries' mask words: rows of the lists are valid rows, the selection is applied here
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint64_t moff = moff0;
    for (uint32_t j = 0; j < n_entries; j++) {
        const uint32_t nr = j == 0 ? nr0 : j == 1 ? nr1 : j == 2 ? nr2 : nr3;
        const uint32_t nwords = (nr + 63u) >> 6;
        uint32_t c = 0;
        bool invert = false;  // wave uniform
        const uint64_t* validity = nullptr;
        if (kNot) {
            // the reference inverts the dictionary results only when some dictionary value passes the 32-bucket fingerprint
            // filter of the needle (comparisons.rs:167-180, :644-648); a matching value passes it
            const FlatEntry& E = cold[j];
            validity = E.validity;
            invert = ((any_match >> j) & 1u) != 0 || a.eq_len != 0;  // (`<>` is the complement over the valid rows, no candidate rule)
            for (uint32_t i0 = 0; !invert && i0 < E.d; i0 += kWave) {
                const uint32_t i = i0 + uint32_t(lane);
                const uint32_t fp = i < E.d ? as_global(E.fingerprints)[i] : 0u;
                invert = __ballot((fp & a.needle_fp) == a.needle_fp && i < E.d) != 0;
            }
        }
        for (uint32_t w = uint32_t(lane); w < nwords; w += kWave) {
            uint64_t hitw = pmask[j * (kMaskBytes / 8u) + w];
            if (kNot) {
                const uint32_t rows_left = nr - (w << 6);
                uint64_t keep = rows_left >= 64 ? ~uint64_t(0) : ((uint64_t(1) << rows_left) - 1);
                if (validity) keep &= as_global(validity)[w];
                hitw = invert ? ~hitw & keep : 0;
                if (a.selection) hitw &= as_global(a.selection)[moff + w];
            } else if (a.selection && hitw) {
                hitw &= as_global(a.selection)[moff + w];
            }
            if (a.mask) as_global_mut(a.mask)[moff + w] = hitw;
            if (a.hits) pmask[j * (kMaskBytes / 8u) + w] = hitw;  // (the final word: selection and NOT applied)
            c += uint32_t(__popcll(hitw));
        }
        if (a.counts || a.total.d_total_out || a.hits) {
            const uint32_t ct = read_lane(wave_inclusive_sum(c), kWave - 1);
            if (lane == 0 && a.counts) as_global_mut(a.counts)[first_entry + j] = ct;
            wave_hits += ct;
        }
        moff += nwords;
    }
    finish(wave_hits);
}

hipError_t launch_flat(int n_sig, bool negated, const FlatArgs& a, hipStream_t stream) {
    if (a.n_slots == 0) return hipSuccess;
    typedef void (*Kern)(FlatArgs);
#define LC_FLAT_ROW(NOT, BIG)                                                                                                  \
    {k_like_flat<1, NOT, BIG>, k_like_flat<2, NOT, BIG>, k_like_flat<3, NOT, BIG>, k_like_flat<4, NOT, BIG>,                  \
     k_like_flat<5, NOT, BIG>, k_like_flat<6, NOT, BIG>, k_like_flat<7, NOT, BIG>, k_like_flat<8, NOT, BIG>}
    static const Kern table[2][2][kMaxSigProbe] = {{LC_FLAT_ROW(false, false), LC_FLAT_ROW(true, false)},
                                                   {LC_FLAT_ROW(false, true), LC_FLAT_ROW(true, true)}};
#undef LC_FLAT_ROW
    const bool big = a.mask_bytes > kPostLdsRows / 8u;
    const Kern kern = table[big ? 1 : 0][negated ? 1 : 0][n_sig - 1];
    const size_t lds = automaton_image_bytes(a.nl) +
                       kFlatWaves * (kFlatMaxE * size_t(big ? a.mask_bytes : kPostLdsRows / 8u) + kFlatMaxE * 64u + kFlatCap * 4u + 80u);
    const uint32_t wgs = (a.n_slots + kFlatWaves - 1u) / kFlatWaves;
    const uint32_t grid = (wgs + 7u) / 8u * 8u;
    if (lds > 64 * 1024) {  // needles of 48-63 bytes (the image alone is 49-64 KB), entries of more than 8,192 rows
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 160 * 1024 - 512);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kFlatWaves * 64), lds, stream, a);
    return hipGetLastError();
}

// Builder of the scan-level slices.  Workgroup (x, g) takes tile x of group g: kFbWords = 16 consecutive signature words
// (1,024 dictionary values, of up to kFlatMaxE entries — a tile follows the group's words, not an entry's) and stores, for
// every slice, those 16 words as ONE aligned 128-byte line.  That is the point of the shape: the first builders stored 32
// or 64 bytes per slice and workgroup, 1.96 GB in 2.1-2.3 ms of a 4.7 ms build, while the same bytes stored as whole lines take
// 0.3 ms (measured with the decode switched off, profiles/r5/ablation_flat_build.txt) — and since a group's tiles cover every
// word of its slices, padding included, nothing has to be zeroed beforehand.
// The decode: a lane walks a CHUNK of a value — at most kFbChunk compressed bytes — through the symbol table and sets the
// value's bit of every slice word it touches straight in LDS (the transposition the index needs happens by addressing; an
// LDS atomic, 32-bit: the 64-bit form runs at a fraction of the rate).  Chunks, not values, because a wave lasts as long as its
// longest lane and a workgroup as long as its slowest wave: with a lane per value the 64 values of a wave hold URLs of 20 and of
// 300 bytes (2.6x the mean); handing the values out sorted by length evens the lanes of a wave but leaves fifteen waves
// waiting for the one with the longest URLs.  An FSST stream can be entered at any code boundary (the parity of the run of
// escape bytes in front of a position says whether it is one), the bigram across the cut needs the last byte of the
// symbol before it, and OR is idempotent — so the chunks of a value are independent work items of bounded, equal size.
struct FlatBuildArgs {
    const FlatGroup* groups;
    const DevSymtab* symtabs;
    uint64_t* slices;
    uint64_t slice_stride;  // words from a group's words of slice s to its words of slice s + 1
    uint64_t group_stride;  // ... from group g's words of slice 0 to group g + 1's
};
constexpr uint32_t kFbWords = 16;                 // words of a tile: one 128-byte line per slice
constexpr uint32_t kFbValues = 64u * kFbWords;    // values of a tile
constexpr uint32_t kFbThreads = kFbValues;        // set-up: a lane per value
#ifndef LC_FB_CHUNK
#define LC_FB_CHUNK 16
#endif
constexpr uint32_t kFbChunk = LC_FB_CHUNK;        // compressed bytes of a work item
struct FbOffsets {  // what str_offset_pair reads of an entry
    const uint8_t* residuals;
    int32_t slope, intercept;
    uint32_t offset_bytes;
};
#ifndef LC_FB_ROWPAD
#define LC_FB_ROWPAD 2u
#endif
template <bool kUni> constexpr uint32_t fb_row_words() { return (kUni ? 256u : uint32_t(kFlatBits)) + LC_FB_ROWPAD; }  // (+2: the store loop's bank spread)
constexpr size_t kFbPoliteLds = 2048;  // extra dynamic LDS of a "polite" build: 2 x (81,152 + 80 + 2,048) > 160 KiB, one workgroup per CU
template <bool kUni> constexpr size_t fb_lds_bytes() {
    // stage + per-code table (8 bytes) + value ranges + first items (entry index in the top bits) + per-code first | last | len
    return size_t(kFbWords) * fb_row_words<kUni>() * 8u + 256u * 8u + size_t(kFbValues) * 8u + size_t(kFbValues) * 4u + 256u * 4u;
}
// bytes equal to 255 immediately in front of position p of a value that starts at `start`: odd = p follows an escape marker
__device__ __forceinline__ uint32_t fb_escape_run(const uint8_t* __restrict__ f, uint32_t start, uint32_t p) {
    uint32_t k = 0;
    while (p - k > start && f[p - 1u - k] == 255u) k++;
    return k;
}
static_assert(kFbValues == 1024, "fb_value swaps the two 5-bit halves of a slot number");
__device__ __forceinline__ uint32_t fb_value(uint32_t slot) { return ((slot & 31u) << 5) | (slot >> 5); }
// kUni: the 256 unigram slices (bit = the byte itself) instead of the kFlatBits bigram slices.
#ifndef LC_FB_SLOTS
#define LC_FB_SLOTS 1
#endif
// kSlots: how the pairs INSIDE a symbol are set — seven predicated slots at compile-time bit positions, or a loop over the
// symbol's length (see the decode below).
template <bool kUni, bool kSlots = false>
__global__ __launch_bounds__(kFbThreads) __attribute__((amdgpu_num_sgpr(80))) void k_flat_build(FlatBuildArgs a) {
    constexpr uint32_t kBits = kUni ? 256u : uint32_t(kFlatBits);
    constexpr uint32_t kRow = fb_row_words<kUni>();
    extern __shared__ __align__(16) uint8_t fb_smem[];
    // [column][slice]: bit (j & 63) of word [j >> 6][s] = value j of the tile has signature bit s (slice fastest: the words the
    // lanes of a wave OR into sit 8 bytes apart)
    uint64_t* stage = reinterpret_cast<uint64_t*>(fb_smem);
    // Per code (round 6): what a code contributes is known before any value is read.  s_tab: the bigram slices INSIDE the symbol,
    // 9 bits each, in order (unigram build: the symbol's bytes themselves); s_meta: first byte | last byte << 8 | length << 16.
    // Walking the symbol's bytes and hashing every pair in the decode loop cost 14 instructions per byte — per LONGEST symbol
    // among the wave's 64 lanes, since the loop diverges on the length; with the table it is 6 per inner bigram.
    static_assert(kFlatBits <= 512, "9 bits per bigram slice in s_tab");
    static_assert(kFlatMaxE <= 4, "the entry index of a value rides in the top two bits of s_first");
    uint64_t* s_tab = stage + size_t(kFbWords) * kRow;
    uint32_t* s_range = reinterpret_cast<uint32_t*>(s_tab + 256);   // (start, stop) of the tile's values
    uint32_t* s_first = s_range + 2u * kFbValues;                   // first work item of value j (exclusive prefix of the chunk
                                                                    // counts) | entry (0 .. kFlatMaxE-1) << 30
    uint32_t* s_meta = s_first + kFbValues;
    constexpr uint32_t kFirstMask = 0x3FFFFFFFu;
    __shared__ uint32_t s_wave_tot[kFbThreads / 64u + 1u];
    const uint32_t gi = blockIdx.y, tile = blockIdx.x, t = threadIdx.x;
    const FlatGroup& G = a.groups[gi];
    const uint32_t n_entries = G.n_entries;  // (0: a padding record — its lines are stored as zeros)
    // Slot t holds value fb_value(t) of the tile: neighbouring slots — the lanes of a wave work on neighbouring items — hold
    // values of DIFFERENT (column, half) words, so that the lanes that reach the same bigram at the same step (every URL starts
    // with "http://") hit 32 LDS words, not one.
    const uint32_t u = fb_value(t);
    // this slot's value: bit (u & 63) of word gword of the group -> entry and value index
    const uint32_t gword = tile * kFbWords + (u >> 6);
    uint32_t ent = 0, v = 0;
    bool valid = false;
    for (uint32_t e = 0; e < kFlatMaxE; e++) {
        if (e < n_entries) {
            const uint32_t wo = G.word_off[e], de = G.e[e].d;
            if (gword >= wo && gword < wo + ((de + 63u) >> 6)) {
                ent = e;
                v = (gword - wo) * 64u + (u & 63u);
                valid = v < de;
            }
        }
    }
    // the value's compressed range first: its loads are in flight while LDS is set up
    uint32_t start0 = 0, stop0 = 0;
    if (valid) {
        const FlatEntry& fe = G.e[ent];
        const FbOffsets od{fe.residuals, fe.slope, fe.intercept, fe.ob_sp & 0xFFu};
        str_offset_pair(od, v, start0, stop0);
    }
    const DevSymtab& st = a.symtabs[G.slot];
    if (t < 256u) {
        const uint64_t sym = st.sym[t];
        const uint32_t len = st.len[t];
        uint64_t packed = 0;
        for (uint32_t k = 0; k + 1u < len; k++)
            packed |= uint64_t(flat_bigram_bit(uint32_t(sym >> (8u * k)) & 0xFFu, uint32_t(sym >> (8u * k + 8u)) & 0xFFu)) << (9u * k);
        s_tab[t] = kUni ? sym : packed;
        s_meta[t] = (uint32_t(sym) & 0xFFu) | ((len ? uint32_t(sym >> (8u * (len - 1u))) & 0xFFu : 0u) << 8) | (len << 16);
    }
    {
        uint4* z = reinterpret_cast<uint4*>(stage);
        for (uint32_t i = t; i < kFbWords * kRow / 2u; i += kFbThreads) z[i] = uint4{0, 0, 0, 0};
    }
    const int lane = lane_id(), wave = wave_id();
    // work items: chunk c of value j covers compressed bytes [start + c kFbChunk, start + (c + 1) kFbChunk) of it
    // (stop < start cannot pass staging's checks; should it ever, the value contributes nothing instead of 2^28 work items)
    const uint32_t n_chunks = (valid && stop0 > start0) ? (stop0 - start0 + kFbChunk - 1u) / kFbChunk : 0u;
    const uint32_t incl = wave_inclusive_sum(n_chunks);
    if (lane == 63) s_wave_tot[wave] = incl;
    s_range[2u * t] = start0;
    s_range[2u * t + 1u] = stop0;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kFbThreads / 64u; w++) {
        const uint32_t x = s_wave_tot[w];
        before += w < uint32_t(wave) ? x : 0u;
        total += x;
    }
    s_first[t] = ((before + incl - n_chunks) & kFirstMask) | (ent << 30);
    __syncthreads();
#if defined(LC_FB_STOP) && LC_FB_STOP == 1  // timing aid (wrong index): no decode — set-up, zeroing and the stores
    total = 0;
#endif
    for (uint32_t i = t; i < total; i += kFbThreads) {
        // the value item i belongs to: the last j with s_first[j] <= i (values without bytes share their successor's first item)
        uint32_t lo = 0, hi = kFbValues;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((s_first[mid] & kFirstMask) <= i) lo = mid; else hi = mid;
        }
        const uint32_t first_ent = s_first[lo];
        const uint32_t slot = lo, c = i - (first_ent & kFirstMask), j = fb_value(slot);
        uint32_t* col = reinterpret_cast<uint32_t*>(stage + size_t(j >> 6) * kRow) + ((j >> 5) & 1u);  // the value's column, its half
        const uint32_t mybit = 1u << (j & 31u);
        const uint32_t start = s_range[2u * slot], stop = s_range[2u * slot + 1u];
        const uint8_t* fsst = G.e[first_ent >> 30].fsst;
        // cut positions move one byte back when they would fall between an escape marker and its literal
        const uint32_t q0 = start + c * kFbChunk;
        uint32_t p0 = q0, p1 = min(stop, q0 + kFbChunk);
        // Round 6: the item's bytes are requested ONCE, together — the 8 bytes in front of the cut and the chunk's 16 (three
        // independent loads; a chunk behind the first begins at least 16 bytes into its value, so the window lies inside it, and
        // every blob has 256 bytes of slack behind it) — and the cut logic works on registers.  It used to be a chain of
        // dependent single-byte loads (the byte in front of each cut, the byte in front of that, then the chunk's words): ~4
        // round trips per item at 5 items per lane, which is what the 3.4 ms of decode were made of.
        static_assert(LC_FB_CHUNK == 16, "the register window of k_flat_build is 8 + 16 bytes");
        const uint64_t wp = c != 0 ? load_unaligned<uint64_t>(fsst + q0 - 8u) : 0;
        const uint64_t w0 = q0 < stop ? load_unaligned<uint64_t>(fsst + q0) : 0;
        const uint64_t w1 = q0 + 8u < stop ? load_unaligned<uint64_t>(fsst + q0 + 8u) : 0;
        // byte i of the window, i in [-8, 16) relative to q0 (the general form: only runs of escape bytes come here)
        auto wbyte = [&](int i) -> uint32_t {
            return uint32_t((i < 0 ? wp >> (8 * (i + 8)) : (i < 8 ? w0 >> (8 * i) : w1 >> (8 * (i - 8)))) & 0xFFu);
        };
        // bytes equal to 255 immediately in front of window position i; -1: the run reaches the window's first byte (for the
        // first chunk the value's start bounds it, like fb_escape_run)
        const int wlo = c != 0 ? -8 : 0;
        auto wrun = [&](int i) -> int {
            int k = 0;
            while (i - 1 - k >= wlo && wbyte(i - 1 - k) == 255u) k++;
            return (i - 1 - k < wlo && c != 0) ? -1 : k;
        };
        int prev = -1;
        bool slow = false;  // a run of escape bytes longer than the window (never seen outside the fuzz tests): the old way
        int i0 = 0, i1 = int(p1 - q0);
        // the common case costs three byte compares: no escape byte next to either cut
        const uint32_t b_m1 = uint32_t(wp >> 56), b_m2 = uint32_t(wp >> 48) & 0xFFu, b_15 = uint32_t(w1 >> 56);
#ifndef LC_FB_MUTATE  // (-DLC_FB_MUTATE: the cuts left where they fall — the mutant tests/test_gpu_round5.py must catch)
        if (c != 0 && b_m1 == 255u) {
            const int r = wrun(0);
            if (r < 0) slow = true; else i0 -= r & 1;
        }
        if (p1 < stop && b_15 == 255u) {  // (p1 < stop: the chunk is whole, its last byte is window byte 15)
            const int r = wrun(i1);
            if (r < 0) slow = true; else i1 -= r & 1;
        }
#endif
        if (!slow && !kUni && c != 0) {  // the last byte in front of the cut: a literal, or the last byte of a symbol
            uint32_t bb = b_m1;
            int r = 0;
            if (i0 != 0 || b_m2 == 255u) {
                bb = wbyte(i0 - 1);
                r = wrun(i0 - 1);
            }
            if (r < 0) slow = true;
            else prev = (r & 1) ? int(bb) : int((s_meta[bb] >> 8) & 0xFFu);
        }
        auto set_bit = [&](uint32_t bit) { atomicOr(col + 2u * bit, mybit); };
        auto emit = [&](uint32_t code, bool& escaped) {
            if (escaped) {  // a literal byte
                escaped = false;
                if (kUni) set_bit(code);
                else if (prev >= 0) set_bit(flat_bigram_bit(uint32_t(prev), code));
                prev = int(code);
                return;
            }
            if (code == 255u) { escaped = true; return; }
            const uint32_t m = s_meta[code], len = m >> 16;
            if (len == 0) return;
            const uint64_t tab = s_tab[code];
            uint32_t lo = uint32_t(tab), hi = uint32_t(tab >> 32);
            if constexpr (kUni) {
                for (uint32_t q = 0; q < len; q++) {
                    set_bit(lo & 0xFFu);
                    lo = __builtin_amdgcn_alignbyte(hi, lo, 1);
                    hi >>= 8;
                }
            } else {
                if (prev >= 0) set_bit(flat_bigram_bit(uint32_t(prev), m & 0xFFu));  // the pair across the code boundary
                // The pairs inside the symbol.  A loop over len - 1 runs, for the whole wave, as long as its longest symbol and
                // pays a shift pair and a branch per turn; seven predicated slots at compile-time bit positions execute 25 % fewer
                // VALU instructions (1.36 against 1.83 x 10^9 per 100 M-row column) — and at first ran SLOWER at two workgroups
                // per CU (4.7 against 3.2 ms) with the SQ counters showing the wave-cycles of ONE workgroup per CU: the slots
                // took the kernel from 75 to 81 SGPRs, which are allocated as 96 (+16 for the trap handler) and leave a SIMD 7
                // wave slots instead of the 8 that two 16-wave workgroups need.  Hence amdgpu_num_sgpr(80) on the kernel: 2.85 ms.
                // (-DLC_FB_SLOTS=0: the loop, 3.2 ms.)
                if constexpr (kSlots) {
                    if (len > 1u) set_bit(lo & 0x1FFu);
                    if (len > 2u) set_bit((lo >> 9) & 0x1FFu);
                    if (len > 3u) set_bit((lo >> 18) & 0x1FFu);
                    if (len > 4u) set_bit(__builtin_amdgcn_alignbit(hi, lo, 27) & 0x1FFu);
                    if (len > 5u) set_bit((hi >> 4) & 0x1FFu);
                    if (len > 6u) set_bit((hi >> 13) & 0x1FFu);
                    if (len > 7u) set_bit((hi >> 22) & 0x1FFu);
                } else {
                    for (uint32_t q = 1; q < len; q++) {
                        set_bit(lo & 0x1FFu);
                        lo = __builtin_amdgcn_alignbit(hi, lo, 9);
                        hi >>= 9;
                    }
                }
            }
            prev = int((m >> 8) & 0xFFu);
        };
        bool escaped = false;
        if (!slow) {
            // the chunk's bytes as dwords that begin at the (possibly moved) cut: four bytes per dword at compile-time positions
            uint32_t d[5] = {uint32_t(w0), uint32_t(w0 >> 32), uint32_t(w1), uint32_t(w1 >> 32), 0u};
            if (i0 != 0) {  // one byte earlier: everything moves up by a byte, the marker in front comes in
                d[4] = d[3] >> 24;
                d[3] = __builtin_amdgcn_alignbyte(d[3], d[2], 3);
                d[2] = __builtin_amdgcn_alignbyte(d[2], d[1], 3);
                d[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 3);
                d[0] = (d[0] << 8) | b_m1;
            }
            const uint32_t nbytes = uint32_t(i1 - i0);
#pragma unroll
            for (uint32_t k = 0; k < 5u; k++) {
                if (4u * k >= nbytes) break;
                const uint32_t x = d[k];
                emit(x & 0xFFu, escaped);
                if (4u * k + 1u < nbytes) emit((x >> 8) & 0xFFu, escaped);
                if (4u * k + 2u < nbytes) emit((x >> 16) & 0xFFu, escaped);
                if (4u * k + 3u < nbytes) emit(x >> 24, escaped);
            }
        } else {
            p0 = q0;
            p1 = min(stop, q0 + kFbChunk);
            prev = -1;
#ifndef LC_FB_MUTATE
            if (c != 0) p0 -= fb_escape_run(fsst, start, p0) & 1u;
            if (p1 < stop) p1 -= fb_escape_run(fsst, start, p1) & 1u;
#endif
            if (!kUni && c != 0) {
                const uint32_t bb = fsst[p0 - 1u];
                const bool literal = (fb_escape_run(fsst, start, p0 - 1u) & 1u) != 0;
                prev = literal ? int(bb) : int((s_meta[bb] >> 8) & 0xFFu);
            }
            for (uint32_t pp = p0; pp < p1; pp++) emit(uint32_t(fsst[pp]), escaped);
        }
    }
    __syncthreads();
    // 16 lanes store the 16 words of one slice (a whole 128-byte line), a wave four slices per instruction
    uint64_t* out = a.slices + size_t(gi) * a.group_stride + size_t(tile) * kFbWords;
    for (uint32_t i = t; i < kBits * kFbWords; i += kFbThreads) {
        const uint32_t s = i / kFbWords, cc = i % kFbWords;
        out[size_t(s) * a.slice_stride + cc] = stage[size_t(cc) * kRow + s];
    }
}

// Where index memory comes from.  The index of a whole column is gigabytes and is an allocation of its own; the index of a
// scan over ONE row group (the reference's granularity: a reader per row group) is a few megabytes, and a host that works at
// that granularity creates and drops hundreds of such scans: as allocations of their own those were a hipMalloc per build and a
// hipFree — 3 ms, device-wide — per dropped index.  Up to kIndexPoolMax they come out of the context's scratch chunks instead.
constexpr uint64_t kIndexPoolMax = uint64_t(16) << 20;
static void* index_mem_alloc(lc_ctx* ctx, uint64_t bytes) {
    if (bytes <= kIndexPoolMax) return pool_alloc(ctx, size_t(bytes));
    void* p = nullptr;
    if (hipMalloc(&p, size_t(bytes)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
static void index_mem_free(lc_ctx* ctx, void* p) {  // (by ownership, not by size: a build that failed half way has recorded none)
    if (p && !pool_release_if_owned(ctx, p)) (void)hipFree(p);
}

// Index memory is HBM the caller's budget must cover: max_hbm_bytes bounds slabs + indexes, LC_OPT_LIKE_INDEX_BUDGET_BYTES the
// indexes alone, and an index never takes more than half of what the device has free.  index_reserve answers "may `bytes` of
// index be allocated now?" and, when yes, CHARGES them to ctx->index_bytes before the caller allocates (two builders cannot
// both pass; the caller gives the bytes back if its hipMalloc fails).  allow_evict: the cached indexes of destroyed scans go
// first, oldest first, one at a time — only their index memory, the pipeline (records, plans) stays cached; what live scans
// hold stays, and the entry-level index then serves the asking scan.  *why: 0 reserved, 1 the answer was "no" only because
// cached indexes of other scans fill the budget (and allow_evict was not given), 2 no room.
bool index_reserve(lc_ctx* ctx, uint64_t bytes, bool allow_evict, int* why) {
    if (why) *why = 2;  // (no room, unless found otherwise below)
    std::lock_guard<std::mutex> g(ctx->index_reserve_mu);
    for (;;) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
        const uint64_t ib = ctx->index_bytes.load(), lim = ctx->like_index_budget.load();
        const bool fits = bytes <= free_b / 2 && !(ctx->max_hbm && ctx->staged_bytes.load() + ib + bytes > ctx->max_hbm) &&
                          !(lim && ib + bytes > lim);
        if (fits) {
            ctx->index_bytes += bytes;
            if (why) *why = 0;
            return true;
        }
        // victims, oldest first: the pipelines of destroyed scans, then those of the scans the context keeps for the next
        // lc_scan_create over their list (idle by definition: handing one out takes the same lock)
        bool victim = false;
        {
            std::lock_guard<std::mutex> g2(ctx->like_orphans_mu);
            for (LikePipeline* q : ctx->like_orphans)
                if (q->d_slices || q->d_uni) {
                    victim = true;
                    if (allow_evict) drop_flat_index(ctx, q);  // (under the orphans' lock: nobody adopts it meanwhile)
                    break;
                }
        }
        if (!victim) {
            std::lock_guard<std::mutex> g2(ctx->scan_cache_mu);
            for (lc_scan* c : ctx->list_cache) {
                LikePipeline* q = c->like;
                if (!q || !(q->d_slices || q->d_uni) || q->flat_state.load() == 1 || q->flat_state.load() == 2 ||
                    q->uni_state.load() == 1 || q->uni_state.load() == 2)
                    continue;
                victim = true;
                if (allow_evict) drop_flat_index(ctx, q);
                break;
            }
        }
        if (!victim) return false;  // what is left belongs to live scans
        if (!allow_evict) {
            if (why) *why = 1;
            return false;
        }
    }
}

// The scan-level index of k_like_flat: groups of consecutive entries (one symbol table, <= kFlatMaxE entries, <= 128
// signature words), their records, and the kFlatBits slices built from the dictionary values.  A scan whose index does not
// fit (more than half of the free device memory) keeps k_like_lean.  Fills the flat fields of `lp` — the scan's pipeline when
// the caller waits for the build, a stand-in that the evaluating thread merges later when the builder thread runs it.
// `polite` (the builder thread's builds): the kernel runs with ONE workgroup per CU instead of two.  A workgroup is 16 waves and
// two of them take every wave slot of a CU for the ~4 ms the build of a 100 M-row column lasts — measured: a query launched
// while the build ran waited until it had finished (4.2 ms for a 30 us kernel).  With one per CU (a kilobyte more of dynamic LDS
// makes the second one not fit) half of the slots stay free for the queries the build must not hold up.
lc_status build_flat(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream, bool allow_evict = true, bool polite = false) {
    LC_PHASE("build_flat (all; builder thread when async)");
    lp->flat = false;
    lp->flat_tried = true;
    lp->flat_why = 0;
    std::vector<FlatGroup> groups;
    std::vector<uint32_t> dst_word(s->n, 0);
    uint32_t n_groups = 0;
    // LDS mask words of an entry: 1 KB per 8,192 rows of the scan's largest entry; a group holds as many entries as 16 KB of
    // them allow (4 x 16 KB per workgroup beside a 64 KB automaton image at most)
    uint32_t max_rows = 1;
    for (const Entry& e : s->meta) max_rows = std::max(max_rows, e.sd.n);
    const uint32_t mask_bytes = (max_rows + kPostLdsRows - 1u) / kPostLdsRows * (kPostLdsRows / 8u);
    const uint32_t max_e = std::min<uint32_t>(kFlatMaxE, std::max<uint32_t>(1u, 16384u / mask_bytes));
    if (mask_bytes > 8192u) return LC_OK;
    lp->flat_mask_bytes = mask_bytes;
    bool eq_ok = true;  // `=` / `<>` through the index need every entry's prefix keys and a shared prefix that fits 24 bits
    auto pad_batch = [&]() {  // a workgroup's groups share a symbol table: close the batch with empty records
        while (groups.size() % kFlatWaves) {
            FlatGroup g;
            std::memset(&g, 0, sizeof(g));
            g.slot = groups.back().slot;
            g.first_entry = groups.back().first_entry + groups.back().n_entries;
            g.mask_word_off = s->seg_offsets[g.first_entry];
            for (uint32_t k = 0; k < kFlatMaxE; k++) g.word_off[k] = 0xFFFFFFFFu;
            groups.push_back(g);
        }
    };
    for (uint32_t b = 0; b < s->n;) {
        FlatGroup g;
        std::memset(&g, 0, sizeof(g));
        for (uint32_t k = 0; k < kFlatMaxE; k++) g.word_off[k] = 0xFFFFFFFFu;
        g.first_entry = b;
        g.slot = s->meta[b].sd.symtab_slot;
        g.mask_word_off = s->seg_offsets[b];
        uint32_t words = 0, i = b;
        while (i < s->n && i - b < max_e && s->meta[i].sd.symtab_slot == g.slot) {
            const StrDesc& d = s->meta[i].sd;
            const uint32_t nw = (d.d + 63u) / 64u;
            if (words + nw > kFlatGroupWords) break;
            // (consecutive entries of a scan have consecutive mask segments: the kernel derives them from the first)
            if (i > b && d.mask_word_off != s->seg_offsets[i]) return fail(LC_ERR_INVALID, "scan mask segments are not consecutive");
            const uint32_t j = i - b;
            g.word_off[j] = words;
            g.n_rows[j] = d.n;
            g.e[j] = FlatEntry{d.residuals, d.fsst, d.postings, d.validity, d.fingerprints, d.slope, d.intercept, d.d,
                               uint32_t(d.offset_bytes) | (std::min<uint32_t>(d.shared_prefix_len, 0xFFFFFFu) << 8), d.prefix_keys};
            if (d.shared_prefix_len >= 0xFFFFFFu || !d.prefix_keys) eq_ok = false;
            words += nw;
            i++;
        }
        if (i == b) return LC_OK;  // (a dictionary of more than 8192 values: not eligible, k_like_lean stays)
        g.word_off[0] = 0;
        g.n_entries = i - b;
        g.n_words = words;
        if (!groups.empty() && groups.back().slot != g.slot) pad_batch();
        groups.push_back(g);
        n_groups++;
        b = i;
    }
    if (groups.empty()) return LC_OK;
    pad_batch();
    // every group takes the same number of words inside a slice (the wave's address follows from its index): the largest
    // group of the scan, rounded up to whole 128-byte lines (the builder stores lines; the probe loads 16 bytes per lane)
    uint32_t gw = kFbWords;
    for (const FlatGroup& g : groups) gw = std::max(gw, (g.n_words + kFbWords - 1u) & ~(kFbWords - 1u));
    for (size_t gi = 0; gi < groups.size(); gi++)
        for (uint32_t j = 0; j < groups[gi].n_entries; j++)
            dst_word[groups[gi].first_entry + j] = uint32_t(gi * flat_group_stride(kFlatBits, gw) + groups[gi].word_off[j]);
    if (uint64_t(groups.size()) * flat_group_stride(kFlatBits, gw) > 0xFFFFFFFFull) return LC_OK;
    // (the unigram index has 256 slices: its own first words, behind the bigram ones in the same device array)
    dst_word.resize(size_t(s->n) * 2);
    for (size_t gi = 0; gi < groups.size(); gi++)
        for (uint32_t j = 0; j < groups[gi].n_entries; j++)
            dst_word[s->n + groups[gi].first_entry + j] = uint32_t(gi * flat_group_stride(256u, gw) + groups[gi].word_off[j]);
    const uint64_t slice_words = uint64_t(groups.size()) * gw;
    const uint64_t bytes = slice_words * 8u * uint64_t(kFlatBits);
    {
        LC_PHASE("build_flat: host grouping done; reserve");
    }
    if (!index_reserve(ctx, bytes, allow_evict, &lp->flat_why)) return LC_OK;  // no room: the entry-level index serves
    struct Reservation {  // given back unless the index ends up in place
        lc_ctx* c; uint64_t b; bool keep;
        ~Reservation() { if (!keep) c->index_bytes -= b; }
    } reservation{ctx, bytes, false};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    (void)hipEventCreate(&ev0);
    (void)hipEventCreate(&ev1);
    uint32_t* d_dst = static_cast<uint32_t*>(pool_alloc(ctx, size_t(s->n) * 8));  // (kept: [bigram | unigram] first words)
    // the group records (320 bytes each) and the word offsets go through ONE pinned block (copies from pageable vectors are
    // staged by the runtime itself, in blocking pieces)
    const size_t g_bytes = groups.size() * sizeof(FlatGroup), d_bytes = size_t(s->n) * 8;
    uint8_t* h_stage = static_cast<uint8_t*>(host_pool_alloc(ctx, g_bytes + d_bytes));
    struct Tmp {
        lc_ctx* c; hipStream_t st; hipEvent_t a, b; void* pinned;
        ~Tmp() {
            (void)hipStreamSynchronize(st);
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
            host_pool_release(c, pinned);
        }
    } tmp{ctx, stream, ev0, ev1, h_stage};
    if (!d_dst || !h_stage) return fail(LC_ERR_OOM, "flat index build: staging");
    std::memcpy(h_stage, groups.data(), g_bytes);
    std::memcpy(h_stage + g_bytes, dst_word.data(), d_bytes);
    lp->d_dst_word = d_dst;
    lp->slice_words = slice_words;
    LC_PHASE("build_flat: hipMalloc + launch + wait");
    hipError_t e_malloc;
    {
        LC_PHASE("build_flat: hipMalloc of the slices");
        lp->d_slices = static_cast<uint64_t*>(index_mem_alloc(ctx, bytes));
        e_malloc = lp->d_slices ? hipSuccess : hipErrorOutOfMemory;
    }
    if (e_malloc != hipSuccess) {
        lp->d_slices = nullptr;
        lp->flat_why = 2;
        return LC_OK;  // no room: the entry-level index serves
    }
    lp->d_groups = static_cast<FlatGroup*>(pool_alloc(ctx, groups.size() * sizeof(FlatGroup)));
    if (!lp->d_groups) return fail(LC_ERR_OOM, "hipMalloc (flat group records)");
    if (ev0) LC_HIP(hipEventRecord(ev0, stream));
    LC_HIP(hipMemcpyAsync(lp->d_groups, h_stage, g_bytes, hipMemcpyHostToDevice, stream));
    LC_HIP(hipMemcpyAsync(d_dst, h_stage + g_bytes, d_bytes, hipMemcpyHostToDevice, stream));
    // (no memset: the tiles of a group cover every word of its slices, padding included)
    FlatBuildArgs ba{lp->d_groups, s->d_symtabs, lp->d_slices, flat_slice_stride(uint32_t(groups.size()), gw),
                     flat_group_stride(kFlatBits, gw)};
    LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_build<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(fb_lds_bytes<false>() + kFbPoliteLds)));
    LC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_flat_build<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(fb_lds_bytes<false>() + kFbPoliteLds)));
    if (polite)
        hipLaunchKernelGGL((k_flat_build<false, true>), dim3(gw / kFbWords, uint32_t(groups.size())), dim3(kFbThreads),
                           fb_lds_bytes<false>() + kFbPoliteLds, stream, ba);
    else
        hipLaunchKernelGGL((k_flat_build<false, LC_FB_SLOTS != 0>), dim3(gw / kFbWords, uint32_t(groups.size())), dim3(kFbThreads),
                           fb_lds_bytes<false>(), stream, ba);
    LC_HIP(hipGetLastError());
    if (ev1) LC_HIP(hipEventRecord(ev1, stream));
    LC_HIP(hipStreamSynchronize(stream));  // the vectors are locals
    float ms = 0;
    if (ev0 && ev1 && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) lp->flat_build_ms = ms;
    lp->n_groups = n_groups;
    lp->n_group_slots = uint32_t(groups.size());
    lp->group_words = gw;
    lp->probe_words = 2;
    for (const FlatGroup& g : groups) lp->probe_words = std::max(lp->probe_words, (g.n_words + 1u) & ~1u);
    lp->slices_bytes = bytes;
    reservation.keep = true;
    lp->flat = true;
    lp->eq_ok = eq_ok;
    lp->d_symtabs = s->d_symtabs;
    return LC_OK;
}

// The unigram index of a scan that has its bigram index (same groups, same word offsets): 256 slices, 32 bytes per
// dictionary value (0.98 GB and one more pass over the dictionaries for the 100 M-row URL column).  A value holds the byte b
// exactly when its bit in slice b is set, so a 1-byte LIKE needs no walk: k_like_scanall<kUni> copies the entry's words of
// ONE slice and maps the rows through the keys.
lc_status build_unigram(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream, uint64_t** out_uni, double* out_ms) {
    *out_uni = nullptr;
    *out_ms = 0;
    if (!lp->flat || !lp->d_dst_word || lp->slice_words == 0) return LC_OK;
    const uint64_t bytes = lp->slice_words * 8u * 256u;
    // (the cached indexes of other scans go first, like for the bigram index; what live scans hold stays: the walkers serve)
    if (!index_reserve(ctx, bytes, true, nullptr)) return LC_OK;
    uint64_t* d_uni = static_cast<uint64_t*>(index_mem_alloc(ctx, bytes));
    if (!d_uni) {
        ctx->index_bytes -= bytes;
        return LC_OK;  // no room: the walkers serve
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    (void)hipEventCreate(&ev0);
    (void)hipEventCreate(&ev1);
    struct Tmp {
        hipEvent_t a, b;
        ~Tmp() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } tmp{ev0, ev1};
    *out_uni = d_uni;  // (the caller owns it — and the bytes charged to the context — whatever happens below)
    if (ev0) LC_HIP(hipEventRecord(ev0, stream));
    FlatBuildArgs ba{lp->d_groups, s->d_symtabs, d_uni, flat_slice_stride(lp->n_group_slots, lp->group_words),
                     flat_group_stride(256u, lp->group_words)};
    hipLaunchKernelGGL(k_flat_build<true>, dim3(lp->group_words / kFbWords, lp->n_group_slots), dim3(kFbThreads),
                       fb_lds_bytes<true>(), stream, ba);
    LC_HIP(hipGetLastError());
    if (ev1) LC_HIP(hipEventRecord(ev1, stream));
    LC_HIP(hipStreamSynchronize(stream));
    float ms = 0;
    if (ev0 && ev1 && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) *out_ms = ms;
    return LC_OK;
}

// per-workgroup records, built once per scan
lc_status build_index(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream) {
    LC_PHASE("like: workgroup records (lean)");
    lp->built = true;
    lp->eligible = false;
    lp->uids = s->uids;
    if (!s->is_str || s->n == 0) return LC_OK;
    // (from the scan's compact arrays: an all-null entry has no dictionary — no candidates, its mask words are zero — and needs
    // no index; every other entry must carry signatures, row lists and fingerprints)
    if (!s->str_index_everywhere || s->max_dict_rows > kPostMaxRows) return LC_OK;
    lp->lean_ok = s->max_str_rows <= kPostLdsRows;  // (entries of more than 8,192 rows: only k_like_flat, whose LDS mask size is the scan's)
    // consecutive entries, at most kLeanWaves * kLeanE, never across a symbol-table change: the host says where each record
    // begins (through a pinned block that lives as long as the pipeline: no wait for the copy), k_lean_records fills them in
    std::vector<uint32_t> begins;
    for (uint32_t b = 0, i = 1; i <= s->n; i++) {
        if (i == s->n || i - b == kLeanWaves * kLeanE || s->symtab_slots[i] != s->symtab_slots[b]) {
            begins.push_back(b);
            b = i;
        }
    }
    begins.push_back(s->n);
    lp->n_lean = uint32_t(begins.size() - 1);
    lp->d_lean = static_cast<LeanRec*>(pool_alloc(ctx, std::max<size_t>(lp->n_lean, 1) * sizeof(LeanRec)));
    lp->d_total_acc = static_cast<unsigned long long*>(pool_alloc(ctx, size_t(kTotalWords) * 8));
    lp->d_lean_begins = static_cast<uint32_t*>(pool_alloc(ctx, begins.size() * 4));
    lp->h_lean_begins = host_pool_alloc(ctx, begins.size() * 4);
    if (!lp->d_lean || !lp->d_total_acc || !lp->d_lean_begins || !lp->h_lean_begins) return fail(LC_ERR_OOM, "hipMalloc (LIKE records)");
    std::memcpy(lp->h_lean_begins, begins.data(), begins.size() * 4);
    LC_HIP(hipMemcpyAsync(lp->d_lean_begins, lp->h_lean_begins, begins.size() * 4, hipMemcpyHostToDevice, stream));
    LC_HIP(hipMemsetAsync(lp->d_total_acc, 0, size_t(kTotalWords) * 8, stream));  // once: launches leave it zero
    {
        const uint64_t threads = uint64_t(lp->n_lean) * 9u;
        hipLaunchKernelGGL(k_lean_records, dim3(uint32_t((threads + 255u) / 256u)), dim3(256), 0, stream,
                           static_cast<const StrDesc*>(s->d_descs), lp->d_lean_begins, lp->n_lean, lp->d_lean);
        LC_HIP(hipGetLastError());
    }
    // (later evaluations on OTHER streams are ordered behind this one by scan_enter_stream's drain of the previous stream)
    lp->eligible = true;
    return LC_OK;
}

lc_status run_lean(LikePipeline* lp, const StrPred& p, const ScanLaunch& L, hipStream_t stream,
                   unsigned long long* d_stats = nullptr, bool force_like = false) {
    LeanArgs la{};
    la.recs = lp->d_lean;
    la.n_recs = lp->n_lean;
    la.automata = p.automata;
    la.automaton_stride = p.automaton_stride;
    la.nl = p.needle_len;
    for (int k = 0; k < kMaxSigProbeWide; k++) la.sig_bits[k] = p.sig_wide[k < int(p.n_sig_wide) ? k : 0];
    la.n_extra = p.n_sig_wide > uint32_t(kMaxSigProbe) ? p.n_sig_wide - uint32_t(kMaxSigProbe) : 0u;
    la.needle_fp = p.needle_fp;
    la.selection = L.d_selection;
    la.mask = L.d_hit;
    la.counts = L.d_counts;
    la.stats = d_stats;
    la.total.d_total_acc = lp->d_total_acc;
    la.total.d_total_out = L.d_total_out;
    LC_HIP(launch_lean(int(std::min<uint32_t>(p.n_sig_wide, uint32_t(kMaxSigProbe))), p.op == LC_OP_NOT_LIKE && !force_like, la,
                       lp->n_lean, stream));
    return LC_OK;
}

// the needle's bigrams as bits of the scan-level signatures, in the order make_str_pred takes them (both ends, then the
// midpoints of the remaining gaps: every prefix of the list covers the needle evenly)
uint32_t flat_needle_bits(const std::vector<uint8_t>& needle, uint16_t (&bits)[kMaxSigProbeWide]) {
    uint32_t n = 0;
    const size_t il = needle.size();
    if (il < 2) return 0;
    std::vector<size_t> order;
    const size_t last = il - 2;
    std::vector<uint8_t> seen(last + 1, 0);
    std::vector<std::pair<size_t, size_t>> gaps;
    auto take = [&](size_t k) { if (!seen[k]) { seen[k] = 1; order.push_back(k); } };
    take(0);
    take(last);
    gaps.push_back({0, last});
    for (size_t g = 0; g < gaps.size(); g++) {
        const size_t lo = gaps[g].first, hi = gaps[g].second;
        if (hi - lo < 2) continue;
        const size_t mid = lo + (hi - lo) / 2;
        take(mid);
        gaps.push_back({lo, mid});
        gaps.push_back({mid, hi});
    }
    for (size_t k : order) {
        if (n >= uint32_t(kMaxSigProbeWide)) break;
        const uint16_t bit = uint16_t(flat_bigram_bit(needle[k], needle[k + 1]));
        bool dup = false;
        for (uint32_t q = 0; q < n; q++) dup |= bits[q] == bit;
        if (!dup) bits[n++] = bit;
    }
    for (uint32_t q = n; q < uint32_t(kMaxSigProbeWide); q++) bits[q] = bits[0];
    return n;
}

lc_status run_flat(LikePipeline* lp, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                   unsigned long long* d_stats = nullptr, bool force_like = false, uint32_t max_probe = kMaxSigProbeWide) {
    const StrPred& p = sp.p;
    FlatArgs fa{};
    fa.groups = lp->d_groups;
    fa.n_slots = lp->n_group_slots;
    fa.slices = lp->d_slices;
    fa.slice_stride = flat_slice_stride(lp->n_group_slots, lp->group_words);
    fa.group_stride = flat_group_stride(kFlatBits, lp->group_words);
    fa.group_words = lp->probe_words;
    fa.mask_bytes = lp->flat_mask_bytes;
    fa.eq_len = force_like ? 0u : p.eq_len;
    fa.symtabs = lp->d_symtabs;
    fa.eq_lit = p.needle;
    fa.automata = p.automata;
    fa.automaton_stride = p.automaton_stride;
    fa.nl = p.needle_len;
    const uint32_t nb = std::min(flat_needle_bits(sp.needle, fa.sig_bits), std::max(max_probe, 1u));
    for (uint32_t q = nb; q < uint32_t(kMaxSigProbeWide); q++) fa.sig_bits[q] = fa.sig_bits[0];
    fa.n_extra = nb > uint32_t(kMaxSigProbe) ? nb - uint32_t(kMaxSigProbe) : 0u;
    fa.needle_fp = p.needle_fp;
    fa.selection = L.d_selection;
    fa.mask = L.mask_optional ? nullptr : L.d_hit;
    fa.counts = L.d_counts;
    fa.stats = d_stats;
    fa.hits = L.d_hits;
    fa.hits_cap = L.hits_cap;
    fa.n_hits = L.d_n_hits;
    fa.hit_first = L.d_hit_first;
    fa.hits_parts = L.hits_parts > 1u ? kHitParts : 1u;
    fa.total.d_total_acc = lp->d_total_acc;
    fa.total.d_total_out = L.d_total_out;
    LC_HIP(launch_flat(int(std::min<uint32_t>(nb, uint32_t(kMaxSigProbe))), p.op == LC_OP_NOT_LIKE && !force_like, fa, stream));
    return LC_OK;
}

// [COUNT(*), candidates, candidate bytes, matching values] out of a planning run's raw counters
static void sum_stats(const uint64_t* raw, uint64_t (&res)[4]) {
    res[0] = raw[0];
    res[1] = res[2] = res[3] = 0;
    for (uint32_t k = 0; k < kStatShards; k++) {
        res[1] += raw[kStatStride + kStatStride * k];
        res[2] += raw[kStatStride + kStatStride * k + 1];
        res[3] += raw[kStatStride + kStatStride * k + 2];
    }
}

// one trial evaluation into scratch: how many rows does the needle hit?
lc_status make_plan(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, const StrPredHost& sp, hipStream_t stream, LikePlan* plan) {
    LC_PHASE("like: plan (trial + round trip)");
    plan->needle = sp.needle;
    plan->use_lean = false;
    const uint64_t words = std::max<uint64_t>(s->seg_offsets.back(), 1);
    uint64_t* d_scratch = static_cast<uint64_t*>(pool_alloc(ctx, (words + kStatWords) * 8));  // mask | COUNT(*) + the statistics
    struct Tmp {
        lc_ctx* c; void* p; hipStream_t st;
        ~Tmp() { (void)hipStreamSynchronize(st); pool_release(c, p); }
    } tmp{ctx, d_scratch, stream};
    if (!d_scratch) return fail(LC_ERR_OOM, "hipMalloc (LIKE plan)");
    ScanLaunch L{};
    L.d_hit = d_scratch;
    L.d_total_out = d_scratch + words;
    LC_HIP(hipMemsetAsync(d_scratch + words, 0, kStatWords * 8, stream));
    // (always as LIKE: the plan belongs to the needle, NOT LIKE is selective exactly when LIKE is)
    unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(d_scratch + words + kStatStride);
    const bool use_flat = lp->flat && (ctx->like_path == 0 || ctx->like_path == 4);
    const auto t_begin = std::chrono::steady_clock::now();
    uint64_t* raw = static_cast<uint64_t*>(host_pool_alloc(ctx, kStatWords * 8));  // (pinned: a copy into pageable memory is staged and slow)
    uint64_t res[4] = {0, 0, 0, 0};
    struct ResTmp {
        lc_ctx* c; void* p;
        ~ResTmp() { host_pool_release(c, p); }
    } res_tmp{ctx, raw};
    if (!raw) return fail(LC_ERR_OOM, "hipHostMalloc (LIKE plan)");
    auto trial = [&](uint32_t n_probe) -> lc_status {
        LC_HIP(hipMemsetAsync(d_scratch + words, 0, kStatWords * 8, stream));
        const lc_status rc = use_flat ? run_flat(lp, sp, L, stream, d_stats, true, n_probe) : run_lean(lp, sp.p, L, stream, d_stats, true);
        if (rc != LC_OK) return rc;
        LC_HIP(hipMemcpyAsync(raw, d_scratch + words, kStatWords * 8, hipMemcpyDeviceToHost, stream));
        LC_HIP(hipStreamSynchronize(stream));
        sum_stats(raw, res);
        return LC_OK;
    };
    uint16_t bits[kMaxSigProbeWide];
    const uint32_t nb = use_flat ? flat_needle_bits(sp.needle, bits) : 0u;
    plan->n_probe = std::min<uint32_t>(nb, uint32_t(kMaxSigProbe));
    plan->flat_planned = use_flat;
    lc_status rc = trial(use_flat ? plan->n_probe : 0u);
    if (rc != LC_OK) return rc;
    if (use_flat && nb > uint32_t(kMaxSigProbe)) {
        // A long needle has more bigrams than the eight a wave reads in its first round.  Every further slice costs the scan
        // 8 bytes per 64 dictionary values and removes candidates; what is cheaper depends on the data ('%yandex.ru/search%':
        // 74 -> 37 candidates per entry with all 15 — worth it; a 34-byte needle that hits nothing: 1.1 -> 0 — not worth
        // 27 MB more).  Priced with the measured constants of this kernel: ~0.75 us per slice and 12,207 entries against
        // ~0.2 ns per candidate walked.
        const uint64_t cand8 = res[1];
        uint64_t keep[4];
        std::memcpy(keep, res, sizeof(keep));
        rc = trial(nb);
        if (rc != LC_OK) return rc;
        const double slice_ns = 750.0 * double(lp->n_group_slots) / 4096.0, cand_ns = 0.2;
        const double cost8 = double(cand8) * cand_ns, cost_all = double(res[1]) * cand_ns + double(nb - kMaxSigProbe) * slice_ns;
        if (cost_all < cost8) plan->n_probe = nb;
        else std::memcpy(res, keep, sizeof(keep));
    }
    plan->hits = res[0];
    plan->n_cand = res[1];
    plan->cand_bytes = res[2];
    plan->matches = res[3];
    plan->plan_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    plan->use_lean = plan->hits * 1024u <= uint64_t(kMaxHitsPer1024) * std::max<uint64_t>(s->total_rows, 1024);
    return LC_OK;
}

}  // namespace

// Plans made on the fly (LikePlan::pending): what they hold while the figures are under way, and their settlement.
constexpr uint32_t kPlanSlots = 256;
// (caller holds ctx->plan_slots_mu) the arena: 4 MB of device and of pinned memory, once per context
static void plan_slots_alloc_locked(lc_ctx* ctx) {
    if (ctx->plan_slots_tried) return;
    LC_PHASE("plan slots: first allocation");
    ctx->plan_slots_tried = true;
    void* d = pool_alloc(ctx, size_t(kPlanSlots) * kStatWords * 8);
    void* h = d ? host_pool_alloc(ctx, size_t(kPlanSlots) * kStatWords * 8) : nullptr;
    if (d && h) {
        ctx->plan_slots_d = static_cast<unsigned long long*>(d);
        ctx->plan_slots_h = static_cast<uint64_t*>(h);
        ctx->plan_slot_ev.assign(kPlanSlots, nullptr);
        for (uint32_t i = kPlanSlots; i-- > 0;) ctx->plan_slots_free.push_back(i);
    } else if (d) {
        pool_release(ctx, d);
    }
}
// lc_ctx_create: the pinned half of the arena is a hipHostMalloc of its own (0.3-0.5 ms) — it was part of the first LIKE
// evaluation of a context
void plan_slots_prime(lc_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->plan_slots_mu);
    plan_slots_alloc_locked(ctx);
    // ... and the runtime's own first-use work for what a plan's evaluation issues besides kernels — a memset of device memory, a
    // copy into pinned memory, an event — is done here, once, instead of behind the first LIKE of the process
    if (ctx->plan_slots_d && ctx->plan_slots_h) {
        hipEvent_t e = nullptr;
        (void)hipMemsetAsync(ctx->plan_slots_d, 0, kStatWords * 8, nullptr);
        (void)hipMemcpyAsync(ctx->plan_slots_h, ctx->plan_slots_d, kStatWords * 8, hipMemcpyDeviceToHost, nullptr);
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
            (void)hipEventRecord(e, nullptr);
            (void)hipEventSynchronize(e);
            (void)hipEventDestroy(e);
        }
        (void)hipGetLastError();
    }
}
static bool plan_slot_take(lc_ctx* ctx, LikePlan* q) {
    std::lock_guard<std::mutex> g(ctx->plan_slots_mu);
    plan_slots_alloc_locked(ctx);
    if (ctx->plan_slots_free.empty()) return false;  // (every slot is under way: this needle gets a trial run instead)
    const uint32_t i = ctx->plan_slots_free.back();
    if (!ctx->plan_slot_ev[i] && hipEventCreateWithFlags(&ctx->plan_slot_ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    ctx->plan_slots_free.pop_back();
    q->slot = int32_t(i);
    q->ev = ctx->plan_slot_ev[i];
    q->d_res = ctx->plan_slots_d + size_t(i) * kStatWords;
    q->h_res = ctx->plan_slots_h + size_t(i) * kStatWords;
    return true;
}
static void release_pending_plan(lc_ctx* ctx, LikePlan* q) {
    if (q->slot < 0) return;
    if (q->pending) (void)hipEventSynchronize(q->ev);  // (the kernel may still be adding to d_res)
    {
        std::lock_guard<std::mutex> g(ctx->plan_slots_mu);
        ctx->plan_slots_free.push_back(uint32_t(q->slot));
    }
    q->slot = -1;
    q->ev = nullptr;
    q->d_res = nullptr;
    q->h_res = nullptr;
    q->pending = false;
}
void plan_slots_destroy(lc_ctx* ctx) {  // lc_ctx_destroy
    std::lock_guard<std::mutex> g(ctx->plan_slots_mu);
    for (hipEvent_t e : ctx->plan_slot_ev)
        if (e) (void)hipEventDestroy(e);
    ctx->plan_slot_ev.clear();
    if (ctx->plan_slots_d) pool_release(ctx, ctx->plan_slots_d);
    if (ctx->plan_slots_h) host_pool_release(ctx, ctx->plan_slots_h);
    ctx->plan_slots_d = nullptr;
    ctx->plan_slots_h = nullptr;
    ctx->plan_slots_free.clear();
}
// true when the plan's figures are in (block: wait for them)
static bool settle_plan(lc_ctx* ctx, const lc_scan* s, LikePlan* q, bool block) {
    if (!q->pending) return true;
    if (!block) {
        const hipError_t e = hipEventQuery(q->ev);
        if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    } else {
        (void)hipEventSynchronize(q->ev);
    }
    uint64_t res[4];
    sum_stats(q->h_res, res);
    q->hits = res[0];
    q->n_cand = res[1];
    q->cand_bytes = res[2];
    q->matches = res[3];
    q->use_lean = q->hits * 1024u <= uint64_t(kMaxHitsPer1024) * std::max<uint64_t>(s->total_rows, 1024);
    release_pending_plan(ctx, q);
    return true;
}

// A destroyed scan's pipeline waits here for the next scan over the same publications of the same entries (same uids in the
// same order: same blobs, same mask layout — the records hold pointers into the entries, which the adopting scan pins like
// the scan that built them did): its workgroup records, its plans, its heat and — when it has one — its scan-level index.  At
// most LC_OPT_LIKE_INDEX_CACHE of them (default 4) and a quarter of the device's memory in indexes; the oldest goes first.
static uint64_t pipeline_bytes(const LikePipeline* lp) {
    return (lp->d_slices ? lp->slices_bytes : 0u) + (lp->d_uni ? lp->slice_words * 8u * 256u : 0u);
}
// the flat fields of `src` (a builder job's stand-in) move into `dst`
static void adopt_flat_fields(LikePipeline* dst, LikePipeline* src) {
    dst->flat_mask_bytes = src->flat_mask_bytes;
    dst->flat = src->flat;
    dst->flat_tried = true;
    dst->eq_ok = src->eq_ok;
    dst->d_symtabs = src->d_symtabs;
    dst->d_slices = src->d_slices;
    dst->d_groups = src->d_groups;
    dst->n_groups = src->n_groups;
    dst->n_group_slots = src->n_group_slots;
    dst->group_words = src->group_words;
    dst->probe_words = src->probe_words;
    dst->slices_bytes = src->slices_bytes;
    dst->flat_build_ms = src->flat_build_ms;
    dst->d_dst_word = src->d_dst_word;
    dst->slice_words = src->slice_words;
    src->d_slices = nullptr;
    src->d_groups = nullptr;
    src->d_dst_word = nullptr;
    src->slices_bytes = 0;
}
// The results of finished builder jobs move into the pipeline.  Caller holds the scan's lock (or owns the pipeline alone).
static void promote_builds(lc_ctx* ctx, LikePipeline* lp) {
    if (lp->flat_state.load(std::memory_order_acquire) == 2) {
        if (lp->flat_job.valid()) lp->flat_job.get();
        LikePipeline* pend = lp->flat_pending;
        lp->flat_pending = nullptr;
        int why = 0;
        if (pend) {
            why = pend->flat_why;
            if (pend->flat) adopt_flat_fields(lp, pend);
            like_pipeline_destroy(ctx, pend);  // (what a failed build left behind)
        }
        // an attempt that only found the budget full of other scans' cached indexes is repeated (with eviction) once this scan
        // has proven hot, one that found no room once index memory has been given back somewhere; every other outcome (built,
        // not eligible) is final for this pipeline
        lp->flat_needs_evict = !lp->flat && why == 1;
        lp->flat_no_room = !lp->flat && why == 2;
        lp->flat_state.store(lp->flat || why == 0 ? 3 : 0, std::memory_order_release);
    }
    if (lp->uni_state.load(std::memory_order_acquire) == 2) {
        if (lp->uni_job.valid()) lp->uni_job.get();
        lp->d_uni = lp->uni_pending;
        lp->uni_build_ms = lp->uni_pending_ms;
        lp->uni_pending = nullptr;
        lp->uni_tried = true;
        lp->uni_state.store(3, std::memory_order_release);
    }
}
// blocks until the jobs in flight for this pipeline have finished, then moves their results in
static void settle_builds(lc_ctx* ctx, LikePipeline* lp) {
    if (lp->flat_job.valid()) lp->flat_job.wait();
    if (lp->uni_job.valid()) lp->uni_job.wait();
    promote_builds(ctx, lp);
}
void like_pipeline_wait(lc_scan* s) {
    if (!s->like) return;
    settle_builds(s->ctx, s->like);
    for (LikePlan& q : s->like->plans) (void)settle_plan(s->ctx, s, &q, true);
}
// frees the index memory of a CACHED pipeline (index_reserve: budget eviction); the pipeline itself stays cached and may build
// its index again.  Caller holds the lock that keeps the pipeline idle (ctx->like_orphans_mu / ctx->scan_cache_mu).
void drop_flat_index(lc_ctx* ctx, LikePipeline* lp) {
    ctx->index_events++;
    if (lp->d_slices) { index_mem_free(ctx, lp->d_slices); ctx->index_bytes -= lp->slices_bytes; }
    if (lp->d_uni) { index_mem_free(ctx, lp->d_uni); ctx->index_bytes -= lp->slice_words * 8u * 256u; }
    pool_release(ctx, lp->d_groups);
    pool_release(ctx, lp->d_dst_word);
    lp->d_slices = nullptr;
    lp->d_uni = nullptr;
    lp->d_groups = nullptr;
    lp->d_dst_word = nullptr;
    lp->slices_bytes = 0;
    lp->slice_words = 0;
    lp->flat = lp->flat_tried = lp->uni_tried = false;
    lp->flat_needs_evict = true;  // (it lost its index to the budget: it may take one back once it is hot again)
    lp->like_evals = 0;
    lp->flat_state.store(0);
    lp->uni_state.store(0);
}
static LikePipeline* like_pipeline_adopt(lc_ctx* ctx, const lc_scan* s) {
    std::lock_guard<std::mutex> g(ctx->like_orphans_mu);
    for (size_t i = ctx->like_orphans.size(); i-- > 0;) {
        LikePipeline* lp = ctx->like_orphans[i];
        if (lp->uids.size() != s->uids.size()) continue;
        const bool same = std::memcmp(lp->uids.data(), s->uids.data(), lp->uids.size() * 8) == 0;
        if (!same) continue;
        ctx->like_orphans.erase(ctx->like_orphans.begin() + long(i));
        return lp;
    }
    return nullptr;
}
void like_pipeline_orphan(lc_ctx* ctx, LikePipeline* lp) {
    if (!lp) return;
    settle_builds(ctx, lp);  // (the builder reads the scan that is going away)
    if (lp->d_slices || lp->d_uni) ctx->index_events++;  // (its index memory is reclaimable from now on)
    if (!lp->built || !lp->eligible || ctx->like_index_cache.load() == 0) { like_pipeline_destroy(ctx, lp); return; }
    std::vector<LikePipeline*> out;
    {
        std::lock_guard<std::mutex> g(ctx->like_orphans_mu);
        ctx->like_orphans.push_back(lp);
        size_t total_b = 0, free_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
        uint64_t held = 0;
        for (const LikePipeline* q : ctx->like_orphans) held += pipeline_bytes(q);
        const uint64_t lim = ctx->like_index_budget.load();
        uint64_t alive = ctx->index_bytes.load();  // (a running total: the victims are only destroyed below)
        while (!ctx->like_orphans.empty() && (ctx->like_orphans.size() > size_t(ctx->like_index_cache.load()) || held > total_b / 4 ||
                                               (lim && alive > lim && held > 0))) {
            const uint64_t vb = pipeline_bytes(ctx->like_orphans.front());
            held -= vb;
            alive -= std::min(alive, vb);
            out.push_back(ctx->like_orphans.front());
            ctx->like_orphans.erase(ctx->like_orphans.begin());
        }
    }
    for (LikePipeline* q : out) like_pipeline_destroy(ctx, q);
}
void like_orphans_clear(lc_ctx* ctx) {
    std::vector<LikePipeline*> out;
    {
        std::lock_guard<std::mutex> g(ctx->like_orphans_mu);
        out.swap(ctx->like_orphans);
    }
    for (LikePipeline* q : out) like_pipeline_destroy(ctx, q);
}

void like_pipeline_destroy(lc_ctx* ctx, LikePipeline* lp) {
    if (!lp) return;
    settle_builds(ctx, lp);
    for (LikePlan& q : lp->plans) release_pending_plan(ctx, &q);
    pool_release(ctx, lp->d_lean);
    pool_release(ctx, lp->d_lean_begins);
    host_pool_release(ctx, lp->h_lean_begins);
    pool_release(ctx, lp->d_total_acc);
    pool_release(ctx, lp->d_groups);
    pool_release(ctx, lp->d_dst_word);
    if (lp->d_slices || lp->d_uni) ctx->index_events++;
    if (lp->d_slices) { index_mem_free(ctx, lp->d_slices); ctx->index_bytes -= lp->slices_bytes; }
    if (lp->d_uni) { index_mem_free(ctx, lp->d_uni); ctx->index_bytes -= lp->slice_words * 8u * 256u; }
    delete lp;
}

void like_pipeline_info(const lc_scan* s, uint64_t* bigram_bytes, uint64_t* unigram_bytes, double* build_ms, uint32_t* n_plans,
                        int32_t* build_pending) {
    const LikePipeline* lp = s->like;
    // (results of a finished job that the next evaluation will move in count as pending: their fields are not in place yet)
    *build_pending = lp && ((lp->flat_state.load() == 1 || lp->flat_state.load() == 2) || (lp->uni_state.load() == 1 || lp->uni_state.load() == 2)) ? 1 : 0;
    *bigram_bytes = lp && lp->d_slices ? lp->slices_bytes : 0;
    *unigram_bytes = lp && lp->d_uni ? lp->slice_words * 8u * 256u : 0;
    *build_ms = lp ? lp->flat_build_ms + lp->uni_build_ms : 0.0;
    *n_plans = lp ? uint32_t(lp->plans.size()) : 0u;
}

// One line on how `LIKE '%needle%'` was / would be evaluated on this scan (lc_scan_explain).  Caller holds s->mu.
std::string like_pipeline_explain(const lc_scan* s, const StrPredHost& sp) {
    const LikePipeline* lp = s->like;
    const int path = s->ctx->like_path;
    if (sp.p.mode == 1 && sp.p.needle_len == 1) {
        if (lp && lp->d_uni && (path == 0 || path == 4) && s->n >= s->ctx->like_pipeline_min_entries) {
            char buf[256];
            std::snprintf(buf, sizeof(buf), "k_like_scanall<unigram> (1-byte needle: exact from the scan-level unigram index, no walk; rows "
                          "through the keys; index %.2f GB built in %.1f ms)", double(lp->slice_words) * 8 * 256 / 1e9, lp->uni_build_ms);
            return buf;
        }
        return "k_str_pred (1-byte needle: no bigram, every fingerprint candidate walked by the many-candidate walkers)";
    }
    if (sp.p.mode == 1 && sp.p.eq_len != 0) {
        const bool flat_eq = lp && lp->built && lp->eligible && lp->flat && lp->eq_ok && (path == 0 || path == 4) &&
                             s->n >= s->ctx->like_pipeline_min_entries;
        if (!flat_eq) return "k_str_pred";
        for (const LikePlan& q : lp->plans)
            if (q.needle == sp.needle) {
                if (!q.use_lean && path != 4) return "k_str_pred (the literal's substring candidates are not selective)";
                char buf[256];
                std::snprintf(buf, sizeof(buf), "k_like_flat (string equality through the scan-level index: the %llu values that contain "
                              "the literal (%.2f per entry), then their length)", (unsigned long long)q.n_cand,
                              double(q.n_cand) / double(std::max<uint32_t>(s->n, 1)));
                return buf;
            }
        return "k_str_pred (literal not planned)";
    }
    if (sp.p.mode == 1 && sp.p.verify_len != 0) return "k_str_pred (needle over 63 bytes: automaton over its first 63, accepted values matched against the pattern)";
    if (path == 1 || path == 5 || s->n < s->ctx->like_pipeline_min_entries) return "k_str_pred";
    if (!lp || !lp->built) return "k_str_pred (scan not evaluated yet)";
    if (!lp->eligible) return "k_str_pred (entries without signature index / row lists)";
    if (!lp->lean_ok && !(lp->flat && (path == 0 || path == 4)))
        return "k_str_pred (entries of more than 8192 rows and no scan-level index)";
    if (path == 3) return "k_like_lean (forced for every needle)";
    const bool use_flat = lp->flat && (path == 0 || path == 4);
    if (path == 4 && use_flat) return "k_like_flat (forced for every needle)";
    if (path == 4) return "k_like_lean (forced for every needle; no scan-level index)";
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle) {
            char buf[384];
            int k = std::snprintf(buf, sizeof(buf), "%s: %llu candidates (%.2f per entry), %llu matching values, %llu hit rows at plan time",
                                  q.use_lean ? (use_flat ? "k_like_flat" : "k_like_lean") : "k_str_pred (needle not selective)",
                                  (unsigned long long)q.n_cand, double(q.n_cand) / double(std::max<uint32_t>(s->n, 1)),
                                  (unsigned long long)q.matches, (unsigned long long)q.hits);
            if (use_flat && k > 0 && size_t(k) < sizeof(buf))
                std::snprintf(buf + k, sizeof(buf) - size_t(k), "; %u slices probed, planned in %.2f ms; scan-level index %.2f GB built in %.1f ms",
                              q.n_probe, double(q.plan_ms), double(lp->slices_bytes) / 1e9, lp->flat_build_ms);
            return buf;
        }
    return "k_str_pred (needle not planned)";
}

// Bytes k_like_lean itself has to move for one evaluation (the numerator of an honest HBM-roofline fraction, like
// lc_scan_traffic_model's figure for k_str_pred): per entry its 64-byte record share and the needle's signature slices;
// per candidate its offset pair (8) and its compressed bytes; per matching value its list bounds (4) and 2 bytes per
// row; the entry's mask words out (+ the selection words of entries with hits).  0 when k_str_pred takes this needle.
// Caller holds s->mu.
uint64_t like_pipeline_bytes(const lc_scan* s, const StrPredHost& sp, bool with_counts, uint32_t sparse_flags) {
    const LikePipeline* lp = s->like;
    if (!lp || !lp->eligible || s->ctx->like_path == 1 || sp.p.verify_len != 0) return 0;
    if (sp.p.mode == 1 && sp.p.needle_len == 1) {
        // k_like_scanall<unigram>: per entry its descriptor, its words of ONE slice, the keys (2 n; an upper bound: entries
        // without a matching value skip them), validity words of nullable entries, mask words out
        if (!lp->d_uni || !(s->ctx->like_path == 0 || s->ctx->like_path == 4)) return 0;
        uint64_t b = with_counts ? uint64_t(s->n) * 4 : 0;
        for (const Entry& e : s->meta) {
            const uint64_t words = (uint64_t(e.len) + 63) / 64;
            b += sizeof(StrDesc) + 4 + uint64_t((e.sd.d + 63u) / 64u) * 8 + 2ull * e.len + words * 8 * (1 + (e.nullable ? 1 : 0));
        }
        return b;
    }
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle && (q.use_lean || s->ctx->like_path == 3 || s->ctx->like_path == 4)) {
            if (lp->flat && (s->ctx->like_path == 0 || s->ctx->like_path == 4)) {
                // k_like_flat: per group its record line and 1 KB of every probed slice; per group with a candidate the
                // entries' walk fields; per candidate its offset pair (8), list bounds (4) and compressed bytes; 2 bytes
                // per hit row; the mask words out
                uint16_t bits[kMaxSigProbeWide];
                const uint64_t nb = std::min<uint64_t>(flat_needle_bits(sp.needle, bits), q.n_probe ? q.n_probe : kMaxSigProbeWide);
                // sparse_flags: 2 = the caller takes no mask (k_like_flat then stores no mask words), 4 = it takes the hit list
                // (8 bytes per hit row)
                uint64_t b = uint64_t(lp->n_group_slots) * (kFlatHotBytes + nb * lp->probe_words * 8) +
                             ((sparse_flags & 2u) ? 0 : s->seg_offsets.back() * 8) + ((sparse_flags & 4u) ? q.hits * 8 : 0) +
                             (with_counts ? uint64_t(s->n) * 4 : 0);
                b += std::min<uint64_t>(q.n_cand, lp->n_groups) * 64 * kFlatMaxE + q.n_cand * 12 + q.cand_bytes + q.hits * 2;
                if (sp.p.eq_len != 0) b += q.matches * 8;  // `=`: the prefix key of every value that contains the literal
                return b;
            }
            uint64_t b = uint64_t(lp->n_lean) * 16 + uint64_t(s->n) * 64 + s->seg_offsets.back() * 8 + (with_counts ? uint64_t(s->n) * 4 : 0);
            for (const Entry& e : s->meta) b += uint64_t((e.sd.d + 63u) / 64u) * 8 * sp.p.n_sig_wide;
            b += q.n_cand * 8 + q.cand_bytes + q.matches * 4 + q.hits * 2;
            return b;
        }
    return 0;
}

// The scan-level index of a scan that does not have it yet.  LC_OPT_LIKE_INDEX_ASYNC = 0: built now, the caller waits (the
// behaviour before round 6).  Otherwise the context's builder thread builds it on its own stream while this and the next
// evaluations run over the entry-level index, the way the reference keeps its prefilter construction out of the read path
// (byte_view_array/conversions.rs:353-355: at insert time).  The first attempt never evicts another scan's cached index; a scan
// that was turned away for that reason tries again, with eviction, once it has served kEvictAfterEvals evaluations.
// Caller holds s->mu.
// -DLC_FB_POLITE=1: the builder thread's builds run ONE workgroup per CU (build_flat's `polite`).  That form was this round's
// first answer to "a query launched during a build waited 4.2 ms for its 30 us kernel"; the answer that holds is the streams'
// priorities — queries run on streams of the highest priority, the builder's has the lowest, and a build workgroup lives ~50 us,
// so a query's workgroups take the next free slots.  Measured with scripts/query_during_build.py (COUNT(*) queries on one table
// while another table's index is built, 8 ms windows): two workgroups per CU — queries median 23-26 us (quiet device: 23.4), max
// 66-81 us, build kernel 4.0 ms; one per CU — median 28-38 us, max 46-64 us, build kernel 5.0 ms.  The full-occupancy build
// disturbs the median less (it is over sooner) and the worst query by 20 us more: it is the default.
#ifndef LC_FB_POLITE
#define LC_FB_POLITE 0
#endif
static lc_status want_flat_index(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream) {
    if (!lp->eligible || lp->flat) return LC_OK;
    LC_PHASE("like: want_flat_index");
    const uint64_t events = ctx->index_events.load();
    if (lp->flat_no_room) {  // room is looked for again only after index memory has been given back somewhere
        if (lp->flat_retry_events == events) return LC_OK;
        lp->flat_no_room = false;
        lp->flat_tried = false;
        if (lp->flat_state.load(std::memory_order_acquire) == 3) lp->flat_state.store(0, std::memory_order_release);
    }
    if (lp->flat_state.load(std::memory_order_acquire) != 0) return LC_OK;
    lp->flat_retry_events = events;
    if (!ctx->like_index_async.load()) {
        if (lp->flat_tried) return LC_OK;
        const lc_status st = build_flat(ctx, s, lp, stream);
        lp->flat_no_room = !lp->flat && lp->flat_why == 2;
        lp->flat_state.store(3, std::memory_order_release);
        return st;
    }
    if (lp->flat_needs_evict && lp->like_evals < kEvictAfterEvals) return LC_OK;
    const bool allow_evict = lp->flat_needs_evict;
    LikePipeline* pend = new LikePipeline();
    lp->flat_pending = pend;
    lp->flat_state.store(1, std::memory_order_release);
    if (!lp->gate) lp->gate = std::make_shared<BuildGate>();
    std::shared_ptr<BuildGate> gate = lp->gate;
    lp->flat_job = builder_submit(ctx, [ctx, s, lp, pend, allow_evict, gate](hipStream_t st) {
        build_gate_pass(gate);
        try {
            (void)build_flat(ctx, s, pend, st, allow_evict, LC_FB_POLITE != 0);
        } catch (...) {
            pend->flat = false;
        }
        lp->flat_state.store(2, std::memory_order_release);
    });
    return LC_OK;
}
static lc_status want_unigram_index(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream) {
    if (!lp->eligible || !lp->flat || lp->d_uni || lp->uni_tried || lp->uni_state.load(std::memory_order_acquire) != 0) return LC_OK;
    if (!ctx->like_index_async.load()) {
        lp->uni_tried = true;
        lp->uni_state.store(3, std::memory_order_release);
        return build_unigram(ctx, s, lp, stream, &lp->d_uni, &lp->uni_build_ms);
    }
    lp->uni_state.store(1, std::memory_order_release);
    if (!lp->gate) lp->gate = std::make_shared<BuildGate>();
    std::shared_ptr<BuildGate> gate = lp->gate;
    lp->uni_job = builder_submit(ctx, [ctx, s, lp, gate](hipStream_t st) {
        build_gate_pass(gate);
        uint64_t* u = nullptr;
        double ms = 0;
        try {
            (void)build_unigram(ctx, s, lp, st, &u, &ms);
        } catch (...) {
        }
        lp->uni_pending = u;
        lp->uni_pending_ms = ms;
        lp->uni_state.store(2, std::memory_order_release);
    });
    return LC_OK;
}

// Caller holds s->mu and has built the automata of `sp` (sp.p.automata).  *handled = true: the evaluation was launched.
static lc_status like_pipeline_eval_gated(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                                          bool* handled, bool* many_candidates);
lc_status like_pipeline_eval(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                             bool* handled, bool* many_candidates) {
    struct OpenGate {  // whatever way the call leaves: a build it submitted may start once what it launched has run
        lc_scan* s;
        hipStream_t stream;
        ~OpenGate() {
            LikePipeline* lp = s->like;
            if (!lp || !lp->gate) return;
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess && hipEventRecord(e, stream) == hipSuccess) {
                lp->gate->ev = e;
            } else {
                (void)hipGetLastError();
                if (e) (void)hipEventDestroy(e);
            }
            lp->gate->open.store(1, std::memory_order_release);
            lp->gate.reset();
        }
    } open_gate{s, stream};
    return like_pipeline_eval_gated(ctx, s, sp, L, stream, handled, many_candidates);
}
static lc_status like_pipeline_eval_gated(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                                          bool* handled, bool* many_candidates) {
    *handled = false;
    *many_candidates = false;
    const StrPred& p = sp.p;
    // a 1-byte needle has no bigram: its candidates are whatever the 32-bucket fingerprint lets through, which for a byte
    // that occurs in the column is most of every dictionary (the many-candidate kernel falls back to the lane-parallel
    // walk by itself for an entry with less than a wave of candidates)
    if (p.mode == 1 && p.needle_len == 1 && (p.op == LC_OP_LIKE || p.op == LC_OP_NOT_LIKE)) {
        *many_candidates = true;
        // ... unless the scan has (or can get) the unigram index: the needle byte's slice IS the dictionary result
        const bool want = (ctx->like_path == 0 || ctx->like_path == 4) && s->n >= ctx->like_pipeline_min_entries &&
                          s->d_wg_ranges && s->n_wg_ranges > 0 && !L.d_cand_bytes && !L.d_own_bytes && !LC_ABL(p.debug_flags != 0);
        if (want) {
            if (!s->like) s->like = like_pipeline_adopt(ctx, s);
            if (!s->like) s->like = new LikePipeline();
            LikePipeline* lp = s->like;
            if (!lp->built) {
                const lc_status st = build_index(ctx, s, lp, stream);
                if (st != LC_OK) return st;
            }
            promote_builds(ctx, lp);
            lp->like_evals++;
            {
                lc_status st = want_flat_index(ctx, s, lp, stream);
                if (st == LC_OK) st = want_unigram_index(ctx, s, lp, stream);
                if (st != LC_OK) return st;
            }
            if (lp->eligible && lp->flat && lp->d_uni) {
                LC_HIP(launch_like_scanall(s->d_wg_ranges, s->n_wg_ranges, p, L, L.d_total_acc, stream,
                                           lp->d_uni + uint64_t(sp.needle[0]) * flat_slice_stride(lp->n_group_slots, lp->group_words),
                                           lp->d_dst_word + s->n));
                *handled = true;
                s->last_like_kernel = LC_LIKE_KERNEL_UNIGRAM;
                return LC_OK;
            }
        }
    }
    // (needles over 63 bytes: their matches are verified against the whole pattern, which k_str_pred does)
    if (p.mode != 1 || (p.op != LC_OP_LIKE && p.op != LC_OP_NOT_LIKE) || !p.use_fingerprints || p.n_sig_bits == 0 || p.needle_len < 2 ||
        automaton_image_bytes(p.needle_len) == 0 || p.verify_len != 0 || L.d_valid || L.d_cand_bytes || L.d_own_bytes || LC_ABL(p.debug_flags != 0))
        return LC_OK;
    if (s->n < ctx->like_pipeline_min_entries || ctx->like_path == 1 || ctx->like_path == 5) return LC_OK;
    LC_PHASE("like_pipeline_eval (all)");
    if (!s->like) {
        LC_PHASE("like: adopt");
        s->like = like_pipeline_adopt(ctx, s);
    }
    if (!s->like) s->like = new LikePipeline();
    LikePipeline* lp = s->like;
    if (!lp->built) {
        const lc_status st = build_index(ctx, s, lp, stream);
        if (st != LC_OK) return st;
    }
    if (!lp->eligible) return LC_OK;
    {
        LC_PHASE("like: promote");
        promote_builds(ctx, lp);
    }
    lp->like_evals++;
    const bool want_flat = ctx->like_path == 0 || ctx->like_path == 4;
    if (want_flat) {
        const lc_status st = want_flat_index(ctx, s, lp, stream);
        if (st != LC_OK) return st;
    }
    const bool use_flat = want_flat && lp->flat;
    if (!use_flat && !lp->lean_ok) return LC_OK;  // entries of more than 8,192 rows without the scan-level index: k_str_pred
    if (p.eq_len != 0 && !(use_flat && lp->eq_ok)) return LC_OK;  // `=` / `<>`: the scan-level kernel only
    LikePlan* plan = nullptr;
    for (LikePlan& q : lp->plans)
        if (q.needle == sp.needle) plan = &q;
    if (plan) {
        LC_PHASE("like: settle plan");
        (void)settle_plan(ctx, s, plan, false);
    }
    // A plan is due for a needle the scan has not seen, and once more when the scan-level index has arrived under a plan made
    // on the entry-level one.  Where the evaluation asked for IS what a trial would run — a plain LIKE over every row — it is
    // launched with the statistics counters attached and the plan settles later (no trial, no host round trip: 0.37 ms of a
    // 12,207-entry scan's first evaluation); a long needle on the scan-level index keeps the two-trial choice of its slices.
    uint16_t nbits[kMaxSigProbeWide];
    const uint32_t nb = use_flat ? flat_needle_bits(sp.needle, nbits) : 0u;
    const bool due = !plan || (use_flat && !plan->flat_planned && !plan->pending);
    bool on_the_fly = due && p.op == LC_OP_LIKE && !L.d_selection && p.eq_len == 0 && (!use_flat || nb <= uint32_t(kMaxSigProbe)) &&
                      ctx->like_path != 3 && ctx->like_path != 4;
    bool created = false;
    if (due && !plan) {
        created = true;
        if (lp->plans.size() >= kMaxPlans) {  // least recently used goes
            size_t victim = 0;
            for (size_t i = 1; i < lp->plans.size(); i++)
                if (lp->plans[i].last_use < lp->plans[victim].last_use) victim = i;
            release_pending_plan(ctx, &lp->plans[victim]);
            lp->plans.erase(lp->plans.begin() + long(victim));
        }
        lp->plans.emplace_back();
        plan = &lp->plans.back();
        plan->needle = sp.needle;
        plan->use_lean = true;
    }
    if (on_the_fly && !plan_slot_take(ctx, plan)) on_the_fly = false;  // (no slot free: a trial run plans this needle)
    if (on_the_fly) {
        LC_PHASE("like: on-the-fly plan + launch");
        plan->pending = true;
        plan->flat_planned = use_flat;
        plan->n_probe = use_flat ? std::min<uint32_t>(nb, uint32_t(kMaxSigProbe)) : 0u;
        plan->last_use = ++lp->tick;
        LC_HIP(hipMemsetAsync(plan->d_res, 0, kStatWords * 8, stream));
        ScanLaunch L2 = L;
        if (!L2.d_total_out) L2.d_total_out = reinterpret_cast<uint64_t*>(plan->d_res);
        const lc_status st = use_flat ? run_flat(lp, sp, L2, stream, plan->d_res + kStatStride, false, plan->n_probe)
                                      : run_lean(lp, p, L2, stream, plan->d_res + kStatStride);
        if (st != LC_OK) return st;
        LC_HIP(hipMemcpyAsync(plan->h_res + 1, plan->d_res + 1, (kStatWords - 1u) * 8, hipMemcpyDeviceToHost, stream));
        LC_HIP(hipMemcpyAsync(plan->h_res, L2.d_total_out, 8, hipMemcpyDeviceToHost, stream));
        LC_HIP(hipEventRecord(plan->ev, stream));
        *handled = true;
        s->last_native_hits = use_flat && L.d_hits != nullptr;
        s->last_like_kernel = use_flat ? LC_LIKE_KERNEL_FLAT : LC_LIKE_KERNEL_LEAN;
        return LC_OK;
    }
    if (due) {
        LikePlan again;
        const lc_status st = make_plan(ctx, s, lp, sp, stream, &again);
        if (st != LC_OK) {
            if (created) lp->plans.pop_back();
            return st;
        }
        *plan = std::move(again);
    }
    plan->last_use = ++lp->tick;
    // a needle the plan found unselective: k_str_pred takes it, and with at least a wave of candidates per entry its
    // sequential walker (every lane works through its own share of the list) beats the lane-parallel one
    if (!plan->use_lean) *many_candidates = plan->n_cand >= uint64_t(kWave) * s->n;
    if (!plan->use_lean && ctx->like_path != 3 && ctx->like_path != 4) return LC_OK;
    LC_PHASE("like: launch");
    const lc_status st = use_flat ? run_flat(lp, sp, L, stream, nullptr, false, plan->n_probe ? plan->n_probe : kMaxSigProbeWide)
                                  : run_lean(lp, p, L, stream);
    if (st == LC_OK) {
        *handled = true;
        s->last_native_hits = use_flat && L.d_hits != nullptr;  // (k_like_flat appends the hit list itself)
        s->last_like_kernel = use_flat ? LC_LIKE_KERNEL_FLAT : LC_LIKE_KERNEL_LEAN;
    }
    return st;
}

hipError_t warm_code_object_like_pipeline() {  // (see warm_code_object_kernels)
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_flat_build<false>));
}

}  // namespace lc
