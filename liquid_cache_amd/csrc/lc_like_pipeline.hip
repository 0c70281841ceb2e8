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
// a needle is "selective" (worth this kernel) up to this many hit rows per 1024 rows of the scan
constexpr uint32_t kMaxHitsPer1024 = 16;

struct LikePlan {
    std::vector<uint8_t> needle;
    bool use_lean = false;
    uint64_t hits = 0, n_cand = 0, cand_bytes = 0, matches = 0;  // of the trial run (byte accounting, EXPLAIN)
    uint64_t last_use = 0;
};

struct LikePipeline {
    bool built = false, eligible = false;
    LeanRec* d_lean = nullptr;  // one record per workgroup
    uint32_t n_lean = 0;
    unsigned long long* d_total_acc = nullptr;
    std::vector<LikePlan> plans;
    uint64_t tick = 0;
};

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
    constexpr uint32_t kMaskBytes = kPostMaxRows / 8u;
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
                    const bool changed = prev != s_in;
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
                    atomicAdd(a.stats, (unsigned long long)min(count - b0, uint32_t(kWave)));
                    atomicAdd(a.stats + 1, (unsigned long long)lb);
                    atomicAdd(a.stats + 2, (unsigned long long)__popcll(matched));
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

hipError_t launch_lean(int n_sig, bool negated, const LeanArgs& a, uint32_t n_recs, hipStream_t stream) {
    if (n_recs == 0) return hipSuccess;
    typedef void (*Kern)(LeanArgs);
    static const Kern table[2][kMaxSigProbe] = {
        {k_like_lean<1, false>, k_like_lean<2, false>, k_like_lean<3, false>, k_like_lean<4, false>, k_like_lean<5, false>,
         k_like_lean<6, false>, k_like_lean<7, false>, k_like_lean<8, false>},
        {k_like_lean<1, true>, k_like_lean<2, true>, k_like_lean<3, true>, k_like_lean<4, true>, k_like_lean<5, true>,
         k_like_lean<6, true>, k_like_lean<7, true>, k_like_lean<8, true>}};
    const size_t lds = automaton_image_bytes(a.nl) + kLeanWaves * (kLeanE * (kPostMaxRows / 8u) + kLeanCap * 4u + 80u);
    const uint32_t grid = LC_LEAN_XCD ? (n_recs + 7u) / 8u * 8u : n_recs;
    hipLaunchKernelGGL(table[negated ? 1 : 0][n_sig - 1], dim3(grid), dim3(kLeanWaves * 64), lds, stream, a);
    return hipGetLastError();
}

// per-workgroup records, built once per scan
lc_status build_index(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream) {
    lp->built = true;
    lp->eligible = false;
    if (!s->is_str || s->n == 0) return LC_OK;
    for (const Entry& e : s->meta) {
        if (e.sd.d == 0) continue;  // an all-null entry has no dictionary: no candidates, its mask words are zero
        if (!e.sd.signatures || !e.sd.postings || !e.sd.fingerprints || e.sd.n > kPostMaxRows) return LC_OK;
    }
    // consecutive entries, at most kLeanWaves * kLeanE, never across a symbol-table change
    std::vector<LeanRec> lean;
    for (uint32_t b = 0, i = 1; i <= s->n; i++) {
        if (i == s->n || i - b == kLeanWaves * kLeanE || s->meta[i].sd.symtab_slot != s->meta[b].sd.symtab_slot) {
            LeanRec r;
            std::memset(&r, 0, sizeof(r));
            r.begin = b;
            r.end = i;
            r.slot = s->meta[b].sd.symtab_slot;
            for (uint32_t k = b; k < i; k++) {
                const StrDesc& d = s->meta[k].sd;
                r.e[k - b] = LeanEntry{d.signatures, d.residuals, d.fsst, d.postings, d.mask_word_off, d.slope, d.intercept,
                                       d.d, d.n, d.offset_bytes, (d.d + 63u) / 64u, d.validity, d.fingerprints};
            }
            lean.push_back(r);
            b = i;
        }
    }
    lp->n_lean = uint32_t(lean.size());
    lp->d_lean = static_cast<LeanRec*>(pool_alloc(ctx, std::max<size_t>(lean.size(), 1) * sizeof(LeanRec)));
    lp->d_total_acc = static_cast<unsigned long long*>(pool_alloc(ctx, size_t(kTotalWords) * 8));
    if (!lp->d_lean || !lp->d_total_acc) return fail(LC_ERR_OOM, "hipMalloc (LIKE records)");
    LC_HIP(hipMemcpyAsync(lp->d_lean, lean.data(), lean.size() * sizeof(LeanRec), hipMemcpyHostToDevice, stream));
    LC_HIP(hipMemsetAsync(lp->d_total_acc, 0, size_t(kTotalWords) * 8, stream));  // once: launches leave it zero
    LC_HIP(hipStreamSynchronize(stream));  // `lean` is a local
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

// one trial evaluation into scratch: how many rows does the needle hit?
lc_status make_plan(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, const StrPredHost& sp, hipStream_t stream, LikePlan* plan) {
    plan->needle = sp.needle;
    plan->use_lean = false;
    const uint64_t words = std::max<uint64_t>(s->seg_offsets.back(), 1);
    uint64_t* d_scratch = static_cast<uint64_t*>(pool_alloc(ctx, words * 8 + 32));  // mask | COUNT(*) | 3 statistics
    struct Tmp {
        lc_ctx* c; void* p; hipStream_t st;
        ~Tmp() { (void)hipStreamSynchronize(st); pool_release(c, p); }
    } tmp{ctx, d_scratch, stream};
    if (!d_scratch) return fail(LC_ERR_OOM, "hipMalloc (LIKE plan)");
    ScanLaunch L{};
    L.d_hit = d_scratch;
    L.d_total_out = d_scratch + words;
    LC_HIP(hipMemsetAsync(d_scratch + words, 0, 32, stream));
    // (always as LIKE: the plan belongs to the needle, NOT LIKE is selective exactly when LIKE is)
    const lc_status rc = run_lean(lp, sp.p, L, stream, reinterpret_cast<unsigned long long*>(d_scratch + words + 1), true);
    if (rc != LC_OK) return rc;
    uint64_t res[4] = {0, 0, 0, 0};
    LC_HIP(hipMemcpyAsync(res, d_scratch + words, 32, hipMemcpyDeviceToHost, stream));
    LC_HIP(hipStreamSynchronize(stream));
    plan->hits = res[0];
    plan->n_cand = res[1];
    plan->cand_bytes = res[2];
    plan->matches = res[3];
    plan->use_lean = plan->hits * 1024u <= uint64_t(kMaxHitsPer1024) * std::max<uint64_t>(s->total_rows, 1024);
    return LC_OK;
}

}  // namespace

void like_pipeline_destroy(lc_ctx* ctx, LikePipeline* lp) {
    if (!lp) return;
    pool_release(ctx, lp->d_lean);
    pool_release(ctx, lp->d_total_acc);
    delete lp;
}

// One line on how `LIKE '%needle%'` was / would be evaluated on this scan (lc_scan_explain).  Caller holds s->mu.
std::string like_pipeline_explain(const lc_scan* s, const StrPredHost& sp) {
    const LikePipeline* lp = s->like;
    const int path = s->ctx->like_path;
    if (sp.p.mode == 1 && sp.p.needle_len == 1) return "k_str_pred (1-byte needle: no bigram, every fingerprint candidate walked by the many-candidate walkers)";
    if (path == 1 || s->n < s->ctx->like_pipeline_min_entries) return "k_str_pred";
    if (!lp || !lp->built) return "k_str_pred (scan not evaluated yet)";
    if (!lp->eligible) return "k_str_pred (entries without signature index / row lists)";
    if (path == 3) return "k_like_lean (forced for every needle)";
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle) {
            char buf[256];
            std::snprintf(buf, sizeof(buf), "%s: %llu candidates (%.1f per entry), %llu matching values, %llu hit rows at plan time",
                          q.use_lean ? "k_like_lean" : "k_str_pred (needle not selective)", (unsigned long long)q.n_cand,
                          double(q.n_cand) / double(std::max<uint32_t>(s->n, 1)), (unsigned long long)q.matches,
                          (unsigned long long)q.hits);
            return buf;
        }
    return "k_str_pred (needle not planned)";
}

// Bytes k_like_lean itself has to move for one evaluation (the numerator of an honest HBM-roofline fraction, like
// lc_scan_traffic_model's figure for k_str_pred): per entry its 64-byte record share and the needle's signature slices;
// per candidate its offset pair (8) and its compressed bytes; per matching value its list bounds (4) and 2 bytes per
// row; the entry's mask words out (+ the selection words of entries with hits).  0 when k_str_pred takes this needle.
// Caller holds s->mu.
uint64_t like_pipeline_bytes(const lc_scan* s, const StrPredHost& sp, bool with_counts) {
    const LikePipeline* lp = s->like;
    if (!lp || !lp->eligible || s->ctx->like_path == 1) return 0;
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle && (q.use_lean || s->ctx->like_path == 3)) {
            uint64_t b = uint64_t(lp->n_lean) * 16 + uint64_t(s->n) * 64 + s->seg_offsets.back() * 8 + (with_counts ? uint64_t(s->n) * 4 : 0);
            for (const Entry& e : s->meta) b += uint64_t((e.sd.d + 63u) / 64u) * 8 * sp.p.n_sig_wide;
            b += q.n_cand * 8 + q.cand_bytes + q.matches * 4 + q.hits * 2;
            return b;
        }
    return 0;
}

// Caller holds s->mu and has built the automata of `sp` (sp.p.automata).  *handled = true: the evaluation was launched.
lc_status like_pipeline_eval(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                             bool* handled, bool* many_candidates) {
    *handled = false;
    *many_candidates = false;
    const StrPred& p = sp.p;
    // a 1-byte needle has no bigram: its candidates are whatever the 32-bucket fingerprint lets through, which for a byte
    // that occurs in the column is most of every dictionary (the many-candidate kernel falls back to the lane-parallel
    // walk by itself for an entry with less than a wave of candidates)
    if (p.mode == 1 && p.needle_len == 1 && (p.op == LC_OP_LIKE || p.op == LC_OP_NOT_LIKE)) *many_candidates = true;
    if (p.mode != 1 || (p.op != LC_OP_LIKE && p.op != LC_OP_NOT_LIKE) || !p.use_fingerprints || p.n_sig_bits == 0 || p.needle_len < 2 ||
        automaton_image_bytes(p.needle_len) == 0 || L.d_valid || L.d_cand_bytes || L.d_own_bytes || LC_ABL(p.debug_flags != 0))
        return LC_OK;
    if (s->n < ctx->like_pipeline_min_entries || ctx->like_path == 1) return LC_OK;
    if (!s->like) s->like = new LikePipeline();
    LikePipeline* lp = s->like;
    if (!lp->built) {
        const lc_status st = build_index(ctx, s, lp, stream);
        if (st != LC_OK) return st;
    }
    if (!lp->eligible) return LC_OK;
    LikePlan* plan = nullptr;
    for (LikePlan& q : lp->plans)
        if (q.needle == sp.needle) plan = &q;
    if (!plan) {
        if (lp->plans.size() >= kMaxPlans) {  // least recently used goes
            size_t victim = 0;
            for (size_t i = 1; i < lp->plans.size(); i++)
                if (lp->plans[i].last_use < lp->plans[victim].last_use) victim = i;
            lp->plans.erase(lp->plans.begin() + long(victim));
        }
        LikePlan fresh;
        const lc_status st = make_plan(ctx, s, lp, sp, stream, &fresh);
        if (st != LC_OK) return st;
        lp->plans.push_back(std::move(fresh));
        plan = &lp->plans.back();
    }
    plan->last_use = ++lp->tick;
    // a needle the plan found unselective: k_str_pred takes it, and with at least a wave of candidates per entry its
    // sequential walker (every lane works through its own share of the list) beats the lane-parallel one
    if (!plan->use_lean) *many_candidates = plan->n_cand >= uint64_t(kWave) * s->n;
    if (!plan->use_lean && ctx->like_path != 3) return LC_OK;
    const lc_status st = run_lean(lp, p, L, stream);
    if (st == LC_OK) *handled = true;
    return st;
}

}  // namespace lc
