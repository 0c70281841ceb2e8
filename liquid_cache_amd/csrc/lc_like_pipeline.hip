// Scan-level pipeline for SELECTIVE `LIKE '%needle%'` over byte-view columns that carry the bigram signature index and
// the inverted row lists (the headline scan: ClickBench q20 / "Q21", URL LIKE '%google%').
//
// What it replaces: k_str_pred puts ONE WAVE on every entry and takes it through five dependent round trips (descriptor
// -> signature slices -> candidate offsets -> compressed bytes -> row lists) with ~1,300 instructions, most of which run
// once per entry whatever the entry holds; measured (round 2) it is bound by that chain, not by bandwidth.  Here the
// same work is cut into two throughput-shaped kernels over the WHOLE scan:
//
//   k_like_probe   one lane per 64 dictionary values of the scan (a flat index over all entries, built once per scan):
//                  AND of the needle's signature slices (coalesced 8-byte loads), candidates compacted into ONE list for
//                  the scan, written at positions known from the plan (no atomics; the list is ordered by entry, hence
//                  grouped by symbol table).  The same kernel zero-fills the hit mask and the per-entry counts.
//   k_like_walk    one wave per 64 candidates of ONE symbol table, whatever entries they belong to: the needle's
//                  automaton folded over the table's FSST symbols sits in LDS (k_str_automata's image), the candidates'
//                  compressed bytes are cut into 8-byte words and walked one lane per word (the walker of k_str_pred,
//                  exact at its fixpoint); the rows of a matching dictionary value are read from the entry's inverted
//                  row list and OR-ed into the mask with far atomics (a selective needle matches a few thousand rows
//                  of 100 M), hits are added to the per-entry counts and to the fused COUNT(*).
//
// Reference counterpart: LiquidByteViewArray::compare_like_substring — fingerprint filter, decode + memmem of the
// candidates, map_dictionary_results_to_array_results (byte_view_array/comparisons.rs:159-183, 325-347, 598-651).
// Results are identical (the candidates of the signature AND are a superset of the matches, the walk is exact).
//
// The PLAN (once per scan and needle, cached): the probe runs in count mode, the per-wave candidate counts come back to
// the host (the one synchronisation), their prefix sums become the write positions and the chunk schedule of the walk;
// then one trial run counts the hits.  Entries are immutable while a scan pins them, so the counts are a property of
// (scan, needle) and every later evaluation runs without a host round trip.  Needles that are not selective (many
// candidates or many hit rows: the atomics would dominate) keep k_str_pred, as do scans with entries that lack the
// index, NOT LIKE, validity outputs and the byte-accounting pass.
#include "lc_device.hpp"
#include "lc_internal.hpp"

namespace lc {

// one record per flat word (64 dictionary values of one entry): everything the probe needs in ONE coalesced 16-byte load
struct alignas(16) FlatRec {
    uint64_t sig_addr;  // address of this word in slice 0 of the entry's signatures; slice b is nw words further per b
    uint32_t entry;     // scan index of the entry, 0xFFFFFFFF: padding (keeps a probe wave inside one symbol table)
    uint16_t nw;        // ceil(d / 64)
    uint16_t w;         // index of this word inside the entry
};
static_assert(sizeof(FlatRec) == 16, "FlatRec layout");

struct alignas(16) LikeEntryRef {
    const uint8_t* fsst;
    const uint8_t* residuals;
    const uint16_t* postings;
    uint64_t mask_word_off;
    int32_t slope, intercept;
    uint32_t offset_bytes, d;
};
static_assert(sizeof(LikeEntryRef) == 48, "LikeEntryRef layout");

// a candidate as the walk wants it: where its compressed bytes are — no descriptor, no offset load left to do
struct alignas(16) CandRec {
    uint64_t abs_start;  // address of the first compressed byte
    uint32_t len;        // compressed bytes
    uint32_t key;        // dictionary key inside its entry
};
static_assert(sizeof(CandRec) == 16, "CandRec layout");

struct LikeChunk {
    uint32_t first, count;  // candidates [first, first + count) of the scan's list: one wave, one symbol table, and (unless
                            // a single value is longer) at most 64 eight-byte words = ONE pass of the lane-parallel walk
};

// ---- the lean one-kernel form (k_like_lean): what a workgroup of four waves needs for its (at most four) entries of ONE
// symbol table, fetched with scalar loads from an address that follows from blockIdx alone
struct alignas(16) LeanEntry {
    const uint64_t* sig;
    const uint8_t* residuals;
    const uint8_t* fsst;
    const uint16_t* postings;
    uint64_t mask_word_off;
    int32_t slope, intercept;
    uint32_t d, n;
    uint32_t offset_bytes, nw;
};
static_assert(sizeof(LeanEntry) == 64, "LeanEntry layout");
struct alignas(16) LeanRec {
    uint32_t begin, end;  // entries [begin, end) of the scan, end - begin <= 4: wave w takes entry begin + w
    uint32_t slot;        // their symbol table
    uint32_t pad;
    LeanEntry e[4];
};
static_assert(sizeof(LeanRec) == 272, "LeanRec layout");
constexpr uint32_t kLeanCap = 512;  // candidate keys a wave lists in LDS before it walks them
#ifndef LC_LEAN_WAVES
#define LC_LEAN_WAVES 4
#endif
constexpr uint32_t kLeanWaves = LC_LEAN_WAVES;  // waves (= entries) per workgroup of k_like_lean: they share the LDS automaton
static_assert(kLeanWaves >= 1 && kLeanWaves <= 4, "a LeanRec holds four entries");
// -DLC_LEAN_STOP=n (variant builds only, results are WRONG): leave the kernel after phase n — 1 probe, 2 offset pairs,
// 3 compressed words — to measure where the time goes
#ifndef LC_LEAN_STOP
#define LC_LEAN_STOP 0
#endif

constexpr uint32_t kProbeThreads = 256;
constexpr uint32_t kProbeRound = 128;                    // candidates a probe wave redistributes through LDS at a time
constexpr uint32_t kWalkWaves = 4;                       // waves (= chunks) per workgroup of the walk; they share the LDS automaton
constexpr uint32_t kMaxPlans = 8;
// a needle is "selective" (worth the pipeline) up to this many signature candidates per entry on average and this many
// hit rows per 1024 rows of the scan; beyond, k_str_pred's per-entry key mapping is the better algorithm
constexpr uint32_t kMaxCandPerEntry = 48;
constexpr uint32_t kMaxHitsPer1024 = 16;

struct LikePlan {
    std::vector<uint8_t> needle;
    bool use_pipeline = false;
    uint32_t n_cand = 0, n_chunks = 0, n_wgs = 0;
    uint64_t hits = 0;
    uint64_t cand_bytes = 0;         // compressed bytes of the candidates (byte accounting)
    uint64_t matches = 0;            // dictionary values that matched
    uint32_t* d_wave_off = nullptr;  // write position of every probe wave (n_k1_waves)
    LikeChunk* d_chunks = nullptr;   // kWalkWaves per workgroup (padded with empty chunks where the table changes)
    uint32_t* d_wg_slot = nullptr;   // symbol-table slot of every walk workgroup
    CandRec* d_cand = nullptr;
    uint32_t* d_cand_entry = nullptr;
    uint64_t last_use = 0;
};

struct LikePipeline {
    bool built = false, eligible = false;
    uint32_t n_flat = 0;         // flat words incl. the padding that keeps a probe wave inside one symbol table
    uint32_t n_k1_waves = 0;
    FlatRec* d_flat = nullptr;
    LikeEntryRef* d_refs = nullptr;
    LeanRec* d_lean = nullptr;   // one record per workgroup of k_like_lean
    uint32_t n_lean = 0;
    unsigned long long* d_total_acc = nullptr;
    std::vector<LikePlan> plans;
    uint64_t tick = 0;
};

namespace {

struct ProbeArgs {
    const FlatRec* flat;
    const LikeEntryRef* refs;
    uint32_t n_flat;
    uint32_t n_entries;
    uint16_t sig_bits[kMaxSigProbe];
    uint32_t* wave_count;      // count mode
    const uint32_t* wave_off;  // fill mode
    CandRec* cand;
    uint32_t* cand_entry;
    uint64_t* mask;            // fill mode: zero-filled here
    uint64_t mask_words;
    uint32_t* counts;          // optional, zero-filled here
    uint64_t* total_zero;      // optional: COUNT(*) word to clear (scans without a single candidate)
};

// N = distinct signature bits of the needle (1..8): every slice load is issued before the first use, none is repeated
template <int N, bool kCount>
__global__ __launch_bounds__(kProbeThreads) void k_like_probe(ProbeArgs a) {
    __shared__ uint32_t lds_entry[kProbeThreads / 64][kProbeRound];
    __shared__ uint16_t lds_key[kProbeThreads / 64][kProbeRound];
    const uint32_t gtid = blockIdx.x * kProbeThreads + threadIdx.x;
    if (!kCount) {
        // the hit mask starts all clear (the walk ORs the hit rows in); 16-byte coalesced stores over the whole grid
        const uint32_t nthreads = gridDim.x * kProbeThreads;
        GlobalMutPtr<u32x4> m4 = reinterpret_cast<GlobalMutPtr<u32x4>>(as_global_mut(a.mask));
        const uint64_t n16 = a.mask_words >> 1;
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (uint64_t i = gtid; i < n16; i += nthreads) m4[i] = z;
        if ((a.mask_words & 1u) && gtid == 0) as_global_mut(a.mask)[a.mask_words - 1] = 0;
        if (a.counts)
            for (uint32_t i = gtid; i < a.n_entries; i += nthreads) as_global_mut(a.counts)[i] = 0;
        if (a.total_zero && gtid == 0) as_global_mut(a.total_zero)[0] = 0;
    }
    if (gtid >= a.n_flat) return;  // n_flat is a multiple of 64: whole waves leave
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    const u32x4 fr = *reinterpret_cast<GlobalPtr<u32x4>>(as_global(a.flat + gtid));
    const uint32_t e = fr.z, nw = fr.w & 0xFFFFu, w = fr.w >> 16;
    uint64_t m = 0;
    if (e != 0xFFFFFFFFu) {
        const uint64_t* sig = reinterpret_cast<const uint64_t*>(uint64_t(fr.x) | (uint64_t(fr.y) << 32));
        uint64_t sv[N];
#pragma unroll
        for (int k = 0; k < N; k++) sv[k] = as_global(sig)[size_t(a.sig_bits[k]) * nw];
        m = sv[0];
#pragma unroll
        for (int k = 1; k < N; k++) m &= sv[k];
        // (no bit beyond the dictionary can be set: the index builders only set bits of real values and lc_stage_indexed
        // rejects blobs that do otherwise)
    }
    const uint32_t cnt = uint32_t(__popcll(m));
    const uint32_t incl = wave_inclusive_sum(cnt);
    const uint32_t total = read_lane(incl, kWave - 1);
    const uint32_t wv = gtid >> 6;
    if (kCount) {
        if (lane == 0) as_global_mut(a.wave_count)[wv] = total;
        return;
    }
    if (total == 0) return;
    const uint32_t base = uint32_t(__builtin_amdgcn_readfirstlane(int(as_global(a.wave_off)[wv])));
    // The wave's candidates are spread over its lanes first (LDS), so that the two dependent loads every candidate needs —
    // its entry's reference and its offset pair — are issued by 64 lanes at once instead of inside a divergent loop.
    for (uint32_t r0 = 0; r0 < total; r0 += kProbeRound) {
        uint32_t o = incl - cnt;
        uint64_t mm = m;
        while (mm) {
            const uint32_t bit = uint32_t(__ffsll((long long)mm)) - 1u;
            mm &= mm - 1;
            const uint32_t pos = o++;
            if (pos >= r0 && pos < r0 + kProbeRound) {
                lds_entry[wave][pos - r0] = e;
                lds_key[wave][pos - r0] = uint16_t(w * 64u + bit);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t n_round = min(kProbeRound, total - r0);
        for (uint32_t j = uint32_t(lane); j < n_round; j += kWave) {
            const uint32_t ce = lds_entry[wave][j], key = lds_key[wave][j];
            GlobalPtr<u32x4> rp = reinterpret_cast<GlobalPtr<u32x4>>(as_global(a.refs + ce));
            const u32x4 r0v = rp[0], r2v = rp[2];
            const uint64_t fsst = uint64_t(r0v.x) | (uint64_t(r0v.y) << 32);
            const uint8_t* residuals = reinterpret_cast<const uint8_t*>(uint64_t(r0v.z) | (uint64_t(r0v.w) << 32));
            const uint32_t slope = r2v.x, intercept = r2v.y, ob = r2v.z;
            const uint64_t v = load_unaligned<uint64_t>(residuals + size_t(key) * ob);
            const uint32_t sh = 32u - 8u * ob;
            const int32_t q0 = int32_t(uint32_t(v) << sh) >> sh;
            const int32_t q1 = int32_t(uint32_t(v >> (8u * ob)) << sh) >> sh;
            const uint32_t start = slope * key + intercept + uint32_t(q0);
            const uint32_t stop = slope * (key + 1u) + intercept + uint32_t(q1);
            const uint64_t abs_start = fsst + start;
            const u32x4 out = {uint32_t(abs_start), uint32_t(abs_start >> 32), stop - start, key};
            *reinterpret_cast<GlobalMutPtr<u32x4>>(as_global_mut(a.cand + base + r0 + j)) = out;
            as_global_mut(a.cand_entry)[base + r0 + j] = ce;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

struct WalkArgs {
    const LikeChunk* chunks;
    const uint32_t* wg_slot;
    const CandRec* cand;
    const uint32_t* cand_entry;
    const LikeEntryRef* refs;
    const uint8_t* automata;
    uint32_t automaton_stride;
    uint32_t nl;
    const uint64_t* selection;
    uint64_t* mask;
    uint32_t* counts;
    unsigned long long* stats;  // plan run only: {compressed bytes of the candidates, matching dictionary values}
    ScanLaunch total;  // d_total_acc / d_total_out only
};

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = uint32_t(__shfl(int(uint32_t(v)), src, kWave));
    const uint32_t hi = uint32_t(__shfl(int(uint32_t(v >> 32)), src, kWave));
    return uint64_t(lo) | (uint64_t(hi) << 32);
}

__global__ __launch_bounds__(kWalkWaves * 64) void k_like_walk(WalkArgs a) {
    // dynamic LDS: [automaton image (u16 row addresses, built for LDS address 0)][per wave: 64 hit flags + head mask]
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    const uint32_t nl = a.nl;
    const uint32_t tbl_bytes = automaton_image_bytes(nl);
    const uint32_t slot = a.wg_slot[blockIdx.x];
    {
        const uint8_t* src = a.automata + size_t(slot) * a.automaton_stride + automaton_u8_bytes(nl);
        for (uint32_t c = wave * 1024u; c < tbl_bytes; c += kWalkWaves * 1024u)
            async_copy16(src + c + uint32_t(lane) * 16u, smem + c);
    }
    const LikeChunk ch = a.chunks[blockIdx.x * kWalkWaves + wave];
    uint8_t* hitflag = smem + tbl_bytes + wave * 80u;
    uint64_t* headmask = reinterpret_cast<uint64_t*>(hitflag + 64);
    // the image holds absolute LDS addresses computed for a table at LDS address 0: this kernel has no static LDS, so its
    // dynamic segment starts there (a toolchain that placed it elsewhere would make every lookup wrong: stop loudly)
    const uint32_t row0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    if (row0 != 0u) __builtin_trap();
    const uint32_t hitrow = row0 + nl * 512u;

    // this lane's candidate: one 16-byte record says where its compressed bytes are
    const bool cl = uint32_t(lane) < ch.count;
    uint64_t abs_start = 0;
    uint32_t len = 0, key = 0, e = 0;
    if (cl) {
        const u32x4 c = *reinterpret_cast<GlobalPtr<u32x4>>(as_global(a.cand + ch.first + uint32_t(lane)));
        abs_start = uint64_t(c.x) | (uint64_t(c.y) << 32);
        len = c.z;
        key = c.w;
        e = as_global(a.cand_entry)[ch.first + uint32_t(lane)];
    }
    // ---- lane-parallel walk: one lane per 8-byte word of every candidate (see "the lane-parallel LIKE walker" in
    // lc_kernels.hip: automaton states are corrected across neighbouring lanes to a fixpoint, and a match counts only
    // there).  The plan cut the chunks so that this loop runs once unless a single value is longer than 512 bytes.
    const uint32_t words = cl ? max(1u, (len + 7u) >> 3) : 0u;
    const uint32_t incl = wave_inclusive_sum(words);
    const uint32_t off = incl - words;
    const uint32_t total = read_lane(incl, kWave - 1);
    hitflag[lane] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the image DMA has landed
    __syncthreads();
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
        const uint64_t o_abs = shfl_u64(abs_start, int(r));
        const uint32_t k = t0 + uint32_t(lane) - o_off;  // word index within the value
        const uint32_t p = 8u * k;
        const uint32_t rem = live && p < o_len ? o_len - p : 0u;
        uint64_t wd = 0;
        if (rem) wd = load_unaligned<uint64_t>(reinterpret_cast<const uint8_t*>(o_abs + p));
        const bool first = k == 0;
        auto walk_task = [&](uint32_t s) {
            uint32_t x[8];
            const uint32_t lo = uint32_t(wd), hi = uint32_t(wd >> 32);
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = (((q < 4 ? lo : hi) >> (8 * (q & 3))) & 0xFFu) << 1;
            return walk8(s, x, rem);
        };
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
        const bool hit = en == hitrow;
        carry_state = read_lane(en, kWave - 1);
        if (carry_state == hitrow) carry_state = row0;
        if (hit && live) hitflag[r] = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool res = cl && hitflag[lane] != 0;
    uint64_t matched = __ballot(res);
    uint64_t wave_hits = 0;
    if (a.stats) {
        const uint64_t lb = wave_sum_u64(uint64_t(len));
        if (lane == 0) {
            atomicAdd(a.stats, (unsigned long long)lb);
            atomicAdd(a.stats + 1, (unsigned long long)__popcll(matched));
        }
    }
    if (matched) {
        // ---- rows of the matching dictionary values, from the entries' inverted row lists (matches are rare: the
        // entry's reference is only fetched now, by the matching lanes)
        uint32_t o0 = 0, o1 = 0, dlen = 0;
        uint64_t post_bits = 0, mask_off = 0;
        if (res) {
            GlobalPtr<u32x4> rp = reinterpret_cast<GlobalPtr<u32x4>>(as_global(a.refs + e));
            const u32x4 r1 = rp[1], r2 = rp[2];
            post_bits = uint64_t(r1.x) | (uint64_t(r1.y) << 32);
            mask_off = uint64_t(r1.z) | (uint64_t(r1.w) << 32);
            dlen = r2.w;
            const uint32_t v = load_unaligned<uint32_t>(reinterpret_cast<const uint8_t*>(post_bits) + 2u * size_t(key));
            o0 = v & 0xFFFFu;
            o1 = v >> 16;
        }
        while (matched) {
            const int ml = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)matched)) - 1);
            matched &= matched - 1;
            const uint32_t b = read_lane(o0, ml), e1 = read_lane(o1, ml);
            const uint64_t pb = uniform_u64(shfl_u64(post_bits, ml));
            const uint64_t moff = uniform_u64(shfl_u64(mask_off, ml));
            const uint32_t dd = read_lane(dlen, ml);
            const uint16_t* prow = reinterpret_cast<const uint16_t*>(pb) + dd + 1u;
            uint32_t c = 0;
            for (uint32_t rr = b + uint32_t(lane); rr < e1; rr += kWave) {
                const uint32_t row = as_global(prow)[rr];
                const uint64_t bit = uint64_t(1) << (row & 63u);
                bool on = true;
                if (a.selection) on = (as_global(a.selection)[moff + (row >> 6)] & bit) != 0;
                if (on) {
                    atomicOr(reinterpret_cast<unsigned long long*>(a.mask + moff + (row >> 6)), (unsigned long long)bit);
                    c++;
                }
            }
            const uint64_t ct = uniform_u64(wave_sum_u64(uint64_t(c)));
            if (a.counts && lane == 0 && ct) atomicAdd(a.counts + read_lane(e, ml), uint32_t(ct));
            wave_hits += ct;
        }
    }
    if (a.total.d_total_out && lane == 0)
        total_contribute(a.total, blockIdx.x * kWalkWaves + wave, gridDim.x * kWalkWaves, wave_hits);
}

// ------------------------------------------------------------------------------------------------------------------
// k_like_lean: the same evaluation in ONE kernel, one wave per entry — k_str_pred's kSigOnly variant reduced to what a
// selective LIKE over indexed entries needs (~350 instructions per entry instead of ~1,300: k_str_pred is bound by
// instruction issue, 12,207 waves x 1,300 instructions x 4 cycles / 1,024 SIMDs = 26 us at 2.4 GHz).  No plan, no host
// round trip; correct for every needle (candidates beyond the LDS list are walked in further rounds), fastest for
// selective ones.  Chain of a wave: record (scalar) -> signature slices -> offset pairs -> compressed words -> walk ->
// list bounds -> rows -> mask words.
struct LeanArgs {
    const LeanRec* recs;
    const uint8_t* automata;
    uint32_t automaton_stride;
    uint32_t nl;
    uint16_t sig_bits[kMaxSigProbe];
    const uint64_t* selection;
    uint64_t* mask;
    uint32_t* counts;
    ScanLaunch total;  // d_total_acc / d_total_out only
};
using ConstLeanPtr = const __attribute__((address_space(4))) LeanEntry*;

template <int N>
__global__ __launch_bounds__(kLeanWaves * 64, 32 / kLeanWaves) void k_like_lean(LeanArgs a) {
    // dynamic LDS: [automaton image][per wave: 128 mask words | kLeanCap u16 candidate keys | 64 hit flags + head mask]
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t kPerWave = kPostMaxRows / 8u + kLeanCap * 2u + 80u;
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    const uint32_t nl = a.nl;
    const uint32_t tbl_bytes = automaton_image_bytes(nl);
    const LeanRec* rec = a.recs + blockIdx.x;
    const uint32_t begin = rec->begin, end = rec->end;
    {
        const uint8_t* src = a.automata + size_t(rec->slot) * a.automaton_stride + automaton_u8_bytes(nl);
        for (uint32_t c = wave * 1024u; c < tbl_bytes; c += kLeanWaves * 1024u) async_copy16(src + c + uint32_t(lane) * 16u, smem + c);
    }
    const uint32_t row0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    if (row0 != 0u) __builtin_trap();  // the image holds absolute LDS addresses computed for address 0
    const uint32_t hitrow = row0 + nl * 512u;
    uint8_t* wbase = smem + tbl_bytes + wave * kPerWave;
    uint64_t* pmask = reinterpret_cast<uint64_t*>(wbase);
    uint16_t* list = reinterpret_cast<uint16_t*>(wbase + kPostMaxRows / 8u);
    uint8_t* hitflag = wbase + kPostMaxRows / 8u + kLeanCap * 2u;
    uint64_t* headmask = reinterpret_cast<uint64_t*>(hitflag + 64);
    const uint32_t entry = begin + wave;
    uint64_t wave_hits = 0;
    if (entry >= end) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, blockIdx.x * kLeanWaves + wave, gridDim.x * kLeanWaves, 0);
        return;
    }
    ConstLeanPtr E = reinterpret_cast<ConstLeanPtr>(reinterpret_cast<uintptr_t>(&rec->e[wave]));
    const uint32_t nw = E->nw, n_rows = E->n;
    const uint32_t nwords = (n_rows + 63u) >> 6;
    // mask words of the entry start clear in LDS (16 bytes per lane = 1 KB)
    reinterpret_cast<uint4*>(pmask)[lane] = make_uint4(0, 0, 0, 0);
    bool synced = false;
    uint32_t n_list = 0;

    // walk candidates list[0 .. count): 64 per batch, one lane per 8-byte word; rows of the matches go into pmask
    auto walk_list = [&](uint32_t count) {
        for (uint32_t b0 = 0; b0 < count; b0 += kWave) {
            const uint32_t j = b0 + uint32_t(lane);
            const bool cl = j < count;
            const uint32_t key = cl ? uint32_t(list[j]) : 0u;
            uint32_t start = 0, len = 0;
            if (cl) {
                const uint32_t ob = E->offset_bytes;
                const uint64_t v = load_unaligned<uint64_t>(E->residuals + size_t(key) * ob);
                const uint32_t sh = 32u - 8u * ob;
                const int32_t q0 = int32_t(uint32_t(v) << sh) >> sh;
                const int32_t q1 = int32_t(uint32_t(v >> (8u * ob)) << sh) >> sh;
                start = uint32_t(E->slope) * key + uint32_t(E->intercept) + uint32_t(q0);
                len = uint32_t(E->slope) * (key + 1u) + uint32_t(E->intercept) + uint32_t(q1) - start;
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
                const uint32_t o_start = uint32_t(__shfl(int(start), int(r), kWave));
                const uint32_t o_len = uint32_t(__shfl(int(len), int(r), kWave));
                const uint32_t k = t0 + uint32_t(lane) - o_off;
                const uint32_t p = 8u * k;
                const uint32_t rem = live && p < o_len ? o_len - p : 0u;
                uint64_t wd = 0;
                if (rem) wd = load_unaligned<uint64_t>(E->fsst + o_start + p);
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
            if (matched) {
                // rows of the matching dictionary values from the entry's inverted row lists, into the LDS mask words
                uint32_t o0 = 0, o1 = 0;
                if (res) {
                    const uint32_t v = load_unaligned<uint32_t>(reinterpret_cast<const uint8_t*>(E->postings) + 2u * size_t(key));
                    o0 = v & 0xFFFFu;
                    o1 = v >> 16;
                }
                const uint16_t* prow = E->postings + E->d + 1u;
                while (matched) {
                    const int ml = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)matched)) - 1);
                    matched &= matched - 1;
                    const uint32_t b = read_lane(o0, ml), e1 = read_lane(o1, ml);
                    for (uint32_t rr = b + uint32_t(lane); rr < e1; rr += kWave) {
                        const uint32_t row = as_global(prow)[rr];
                        atomicOr(reinterpret_cast<unsigned long long*>(&pmask[row >> 6]), 1ull << (row & 63u));
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    // ---- probe: AND of the needle's signature slices, 64 words (4,096 dictionary values) per round
    for (uint32_t w0 = 0; w0 < nw; w0 += kWave) {
        const uint32_t w = w0 + uint32_t(lane);
        uint64_t m = 0;
        if (w < nw) {
            uint64_t sv[N];
#pragma unroll
            for (int k = 0; k < N; k++) sv[k] = as_global(E->sig)[size_t(a.sig_bits[k]) * nw + w];
            m = sv[0];
#pragma unroll
            for (int k = 1; k < N; k++) m &= sv[k];
        }
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
                if ((ms >> lane) & 1u) list[lanes_below(ms)] = uint16_t((w0 + uint32_t(sl)) * 64u + uint32_t(lane));
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
            list[o++] = uint16_t(w * 64u + bit);
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
    // ---- the entry's mask words: rows of the lists are valid rows, the selection is applied here
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint64_t moff = E->mask_word_off;
    uint32_t c = 0;
    for (uint32_t w = uint32_t(lane); w < nwords; w += kWave) {
        uint64_t hitw = pmask[w];
        if (a.selection && hitw) hitw &= as_global(a.selection)[moff + w];
        as_global_mut(a.mask)[moff + w] = hitw;
        c += uint32_t(__popcll(hitw));
    }
    if (a.counts || a.total.d_total_out) {
        const uint32_t ct = read_lane(wave_inclusive_sum(c), kWave - 1);
        if (lane == 0 && a.counts) as_global_mut(a.counts)[entry] = ct;
        wave_hits = ct;
    }
    if (a.total.d_total_out && lane == 0) total_contribute(a.total, blockIdx.x * kLeanWaves + wave, gridDim.x * kLeanWaves, wave_hits);
}

hipError_t launch_lean(int n_sig, const LeanArgs& a, uint32_t n_recs, hipStream_t stream) {
    if (n_recs == 0) return hipSuccess;
    typedef void (*Kern)(LeanArgs);
    static const Kern table[kMaxSigProbe] = {k_like_lean<1>, k_like_lean<2>, k_like_lean<3>, k_like_lean<4>,
                                             k_like_lean<5>, k_like_lean<6>, k_like_lean<7>, k_like_lean<8>};
    const size_t lds = automaton_image_bytes(a.nl) + kLeanWaves * (kPostMaxRows / 8u + kLeanCap * 2u + 80u);
    hipLaunchKernelGGL(table[n_sig - 1], dim3(n_recs), dim3(kLeanWaves * 64), lds, stream, a);
    return hipGetLastError();
}

template <bool kCount>
hipError_t launch_probe(int n_sig, const ProbeArgs& a, hipStream_t stream) {
    uint32_t grid = (a.n_flat + kProbeThreads - 1) / kProbeThreads;
    if (grid == 0 && kCount) return hipSuccess;
    if (grid == 0) grid = 1;  // fill mode still clears the mask and the counts
    typedef void (*Kern)(ProbeArgs);
    static const Kern table[kMaxSigProbe] = {k_like_probe<1, kCount>, k_like_probe<2, kCount>, k_like_probe<3, kCount>,
                                             k_like_probe<4, kCount>, k_like_probe<5, kCount>, k_like_probe<6, kCount>,
                                             k_like_probe<7, kCount>, k_like_probe<8, kCount>};
    hipLaunchKernelGGL(table[n_sig - 1], dim3(grid), dim3(kProbeThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_walk(const WalkArgs& a, uint32_t n_wgs, hipStream_t stream) {
    if (n_wgs == 0) return hipSuccess;
    const size_t lds = automaton_image_bytes(a.nl) + kWalkWaves * 80u;
    hipLaunchKernelGGL(k_like_walk, dim3(n_wgs), dim3(kWalkWaves * 64), lds, stream, a);
    return hipGetLastError();
}

void free_plan(lc_ctx* ctx, LikePlan& p) {
    pool_release(ctx, p.d_wave_off);
    pool_release(ctx, p.d_chunks);
    pool_release(ctx, p.d_wg_slot);
    pool_release(ctx, p.d_cand);
    pool_release(ctx, p.d_cand_entry);
    p.d_wave_off = nullptr;
    p.d_chunks = nullptr;
    p.d_wg_slot = nullptr;
    p.d_cand = nullptr;
    p.d_cand_entry = nullptr;
}

// flat index over the scan's dictionaries, built once per scan
lc_status build_index(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, hipStream_t stream) {
    lp->built = true;
    lp->eligible = false;
    if (!s->is_str || s->n == 0) return LC_OK;
    for (const Entry& e : s->meta) {
        if (e.sd.d == 0) continue;  // an all-null entry has no dictionary: no candidates, its mask words stay zero
        if (!e.sd.signatures || !e.sd.postings || e.sd.n > kPostMaxRows) return LC_OK;
    }
    std::vector<FlatRec> flat;
    std::vector<LikeEntryRef> refs(s->n);
    const FlatRec pad{0, 0xFFFFFFFFu, 0, 0};
    uint32_t prev_slot = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < s->n; i++) {
        const StrDesc& d = s->meta[i].sd;
        const uint32_t nw = (d.d + 63u) / 64u;
        if (nw && d.symtab_slot != prev_slot) {
            // a probe wave (64 flat words) never spans two symbol tables: the candidate list is then grouped by table
            while (flat.size() % 64) flat.push_back(pad);
            prev_slot = d.symtab_slot;
        }
        if (uint64_t(flat.size()) + nw + 64 > 0xFFFFFFF0ull) return LC_OK;
        refs[i] = LikeEntryRef{d.fsst, d.residuals, d.postings, d.mask_word_off, d.slope, d.intercept, d.offset_bytes, d.d};
        for (uint32_t w = 0; w < nw; w++)
            flat.push_back(FlatRec{uint64_t(reinterpret_cast<uintptr_t>(d.signatures + w)), i, uint16_t(nw), uint16_t(w)});
    }
    while (flat.size() % 64) flat.push_back(pad);
    lp->n_flat = uint32_t(flat.size());
    lp->n_k1_waves = lp->n_flat / 64;
    // k_like_lean: one record per workgroup — consecutive entries, at most four, never across a symbol-table change
    std::vector<LeanRec> lean;
    for (uint32_t b = 0, i = 1; i <= s->n; i++) {
        if (i == s->n || i - b == kLeanWaves || s->meta[i].sd.symtab_slot != s->meta[b].sd.symtab_slot) {
            LeanRec r;
            std::memset(&r, 0, sizeof(r));
            r.begin = b;
            r.end = i;
            r.slot = s->meta[b].sd.symtab_slot;
            for (uint32_t k = b; k < i; k++) {
                const StrDesc& d = s->meta[k].sd;
                r.e[k - b] = LeanEntry{d.signatures, d.residuals, d.fsst, d.postings, d.mask_word_off, d.slope, d.intercept,
                                       d.d, d.n, d.offset_bytes, (d.d + 63u) / 64u};
            }
            lean.push_back(r);
            b = i;
        }
    }
    lp->n_lean = uint32_t(lean.size());
    lp->d_lean = static_cast<LeanRec*>(pool_alloc(ctx, std::max<size_t>(lean.size(), 1) * sizeof(LeanRec)));
    if (!lp->d_lean) return fail(LC_ERR_OOM, "hipMalloc (LIKE pipeline index)");
    LC_HIP(hipMemcpyAsync(lp->d_lean, lean.data(), lean.size() * sizeof(LeanRec), hipMemcpyHostToDevice, stream));
    lp->d_flat = static_cast<FlatRec*>(pool_alloc(ctx, std::max<size_t>(flat.size(), 1) * sizeof(FlatRec)));
    lp->d_refs = static_cast<LikeEntryRef*>(pool_alloc(ctx, size_t(s->n) * sizeof(LikeEntryRef)));
    lp->d_total_acc = static_cast<unsigned long long*>(pool_alloc(ctx, size_t(kTotalWords) * 8));
    if (!lp->d_flat || !lp->d_refs || !lp->d_total_acc) return fail(LC_ERR_OOM, "hipMalloc (LIKE pipeline index)");
    LC_HIP(hipMemcpyAsync(lp->d_flat, flat.data(), flat.size() * sizeof(FlatRec), hipMemcpyHostToDevice, stream));
    LC_HIP(hipMemcpyAsync(lp->d_refs, refs.data(), refs.size() * sizeof(LikeEntryRef), hipMemcpyHostToDevice, stream));
    LC_HIP(hipMemsetAsync(lp->d_total_acc, 0, size_t(kTotalWords) * 8, stream));  // once: launches leave it zero
    LC_HIP(hipStreamSynchronize(stream));  // the host vectors are locals
    lp->eligible = true;
    return LC_OK;
}

void fill_probe_args(const lc_scan* s, const LikePipeline* lp, const StrPred& p, ProbeArgs* a) {
    *a = ProbeArgs{};
    a->flat = lp->d_flat;
    a->refs = lp->d_refs;
    a->n_flat = lp->n_flat;
    a->n_entries = s->n;
    for (int k = 0; k < kMaxSigProbe; k++) a->sig_bits[k] = p.sig_bits[k];
}

lc_status run_probe(lc_scan* s, LikePipeline* lp, const LikePlan& plan, const StrPred& p, const ScanLaunch& L, hipStream_t stream) {
    ProbeArgs pa;
    fill_probe_args(s, lp, p, &pa);
    pa.wave_off = plan.d_wave_off;
    pa.cand = plan.d_cand;
    pa.cand_entry = plan.d_cand_entry;
    pa.mask = L.d_hit;
    pa.mask_words = s->seg_offsets.back();
    pa.counts = L.d_counts;
    pa.total_zero = plan.n_wgs == 0 ? L.d_total_out : nullptr;
    LC_HIP(launch_probe<false>(int(p.n_sig_bits), pa, stream));
    return LC_OK;
}

lc_status run(lc_scan* s, LikePipeline* lp, const LikePlan& plan, const StrPred& p, const ScanLaunch& L, hipStream_t stream,
              unsigned long long* d_stats = nullptr) {
    const lc_status ps = run_probe(s, lp, plan, p, L, stream);
    if (ps != LC_OK) return ps;
    WalkArgs wa{};
    wa.chunks = plan.d_chunks;
    wa.wg_slot = plan.d_wg_slot;
    wa.cand = plan.d_cand;
    wa.cand_entry = plan.d_cand_entry;
    wa.refs = lp->d_refs;
    wa.automata = p.automata;
    wa.automaton_stride = p.automaton_stride;
    wa.nl = p.needle_len;
    wa.selection = L.d_selection;
    wa.mask = L.d_hit;
    wa.counts = L.d_counts;
    wa.stats = d_stats;
    wa.total.d_total_acc = lp->d_total_acc;
    wa.total.d_total_out = L.d_total_out;
    LC_HIP(launch_walk(wa, plan.n_wgs, stream));
    return LC_OK;
}

lc_status run_lean(LikePipeline* lp, const StrPred& p, const ScanLaunch& L, hipStream_t stream) {
    LeanArgs la{};
    la.recs = lp->d_lean;
    la.automata = p.automata;
    la.automaton_stride = p.automaton_stride;
    la.nl = p.needle_len;
    for (int k = 0; k < kMaxSigProbe; k++) la.sig_bits[k] = p.sig_bits[k];
    la.selection = L.d_selection;
    la.mask = L.d_hit;
    la.counts = L.d_counts;
    la.total.d_total_acc = lp->d_total_acc;
    la.total.d_total_out = L.d_total_out;
    LC_HIP(launch_lean(int(p.n_sig_bits), la, lp->n_lean, stream));
    return LC_OK;
}

lc_status make_plan(lc_ctx* ctx, lc_scan* s, LikePipeline* lp, const StrPredHost& sp, hipStream_t stream, LikePlan* plan) {
    plan->needle = sp.needle;
    plan->use_pipeline = false;
    const uint64_t words = std::max<uint64_t>(s->seg_offsets.back(), 1);
    // scratch: per-wave counts | trial mask | {COUNT(*), candidate bytes, matches}
    uint32_t* d_wave_count = static_cast<uint32_t*>(pool_alloc(ctx, std::max<size_t>(lp->n_k1_waves, 1) * 4));
    uint64_t* d_scratch = static_cast<uint64_t*>(pool_alloc(ctx, words * 8 + 24));
    struct Tmp {
        lc_ctx* c; void* p; void* q; hipStream_t st;
        ~Tmp() { (void)hipStreamSynchronize(st); pool_release(c, p); pool_release(c, q); }
    } tmp{ctx, d_wave_count, d_scratch, stream};
    if (!d_wave_count || !d_scratch) return fail(LC_ERR_OOM, "hipMalloc (LIKE plan)");
    // 1. count: candidates of every probe wave -> write positions
    ProbeArgs pa;
    fill_probe_args(s, lp, sp.p, &pa);
    pa.wave_count = d_wave_count;
    LC_HIP(launch_probe<true>(int(sp.p.n_sig_bits), pa, stream));
    std::vector<uint32_t> wc(lp->n_k1_waves, 0);
    LC_HIP(hipMemcpyAsync(wc.data(), d_wave_count, size_t(lp->n_k1_waves) * 4, hipMemcpyDeviceToHost, stream));
    LC_HIP(hipStreamSynchronize(stream));
    std::vector<uint32_t> woff(lp->n_k1_waves, 0);
    uint64_t total = 0;
    for (uint32_t wv = 0; wv < lp->n_k1_waves; wv++) {
        woff[wv] = uint32_t(total);
        total += wc[wv];
        if (total > uint64_t(kMaxCandPerEntry) * s->n + 4096) {
            plan->n_cand = uint32_t(total);
            return LC_OK;  // not selective: k_str_pred keeps it
        }
    }
    plan->n_cand = uint32_t(total);
    plan->d_wave_off = static_cast<uint32_t*>(pool_alloc(ctx, std::max<size_t>(woff.size(), 1) * 4));
    plan->d_cand = static_cast<CandRec*>(pool_alloc(ctx, std::max<uint64_t>(total, 1) * sizeof(CandRec)));
    plan->d_cand_entry = static_cast<uint32_t*>(pool_alloc(ctx, std::max<uint64_t>(total, 1) * 4));
    auto bail = [&](lc_status st) { free_plan(ctx, *plan); return st; };
    if (!plan->d_wave_off || !plan->d_cand || !plan->d_cand_entry) return bail(fail(LC_ERR_OOM, "hipMalloc (LIKE plan)"));
    if (hipMemcpyAsync(plan->d_wave_off, woff.data(), woff.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        return bail(fail(LC_ERR_DEVICE, "hipMemcpy (LIKE plan)"));
    // 2. emit the candidate records once and cut them into chunks: consecutive candidates of ONE symbol table with at most
    //    64 eight-byte words in all (one pass of the walk), four chunks per workgroup
    ScanLaunch L{};
    L.d_hit = d_scratch;
    L.d_total_out = d_scratch + words;
    plan->n_wgs = 1;  // (keeps the probe from clearing the total word: the trial below owns it)
    lc_status rc = run_probe(s, lp, *plan, sp.p, L, stream);
    if (rc != LC_OK) return bail(rc);
    std::vector<CandRec> recs(total);
    std::vector<uint32_t> ents(total);
    if ((total && (hipMemcpyAsync(recs.data(), plan->d_cand, total * sizeof(CandRec), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                   hipMemcpyAsync(ents.data(), plan->d_cand_entry, total * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)) ||
        hipStreamSynchronize(stream) != hipSuccess)
        return bail(fail(LC_ERR_DEVICE, "hipMemcpy (LIKE plan)"));
    std::vector<LikeChunk> chunks;
    std::vector<uint32_t> wg_slot;
    uint32_t cur_first = 0, cur_count = 0, cur_words = 0, cur_slot = 0xFFFFFFFFu;
    auto close_chunk = [&]() {
        if (cur_count == 0) return;
        if (chunks.size() % kWalkWaves == 0) wg_slot.push_back(cur_slot);
        chunks.push_back(LikeChunk{cur_first, cur_count});
        cur_count = cur_words = 0;
    };
    auto close_table = [&]() {
        close_chunk();
        while (chunks.size() % kWalkWaves) chunks.push_back(LikeChunk{0, 0});
    };
    for (uint64_t i = 0; i < total; i++) {
        if (ents[i] >= s->n) return bail(fail(LC_ERR_DEVICE, "LIKE plan: candidate list is corrupt"));
        const uint32_t slot = s->meta[ents[i]].sd.symtab_slot;
        const uint32_t wds = std::max<uint32_t>(1u, (recs[i].len + 7u) / 8u);
        if (slot != cur_slot) {
            close_table();
            cur_slot = slot;
        }
        if (cur_count && (cur_words + wds > 64u || cur_count == 64u)) close_chunk();
        if (cur_count == 0) cur_first = uint32_t(i);
        cur_count++;
        cur_words += wds;
    }
    close_table();
    plan->n_chunks = uint32_t(chunks.size());
    plan->n_wgs = uint32_t(chunks.size() / kWalkWaves);
    plan->d_chunks = static_cast<LikeChunk*>(pool_alloc(ctx, std::max<size_t>(chunks.size(), 1) * sizeof(LikeChunk)));
    plan->d_wg_slot = static_cast<uint32_t*>(pool_alloc(ctx, std::max<size_t>(wg_slot.size(), 1) * 4));
    if (!plan->d_chunks || !plan->d_wg_slot) return bail(fail(LC_ERR_OOM, "hipMalloc (LIKE plan)"));
    if (hipMemcpyAsync(plan->d_chunks, chunks.data(), chunks.size() * sizeof(LikeChunk), hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(plan->d_wg_slot, wg_slot.data(), wg_slot.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
        return bail(fail(LC_ERR_DEVICE, "hipMemcpy (LIKE plan)"));
    // 3. trial run into scratch: how many rows does the needle hit?
    rc = hipMemsetAsync(d_scratch + words, 0, 24, stream) == hipSuccess ? LC_OK : fail(LC_ERR_DEVICE, "memset (LIKE plan)");
    if (rc == LC_OK) rc = run(s, lp, *plan, sp.p, L, stream, reinterpret_cast<unsigned long long*>(d_scratch + words + 1));
    uint64_t res3[3] = {0, 0, 0};
    if (rc == LC_OK && hipMemcpyAsync(res3, d_scratch + words, 24, hipMemcpyDeviceToHost, stream) != hipSuccess)
        rc = fail(LC_ERR_DEVICE, "hipMemcpy (LIKE plan)");
    if (rc == LC_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(LC_ERR_DEVICE, "stream (LIKE plan)");
    if (rc != LC_OK) return bail(rc);
    plan->hits = res3[0];
    plan->cand_bytes = res3[1];
    plan->matches = res3[2];
    plan->use_pipeline = plan->hits * 1024u <= uint64_t(kMaxHitsPer1024) * std::max<uint64_t>(s->total_rows, 1024);
    if (!plan->use_pipeline) free_plan(ctx, *plan);
    return LC_OK;
}

}  // namespace

void like_pipeline_destroy(lc_ctx* ctx, LikePipeline* lp) {
    if (!lp) return;
    for (LikePlan& p : lp->plans) free_plan(ctx, p);
    pool_release(ctx, lp->d_flat);
    pool_release(ctx, lp->d_refs);
    pool_release(ctx, lp->d_lean);
    pool_release(ctx, lp->d_total_acc);
    delete lp;
}

// One line on how `LIKE '%needle%'` was / would be evaluated on this scan (lc_scan_explain).  Caller holds s->mu.
std::string like_pipeline_explain(const lc_scan* s, const StrPredHost& sp) {
    const LikePipeline* lp = s->like;
    const int path = s->ctx->like_path;
    if (path == 1 || s->n < s->ctx->like_pipeline_min_entries) return "k_str_pred";
    if (!lp || !lp->built) return "k_str_pred (no scan-level index on this scan yet)";
    if (!lp->eligible) return "k_str_pred (entries without signature index / row lists)";
    if (path == 3) return "k_like_lean (forced for every needle)";
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle) {
            char buf[256];
            if (q.use_pipeline && path == 2)
                std::snprintf(buf, sizeof(buf), "k_like_probe + k_like_walk: %u candidates in %u chunks, %llu hit rows, %u flat words",
                              q.n_cand, q.n_chunks, (unsigned long long)q.hits, lp->n_flat);
            else if (q.use_pipeline)
                std::snprintf(buf, sizeof(buf), "k_like_lean: %u candidates at plan time (%.1f per entry), %llu hit rows", q.n_cand,
                              double(q.n_cand) / double(std::max<uint32_t>(s->n, 1)), (unsigned long long)q.hits);
            else
                std::snprintf(buf, sizeof(buf), "k_str_pred (needle not selective: %u+ candidates, %llu hit rows at plan time)", q.n_cand,
                              (unsigned long long)q.hits);
            return buf;
        }
    return "k_str_pred (needle not planned)";
}

// Bytes the two kernels themselves have to move for one evaluation (the numerator of an honest HBM-roofline fraction, like
// lc_scan_traffic_model's figure for k_str_pred): probe = entry map + the needle's slices + mask / count clears + the
// candidate list; walk = chunk records + candidates + one 48-byte reference per entry that has candidates + offset pairs +
// compressed bytes of the candidates + list bounds and rows of the matches + one 8-byte read-modify-write per hit row.
// 0 when the pipeline does not take this needle.  Caller holds s->mu.
uint64_t like_pipeline_bytes(const lc_scan* s, const StrPredHost& sp, bool with_counts) {
    const LikePipeline* lp = s->like;
    if (!lp || !lp->eligible) return 0;
    for (const LikePlan& q : lp->plans)
        if (q.needle == sp.needle && q.use_pipeline) {
            // probe: 16 B per flat word + the needle's slices + mask / count clears + write positions + per candidate the
            // offset pair (8) and the 16 + 4 bytes of its record; one 48-byte reference per entry that has candidates
            uint64_t flat_valid = 0;
            for (const Entry& e : s->meta) flat_valid += (e.sd.d + 63u) / 64u;
            uint64_t b = uint64_t(lp->n_flat) * 16 + flat_valid * 8 * sp.p.n_sig_bits + s->seg_offsets.back() * 8 +
                         (with_counts ? uint64_t(s->n) * 4 : 0) + uint64_t(lp->n_k1_waves) * 4 + uint64_t(q.n_cand) * (8 + 20) +
                         std::min<uint64_t>(q.n_cand, s->n) * 48;
            // walk: chunk records + workgroup slots + candidate records + compressed bytes + per match its entry's
            // reference (48), list bounds (4) and rows (2 each) + one 8-byte read-modify-write per hit row
            b += uint64_t(q.n_chunks) * 8 + uint64_t(q.n_wgs) * 4 + uint64_t(q.n_cand) * 20 + q.cand_bytes + q.matches * (48 + 4) +
                 q.hits * (2 + 16);
            return b;
        }
    return 0;
}

// Caller holds s->mu and has built the automata of `sp` (sp.p.automata).  *handled = true: the evaluation was launched.
lc_status like_pipeline_eval(lc_ctx* ctx, lc_scan* s, const StrPredHost& sp, const ScanLaunch& L, hipStream_t stream,
                             bool* handled) {
    *handled = false;
    const StrPred& p = sp.p;
    if (p.mode != 1 || p.op != LC_OP_LIKE || !p.use_fingerprints || p.n_sig_bits == 0 || p.needle_len < 2 ||
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
    if (ctx->like_path == 3) {  // forced: the lean kernel for every needle (it is correct for all of them)
        const lc_status st = run_lean(lp, p, L, stream);
        if (st == LC_OK) *handled = true;
        return st;
    }
    LikePlan* plan = nullptr;
    for (LikePlan& q : lp->plans)
        if (q.needle == sp.needle) plan = &q;
    if (!plan) {
        if (lp->plans.size() >= kMaxPlans) {
            // evict the least recently used plan (launches that read its buffers are on this scan's one stream)
            size_t victim = 0;
            for (size_t i = 1; i < lp->plans.size(); i++)
                if (lp->plans[i].last_use < lp->plans[victim].last_use) victim = i;
            (void)hipStreamSynchronize(stream);
            free_plan(ctx, lp->plans[victim]);
            lp->plans.erase(lp->plans.begin() + long(victim));
        }
        LikePlan fresh;
        const lc_status st = make_plan(ctx, s, lp, sp, stream, &fresh);
        if (st != LC_OK) return st;
        lp->plans.push_back(std::move(fresh));
        plan = &lp->plans.back();
    }
    plan->last_use = ++lp->tick;
    if (!plan->use_pipeline) return LC_OK;
    const lc_status st = ctx->like_path == 2 ? run(s, lp, *plan, p, L, stream) : run_lean(lp, p, L, stream);
    if (st == LC_OK) *handled = true;
    return st;
}

}  // namespace lc
