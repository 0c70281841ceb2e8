// k_like_scanall — `[NOT] LIKE '%needle%'` over byte-view entries when MANY dictionary values have to be walked: scans
// without the bigram signature index (the reference's own regime: fingerprint prefilter, then decode + memmem of the
// candidates — byte_view_array/comparisons.rs:159-183, 598-651), columns staged without fingerprints, 1-byte needles and
// needles the plan found unselective.
//
// What it replaces there: the lane-per-VALUE walkers of k_str_pred (like_walk_many / like_walk_stream).  Round 3 measured
// what bounds them: every lane fetches ITS value's bytes with lane-private loads (64 different cache lines per load
// instruction: the address unit, and 1.6x the bytes through HBM because sectors are fetched more than once) and runs a
// cursor per chain (~200 of ~330 instructions per 16-byte block).  This kernel walks the entry's FSST buffer the way it lies
// in memory instead:
//   * lane per 8-byte WORD of the buffer: a pass of the wave reads 512 consecutive bytes with one coalesced load per lane —
//     every byte of the buffer moves once, in order, no offsets are needed to find it;
//   * value boundaries come as a BITMAP over the chunk's bytes ("a value ends behind this byte"), scattered into LDS from
//     the compact offsets by the lanes while the words are in flight; a word's eight flags are one LDS byte;
//   * every lane walks its word through the LDS image of the needle's automaton folded over the FSST symbols
//     (k_str_automata: one lookup per compressed byte, escapes are part of the state) from the START state, resets at the
//     value ends inside the word and records "matched" per end; four passes are interleaved per lane (four independent
//     lookup chains in flight);
//   * the start state of a word is really the end state of the word before it.  That is the start state again unless a
//     match (or an escape) is in progress across the boundary — a few percent of the words.  Those words go to a worklist
//     and are re-walked in dense batches up to their first value end (a reset isolates everything behind it); a changed
//     end state of a word WITHOUT a value end re-queues its successor.  Exact at the fixpoint, like the neighbour-state
//     correction of the lane-parallel walker, but the correction costs one dense pass per ~500 words instead of one per 64;
//   * a match at the r-th value end of the chunk belongs to dictionary value (first value of the chunk + r), corrected for
//     the one empty value a dictionary can hold (a dictionary's values are distinct; entries staged with several empty
//     values are flagged at staging and keep k_str_pred).
// Every dictionary value is walked — the fingerprints are not read (they prune 60 % of the values of a URL column, but
// skipping a value in a coalesced stream saves nothing), except for the reference's NOT LIKE candidate rule
// (comparisons.rs:167-180, 644-648).  Rows are mapped through the keys as in k_str_pred's phase C.
#include "lc_device.hpp"
#include "lc_internal.hpp"

namespace lc {
namespace {

constexpr uint32_t kSaChunkWords = 512;                  // 8-byte words of a chunk (8 passes of the wave)
constexpr uint32_t kSaChunkBytes = kSaChunkWords * 8u;   // 4 KB
constexpr uint32_t kSaPasses = kSaChunkWords / kWave;    // 8
#ifndef LC_SA_ILP
#define LC_SA_ILP 4
#endif
// 1: words with and without a value end are walked separately (round 5, see the kernel: every word without the per-byte end
// logic — 2 VALU per byte —, the 18 % that hold a value end again in dense passes); 0 (shipped): every word with the full
// per-byte logic.  Measured on the 100 M-row URL column through LC_OPT_LIKE_PATH = 5, all ten needle classes bit-identical
// to the oracle in both forms: 592-609 / 585 us hot / cold with the split against 584-610 / 565-582 without.  No gain — and no
// surprise once the instructions are counted: the list of the words with an end, their second fetch and walk and a carry pass
// that reads LDS instead of registers give back most of what the cheap walk saves (~660 against ~750 vector instructions per
// 4 KB chunk).  Kept as an A/B option (profiles/r5/ablation_scanall_split.txt).
// 2 (round 6, built, measured, NOT shipped): INERT WORDS ARE NOT WALKED.  The start state's row of the automaton image says for every code whether
// it moves the automaton out of the start state; a word whose eight bytes all leave it there ends in the start state without
// a hit whatever its value ends are — if it is ENTERED in the start state.  So every word is only CLASSIFIED (one look-up
// per byte into ONE 512-byte row: two dwords per bank, next to conflict free, no dependency between the bytes, ~3 VALU per
// byte against ~9 for the walk), the words with an interesting byte (25 % for '%google%' on URLs) are listed and walked in
// dense passes with the full per-byte logic, and the words that are entered in another state than the start state come out of
// the carry pass below and go through the correction worklist exactly as before (an inert word that is entered mid-match is
// one of them).  The carry pass works on the passes' ballots in scalar registers; the attribution of matches to dictionary
// values is skipped for a chunk without a hit.  Exact (the whole path-5 suite: 24 tests, every needle class).  Measured on the
// 100 M-row URL column, hot / cold (profiles/r6/ab_scanall_inert.txt): '%google%' 587 / 578 us against 574 / 565 for mode 0,
// '%mail%' 952 / 965 against 831 / 837, '%ru/%' 1,155 against 985, a needle whose first byte hardly occurs ('%zzzzqqq%': next
// to every word inert) 544 / 534 against 562 / 549.  The counters say why: VALU wave-instructions per launch fall from
// 2.90 x 10^8 to 2.25 x 10^8 (840 instead of 1,083 per 4 KB chunk) — but SALU ones double (6.4 -> 11.5 x 10^7: the ballots
// and the 128-bit carry arithmetic), LDS instructions grow (4.05 -> 4.72 x 10^7) and with them LDS cycles (1.37 -> 1.59 x 10^8,
// 40 % of them bank conflicts).  Even with NOTHING to walk the chunk costs ~800 instructions: classification 256, lists ~100,
// the carry pass ~160, value ends ~50, addressing ~40, the two dense passes ~175.  The walk was never more than a third of this
// kernel; what bounds it is the per-word bookkeeping of a lane-per-word formulation, whatever the words cost.
#ifndef LC_SA_SPLIT
#define LC_SA_SPLIT 0
#endif
// variant builds only (results WRONG): 1 = no table lookups, 2 = no corrections / attribution, 4 = no placement
#ifndef LC_SA_ABL
#define LC_SA_ABL 0
#endif
[[maybe_unused]] constexpr uint32_t kSaIlp = LC_SA_ILP;  // passes walked in lock step by a lane (independent lookup chains)
// per wave: [dictionary result bitmap (dres_bytes)][ends 512][hit8 512][out16 1024][in16 1024][list 2 x 1024][phase-C stage 512]
// (the words of a chunk are NOT kept in LDS: 4 KB per wave cost two of five workgroups per CU — 12 instead of 20 waves — and
// the few words the worklist re-walks come back from L2)
constexpr uint32_t kSaFixedLds = 512u + 512u + 1024u + 1024u + 2048u + 512u;

struct ScanAllArgs {
    const StrWgRecord* recs;
    uint32_t n_recs;
    const uint8_t* automata;
    uint32_t automaton_stride;
    uint32_t nl;
    uint32_t use_fp;     // NOT LIKE: the reference's candidate rule applies to entries that carry fingerprints
    uint32_t needle_fp;
    uint32_t dres_bytes; // result bitmap bytes per wave (largest dictionary of the scan, multiple of 16)
    const uint64_t* selection;
    uint64_t* mask;
    uint64_t* valid;
    uint32_t* counts;
    ScanLaunch total;
    // kUni (1-byte needles): the needle byte's slice of the scan-level unigram index — bit i of entry e's words (at
    // uni_word[e]) says whether dictionary value i holds the byte — is the dictionary result; nothing is walked
    const uint64_t* uni_slice;
    const uint32_t* uni_word;
};
using ConstRecDesc = const __attribute__((address_space(4))) StrDesc*;

template <bool kNot, bool kUni>
__global__ __launch_bounds__(kThreads) void k_like_scanall(ScanAllArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = lane_id();
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(wave_id()));
    const uint32_t nl = a.nl;
    const uint32_t tbl_bytes = kUni ? 0u : automaton_image_bytes(nl);
    const StrWgRecord* rec = a.recs + blockIdx.x;
    const uint32_t begin = rec->begin, end = rec->end;
    if (!kUni) {   // the workgroup's automaton image (its entries share one symbol table): every wave brings its share
        const uint8_t* src = a.automata + size_t(rec->symtab_slot) * a.automaton_stride + automaton_u8_bytes(nl);
        for (uint32_t c = wave * 1024u; c < tbl_bytes; c += kWavesPerBlock * 1024u) async_copy16(src + c + uint32_t(lane) * 16u, smem + c);
    }
    const uint32_t row0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    if (!kUni && row0 != 0u) __builtin_trap();  // the image holds absolute LDS addresses computed for address 0
    const uint32_t hitrow = row0 + nl * 512u;
    const uint32_t per_wave = a.dres_bytes + kSaFixedLds;
    uint8_t* wbase = smem + tbl_bytes + wave * per_wave;
    uint32_t* dres = reinterpret_cast<uint32_t*>(wbase);
    uint8_t* ends = wbase + a.dres_bytes;
    uint8_t* hit8 = ends + 512;
    uint16_t* out16 = reinterpret_cast<uint16_t*>(hit8 + 512);
    uint16_t* in16 = out16 + 512;
    uint16_t* list_a = in16 + 512;
    uint16_t* list_b = list_a + 512;
    uint8_t* stage = reinterpret_cast<uint8_t*>(list_b + 512);
    const uint32_t unit = blockIdx.x * kWavesPerBlock + wave, n_units = gridDim.x * kWavesPerBlock;
    const uint32_t entry = begin + wave;
    // every wave of the workgroup passes the barrier that publishes the image exactly once
    if (!kUni) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (entry >= end) {
        if (a.total.d_total_out && lane == 0) total_contribute(a.total, unit, n_units, 0);
        return;
    }
    ConstRecDesc dp = reinterpret_cast<ConstRecDesc>(reinterpret_cast<uintptr_t>(&rec->d[wave]));
    const uint32_t D = dp->d, n_rows = dp->n, fsst_len = dp->fsst_len;
    const uint32_t nwords = (n_rows + 63u) >> 6;
    for (uint32_t i = uint32_t(lane) * 4u; i < a.dres_bytes / 4u; i += kWave * 4u)
        *reinterpret_cast<uint4*>(dres + i) = make_uint4(0, 0, 0, 0);
    // NOT LIKE: the reference inverts the dictionary results only when some value passes the 32-bucket fingerprint filter
    uint32_t fp_cand = 0;
    if (kNot && a.use_fp && dp->fingerprints) {
        for (uint32_t i0 = 0; i0 < D && fp_cand == 0; i0 += kWave) {
            const uint32_t i = i0 + uint32_t(lane);
            const uint32_t fp = i < D ? as_global(dp->fingerprints)[i] : 0u;
            fp_cand = __ballot(i < D && (fp & a.needle_fp) == a.needle_fp) != 0 ? 1u : 0u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint64_t any_true = 0;  // wave uniform

    if (kUni) {
        // the slice words ARE the dictionary results (bits beyond D are zero in the index)
        const uint64_t* src = a.uni_slice + a.uni_word[entry];
        const uint32_t nw = (D + 63u) >> 6;
        uint64_t acc = 0;
        for (uint32_t w = uint32_t(lane); w < nw; w += kWave) {
            const uint64_t v = *as_global(src + w);
            reinterpret_cast<uint64_t*>(dres)[w] = v;
            acc |= v;
        }
        any_true = __ballot(acc != 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (D > 0 && fsst_len > 0) {
        const uint8_t* fsst = dp->fsst;
        const uint32_t last_word = (fsst_len - 1u) & ~7u;  // (the section carries 16 bytes of padding)
        uint32_t v_next = 0;        // first dictionary value whose end has not been placed yet (wave uniform)
        uint32_t e0 = 0xFFFFFFFFu;  // index of the dictionary's empty value, if it has one
        uint32_t carry = row0;      // true end state of the last word of the chunk before
        // Nothing the next chunk needs from memory is requested when it is needed: its words and the offset pairs of the next
        // 128 dictionary values are loaded while the current chunk is walked (a wave works through ~22 chunks per entry, and
        // with 12 waves per CU three dependent round trips per chunk were the whole kernel: 640 us per 100 M rows).
        auto load_words = [&](uint32_t cc, uint64_t (&dst)[kSaPasses]) {
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) {
                const uint32_t off = min(cc + 8u * (q * kWave + uint32_t(lane)), last_word);
                dst[q] = *reinterpret_cast<GlobalPtr<uint64_t>>(reinterpret_cast<uintptr_t>(fsst + off));
            }
        };
        auto load_pair = [&](uint32_t i, uint32_t& start, uint32_t& stop) {
            start = 0;
            stop = 0xFFFFFFFFu;
            if (i < D) str_offset_pair(*dp, i, start, stop);
        };
        uint64_t wn[kSaPasses];
        load_words(0, wn);
        uint32_t pf_start[2], pf_stop[2];
        load_pair(uint32_t(lane), pf_start[0], pf_stop[0]);
        load_pair(uint32_t(kWave) + uint32_t(lane), pf_start[1], pf_stop[1]);
        for (uint32_t c0 = 0; c0 < fsst_len; c0 += kSaChunkBytes) {
            uint64_t w[kSaPasses];
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) w[q] = wn[q];
            if (c0 + kSaChunkBytes < fsst_len) load_words(c0 + kSaChunkBytes, wn);
            // ---- value ends of the chunk -> bitmap over its bytes
            reinterpret_cast<uint64_t*>(ends)[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t va = v_next;
            const uint32_t c1 = c0 + kSaChunkBytes;
            for (uint32_t round = 0;; round++) {
                if (LC_SA_ABL & 4) { v_next = D; break; }
                const uint32_t i = v_next + uint32_t(lane);
                uint32_t start, stop;
                if (round < 2) { start = pf_start[round]; stop = pf_stop[round]; }  // (prefetched for va + 64 round + lane)
                else load_pair(i, start, stop);
                const bool placed = i < D && stop <= c1;
                if (placed && stop > start && stop > c0) {  // (stop > c0 unless the offsets decrease: corrupt bytes)
                    const uint32_t pos = stop - 1u - c0;
                    atomicOr(reinterpret_cast<uint32_t*>(ends) + (pos >> 5), 1u << (pos & 31u));
                }
                const uint64_t em = __ballot(placed && stop == start);
                if (em) e0 = v_next + uint32_t(__ffsll((long long)em)) - 1u;
                const uint32_t np = uint32_t(__popcll(__ballot(placed)));
                v_next += np;
                if (np < uint32_t(kWave)) break;
            }
            load_pair(v_next + uint32_t(lane), pf_start[0], pf_stop[0]);
            load_pair(v_next + uint32_t(kWave) + uint32_t(lane), pf_start[1], pf_stop[1]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#if LC_SA_SPLIT == 2
            // ---- phase 1: classify.  interesting = some byte's transition out of the start state is not the start state
            uint32_t n_b = 0;
            uint16_t* list_bw = list_b;  // (free until the corrections; they start after the dense passes)
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) {
                const uint64_t ww = w[q];
                uint32_t acc = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
                    const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                    acc |= lds_u16(row0 + 2u * code) ^ row0;
                }
                const uint32_t wi = q * kWave + uint32_t(lane);
                const bool has = acc != 0;
                const uint64_t bm = __ballot(has);
                if (has) list_bw[n_b + lanes_below(bm)] = uint16_t(wi);
                n_b += uint32_t(__popcll(bm));
                out16[wi] = uint16_t(row0);  // (an inert word entered in the start state: start state out, no hit)
                hit8[wi] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // ---- phase 2: the interesting words, densely, two passes in lock step, from the start state with the full per-byte
            // logic (value ends reset, matches recorded per end); their bytes come back by index (L1 / L2: just streamed)
            uint32_t chunk_hit = 0;  // (lane-local OR of the hit flags written: is there anything to attribute?)
            for (uint32_t b0 = 0; b0 < n_b; b0 += 2u * kWave) {
                uint32_t st[2], hit[2], e8[2], wi2[2];
                uint64_t ww2[2];
                bool act[2];
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    const uint32_t j = b0 + u * kWave + uint32_t(lane);
                    act[u] = j < n_b;
                    wi2[u] = act[u] ? uint32_t(list_bw[j]) : 0u;
                    ww2[u] = *reinterpret_cast<GlobalPtr<uint64_t>>(reinterpret_cast<uintptr_t>(fsst + min(c0 + 8u * wi2[u], last_word)));
                    e8[u] = act[u] ? uint32_t(ends[wi2[u]]) : 0u;
                    st[u] = row0;
                    hit[u] = 0;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
#pragma unroll
                    for (uint32_t u = 0; u < 2; u++) {
                        const uint64_t ww = ww2[u];
                        const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                        const uint32_t t = lds_u16(st[u] + 2u * code);
                        const bool is_end = ((e8[u] >> k) & 1u) != 0;
                        hit[u] |= (is_end && t == hitrow) ? (1u << k) : 0u;
                        st[u] = is_end ? row0 : t;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    if (act[u]) {
                        out16[wi2[u]] = uint16_t(st[u]);
                        hit8[wi2[u]] = uint8_t(hit[u]);
                        chunk_hit |= hit[u];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // ---- phase 3: the carries of the matched state and the worklist, pass by pass in buffer order, on the passes' ballots.
            // M: the word ends matched; P: it holds no value end (the matched state passes through it); O: it ends in a state
            // that is neither the start nor the matched state (its successor has to be re-walked).  The carry of the matched
            // state INTO word l is the carry chain of (M | P) + M; only lanes it reaches touch their word.
            uint32_t n_list = 0;
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) {
                const uint32_t wi = q * kWave + uint32_t(lane);
                const uint32_t stq = uint32_t(out16[wi]);
                const uint32_t e8q = uint32_t(ends[wi]);
                const bool has_end = e8q != 0;
                const uint64_t M = __ballot(stq == hitrow), P = __ballot(!has_end);
                const uint64_t A = M | P;
                const unsigned __int128 sum = (unsigned __int128)A + M + (carry == hitrow ? 1u : 0u);
                const uint64_t im = uint64_t(sum) ^ A ^ M;  // bit l: carry INTO word l = "starts matched"
                uint32_t st_new = stq;
                if (im != 0) {  // (wave uniform: selective needles rarely get here)
                    if ((im >> lane) & 1u) {
                        if (has_end) {
                            const uint32_t hq = uint32_t(hit8[wi]) | (1u << (uint32_t(__ffs(int(e8q))) - 1u));
                            hit8[wi] = uint8_t(hq);
                            chunk_hit |= hq;
                        } else {
                            st_new = hitrow;
                            out16[wi] = uint16_t(hitrow);
                        }
                    }
                }
                // the end state of the word before: the start / matched state needs nothing (the carry chain covers "matched");
                // anything else sends this word to the worklist with that state as its start state
                const uint64_t O = __ballot(st_new != row0 && st_new != hitrow);  // (after the carry: a word the matched state
                                                                                  // passed through ends matched, not odd)
                const bool carry_odd = carry != row0 && carry != hitrow;
                const uint64_t need_m = (O << 1) | (carry_odd ? 1ull : 0ull);
                if (need_m != 0) {
                    const uint32_t prev = lane_shift_up1(st_new, carry);
                    const bool need = ((need_m >> lane) & 1u) != 0;
                    if (need) {
                        in16[wi] = uint16_t(prev);
                        list_a[n_list + lanes_below(need_m)] = uint16_t(wi);
                    }
                    n_list += uint32_t(__popcll(need_m));
                }
                carry = read_lane(st_new, kWave - 1);
            }
#elif LC_SA_SPLIT
            // ---- round 5: the words of a chunk in TWO classes.  A value ends inside 18 % of the words (a URL is ~44 compressed
            // bytes); the per-byte work those words need — end test, match record, reset: ~9 VALU per byte — was paid by every
            // word, because every pass of 64 words holds some of them.  Now every word is walked WITHOUT looking at value ends
            // (extract, add, lookup: 2 VALU per byte — right for the 82 % that hold none), and the words with an end are listed
            // and walked again in dense passes with the full per-byte logic, their bytes fetched by index (L1 / L2: the chunk was
            // just streamed).  The carries and the worklist below read the end states back from LDS.
            uint32_t n_b = 0;
            uint16_t* list_bw = list_b;  // (free until the corrections; they start after the dense passes)
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) {
                const uint32_t wi = q * kWave + uint32_t(lane);
                const bool has = ends[wi] != 0;
                const uint64_t bm = __ballot(has);
                if (has) list_bw[n_b + lanes_below(bm)] = uint16_t(wi);
                n_b += uint32_t(__popcll(bm));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // the first two dense passes' words are requested before the cheap walk starts
            uint32_t bwi[2];
            uint64_t bww[2];
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
                const uint32_t j = u * kWave + uint32_t(lane);
                bwi[u] = j < n_b ? uint32_t(list_bw[j]) : 0u;
                bww[u] = *reinterpret_cast<GlobalPtr<uint64_t>>(reinterpret_cast<uintptr_t>(fsst + min(c0 + 8u * bwi[u], last_word)));
            }
            // phase 1: every word from the start state, value ends ignored, kSaIlp passes in lock step
#pragma unroll
            for (uint32_t q0 = 0; q0 < kSaPasses; q0 += kSaIlp) {
                uint32_t st[kSaIlp];
#pragma unroll
                for (uint32_t u = 0; u < kSaIlp; u++) st[u] = row0;
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
#pragma unroll
                    for (uint32_t u = 0; u < kSaIlp; u++) {
                        const uint64_t ww = w[q0 + u];
                        const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                        st[u] = lds_u16(st[u] + 2u * code);
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kSaIlp; u++) {
                    const uint32_t wi = (q0 + u) * kWave + uint32_t(lane);
                    out16[wi] = uint16_t(st[u]);
                    hit8[wi] = 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // phase 2: the words with a value end, densely, two passes in lock step
            for (uint32_t b0 = 0; b0 < n_b; b0 += 2u * kWave) {
                uint32_t st[2], hit[2], e8[2], wi2[2];
                uint64_t ww2[2];
                bool act[2];
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    const uint32_t j = b0 + u * kWave + uint32_t(lane);
                    act[u] = j < n_b;
                    wi2[u] = b0 == 0 ? bwi[u] : (act[u] ? uint32_t(list_bw[j]) : 0u);
                    ww2[u] = b0 == 0 ? bww[u]
                                     : *reinterpret_cast<GlobalPtr<uint64_t>>(reinterpret_cast<uintptr_t>(fsst + min(c0 + 8u * wi2[u], last_word)));
                    e8[u] = act[u] ? uint32_t(ends[wi2[u]]) : 0u;
                    st[u] = row0;
                    hit[u] = 0;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
#pragma unroll
                    for (uint32_t u = 0; u < 2; u++) {
                        const uint64_t ww = ww2[u];
                        const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                        const uint32_t t = lds_u16(st[u] + 2u * code);
                        const bool is_end = ((e8[u] >> k) & 1u) != 0;
                        hit[u] |= (is_end && t == hitrow) ? (1u << k) : 0u;
                        st[u] = is_end ? row0 : t;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < 2; u++) {
                    if (act[u]) {
                        out16[wi2[u]] = uint16_t(st[u]);
                        hit8[wi2[u]] = uint8_t(hit[u]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // phase 3: the carries of the matched state and the worklist, pass by pass in buffer order
            uint32_t n_list = 0;
#pragma unroll
            for (uint32_t q = 0; q < kSaPasses; q++) {
                const uint32_t wi = q * kWave + uint32_t(lane);
                uint32_t stq = uint32_t(out16[wi]), hitq = uint32_t(hit8[wi]);
                const uint32_t e8q = uint32_t(ends[wi]);
                const bool has_end = e8q != 0;
                const uint64_t M = __ballot(stq == hitrow), P = __ballot(!has_end);
                const uint64_t A = M | P;
                const unsigned __int128 sum = (unsigned __int128)A + M + (carry == hitrow ? 1u : 0u);
                const uint64_t im = uint64_t(sum) ^ A ^ M;  // bit l: carry INTO word l = "starts matched"
                if ((im >> lane) & 1u) {
                    if (has_end) hitq |= 1u << (uint32_t(__ffs(int(e8q))) - 1u);
                    else stq = hitrow;
                }
                const uint32_t prev = lane_shift_up1(stq, carry);  // end state of the word before
                carry = read_lane(stq, kWave - 1);
                const bool need = prev != row0 && prev != hitrow;
                out16[wi] = uint16_t(stq);
                in16[wi] = uint16_t(prev);
                hit8[wi] = uint8_t(hitq);
                const uint64_t m = __ballot(need);
                if (need) list_a[n_list + lanes_below(m)] = uint16_t(wi);
                n_list += uint32_t(__popcll(m));
            }
#else
            // ---- speculative walk: every word from the start state, kSaIlp passes in lock step
            uint32_t n_list = 0;
#pragma unroll
            for (uint32_t q0 = 0; q0 < kSaPasses; q0 += kSaIlp) {
                uint32_t st[kSaIlp], hit[kSaIlp], e8[kSaIlp];
#pragma unroll
                for (uint32_t u = 0; u < kSaIlp; u++) {
                    st[u] = row0;
                    hit[u] = 0;
                    e8[u] = ends[(q0 + u) * kWave + uint32_t(lane)];
                }
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
#pragma unroll
                    for (uint32_t u = 0; u < kSaIlp; u++) {
                        const uint64_t ww = w[q0 + u];
                        const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                        const uint32_t t = (LC_SA_ABL & 1) ? ((st[u] + 2u * code) & 0x1FFEu) : lds_u16(st[u] + 2u * code);
                        const bool is_end = ((e8[u] >> k) & 1u) != 0;
                        hit[u] |= (is_end && t == hitrow) ? (1u << k) : 0u;
                        st[u] = is_end ? row0 : t;
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kSaIlp; u++) {
                    const uint32_t wi = (q0 + u) * kWave + uint32_t(lane);
                    // The matched state is absorbing: a word that starts in it stays in it up to its first value end, whatever
                    // its bytes.  So "matched" travels from a word that ends matched through the words without a value end
                    // behind it like a carry through an adder — im(l) = M(l-1) | (P(l-1) & im(l-1)) — and the carries of
                    // (M | P) + M are exactly that: no lookup, no worklist for the common reason a start state is not the
                    // start state ('%ru/%' matches 89 % of the values, each a few words before its end).
                    const bool has_end = e8[u] != 0;
                    const uint64_t M = __ballot(st[u] == hitrow), P = __ballot(!has_end);
                    const uint64_t A = M | P;
                    const unsigned __int128 sum = (unsigned __int128)A + M + (carry == hitrow ? 1u : 0u);
                    const uint64_t im = uint64_t(sum) ^ A ^ M;  // bit l: carry INTO word l = "starts matched"
                    if ((im >> lane) & 1u) {
                        if (has_end) hit[u] |= 1u << (uint32_t(__ffs(int(e8[u]))) - 1u);
                        else st[u] = hitrow;
                    }
                    const uint32_t prev = lane_shift_up1(st[u], carry);  // end state of the word before
                    carry = read_lane(st[u], kWave - 1);
                    // what is left for the worklist: a partial match or an escape in progress across the word boundary
                    const bool need = prev != row0 && prev != hitrow;
                    out16[wi] = uint16_t(st[u]);
                    in16[wi] = uint16_t(prev);
                    hit8[wi] = uint8_t(hit[u]);
                    const uint64_t m = __ballot(need);
                    if (need) list_a[n_list + lanes_below(m)] = uint16_t(wi);
                    n_list += uint32_t(__popcll(m));
                }
            }
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // ---- corrections: words whose true start state is not the start state, in dense batches, to a fixpoint
            uint16_t* cur = list_a;
            uint16_t* nxt = list_b;
            if (LC_SA_ABL & 2) n_list = 0;
            while (n_list) {
                uint32_t n_next = 0;
                for (uint32_t b0 = 0; b0 < n_list; b0 += kWave) {
                    const uint32_t j = b0 + uint32_t(lane);
                    const bool act = j < n_list;
                    const uint32_t wi = act ? cur[j] : 0u;
                    const uint64_t ww = *reinterpret_cast<GlobalPtr<uint64_t>>(reinterpret_cast<uintptr_t>(fsst + min(c0 + 8u * wi, last_word)));
                    const uint32_t e = ends[wi];
                    const uint32_t fe = e ? uint32_t(__ffs(int(e))) - 1u : 8u;  // bytes 0 .. fe depend on the start state
                    uint32_t s = in16[wi];
#pragma unroll
                    for (uint32_t k = 0; k < 8; k++) {
                        const uint32_t code = (k < 4 ? uint32_t(ww) >> (8 * k) : uint32_t(ww >> 32) >> (8 * (k - 4))) & 0xFFu;
                        const uint32_t t = lds_u16(s + 2u * code);
                        s = k <= fe ? t : s;
                    }
                    bool push = false;
                    if (act) {
                        if (e) {
                            const uint32_t h = (uint32_t(hit8[wi]) & ~(1u << fe)) | (s == hitrow ? (1u << fe) : 0u);
                            hit8[wi] = uint8_t(h);
#if LC_SA_SPLIT == 2
                            chunk_hit |= h;
#endif
                        } else if (s != uint32_t(out16[wi])) {
                            out16[wi] = uint16_t(s);
                            if (wi + 1u < kSaChunkWords) {
                                in16[wi + 1u] = uint16_t(s);
                                push = true;
                            }
                        }
                    }
                    const uint64_t pm = __ballot(push);
                    if (push) nxt[n_next + lanes_below(pm)] = uint16_t(wi + 1u);
                    n_next += uint32_t(__popcll(pm));
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                uint16_t* t2 = cur;
                cur = nxt;
                nxt = t2;
                n_list = n_next;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            carry = uint32_t(out16[kSaChunkWords - 1u]);  // (corrected)
            carry = uint32_t(__builtin_amdgcn_readfirstlane(int(carry)));
            // ---- matches -> dictionary values: the r-th value end of the chunk is value va + r (+ 1 behind the empty value)
            uint32_t run = 0;
#if LC_SA_SPLIT == 2
            const bool attribute = __ballot(chunk_hit != 0) != 0;  // (a chunk without a hit: nothing to attribute)
#else
            const bool attribute = true;
#endif
            for (uint32_t q = 0; q < (((LC_SA_ABL & 2) || !attribute) ? 0u : kSaPasses); q++) {
                const uint32_t wi = q * kWave + uint32_t(lane);
                const uint32_t e = ends[wi];
                uint32_t h = hit8[wi];
                const uint32_t pc = uint32_t(__popc(e));
                const uint32_t incl = wave_inclusive_sum(pc);
                const uint64_t hm = __ballot(h != 0);
                if (hm) {
                    any_true |= hm;
                    while (h) {
                        const uint32_t k = uint32_t(__ffs(int(h))) - 1u;
                        h &= h - 1u;
                        uint32_t idx = va + run + (incl - pc) + uint32_t(__popc(e & ((1u << k) - 1u)));
                        if (e0 >= va && e0 <= idx) idx++;
                        if (idx < D) atomicOr(dres + (idx >> 5), 1u << (idx & 31u));
                    }
                }
                run += read_lane(incl, kWave - 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }

    // ---- rows: hit = (dictionary result of the row's key, inverted for NOT LIKE) AND valid AND selected
    bool invert = false;
    if (kNot) invert = (a.use_fp && dp->fingerprints) ? fp_cand != 0 : true;
    const bool all_false = any_true == 0;
    const bool need_vw = !all_false || invert || a.valid != nullptr;
    const uint32_t xor8 = invert ? 0xFFu : 0u;
    const uint32_t key_max = a.dres_bytes * 8u - 1u;  // keys under null slots may be garbage (clamped, masked by validity)
    const uint64_t word_off = dp->mask_word_off;
    uint32_t hit_count = 0;
    constexpr int KC = 8;
    for (uint32_t pass = 0; pass < n_rows; pass += KC * kWave * 8) {
        u32x4 kv[KC];
        if (!all_false) {
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const uint32_t r0 = pass + uint32_t(k) * kWave * 8 + uint32_t(lane) * 8;
                kv[k] = *reinterpret_cast<GlobalPtr<u32x4>>(as_global(dp->keys) + min(r0, (n_rows - 1u) & ~7u));
            }
        }
        uint64_t vw[KC * 8 / kWave];
#pragma unroll
        for (int h = 0; h < KC * 8 / kWave; h++) {
            const uint32_t widx = (pass >> 6) + uint32_t(h) * kWave + uint32_t(lane);
            const uint32_t wc = min(widx, nwords - 1u);
            uint64_t sv = ~uint64_t(0), vv = ~uint64_t(0);
            if (need_vw && a.selection) sv = *as_global(a.selection + word_off + wc);
            if (need_vw && dp->validity) vv = *as_global(dp->validity + wc);
            const uint32_t rows_left = n_rows - (wc << 6);
            const uint64_t tail = rows_left >= 64 ? ~uint64_t(0) : ((uint64_t(1) << rows_left) - 1);
            vw[h] = widx < nwords ? (sv & vv & tail) : 0;
        }
        if (!all_false) {
#pragma unroll
            for (int k = 0; k < KC; k++) {
                uint32_t bits = 0;
                const uint32_t kw[4] = {kv[k].x, kv[k].y, kv[k].z, kv[k].w};
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t key = (q & 1) ? kw[q >> 1] >> 16 : kw[q >> 1] & 0xFFFFu;
                    const uint32_t kc = min(key, key_max);
                    bits |= ((dres[kc >> 5] >> (kc & 31)) & 1u) << q;
                }
                stage[uint32_t(k) * kWave + uint32_t(lane)] = uint8_t(bits ^ xor8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#pragma unroll
        for (int h = 0; h < KC * 8 / kWave; h++) {
            const uint32_t wl = uint32_t(h) * kWave + uint32_t(lane);
            const uint32_t widx = (pass >> 6) + wl;
            if (widx < nwords) {
                const uint64_t rw = all_false ? (invert ? ~uint64_t(0) : uint64_t(0)) : reinterpret_cast<const uint64_t*>(stage)[wl];
                const uint64_t hitw = rw & vw[h];
                as_global_mut(a.mask)[word_off + widx] = hitw;
                if (a.valid) as_global_mut(a.valid)[word_off + widx] = vw[h];
                hit_count += uint32_t(__popcll(hitw));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    uint64_t wave_hits = 0;
    if (a.counts || a.total.d_total_out) {
        wave_hits = (all_false && !invert) ? 0 : uniform_u64(wave_sum_u64(uint64_t(hit_count)));
        if (lane == 0 && a.counts) as_global_mut(a.counts)[entry] = uint32_t(wave_hits);
    }
    if (a.total.d_total_out && lane == 0) total_contribute(a.total, unit, n_units, wave_hits);
}

}  // namespace

// Launcher: one workgroup per record of the scan (<= 4 entries of one symbol table, one per wave).
hipError_t launch_like_scanall(const StrWgRecord* d_recs, uint32_t n_recs, const StrPred& pred, const ScanLaunch& L,
                               unsigned long long* d_total_acc, hipStream_t stream, const uint64_t* uni_slice,
                               const uint32_t* uni_word) {
    if (n_recs == 0) return hipSuccess;
    ScanAllArgs a{};
    a.recs = d_recs;
    a.n_recs = n_recs;
    a.automata = pred.automata;
    a.automaton_stride = pred.automaton_stride;
    a.nl = pred.needle_len;
    a.use_fp = pred.use_fingerprints ? 1u : 0u;
    a.needle_fp = pred.needle_fp;
    a.dres_bytes = (((std::max<uint32_t>(L.max_dict_len, 1u) + 63u) / 64u) * 8u + 15u) & ~15u;
    a.selection = L.d_selection;
    a.mask = L.d_hit;
    a.valid = L.d_valid;
    a.counts = L.d_counts;
    a.total.d_total_acc = d_total_acc;
    a.total.d_total_out = L.d_total_out;
    a.uni_slice = uni_slice;
    a.uni_word = uni_word;
    const bool uni = uni_slice != nullptr;
    const size_t lds = (uni ? 0 : automaton_image_bytes(pred.needle_len)) + size_t(kWavesPerBlock) * (a.dres_bytes + kSaFixedLds);
    typedef void (*Kern)(ScanAllArgs);
    const Kern kern = uni ? (pred.op == LC_OP_NOT_LIKE ? static_cast<Kern>(k_like_scanall<true, true>) : static_cast<Kern>(k_like_scanall<false, true>))
                          : (pred.op == LC_OP_NOT_LIKE ? static_cast<Kern>(k_like_scanall<true, false>) : static_cast<Kern>(k_like_scanall<false, false>));
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 160 * 1024 - 512);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_recs), dim3(kThreads), lds, stream, a);
    return hipGetLastError();
}

hipError_t warm_code_object_like_scanall() {  // (see warm_code_object_kernels)
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_like_scanall<false, false>));
}

}  // namespace lc
