// Device-side helpers shared by the kernel translation units (lc_kernels.hip, lc_like_pipeline.hip): wave64 cross-lane
// steps on the DPP path, global -> LDS DMA, address-space casts that keep accesses off the FLAT path, the fused COUNT(*)
// accumulator and the 8-byte step of the LIKE automaton walk.  gfx950 only.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "lc_kernels.hpp"

extern "C" __device__ int __llvm_amdgcn_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");

namespace lc {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;

__device__ __forceinline__ int lane_id() { return int(threadIdx.x) & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return int(threadIdx.x) >> 6; }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;  // lane 0 holds the total
}

// Cross-lane steps on the VALU's DPP path (a few cycles) instead of ds_bpermute (an LDS round trip, ~100 cycles):
// the byte-view kernel is bound by the length of its dependent chains.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kCtrl, kRowMask, 0xF, false));
}
// inclusive prefix sum over the 64 lanes (gfx9 DPP: row_shr within rows of 16, then row_bcast15 / row_bcast31)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += dpp_or_zero<0x111, 0xF>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return v;
}
// OR of `v` over the 64 lanes, in every lane (the same DPP ladder, then a broadcast of lane 63)
__device__ __forceinline__ uint32_t wave_or_all(uint32_t v) {
    v |= dpp_or_zero<0x111, 0xF>(v);
    v |= dpp_or_zero<0x112, 0xF>(v);
    v |= dpp_or_zero<0x114, 0xF>(v);
    v |= dpp_or_zero<0x118, 0xF>(v);
    v |= dpp_or_zero<0x142, 0xA>(v);
    v |= dpp_or_zero<0x143, 0xC>(v);
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}
// value of the previous lane (lane 0 gets `first`)
__device__ __forceinline__ uint32_t lane_shift_up1(uint32_t v, uint32_t first) {
    return uint32_t(__builtin_amdgcn_update_dpp(int(first), int(v), 0x138, 0xF, 0xF, false));  // wave_shr:1
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
    const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
    return uint64_t(lo) | (uint64_t(hi) << 32);
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) { return uint32_t(__builtin_amdgcn_readlane(int(v), lane)); }

// global -> LDS DMA of 16 bytes per active lane: lane l writes lds_base + l*16 (lds_base must be wave-uniform)
__device__ __forceinline__ void async_copy16(const void* gsrc_lane, void* lds_base) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                         reinterpret_cast<uintptr_t>(gsrc_lane)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>(
                                         uint32_t(reinterpret_cast<uintptr_t>(lds_base))),
                                     16, 0, 0);
}

// The same for data that a scan reads exactly ONCE (packed column words): non-temporal (aux = 2).  Round 6's policy probe
// (scripts/micro/stream_probe.hip, profiles/r6/stream_probe.txt): a 788 MB read stream runs at 6.0 TB/s with the default policy
// and 6.7-6.8 TB/s non-temporal, LDS-DMA and 16-byte register loads alike.  Measured on the kernels (profiles/r6/ab_stream_nt.txt):
// the LDS-DMA scan of a W = 62 column 150.6 -> 136.5 us L3-cold (-DLC_STREAM_NT=0 restores the default policy); the
// register-resident narrow-integer kernels gain nothing cold (37.5 vs 36.8, 27.2 vs 25.9, 49.6 vs 52.7 us) and lose the
// Infinity-Cache hits of a column that fits it (32.9 vs 29.5 us hot), so their loads keep the default policy
// (-DLC_STREAM_NT_REG=1 is the A/B switch).
#ifndef LC_STREAM_NT
#define LC_STREAM_NT 1
#endif
#ifndef LC_STREAM_NT_REG
#define LC_STREAM_NT_REG 0
#endif
__device__ __forceinline__ void async_copy16_stream(const void* gsrc_lane, void* lds_base) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(
                                         reinterpret_cast<uintptr_t>(gsrc_lane)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>(
                                         uint32_t(reinterpret_cast<uintptr_t>(lds_base))),
                                     16, 0, LC_STREAM_NT ? 2 : 0);
}

// v_writelane_b32 with a compile-time lane.  The LLVM intrinsic, not inline assembly: gfx950 needs two wait states
// between a VALU instruction that writes an SGPR / VCC (v_cmp) and a v_writelane that reads it.  The compiler's hazard
// recognizer provides them (s_nop, or independent work scheduled in between) for its own instructions only; a
// hand-written v_writelane placed right behind the v_cmp reads the PREVIOUS mask (found the hard way: counts stayed
// plausible, bits did not).
template <uint32_t LANE>
__device__ __forceinline__ uint32_t writelane_c(uint32_t sval, uint32_t old) {
    return uint32_t(__llvm_amdgcn_writelane_i32(int(sval), int(LANE), int(old)));
}

// Pointers read out of descriptors are generic to the compiler; these casts make the accesses global_load (own
// vmcnt counter, no coupling with LDS waits) instead of flat_load.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
using GlobalPtr = const __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ GlobalPtr<T> as_global(const T* p) { return (GlobalPtr<T>)p; }
// a register load of read-once data (see async_copy16_stream)
template <typename T>
__device__ __forceinline__ T stream_load(GlobalPtr<T> p) {
    if constexpr (LC_STREAM_NT_REG != 0) return __builtin_nontemporal_load(p);
    else return *p;
}
// stores through a pointer the compiler cannot prove global would be FLAT stores: besides the slower path, a pending flat
// access makes the compiler wait for ALL outstanding loads (vmcnt(0)) at the next use of any loaded value
template <typename T>
using GlobalMutPtr = __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ GlobalMutPtr<T> as_global_mut(T* p) { return (GlobalMutPtr<T>)p; }

// Fused COUNT(*): called by ONE lane of every wave of the launch (`unit` = its index, `n_units` = how many there are)
// with the hits of the entries that wave evaluated.  Two levels of {arrivals : 24 | hits : 40} words: the last arrival
// of a shard forwards the shard's sum to the top word, the last shard writes the total and leaves every word zero
// for the next launch.  One returning far atomic per wave, nothing spins.
__device__ __forceinline__ void total_contribute(const ScanLaunch& L, uint32_t unit, uint32_t n_units, uint64_t hits) {
    constexpr unsigned long long kOne = 1ull << 40, kMask = kOne - 1;
    // shard = unit % n_shards, in_shard = ceil((n_units - shard) / n_shards) — without a division (two 32-bit divisions are
    // ~45 scalar instructions per wave, 5 % of everything a k_like_lean wave executes): with at least kTotalShards units the
    // divisor is the power of two, with fewer every unit is its own shard (unit < n_units)
    static_assert((kTotalShards & (kTotalShards - 1u)) == 0, "kTotalShards is a power of two");
    const bool full = n_units >= kTotalShards;
    const uint32_t n_shards = full ? kTotalShards : n_units;
    const uint32_t shard = full ? (unit & (kTotalShards - 1u)) : unit;
    const uint32_t in_shard = full ? (n_units - shard + kTotalShards - 1u) / kTotalShards : 1u;
    unsigned long long* sw = L.d_total_acc + 8u * (shard + 1u);
    const unsigned long long inc = kOne | (hits & kMask);
    const unsigned long long old = atomicAdd(sw, inc);
    if ((old >> 40) + 1ull != in_shard) return;
    const unsigned long long shard_hits = (old + inc) & kMask;
    atomicExch(sw, 0ull);
    const unsigned long long tinc = kOne | shard_hits;
    const unsigned long long told = atomicAdd(L.d_total_acc, tinc);
    if ((told >> 40) + 1ull != n_shards) return;
    atomicExch(L.d_total_acc, 0ull);
    // an atomic store: the value must land in memory, not in this XCD's L2 behind a later k_alp_patch_fix adjustment
    atomicExch(reinterpret_cast<unsigned long long*>(L.d_total_out), (told + tinc) & kMask);
}

// A hit list as its consumers see it: `parts` partitions (1: the contiguous form) in partition order.  hitlist_prefix fills
// s_prefix[0 .. kHitParts] (LDS, written by the first threads of the workgroup; ends with a workgroup barrier) with the
// number of records in front of every partition and returns the total; hitlist_at maps an index of that order to its
// position in the buffer.
__device__ __forceinline__ uint64_t hitlist_prefix(const unsigned long long* n_hits, uint64_t cap, uint32_t parts, uint64_t* s_prefix) {
    if (threadIdx.x == 0) {
        if (parts <= 1) {
            s_prefix[0] = 0;
            s_prefix[1] = min(uint64_t(*n_hits), cap);
        } else {
            const uint64_t stride = cap / parts;
            uint64_t run = 0;
            for (uint32_t p = 0; p < parts; p++) {
                s_prefix[p] = run;
                run += min(uint64_t(n_hits[p * kHitCounterStride]), stride);
            }
            s_prefix[parts] = run;
        }
    }
    __syncthreads();
    return s_prefix[parts <= 1 ? 1 : parts];
}
__device__ __forceinline__ uint64_t hitlist_at(const uint64_t* s_prefix, uint64_t cap, uint32_t parts, uint64_t i) {
    if (parts <= 1) return i;
    static_assert(kHitParts == 16, "four halving steps");
    uint32_t p = i >= s_prefix[8] ? 8u : 0u;
    p += i >= s_prefix[p + 4] ? 4u : 0u;
    p += i >= s_prefix[p + 2] ? 2u : 0u;
    p += i >= s_prefix[p + 1] ? 1u : 0u;
    return uint64_t(p) * (cap / parts) + (i - s_prefix[p]);
}

template <typename T>
__device__ __forceinline__ T load_unaligned(const uint8_t* p) {
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t*)p, sizeof(T));
    return v;
}

template <typename T>
__device__ __forceinline__ void store_unaligned(uint8_t* p, T v) {
    __builtin_memcpy((__attribute__((address_space(1))) uint8_t*)p, &v, sizeof(T));
}

typedef const __attribute__((address_space(3))) uint8_t* LdsBytePtr;
typedef const __attribute__((address_space(3))) uint16_t* LdsU16Ptr;
__device__ __forceinline__ uint32_t lds_u16(uint32_t addr) { return *reinterpret_cast<LdsU16Ptr>(addr); }
__device__ __forceinline__ uint32_t lds_u8(uint32_t addr) { return *reinterpret_cast<LdsBytePtr>(addr); }

// compressed bytes per lane and pass of the lane-parallel walk.  An entry's ~9 candidates are ~80 8-byte words, so
// 16-byte tasks would make one pass the usual case — measured: no gain (the kernel is bound by instruction issue, not by
// the extra round trip), and the second word's registers cost a wave of occupancy (35 -> 42 us).
#ifndef LC_WALK_TASK_WORDS
#define LC_WALK_TASK_WORDS 1
#endif
constexpr int kTaskWords = LC_WALK_TASK_WORDS;
constexpr uint32_t kTaskBytes = 8u * kTaskWords;
__device__ __forceinline__ uint32_t walk8(uint32_t sb, const uint32_t (&x)[8], uint32_t rem) {
    const uint32_t s1 = lds_u16(sb + x[0]);
    const uint32_t s2 = lds_u16(s1 + x[1]);
    const uint32_t s3 = lds_u16(s2 + x[2]);
    const uint32_t s4 = lds_u16(s3 + x[3]);
    const uint32_t s5 = lds_u16(s4 + x[4]);
    const uint32_t s6 = lds_u16(s5 + x[5]);
    const uint32_t s7 = lds_u16(s6 + x[6]);
    const uint32_t s8 = lds_u16(s7 + x[7]);
    // state after the first min(rem, 8) bytes
    const uint32_t r = min(rem, 8u) - 1u;
    const uint32_t a0 = (r & 1u) ? s2 : s1, a1 = (r & 1u) ? s4 : s3, a2 = (r & 1u) ? s6 : s5, a3 = (r & 1u) ? s8 : s7;
    const uint32_t b0 = (r & 2u) ? a1 : a0, b1 = (r & 2u) ? a3 : a2;
    const uint32_t sel = (r & 4u) ? b1 : b0;
    return rem == 0 ? sb : sel;
}


// offsets of dictionary entries i and i+1 with ONE (unaligned, 8-byte) load: the residual width only selects shifts,
// so there is no branch between the load and its use (sections are padded, reading a few bytes past is safe)
template <typename D>
__device__ __forceinline__ void str_offset_pair(const D& d, uint32_t i, uint32_t& start, uint32_t& stop) {
    const uint32_t ob = d.offset_bytes;  // 1, 2 or 4
    const uint64_t v = load_unaligned<uint64_t>(d.residuals + size_t(i) * ob);
    const uint32_t sh = 32u - 8u * ob;
    const int32_t r0 = int32_t(uint32_t(v) << sh) >> sh;
    const int32_t r1 = int32_t(uint32_t(v >> (8u * ob)) << sh) >> sh;
    start = uint32_t(d.slope) * i + uint32_t(d.intercept) + uint32_t(r0);
    stop = uint32_t(d.slope) * (i + 1u) + uint32_t(d.intercept) + uint32_t(r1);
}



// Decoding iterator over one FSST-compressed value: position in the code stream + byte index inside the current symbol
// (raw/fsst_buffer.rs:642-663: code 255 escapes the next byte; a dangling escape marker is ignored).
struct FsstIter {
    uint32_t pos, stop;  // next code, end of the value
    uint64_t sym;        // bytes of the current symbol
    uint32_t len, k;     // its length, index of the current byte
    bool at_end;
};
__device__ __forceinline__ void fsst_iter_load(FsstIter& it, const DevSymtab& st, const uint8_t* __restrict__ fsst) {
    while (it.pos < it.stop) {
        const uint32_t c = fsst[it.pos++];
        if (c == 255u) {
            if (it.pos >= it.stop) break;  // dangling escape marker: ignored, like the reference decoder
            it.sym = fsst[it.pos++];
            it.len = 1;
        } else {
            it.sym = st.sym[c];
            it.len = st.len[c];
        }
        it.k = 0;
        if (it.len) return;
    }
    it.at_end = true;
}
__device__ __forceinline__ uint32_t fsst_iter_cur(const FsstIter& it) { return uint32_t(it.sym >> (8u * it.k)) & 0xFFu; }
__device__ __forceinline__ void fsst_iter_next(FsstIter& it, const DevSymtab& st, const uint8_t* __restrict__ fsst) {
    if (++it.k == it.len) fsst_iter_load(it, st, fsst);
}

}  // namespace
}  // namespace lc
