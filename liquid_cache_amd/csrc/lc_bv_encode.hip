// Arrow Utf8 / Binary -> LiquidByteViewArray ON THE DEVICE (LiquidByteViewArray::from_string_array / from_binary_array,
// byte_view_array/conversions.rs:260-373; serialization.rs:122-220): the steps the host transcoder (lc_transcode.cpp
// transcode_byte_view) runs per 8192-row batch, as two kernels over a whole row group of batches, one workgroup per batch.
//
//   k_bv_build   dictionary in first-occurrence order (GenericByteDictionaryBuilder::append_option, utils/mod.rs:147-161):
//                every valid row is inserted into an open-addressing table keyed by its bytes; a slot remembers the
//                SMALLEST row that holds its value (atomicMin), so "row r is the first occurrence" is table[slot(r)] == r
//                and the dictionary order is a prefix sum over those flags.
//                Then, per dictionary value: FSST compression with the path's symbol table (the greedy longest-match
//                encoder of lc_fsst.hpp, same decisions, its tables in LDS), the 32-bucket byte fingerprint
//                (fingerprint.rs:33-35), the common prefix with value 0 (conversions.rs:269-307), the uncompressed size;
//                a prefix sum of the compressed lengths gives the offsets, and the compact-offset line (fsst_buffer.rs:
//                267-359) is fitted from exact integer sums — the reference accumulates the same integers in f64, which is
//                exact below 2^53; anything larger is reported and the caller takes the host path.
//   k_bv_pack    writes the entry's sections where the runtime allocated them (sizes are known after k_bv_build): keys,
//                validity, prefix keys (fsst_buffer.rs:175-187), fingerprints, offset residuals, the compressed bytes
//                compacted to their offsets, the shared prefix, and the inverted row lists (lc_kernels.hpp) by a bitonic
//                sort of (key, row) pairs in LDS — rows of a key ascending, exactly the host's counting sort.
//
// The staged entry is byte-identical to what lc_insert_arrow stages (tests compare lc_entry_to_liquid_bytes and
// lc_entry_index_to_bytes of both).
#include "lc_device.hpp"
#include "lc_internal.hpp"

namespace lc {
namespace {

constexpr int kBvThreads = 1024;
constexpr uint32_t kBvEmpty = 0xFFFFFFFFu;

struct EncLds {
    uint64_t long_sym[256];
    uint32_t short2[kDevEncShort2Slots];
    uint16_t short1[256];
    uint8_t long_len[256];
    uint8_t long_code[256];
    uint8_t bucket[kDevEncBuckets + 8];
};

__device__ __forceinline__ bool row_valid(const uint64_t* validity, uint32_t r) {
    return !validity || ((validity[r >> 6] >> (r & 63)) & 1u) != 0;
}

__device__ __forceinline__ uint64_t load_tail(const uint8_t* p, uint32_t avail) {
    // `avail` (1..7) bytes at p, zero extended — the input buffer is padded so an 8-byte load is always in bounds
    const uint64_t w = load_unaligned<uint64_t>(p);
    return w & ((uint64_t(1) << (8u * avail)) - 1);
}

__device__ __forceinline__ uint64_t hash_value(const uint8_t* p, uint32_t len) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) {
        h = (h ^ load_unaligned<uint64_t>(p + i)) * 0xC4CEB9FE1A85EC53ull;
        h ^= h >> 29;
    }
    if (i < len) h = (h ^ load_tail(p + i, len - i)) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 32);
}

__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t len) {
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8)
        if (load_unaligned<uint64_t>(a + i) != load_unaligned<uint64_t>(b + i)) return false;
    return i == len || load_tail(a + i, len - i) == load_tail(b + i, len - i);
}

// block-wide exclusive prefix sum of one value per thread (1024 threads = 16 waves); returns the exclusive sum and the total
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* wave_tot /* 16 */, uint32_t* total) {
    const uint32_t inc = wave_inclusive_sum(v);
    const int w = wave_id(), lane = lane_id();
    __syncthreads();  // wave_tot may still be read from a previous call
    if (lane == kWave - 1) wave_tot[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < kBvThreads / kWave; i++) {
        const uint32_t t = wave_tot[i];
        if (i < w) base += t;
        tot += t;
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kBvThreads) void k_bv_build(const BvEncodeDesc* __restrict__ descs, const DevFsstEncoder* __restrict__ encoders) {
    __shared__ EncLds enc;
    __shared__ uint32_t wave_tot[kBvThreads / kWave];
    __shared__ uint32_t s_sp_len, s_min_res, s_max_res;
    __shared__ unsigned long long s_raw, s_sy, s_sxy;
    __shared__ int32_t s_slope, s_intercept;
    const BvEncodeDesc& a = descs[blockIdx.x];
    const uint32_t tid = threadIdx.x, n = a.n;
    {   // encoder tables -> LDS
        const DevFsstEncoder& g = encoders[a.encoder];
        for (uint32_t i = tid; i < 256; i += kBvThreads) {
            enc.long_sym[i] = g.long_sym[i];
            enc.short1[i] = g.short1[i];
            enc.long_len[i] = g.long_len[i];
            enc.long_code[i] = g.long_code[i];
        }
        for (uint32_t i = tid; i < kDevEncShort2Slots; i += kBvThreads) enc.short2[i] = g.short2[i];
        for (uint32_t i = tid; i < kDevEncBuckets + 1; i += kBvThreads) enc.bucket[i] = g.bucket[i];
        if (tid == 0) { s_sp_len = 0; s_raw = 0; s_sy = 0; s_sxy = 0; s_min_res = 0xFFFFFFFFu; s_max_res = 0; }
    }
    const int32_t base = n ? a.offsets[0] : 0;
    auto row_start = [&](uint32_t r) { return uint32_t(a.offsets[r] - base); };
    auto row_len = [&](uint32_t r) { return uint32_t(a.offsets[r + 1] - a.offsets[r]); };
    // ---- dictionary: insert every valid row (the table was set to kBvEmpty by the host)
    for (uint32_t r = tid; r < n; r += kBvThreads) {
        if (!row_valid(a.validity, r)) continue;
        const uint8_t* p = a.data + row_start(r);
        const uint32_t len = row_len(r);
        uint32_t s = uint32_t(hash_value(p, len)) & a.table_mask;
        for (;;) {
            const uint32_t cur = atomicCAS(&a.table[s], kBvEmpty, r);
            if (cur == kBvEmpty) break;  // claimed
            // `cur` holds the same value for as long as the slot lives: later atomicMins only swap in rows with equal bytes
            if (row_len(cur) == len && bytes_equal(a.data + row_start(cur), p, len)) {
                atomicMin(&a.table[s], r);
                break;
            }
            s = (s + 1) & a.table_mask;
        }
        a.row_slot[r] = s;
    }
    __threadfence_block();
    __syncthreads();
    // ---- dictionary order: prefix sum over "first occurrence" flags; thread t owns rows [t*R, t*R + R)
    const uint32_t R = (n + kBvThreads - 1) / kBvThreads;
    uint32_t firsts = 0;
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t r = tid * R + i;
        if (r < n && row_valid(a.validity, r) && __atomic_load_n(&a.table[a.row_slot[r]], __ATOMIC_RELAXED) == r) firsts++;
    }
    uint32_t d = 0;
    uint32_t k0 = block_exclusive_sum(firsts, wave_tot, &d);
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t r = tid * R + i;
        if (r < n && row_valid(a.validity, r) && __atomic_load_n(&a.table[a.row_slot[r]], __ATOMIC_RELAXED) == r) {
            a.dict_row[k0] = r;
            a.dict_index[r] = k0;
            k0++;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t r = tid; r < n; r += kBvThreads) {
        uint32_t key = 0;
        if (row_valid(a.validity, r)) key = a.dict_index[__atomic_load_n(&a.table[a.row_slot[r]], __ATOMIC_RELAXED)];
        a.keys[r] = uint16_t(key);
    }
    // ---- per dictionary value: size, common prefix with value 0, fingerprint, FSST compression
    if (tid == 0 && d) s_sp_len = row_len(a.dict_row[0]);
    __syncthreads();
    {
        const uint32_t r0 = d ? a.dict_row[0] : 0;
        const uint8_t* p0 = a.data + (d ? row_start(r0) : 0);
        const uint32_t len0 = d ? row_len(r0) : 0;
        uint64_t raw = 0;
        uint32_t sp = len0;
        for (uint32_t k = tid; k < d; k += kBvThreads) {
            const uint32_t r = a.dict_row[k];
            const uint32_t start = row_start(r), len = row_len(r);
            const uint8_t* p = a.data + start;
            raw += len;
            uint32_t c = 0;
            const uint32_t m = min(sp, len);
            while (c < m && p[c] == p0[c]) c++;
            sp = c;
            uint8_t* out = a.comp + 2 * size_t(start);
            uint32_t fp = 0;
            const uint32_t o = dev_enc_compress(enc, [p](uint32_t pos, uint32_t avail) {
                return avail >= 8 ? load_unaligned<uint64_t>(p + pos) : load_tail(p + pos, avail);
            }, len, out, &fp);
            a.clen[k] = o;
            a.fingerprints[k] = fp;
        }
        raw = wave_sum_u64(raw);
        if (lane_id() == 0 && raw) atomicAdd(&s_raw, (unsigned long long)raw);
        if (d) atomicMin(&s_sp_len, sp);
    }
    __threadfence_block();
    __syncthreads();
    // ---- offsets[0..d] = exclusive prefix sum of the compressed lengths; thread t owns entries [t*Q, t*Q + Q)
    const uint32_t nd = d + 1;
    const uint32_t Q = (nd + kBvThreads - 1) / kBvThreads;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < Q; i++) {
        const uint32_t k = tid * Q + i;
        if (k < d) mine += a.clen[k];
    }
    uint32_t total = 0;
    uint32_t run = block_exclusive_sum(mine, wave_tot, &total);
    unsigned long long sy = 0, sxy = 0;
    for (uint32_t i = 0; i < Q; i++) {
        const uint32_t k = tid * Q + i;
        if (k >= nd) break;
        const uint32_t l = k < d ? a.clen[k] : 0;
        a.offsets_out[k] = run;  // (clen[k] was read above: offsets_out is a different array)
        sy += run;
        sxy += (unsigned long long)k * run;
        run += l;
    }
    sy = wave_sum_u64(sy);
    sxy = wave_sum_u64(sxy);
    if (lane_id() == 0) { atomicAdd(&s_sy, sy); atomicAdd(&s_sxy, sxy); }
    __threadfence_block();
    __syncthreads();
    // ---- the line through the offsets (fit_line, fsst_buffer.rs:267-296); no FMA contraction: the host rounds every step
    if (tid == 0) {
#pragma clang fp contract(off)
        int32_t slope = 0, intercept = 0;
        if (nd > 1) {
            const unsigned long long N = nd;
            const double nf = double(N), sx = double(N * (N - 1) / 2), sxx = double(N * (N - 1) * (2 * N - 1) / 6);
            const double fy = double(s_sy), fxy = double(s_sxy);
            const double s = (nf * fxy - sx * fy) / (nf * sxx - sx * sx);
            const double ic = (fy - s * sx) / nf;
            auto sat = [](double x) -> int32_t {
                x = round(x);
                if (x != x) return 0;
                if (x > 2147483647.0) return 2147483647;
                if (x < -2147483648.0) return int32_t(-2147483647 - 1);
                return int32_t(x);
            };
            slope = sat(s);
            intercept = sat(ic);
        }
        s_slope = slope;
        s_intercept = intercept;
    }
    __syncthreads();
    {
        const int32_t slope = s_slope, intercept = s_intercept;
        int32_t mn = 2147483647, mx = int32_t(-2147483647 - 1);
        for (uint32_t k = tid; k < nd; k += kBvThreads) {
            const uint32_t pred = uint32_t(slope) * k + uint32_t(intercept);
            const int32_t res = int32_t(a.offsets_out[k] - pred);
            mn = min(mn, res);
            mx = max(mx, res);
        }
        // order-preserving map to u32 for the LDS atomics
        if (tid < nd) {
            atomicMin(&s_min_res, uint32_t(mn) ^ 0x80000000u);
            atomicMax(&s_max_res, uint32_t(mx) ^ 0x80000000u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int32_t mn = int32_t(s_min_res ^ 0x80000000u), mx = int32_t(s_max_res ^ 0x80000000u);
        BvEncodeStats st{};
        st.d = d;
        st.fsst_len = total;
        st.shared_prefix_len = s_sp_len;
        st.raw_bytes = s_raw;
        st.slope = s_slope;
        st.intercept = s_intercept;
        st.offset_bytes = (mn >= -128 && mx <= 127) ? 1 : (mn >= -32768 && mx <= 32767) ? 2 : 4;
        st.inexact = (s_sxy >> 53) != 0 || (s_sy >> 53) != 0;
        *a.stats = st;
    }
}

// ---------------------------------------------------------------------------------------------------------------- pack
__global__ __launch_bounds__(kBvThreads) void k_bv_pack(const BvEncodeDesc* __restrict__ descs, const BvPackDesc* __restrict__ packs) {
    __shared__ uint32_t pairs[kPostLdsRows];
    const BvEncodeDesc& a = descs[blockIdx.x];
    const BvPackDesc& o = packs[blockIdx.x];
    const uint32_t tid = threadIdx.x, n = a.n, d = o.d;
    const int32_t base = n ? a.offsets[0] : 0;
    auto row_start = [&](uint32_t r) { return uint32_t(a.offsets[r] - base); };
    auto row_len = [&](uint32_t r) { return uint32_t(a.offsets[r + 1] - a.offsets[r]); };
    for (uint32_t r = tid; r < n; r += kBvThreads) o.keys[r] = a.keys[r];
    if (o.validity) {
        const uint32_t words = (n + 63) >> 6;
        for (uint32_t w = tid; w < words; w += kBvThreads) {
            uint64_t v = a.validity[w];
            if (w == words - 1 && (n & 63u)) v &= (uint64_t(1) << (n & 63u)) - 1;
            o.validity[w] = v;
        }
    }
    const uint32_t sp = o.shared_prefix_len;
    for (uint32_t k = tid; k < d; k += kBvThreads) {
        const uint32_t r = a.dict_row[k];
        const uint32_t start = row_start(r), len = row_len(r);
        const uint8_t* p = a.data + start;
        // prefix key: the first 7 bytes after the shared prefix, zero padded; byte 7 = remaining length (255: longer)
        const uint32_t rl = sp < len ? len - sp : 0;
        uint64_t pk = 0;
        if (rl) pk = rl >= 8 ? load_unaligned<uint64_t>(p + sp) : load_tail(p + sp, rl);
        pk &= 0x00FFFFFFFFFFFFFFull;
        pk |= uint64_t(rl >= 255 ? 255u : rl) << 56;
        reinterpret_cast<uint64_t*>(o.prefix_keys)[k] = pk;
        if (o.fingerprints) o.fingerprints[k] = a.fingerprints[k];
        // compressed bytes to their place
        const uint32_t c0 = a.offsets_out[k], c1 = a.offsets_out[k + 1];
        const uint8_t* src = a.comp + 2 * size_t(start);
        for (uint32_t i = 0; i < c1 - c0; i++) o.fsst[c0 + i] = src[i];
    }
    for (uint32_t k = tid; k < d + 1; k += kBvThreads) {
        const uint32_t pred = uint32_t(o.slope) * k + uint32_t(o.intercept);
        const int32_t res = int32_t(a.offsets_out[k] - pred);
        if (o.offset_bytes == 1) o.residuals[k] = uint8_t(int8_t(res));
        else if (o.offset_bytes == 2) reinterpret_cast<uint16_t*>(o.residuals)[k] = uint16_t(int16_t(res));
        else reinterpret_cast<uint32_t*>(o.residuals)[k] = uint32_t(res);
    }
    if (d && sp) {
        const uint8_t* p0 = a.data + row_start(a.dict_row[0]);
        for (uint32_t i = tid; i < sp; i += kBvThreads) o.shared_prefix[i] = p0[i];
    }
    // ---- inverted row lists: (key, row) pairs of the valid rows sorted, invalid rows last
    if (!o.postings) return;
    uint32_t N = 64;
    while (N < n) N <<= 1;  // n <= kPostLdsRows (the runtime only asks for lists then)
    for (uint32_t i = tid; i < N; i += kBvThreads) {
        uint32_t v = 0xFFFFFFFFu;
        if (i < n && row_valid(a.validity, i)) v = (uint32_t(a.keys[i]) << 16) | i;
        pairs[i] = v;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < N / 2; t += kBvThreads) {
                const uint32_t lo = 2 * t - (t & (stride - 1));  // index with bit `stride` clear
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t x = pairs[lo], y = pairs[hi];
                if ((x > y) == up) { pairs[lo] = y; pairs[hi] = x; }
            }
            __syncthreads();
        }
    }
    uint16_t* off = o.postings;
    uint16_t* rows = o.postings + d + 1;
    for (uint32_t i = tid; i < N; i += kBvThreads) {
        const uint32_t v = pairs[i];
        const bool live = v != 0xFFFFFFFFu;
        if (live) rows[i] = uint16_t(v);
        // list bounds: position i starts the lists of every key in (key of i-1, key of i]; the end closes the rest
        const uint32_t key = live ? (v >> 16) : d;
        const uint32_t prev = i == 0 ? 0xFFFFFFFFu : pairs[i - 1];
        if (prev == 0xFFFFFFFFu && i != 0) continue;  // past the end
        const uint32_t first = i == 0 ? 0 : (prev >> 16) + 1;
        for (uint32_t k = first; k <= key && k <= d; k++) off[k] = uint16_t(i);
    }
    if (tid == 0 && N == n && n && pairs[N - 1] != 0xFFFFFFFFu) {
        // every position is live: nothing closed the lists above the last key
        for (uint32_t k = (pairs[N - 1] >> 16) + 1; k <= d; k++) off[k] = uint16_t(n);
    }
}

}  // namespace

hipError_t launch_bv_build(const BvEncodeDesc* d_descs, uint32_t n_arrays, const DevFsstEncoder* d_encoders, hipStream_t stream) {
    if (n_arrays == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bv_build, dim3(n_arrays), dim3(kBvThreads), 0, stream, d_descs, d_encoders);
    return hipGetLastError();
}

hipError_t launch_bv_pack(const BvEncodeDesc* d_descs, const BvPackDesc* d_packs, uint32_t n_arrays, hipStream_t stream) {
    if (n_arrays == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bv_pack, dim3(n_arrays), dim3(kBvThreads), 0, stream, d_descs, d_packs);
    return hipGetLastError();
}

hipError_t warm_code_object_bv_encode() {  // (see warm_code_object_kernels)
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_bv_pack));
}

}  // namespace lc
