// Bench / profiling / test aids (include/liquid_cache_amd_bench.h).  Built into libliquid_cache_amd_bench.so, NOT into the
// product library: the synthetic column generators (lc_synth.cpp), the PMC counter calibration kernels and the host-side
// row-list builder aid live here and reach the product only through its public C ABI.
#include <cstdint>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/liquid_cache_amd.h"
#include "../../include/liquid_cache_amd_bench.h"
#include "lc_host.hpp"
#include "lc_kernels.hpp"

namespace {

template <typename T>
__device__ __forceinline__ T load_unaligned(const uint8_t* p) {
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) uint8_t*)p, sizeof(T));
    return v;
}


// Counter calibration (scripts/pmc_calibrate.py): read a KNOWN number of bytes with a given access shape, so that
// rocprofv3's FETCH_SIZE can be converted into bytes for that shape instead of assuming the factor of another one.
//   k_calib_read<W>          every lane reads W consecutive bytes, lanes adjacent (coalesced): W = 4, 8, 16
//   k_calib_read_scattered8  every lane reads 8 unaligned bytes out of its own 64-byte sector (the FSST word loads and
//                            offset-pair loads of k_str_pred look like this): 64 useful... 8 useful bytes per sector
template <int W>
__global__ __launch_bounds__(256) void k_calib_read(const uint8_t* __restrict__ src, uint64_t bytes, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t n = bytes / W;
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        if constexpr (W == 4) acc ^= reinterpret_cast<const uint32_t*>(src)[i];
        else if constexpr (W == 8) { const uint2 v = reinterpret_cast<const uint2*>(src)[i]; acc ^= v.x ^ v.y; }
        else { const uint4 v = reinterpret_cast<const uint4*>(src)[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_calib_read_scattered8(const uint8_t* __restrict__ src, uint64_t bytes,
                                                                uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t n = bytes / 64;  // one access per 64-byte sector
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += uint64_t(gridDim.x) * 256) {
        const uint64_t v = load_unaligned<uint64_t>(src + i * 64 + 3 + (i % 7) * 7);
        acc ^= uint32_t(v) ^ uint32_t(v >> 32);
    }
    if (acc == 0x9E3779B9u) sink[blockIdx.x] = acc;
}

// Streaming-read probe (lc_probe_stream_read): what a kernel that does nothing but read `bytes` once achieves on this device, so
// that the scan kernels' roofline fractions can be read against the MEASURED ceiling for their byte count, hot and L3-cold.
// Round 6 (scripts/micro/stream_probe.hip, profiles/r6/stream_probe.txt): 16 bytes per lane, EIGHT independent loads in flight
// per lane, NON-TEMPORAL (the data is read once) — 788 MB in 114-117 us = 6.7-6.9 TB/s from a grid of one or two workgroups per
// CU; the round-5 probe (four default-policy loads) stopped at 5.1-5.5 TB/s, below what the scan kernels themselves reached.
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
// The Infinity-Cache flush between "cold" launches reads its scratch with the DEFAULT policy: the lines must be allocated in
// the cache to displace what is there (a non-temporal stream need not be).
__global__ __launch_bounds__(256) void k_flush_read(const uint4* __restrict__ src, uint64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_probe_read(const probe_u32x4* __restrict__ src, uint64_t n16, uint32_t* __restrict__ sink) {
    constexpr int U = 8;
    uint32_t acc = 0;
    const uint64_t stride = uint64_t(gridDim.x) * 256;
    uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        probe_u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < U; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n16; i += stride) { const probe_u32x4 a = __builtin_nontemporal_load(src + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9E3779B9u) sink[blockIdx.x] = acc;
}

}  // namespace

extern "C" {

int32_t lc_probe_stream_read(void* ctx_, uint64_t bytes, int32_t iters, int32_t grid_blocks, double* out_hot_us, double* out_cold_us) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    if (!ctx || bytes < 4096 || iters <= 0 || grid_blocks <= 0 || !out_hot_us || !out_cold_us) return LC_ERR_INVALID;
    lc_device_info info;
    if (lc_device_info_get(ctx, &info) != LC_OK || info.device_id < 0) return LC_ERR_DEVICE;
    if (hipSetDevice(info.device_id) != hipSuccess) return LC_ERR_DEVICE;
    const uint64_t flush_bytes = uint64_t(1) << 30;  // 4 x the 256 MiB Infinity Cache
    uint8_t* d = nullptr;
    uint8_t* f = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), bytes + 65536) != hipSuccess) return LC_ERR_OOM;
    if (hipMalloc(reinterpret_cast<void**>(&f), flush_bytes + 65536) != hipSuccess) { (void)hipFree(d); return LC_ERR_OOM; }
    int32_t rc = LC_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemset(d, 1, bytes + 65536) != hipSuccess || hipMemset(f, 2, flush_bytes + 65536) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        rc = LC_ERR_DEVICE;
    uint32_t* sink = reinterpret_cast<uint32_t*>(d + bytes);
    uint32_t* fsink = reinterpret_cast<uint32_t*>(f + flush_bytes);
    const dim3 grid{uint32_t(grid_blocks)}, block(256);
    double hot = 0, cold = 0;
    for (int pass = 0; pass < 2 && rc == LC_OK; pass++) {  // 0: hot (back to back), 1: cold (flush before every launch)
        double sum = 0;
        for (int i = -2; i < iters && rc == LC_OK; i++) {
            if (pass == 1) hipLaunchKernelGGL(k_flush_read, dim3(4096), block, 0, nullptr, reinterpret_cast<const uint4*>(f), flush_bytes / 16, fsink);
            (void)hipEventRecord(e0, nullptr);
            hipLaunchKernelGGL(k_probe_read, grid, block, 0, nullptr, reinterpret_cast<const probe_u32x4*>(d), bytes / 16, sink);
            (void)hipEventRecord(e1, nullptr);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = LC_ERR_DEVICE; break; }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (i >= 0) sum += double(ms) * 1000.0;
        }
        (pass == 0 ? hot : cold) = sum / iters;
    }
    *out_hot_us = hot;
    *out_cold_us = cold;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    (void)hipFree(f);
    return rc;
}

// Kernel time of lc_scan_eval over a scan, HIP events on the stream the kernels run on — back to back (hot), or with the
// memory-side Infinity Cache flushed before every launch (two read passes of k_probe_read over `flush_bytes` of scratch:
// the cache ends up holding clean scratch lines only).  Through the PUBLIC scan API only: bench infrastructure, not part of
// the product library.
int32_t lc_bench_gather_bytes_hits_timed(void* ctx_, void* scan_, const void* d_hits, const void* d_n_hits, uint64_t capacity_rows,
                                         void* d_views, void* d_data, uint64_t capacity_bytes, void* d_n_bytes, uint32_t flags,
                                         void* stream, int32_t iters, float* out_avg_ms) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    lc_scan* scan = static_cast<lc_scan*>(scan_);
    if (!ctx || !scan || !out_avg_ms || iters <= 0) return LC_ERR_INVALID;
    lc_device_info info;
    if (lc_device_info_get(ctx, &info) != LC_OK || info.device_id < 0) return LC_ERR_DEVICE;
    if (hipSetDevice(info.device_id) != hipSuccess) return LC_ERR_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return LC_ERR_DEVICE;
    int32_t rc = LC_OK;
    if (flags & LC_HITS_COUNTERS_ZEROED) rc = lc_device_memset(ctx, d_n_bytes, 0, 8, st);
    if (rc == LC_OK && hipEventRecord(a, st) != hipSuccess) rc = LC_ERR_DEVICE;
    for (int i = 0; i < iters && rc == LC_OK; i++)
        rc = lc_scan_gather_bytes_hits(ctx, scan, d_hits, d_n_hits, capacity_rows, d_views, nullptr, d_data, capacity_bytes, d_n_bytes,
                                       flags, st);
    float ms = 0;
    if (rc == LC_OK && (hipEventRecord(b, st) != hipSuccess || hipEventSynchronize(b) != hipSuccess ||
                        hipEventElapsedTime(&ms, a, b) != hipSuccess))
        rc = LC_ERR_DEVICE;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *out_avg_ms = ms / float(iters);
    return rc;
}

int32_t lc_bench_eval_timed(void* ctx_, void* scan_, const void* pred_, const void* d_selection, void* d_mask_out,
                            void* d_counts_out, void* stream, int32_t iters, uint64_t flush_bytes, float* out_avg_ms) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    lc_scan* scan = static_cast<lc_scan*>(scan_);
    const lc_predicate* pred = static_cast<const lc_predicate*>(pred_);
    if (!ctx || !scan || !pred || !out_avg_ms || iters <= 0) return LC_ERR_INVALID;
    lc_device_info info;
    if (lc_device_info_get(ctx, &info) != LC_OK || info.device_id < 0) return LC_ERR_DEVICE;
    if (hipSetDevice(info.device_id) != hipSuccess) return LC_ERR_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint8_t* d_flush = nullptr;
    if (flush_bytes) {
        flush_bytes = std::max<uint64_t>(flush_bytes, 1 << 20);
        if (hipMalloc(reinterpret_cast<void**>(&d_flush), flush_bytes + 65536) != hipSuccess) return LC_ERR_OOM;
        if (hipMemset(d_flush, 1, flush_bytes + 65536) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipFree(d_flush);
            return LC_ERR_DEVICE;
        }
    }
    hipEvent_t a = nullptr, b = nullptr;
    int32_t rc = LC_OK;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) rc = LC_ERR_DEVICE;
    double total = 0;
    if (rc == LC_OK && !d_flush) {
        // back to back: one pair of events around all launches
        if (hipEventRecord(a, st) != hipSuccess) rc = LC_ERR_DEVICE;
        for (int i = 0; i < iters && rc == LC_OK; i++) rc = lc_scan_eval(ctx, scan, pred, d_selection, d_mask_out, d_counts_out, st);
        float ms = 0;
        if (rc == LC_OK && (hipEventRecord(b, st) != hipSuccess || hipEventSynchronize(b) != hipSuccess ||
                            hipEventElapsedTime(&ms, a, b) != hipSuccess))
            rc = LC_ERR_DEVICE;
        total = ms;
    }
    for (int i = 0; i < iters && rc == LC_OK && d_flush; i++) {
        for (int pass = 0; pass < 2; pass++)
            hipLaunchKernelGGL(k_flush_read, dim3(2048), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_flush), flush_bytes / 16,
                               reinterpret_cast<uint32_t*>(d_flush + flush_bytes));
        if (hipGetLastError() != hipSuccess || hipEventRecord(a, st) != hipSuccess) { rc = LC_ERR_DEVICE; break; }
        rc = lc_scan_eval(ctx, scan, pred, d_selection, d_mask_out, d_counts_out, st);
        float ms = 0;
        if (rc == LC_OK && (hipEventRecord(b, st) != hipSuccess || hipEventSynchronize(b) != hipSuccess ||
                            hipEventElapsedTime(&ms, a, b) != hipSuccess))
            rc = LC_ERR_DEVICE;
        total += ms;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipStreamSynchronize(st);
    if (d_flush) (void)hipFree(d_flush);
    if (rc == LC_OK) *out_avg_ms = float(total / iters);
    return rc;
}

int32_t lc_calibrate_read(void* ctx_, uint64_t bytes, int32_t shape, int32_t iters) {
    lc_ctx* ctx = static_cast<lc_ctx*>(ctx_);
    if (!ctx || bytes < 4096 || iters <= 0) return LC_ERR_INVALID;
    lc_device_info info;
    if (lc_device_info_get(ctx, &info) != LC_OK || info.device_id < 0) return LC_ERR_DEVICE;
    if (hipSetDevice(info.device_id) != hipSuccess) return LC_ERR_DEVICE;
    void* d = nullptr;
    if (hipMalloc(&d, bytes + 8192) != hipSuccess) return LC_ERR_OOM;
    int32_t rc = LC_OK;
    if (hipMemset(d, 1, bytes + 8192) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = LC_ERR_DEVICE;
    const uint8_t* p = static_cast<const uint8_t*>(d);
    uint32_t* sink = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(d) + bytes);
    const dim3 grid(2048), block(256);
    for (int i = 0; i < iters && rc == LC_OK; i++) {
        switch (shape) {
            case 4: hipLaunchKernelGGL(k_calib_read<4>, grid, block, 0, nullptr, p, bytes, sink); break;
            case 8: hipLaunchKernelGGL(k_calib_read<8>, grid, block, 0, nullptr, p, bytes, sink); break;
            case 16: hipLaunchKernelGGL(k_calib_read<16>, grid, block, 0, nullptr, p, bytes, sink); break;
            case 1008: hipLaunchKernelGGL(k_calib_read_scattered8, grid, block, 0, nullptr, p, bytes, sink); break;
            default: rc = LC_ERR_INVALID;  // unknown access shape (4, 8, 16 or 1008)
        }
        if (rc == LC_OK && hipGetLastError() != hipSuccess) rc = LC_ERR_DEVICE;
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(d);
    return rc;
}

size_t lc_debug_row_lists(const uint16_t* keys, const uint8_t* validity, uint32_t n, uint32_t d, uint16_t* out, size_t cap) {
    if (!keys || !out || n > lc::kPostMaxRows || d == 0) return 0;
    const std::vector<uint16_t> post = lc::build_row_lists(keys, validity, n, d);
    if (post.size() > cap) return 0;
    std::memcpy(out, post.data(), post.size() * 2);
    return post.size();
}

}  // extern "C"
