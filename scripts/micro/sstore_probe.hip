// Do scalar stores work on gfx950, and what do they cost?  (round 6: a v_cmp's ballot lives in an SGPR pair; the narrow-integer
// kernels spend half of their VALU instructions moving such pairs into lanes with v_writelane so that a vector store can write
// them.)  Every wave produces 64 dwords in SGPRs per "pass" and writes them (a) with s_store_dword, (b) via v_writelane + one
// global_store_dwordx2 per lane; the outputs are compared and both forms timed.
//   hipcc --offload-arch=gfx950 -O3 -o sstore_probe scripts/micro/sstore_probe.hip && ./sstore_probe
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

extern "C" __device__ int __llvm_amdgcn_writelane_i32(int, int, int) __asm("llvm.amdgcn.writelane.i32");

template <int R>
__device__ __forceinline__ void sstore_step(uint32_t t, uint32_t bound, uint32_t* out_wave) {
    const uint64_t b = __ballot(t + uint32_t(R) * 0x9E3779B9u <= bound);
    // dword 2R and 2R+1 of the wave's 64-dword block
    asm volatile("s_nop 4\n\ts_store_dword %0, %2, %3\n\ts_store_dword %1, %2, %4"
                 :: "s"(uint32_t(b)), "s"(uint32_t(b >> 32)), "s"(out_wave), "n"(R * 8), "n"(R * 8 + 4) : "memory");
}
template <int... RS>
__device__ __forceinline__ void sstore_steps(std::integer_sequence<int, RS...>, uint32_t t, uint32_t bound, uint32_t* out_wave) {
    (sstore_step<RS>(t, bound, out_wave), ...);
}

__global__ __launch_bounds__(256) void k_sstore(const uint32_t* __restrict__ in, uint32_t bound, uint32_t* __restrict__ out, uint32_t passes) {
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t t = in[(size_t(wave) * passes + p) * 64u + lane];
        uint32_t* ow = reinterpret_cast<uint32_t*>(uintptr_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(out + (size_t(wave) * passes + p) * 64u))))) |
                                                   (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(out + (size_t(wave) * passes + p) * 64u) >> 32))))) << 32));
        sstore_steps(std::make_integer_sequence<int, 32>{}, t, bound, ow);
    }
    asm volatile("s_dcache_wb" ::: "memory");
}

template <int R>
__device__ __forceinline__ void wl_step(uint32_t t, uint32_t bound, uint32_t& X, uint32_t& Y) {
    const uint64_t b = __ballot(t + uint32_t(R) * 0x9E3779B9u <= bound);
    X = uint32_t(__llvm_amdgcn_writelane_i32(int(uint32_t(b)), R, int(X)));        // lane R: dword 2R
    Y = uint32_t(__llvm_amdgcn_writelane_i32(int(uint32_t(b >> 32)), R, int(Y)));  //         dword 2R + 1
}
template <int... RS>
__device__ __forceinline__ void wl_steps(std::integer_sequence<int, RS...>, uint32_t t, uint32_t bound, uint32_t& X, uint32_t& Y) {
    (wl_step<RS>(t, bound, X, Y), ...);
}
__global__ __launch_bounds__(256) void k_writelane(const uint32_t* __restrict__ in, uint32_t bound, uint32_t* __restrict__ out, uint32_t passes) {
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t t = in[(size_t(wave) * passes + p) * 64u + lane];
        uint32_t X = 0, Y = 0;
        wl_steps(std::make_integer_sequence<int, 32>{}, t, bound, X, Y);
        if (lane < 32) reinterpret_cast<uint2*>(out + (size_t(wave) * passes + p) * 64u)[lane] = make_uint2(X, Y);
    }
}

int main() {
    const uint32_t grid = 2048, passes = 64;  // 8192 waves x 64 passes x 64 dwords = 128 MB of output
    const size_t n = size_t(grid) * 4 * passes * 64;
    uint32_t *d_in, *d_a, *d_b;
    CK(hipMalloc(&d_in, n * 4));
    CK(hipMalloc(&d_a, n * 4));
    CK(hipMalloc(&d_b, n * 4));
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; i++) h[i] = uint32_t(i * 2654435761u) ^ uint32_t(i >> 7);
    CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_a, 0xEE, n * 4));
    CK(hipMemset(d_b, 0xDD, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_s = 0, ms_w = 0;
    for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_sstore, dim3(grid), dim3(256), 0, nullptr, d_in, 0x7FFFFFFFu, d_a, passes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_s, e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_writelane, dim3(grid), dim3(256), 0, nullptr, d_in, 0x7FFFFFFFu, d_b, passes);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_w, e0, e1));
    }
    std::vector<uint32_t> a(n), b(n);
    CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += a[i] != b[i];
    std::printf("scalar stores %s: %zu of %zu dwords differ; s_store_dword %.1f us, v_writelane + vector store %.1f us (%.1f MB out)\n",
                bad ? "WRONG" : "ok", bad, n, ms_s * 1e3, ms_w * 1e3, n * 4 / 1e6);
    return bad ? 1 : 0;
}
