// Dependent-load latency on MI355X: random pointer chase over a buffer, 1 lane per wave, W waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint32_t* __restrict__ next, uint32_t n, int steps, uint64_t* out_cycles, uint32_t* sink) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) != 0) return;
    uint32_t p = (w * 2654435761u) % n;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < steps; i++) p = next[size_t(p) * 16];  // 64-byte stride per element
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    out_cycles[w] = t1 - t0;
    sink[w] = p;
}
int main() {
    for (size_t mb : {64, 1024, 4096}) {
        const size_t n = mb * 1024 * 1024 / 64;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> host(n * 16, 0);
        for (size_t i = 0; i < n; i++) host[size_t(perm[i]) * 16] = perm[(i + 1) % n];
        uint32_t* d; hipMalloc(&d, n * 64); hipMemcpy(d, host.data(), n * 64, hipMemcpyHostToDevice);
        for (int waves : {1, 256, 4096, 8192}) {
            uint64_t* dc; uint32_t* ds; hipMalloc(&dc, waves * 8); hipMalloc(&ds, waves * 4);
            const int steps = 200;
            chase<<<(waves + 3) / 4, 256>>>(d, uint32_t(n), steps, dc, ds);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            chase<<<(waves + 3) / 4, 256>>>(d, uint32_t(n), steps, dc, ds);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("   kernel %.1f us by events; ", ms * 1000);
            std::vector<uint64_t> c(waves); hipMemcpy(c.data(), dc, waves * 8, hipMemcpyDeviceToHost);
            if (waves == 1) {
                uint32_t got = 0; hipMemcpy(&got, ds, 4, hipMemcpyDeviceToHost);
                uint32_t p = 0; for (int i = 0; i < steps; i++) p = host[size_t(p) * 16];
                printf("[check got %u want %u] ", got, p);
            }
            double s = 0; for (auto v : c) s += double(v);
            printf("buffer %5zu MB waves %5d: %.0f ns per dependent load\n", mb, waves, s / waves / steps * 10.0);
            hipFree(dc); hipFree(ds);
        }
        hipFree(d);
    }
    return 0;
}
