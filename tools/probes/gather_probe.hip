// Random 4-byte gather ceiling on MI355X as a function of working-set size and loads in
// flight per lane (what bounds the inverse-BWT LF walk).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int K>
__global__ __launch_bounds__(256) void k_chase(const uint32_t *__restrict__ a, uint32_t mask, uint32_t steps,
                                               uint32_t *__restrict__ sink)
{
    uint32_t r[K];
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) r[k] = (g * 2654435761u + k * 40503u) & mask;
    for (uint32_t i = 0; i < steps; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) r[k] = a[r[k]] & mask;
    }
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < K; k++) x ^= r[k];
    if (x == 0xFFFFFFFFu) sink[0] = x;
}

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

template <int K>
static void run(const uint32_t *d, uint32_t mask, uint32_t *sink, int wgs, const char *tag, size_t mib)
{
    const uint32_t steps = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chase<K>, dim3(wgs), dim3(256), 0, 0, d, mask, 16u, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chase<K>, dim3(wgs), dim3(256), 0, 0, d, mask, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double acc = (double)wgs * 256 * K * steps;
    printf("%s ws=%5zu MiB K=%d wgs=%5d : %.2f ms  %.1f G acc/s  (x64B = %.2f TB/s, x128B = %.2f TB/s)\n", tag, mib, K, wgs,
           ms, acc / ms / 1e6, acc * 64 / ms / 1e9, acc * 128 / ms / 1e9);
}

int main()
{
    const size_t maxw = (size_t)1 << 30;             // 4 GiB of u32
    uint32_t *d; hipMalloc(&d, maxw * 4);
    uint32_t *sink; hipMalloc(&sink, 4);
    // pseudo-random successor table: a[i] = hash(i) (not a permutation; fine for a bandwidth probe)
    std::vector<uint32_t> h((size_t)1 << 24);
    uint32_t s = 12345;
    for (auto &x : h) x = lcg(s) ^ (lcg(s) >> 11);
    for (size_t o = 0; o < maxw; o += h.size()) {
        hipMemcpy(d + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (auto &x : h) x = x * 2246822519u + 374761393u;   // vary per 64 MiB slab
        if (o > ((size_t)1 << 28)) { /* reuse same slab content for speed */ }
    }
    // lanes in flight vs throughput at cache-resident working sets (what a windowed LF walk would see)
    for (size_t mib : {16, 32, 64}) {
        const uint32_t mask = (uint32_t)(mib * 262144 - 1);
        for (int wgs : {256, 512, 1024, 2048}) run<1>(d, mask, sink, wgs, "lanes  ", mib);
    }
    for (size_t mib : {4, 16, 32, 64, 128, 256, 1024, 4096}) {
        const uint32_t mask = (uint32_t)(mib * 262144 - 1);
        run<1>(d, mask, sink, 2048, "occ100%", mib);
        run<4>(d, mask, sink, 2048, "occ100%", mib);
        run<4>(d, mask, sink, 8192, "4waves ", mib);
    }
    return 0;
}
