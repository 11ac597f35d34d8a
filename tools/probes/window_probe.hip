// Does a sliding window over many 4 MiB tables behave like a small warm working set?
// NB tables of 2^20 u32 successors (random within the table, written by a GPU kernel just before,
// like the LF tables of the inverse BWT).  Lanes take (table, start) tickets in table-major order
// and chase STEPS successors per ticket.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_fill(uint32_t *a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        a[i] = x & 0xFFFFFu;
    }
}

__global__ __launch_bounds__(256) void k_walk(const uint32_t *__restrict__ a, uint32_t nb, uint32_t per_table,
                                              uint32_t steps, uint32_t *ctr, uint32_t *sink)
{
    const uint32_t total = nb * per_table;
    uint32_t acc = 0;
    for (;;) {
        uint32_t t = 0;
        const uint64_t m = __ballot(1);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (rank == 0) t = atomicAdd(ctr, 64u);
        t = __builtin_amdgcn_readfirstlane(t) + rank;
        if (t >= total) break;
        const uint32_t b = t / per_table, s = t - b * per_table;
        const uint32_t *T = a + ((size_t)b << 20);
        uint32_t r = (s * (0x100000u / per_table)) & 0xFFFFFu;
        for (uint32_t i = 0; i < steps; i++) r = T[r];
        acc ^= r;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main()
{
    const uint32_t NB = 256;
    uint32_t *d, *ctr, *sink;
    hipMalloc(&d, (size_t)NB << 22); hipMalloc(&ctr, 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // pieces per table x steps per piece = 2^20: (8192 x 128) is the shipped splitter spacing, (65536 x 16)
    // the 16-row spacing that keeps only ~8 tables live at full occupancy
    for (uint32_t per_table : {8192u, 65536u})
    for (uint32_t nb : {256u}) {
        for (int wgs : {256, 512, 1024, 2048, 4096}) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, (size_t)NB << 20);   // cold: just written
            hipMemset(ctr, 0, 4);
            const uint32_t steps = (1u << 20) / per_table;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_walk, dim3(wgs), dim3(256), 0, 0, d, nb, per_table, steps, ctr, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("pieces/table=%5u tables=%3u (%4u MiB) wgs=%4d : %.3f ms  %.1f G steps/s\n", per_table, nb, nb * 4, wgs, ms,
                   (double)nb * per_table * steps / ms / 1e6);
        }
    }
    return 0;
}
