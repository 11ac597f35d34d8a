// Does the dispatcher MIX the workgroups of two oversubscribed grids queued on two streams, or does the first grid keep the CUs
// until it runs out of workgroups?  Two launches of the same kernel (40 KB of LDS per workgroup: four per CU; every workgroup
// spins `us` microseconds on s_memrealtime and records when it started), grid A on stream 1, grid B on stream 2 right behind it.
//   mix_probe [rounds = 8] [us = 100]      prints when B's workgroups started relative to A's (quantiles, in units of A's duration)
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/mix_probe tools/probes/mix_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

__global__ __launch_bounds__(256) void k_spin(unsigned long long *start, unsigned long long *end, uint32_t ticks)
{
    __shared__ uint32_t pad[40 * 1024 / 4];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    pad[threadIdx.x] = (uint32_t)t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) end[blockIdx.x] = __builtin_amdgcn_s_memrealtime() + (pad[1] & 0u);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 8, us = argc > 2 ? atoi(argv[2]) : 100;
    const uint32_t nwg = 256 * 4 * rounds;
    unsigned long long *d[4];
    for (auto &p : d) hipMalloc((void **)&p, nwg * 8);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_spin, dim3(nwg), dim3(256), 0, s1, d[0], d[1], (uint32_t)(us * 100));
        hipLaunchKernelGGL(k_spin, dim3(nwg), dim3(256), 0, s2, d[2], d[3], (uint32_t)(us * 100));
        hipDeviceSynchronize();
        std::vector<unsigned long long> a0(nwg), a1(nwg), b0(nwg), b1(nwg);
        hipMemcpy(a0.data(), d[0], nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(a1.data(), d[1], nwg * 8, hipMemcpyDeviceToHost);
        hipMemcpy(b0.data(), d[2], nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(b1.data(), d[3], nwg * 8, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(a0.begin(), a0.end()), ta = *std::max_element(a1.begin(), a1.end());
        const unsigned long long tb = *std::max_element(b1.begin(), b1.end());
        std::sort(b0.begin(), b0.end()); std::sort(a0.begin(), a0.end());
        auto rel = [&](unsigned long long t) { return (double)((long long)t - (long long)t0) / (double)(ta - t0); };
        printf("A: %u workgroups, %.1f us in all; both done after %.1f us.  A's starts (fraction of A's duration): median %.2f, last %.2f;  "
               "B's starts: first %.2f, 10 %% %.2f, median %.2f, 90 %% %.2f\n", nwg, (ta - t0) / 100.0, (tb - t0) / 100.0,
               rel(a0[nwg / 2]), rel(a0[nwg - 1]), rel(b0[0]), rel(b0[nwg / 10]), rel(b0[nwg / 2]), rel(b0[nwg * 9 / 10]));
    }
    return 0;
}
