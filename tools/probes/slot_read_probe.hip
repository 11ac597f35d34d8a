// How fast can a kernel read "the first `used` words of every `stride`-word slot"?  (k_fs_sort's pattern: 512 slots of
// 4096 eight-byte words per block, about half of each in use, one workgroup per slot.)
//   slot_read_probe            prints GB/s for: stride 4096 / 2048 (dense) / 4096+pad words, 8- and 16-byte loads per lane,
//                              one slot per workgroup and 8 slots per workgroup with the next slot's loads issued early
// hipcc --offload-arch=gfx950 -O3 -o tools/probes/slot_read_probe tools/probes/slot_read_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int VEC, int G, bool PRE>
__global__ __launch_bounds__(512) void k_read(const uint64_t *__restrict__ a, size_t stride, uint32_t used, uint64_t *__restrict__ out)
{
    const uint32_t tid = threadIdx.x;
    const size_t slot0 = (size_t)blockIdx.x * G;
    uint64_t acc = 0;
    constexpr int R = 4 / VEC;                                  // rounds: used = 2048 words = 512 threads x 4 words
    uint64_t w[4], wn[4];
    auto fetch = [&](size_t s, uint64_t *d) {
        const uint64_t *p = a + s * stride;
        if (VEC == 1) { for (int r = 0; r < 4; r++) d[r] = p[r * 512 + tid]; }
        else { for (int r = 0; r < 2; r++) { const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(p)[r * 512 + tid]; d[2 * r] = v.x; d[2 * r + 1] = v.y; } }
    };
    if (PRE) fetch(slot0, wn);
    for (int j = 0; j < G; j++) {
        if (PRE) { for (int r = 0; r < 4; r++) w[r] = wn[r]; if (j + 1 < G) fetch(slot0 + j + 1, wn); }
        else fetch(slot0 + j, w);
        for (int r = 0; r < 4; r++) acc += w[r] * (r + 3);
        if (PRE) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else __syncthreads();
    }
    if (acc == 0x1234567) out[blockIdx.x] = acc;
    (void)used; (void)R;
}

int main()
{
    const size_t nslot = 512 * 256;                            // 256 blocks
    const size_t maxstride = 4096 + 512;
    uint64_t *a, *out;
    if (hipMalloc(&a, nslot * maxstride * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, nslot * 8);
    hipMemset(a, 1, nslot * maxstride * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t strides[] = {4096, 2048, 4096 + 32, 4096 + 128, 4096 + 512, 3072};
    for (size_t st : strides) {
        auto run = [&](const char *name, auto kern, int G) {
            float best = 1e9;
            for (int it = 0; it < 4; it++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(nslot / G), dim3(512), 0, 0, a, st, 2048u, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            printf("stride %5zu words  %-34s %7.3f ms  %7.1f GB/s\n", st, name, best, nslot * 2048.0 * 8 / best / 1e6);
        };
        run("8 B/lane, 1 slot/WG", k_read<1, 1, false>, 1);
        run("16 B/lane, 1 slot/WG", k_read<2, 1, false>, 1);
        run("8 B/lane, 8 slots/WG", k_read<1, 8, false>, 8);
        run("8 B/lane, 8 slots/WG, prefetch", k_read<1, 8, true>, 8);
        run("16 B/lane, 8 slots/WG, prefetch", k_read<2, 8, true>, 8);
    }
    return 0;
}
