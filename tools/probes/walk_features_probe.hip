// What costs the LF walk its window-locality gain?  Starting from the bare windowed chase
// (window_probe.hip), features of the real walk are added one at a time:
//   F_VAR   pieces end at rows that are multiples of 128 (geometric lengths) with wave-pooled tickets
//   F_STORE a dword of packed symbols is stored every 4 steps (+ at piece end)
//   F_INFO  a piece record is stored at piece end
// 256 tables x 2^20 successors (1 GiB), lanes = wgs x 256.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_fill(uint32_t *a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        a[i] = (x & 0xFFFFFu) | (x & 0xFF000000u);              // successor | "symbol" in the top byte
    }
}

template <bool F_STORE, bool F_INFO>
__global__ __launch_bounds__(256) void k_walk_var(const uint32_t *__restrict__ a, uint32_t nb, uint32_t *ctr,
                                                  uint32_t *__restrict__ slots, uint32_t *__restrict__ info, uint32_t *sink)
{
    const uint32_t per_table = 8192, total = nb * per_table;
    const uint32_t l = threadIdx.x & 63;
    uint32_t pool_next = 0, pool_end = 0, acc = 0;
    bool active = false, dead = false;
    uint32_t r = 0, len = 0, w = 0, id = 0;
    const uint32_t *T = a;
    for (;;) {
        const uint64_t needm = __ballot(!active && !dead);
        if (needm) {
            if (pool_next == pool_end) {
                uint32_t base = 0;
                if (l == 0) base = atomicAdd(ctr, 256u);
                pool_next = __builtin_amdgcn_readfirstlane(base);
                pool_end = pool_next + 256u;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(needm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needm, 0));
            const uint32_t ticket = pool_next + rank;
            const bool take = !active && !dead && ticket < pool_end;
            const uint32_t np = pool_next + (uint32_t)__popcll(needm);
            pool_next = np < pool_end ? np : pool_end;
            if (take) {
                if (ticket >= total) dead = true;
                else {
                    const uint32_t b = ticket >> 13, s = ticket & 8191u;
                    T = a + ((size_t)b << 20);
                    id = ticket; r = s * 128u; len = 0; w = 0; active = true;
                }
            }
        }
        if (__ballot(active) == 0) { if (__ballot(!dead) == 0) break; continue; }
        if (active) {
            const uint32_t wv = T[r];
            w |= (wv >> 24) << (8 * (len & 3));
            r = wv & 0xFFFFFu;
            len++;
            const bool end = (r & 127u) == 0 || len == 1024;
            if (F_STORE) { if ((len & 3) == 0 || end) { slots[(size_t)id * 256 + ((len - 1) >> 2)] = w; w = 0; } }
            if (end) { if (F_INFO) info[id] = len | (r << 10); acc ^= w ^ r; active = false; }
        }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// variant 3/4: a ticket is a RUN of RUNLEN consecutive splitters: when a piece ends the lane continues with
// the next splitter of its run (no ballot / pool code), a new ticket only every RUNLEN pieces.
// STORE16: 16 packed symbols per uint4 store (every 16 steps and at piece end).
template <int RUNLEN, bool STORE16>
__global__ __launch_bounds__(256) void k_walk_runs(const uint32_t *__restrict__ a, uint32_t nb, uint32_t *ctr,
                                                   uint4 *__restrict__ slots, uint32_t *__restrict__ info, uint32_t *sink)
{
    const uint32_t per_table = 8192, total = nb * per_table;
    const uint32_t l = threadIdx.x & 63;
    uint32_t pool_next = 0, pool_end = 0, acc = 0;
    bool active = false, dead = false;
    uint32_t r = 0, len = 0, id = 0, left = 0, w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    const uint32_t *T = a;
    for (;;) {
        if (!active && !dead && left) {                        // next splitter of my run: no ticket needed
            id++; left--; r = (id & 8191u) * 128u; len = 0; w0 = w1 = w2 = w3 = 0; active = true;
        }
        const uint64_t needm = __ballot(!active && !dead);
        if (needm) {
            if (pool_next == pool_end) {
                uint32_t base = 0;
                if (l == 0) base = atomicAdd(ctr, 64u * RUNLEN);
                pool_next = __builtin_amdgcn_readfirstlane(base);
                pool_end = pool_next + 64u * RUNLEN;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(needm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needm, 0));
            const uint32_t ticket = pool_next + rank * RUNLEN;
            const bool take = !active && !dead && ticket < pool_end;
            const uint32_t np = pool_next + (uint32_t)__popcll(needm) * RUNLEN;
            pool_next = np < pool_end ? np : pool_end;
            if (take) {
                if (ticket >= total) dead = true;
                else {
                    T = a + ((size_t)(ticket >> 13) << 20);
                    id = ticket; left = RUNLEN - 1; r = (ticket & 8191u) * 128u; len = 0; w0 = w1 = w2 = w3 = 0; active = true;
                }
            }
        }
        if (__ballot(active) == 0) { if (__ballot(!dead) == 0) break; continue; }
        if (active) {
            const uint32_t wv = T[r];
            const uint32_t sh = (wv >> 24) << (8 * (len & 3)), q = (len >> 2) & 3;
            w0 |= (q == 0) ? sh : 0u; w1 |= (q == 1) ? sh : 0u; w2 |= (q == 2) ? sh : 0u; w3 |= (q == 3) ? sh : 0u;
            r = wv & 0xFFFFFu;
            len++;
            const bool end = (r & 127u) == 0 || len == 1024;
            if (STORE16 && ((len & 15) == 0 || end)) { slots[(size_t)id * 64 + ((len - 1) >> 4)] = make_uint4(w0, w1, w2, w3); w0 = w1 = w2 = w3 = 0; }
            if (end) { if (STORE16) info[id] = len | (r << 10); acc ^= w0 ^ r; active = false; }
        }
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main()
{
    const uint32_t NB = 256;
    uint32_t *d, *ctr, *sink, *slots, *info;
    hipMalloc(&d, (size_t)NB << 22); hipMalloc(&ctr, 4); hipMalloc(&sink, 4);
    hipMalloc(&slots, (size_t)NB * 8192 * 1024); hipMalloc(&info, (size_t)NB * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 3; variant < 7; variant++)
        for (int wgs : {256, 512, 1024}) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, (size_t)NB << 20);
            hipMemset(ctr, 0, 4);
            hipEventRecord(e0);
            if (variant == 3) hipLaunchKernelGGL((k_walk_runs<4, false>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, (uint4 *)slots, info, sink);
            if (variant == 4) hipLaunchKernelGGL((k_walk_runs<4, true>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, (uint4 *)slots, info, sink);
            if (variant == 5) hipLaunchKernelGGL((k_walk_runs<8, false>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, (uint4 *)slots, info, sink);
            if (variant == 6) hipLaunchKernelGGL((k_walk_runs<8, true>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, (uint4 *)slots, info, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("variant %d (runs of %d, %s) wgs=%4d : %.3f ms\n", variant, variant < 5 ? 4 : 8,
                   (variant & 1) ? "no stores" : "uint4 stores + piece records", wgs, ms);
        }
    for (int variant = 0; variant < 3; variant++)
        for (int wgs : {256, 512, 1024, 2048}) {
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, (size_t)NB << 20);
            hipMemset(ctr, 0, 4);
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL((k_walk_var<false, false>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, slots, info, sink);
            if (variant == 1) hipLaunchKernelGGL((k_walk_var<true, false>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, slots, info, sink);
            if (variant == 2) hipLaunchKernelGGL((k_walk_var<true, true>), dim3(wgs), dim3(256), 0, 0, d, NB, ctr, slots, info, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("variant %d (%s) wgs=%4d : %.3f ms\n", variant,
                   variant == 0 ? "variable pieces, pooled tickets" : variant == 1 ? "+ dword stores" : "+ piece records", wgs, ms);
        }
    return 0;
}
