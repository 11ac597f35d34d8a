// mtf.hip -- Move-to-Front transform of many blocks, gfx950 / wave64.
//
// Computes exactly computeMtfGold (cudpp-inpar/apps/cudpp_testrig/test_compress.cpp:93-125),
// i.e. what the reference's four kernels mtf_reduction / mtf_GLreduction /
// mtf_GLdownsweep / mtf_localscan_lists (kernel/compress_kernel.cuh:1339-2023,
// driver app/compress_app.cu:133-223) compute with one thread per 64 bytes and
// a 256-byte list per thread.
//
// MI355X design: the unit of work is a WAVE and a 4096-byte chunk (= one
// Huffman block, so the per-chunk histogram falls out for free):
//   1. k_mtf_chunk_lists  each wave walks its chunk backwards 64 bytes at a
//      time and emits the chunk's "distinct symbols, most recent first" list
//      (LDS atomicMin on a 256-entry recency table + ballot compaction).
//   2. k_mtf_scan_lists   one wave per block folds the chunk lists left to
//      right with the associative operator  S' = P ++ (S \ P)  to get the MTF
//      list at the start of every chunk (wave-wide filter: 4 ballots / fold).
//   3. k_mtf_encode       one wave per chunk keeps the 256-entry list in 64
//      VGPR lanes x 4 packed bytes; per input byte: SWAR zero-byte test +
//      ballot finds the position, one cross-lane shift rotates the prefix.
#include "glc_device.h"
#include "glc_internal.h"

namespace glc {

constexpr int MTF_WAVES = 4;                       // waves per workgroup

// --- 1. chunk-local recency lists ------------------------------------------
__global__ __launch_bounds__(MTF_WAVES * 64) void k_mtf_chunk_lists(const uint8_t *__restrict__ in,
                                                                   size_t in_stride, uint32_t n,
                                                                   uint8_t *__restrict__ lists,
                                                                   uint16_t *__restrict__ lens,
                                                                   uint32_t max_chunks)
{
    __shared__ uint32_t s_first[MTF_WAVES][256];
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const uint32_t chunk = blockIdx.x * MTF_WAVES + w;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    if (chunk >= nchunks) return;                           // whole wave exits together
    const uint32_t lo = chunk * MTF_CHUNK, hi = min(n, lo + MTF_CHUNK);
    const uint8_t *src = in + (size_t)b * in_stride;
    uint8_t *L = lists + ((size_t)b * max_chunks + chunk) * 256;
    uint32_t *first = s_first[w];
    for (int i = l; i < 256; i += 64) first[i] = 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
    uint32_t len = 0;
    // order index o = hi-1-p : 0 is the most recent byte of the chunk
    for (uint32_t o0 = 0; o0 < hi - lo && len < 256; o0 += 64) {
        const uint32_t o = o0 + l;
        const bool valid = o < hi - lo;
        const uint32_t sym = valid ? src[hi - 1 - o] : 0u;
        if (valid) atomicMin(&first[sym], o);
        __builtin_amdgcn_wave_barrier();
        const bool isnew = valid && first[sym] == o;        // most recent occurrence of sym in the chunk
        const uint64_t bal = __ballot(isnew);
        if (isnew) L[len + mbcnt(bal)] = (uint8_t)sym;
        len += (uint32_t)__popcll(bal);
        __builtin_amdgcn_wave_barrier();
    }
    if (l == 0) lens[(size_t)b * max_chunks + chunk] = (uint16_t)len;
}

// --- 2. exclusive scan of the lists (in place: lists[c] becomes the MTF list
//        in force at the start of chunk c) -----------------------------------
__global__ __launch_bounds__(64) void k_mtf_scan_lists(uint8_t *__restrict__ lists,
                                                       const uint16_t *__restrict__ lens, uint32_t n,
                                                       uint32_t max_chunks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_state[2][256];
    __shared__ __attribute__((aligned(16))) uint8_t s_inp[256];
    const uint32_t b = blockIdx.x, l = threadIdx.x;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    int cur = 0;
    for (int i = l; i < 256; i += 64) s_state[0][i] = (uint8_t)i;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t c = 0; c < nchunks; c++) {
        uint8_t *L = lists + ((size_t)b * max_chunks + c) * 256;
        const uint32_t m = lens[(size_t)b * max_chunks + c];
        // P = chunk-local list (registers), then publish the current state as the start list
        uint32_t p4 = reinterpret_cast<const uint32_t *>(L)[l];       // entries 4l..4l+3 of P
        reinterpret_cast<uint32_t *>(L)[l] = reinterpret_cast<const uint32_t *>(s_state[cur])[l];
        if (c + 1 == nchunks) break;
        // membership table of P
        reinterpret_cast<uint32_t *>(s_inp)[l] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t e = 4 * l + j;
            if (e < m) { const uint8_t sy = (uint8_t)(p4 >> (8 * j)); s_inp[sy] = 1; s_state[cur ^ 1][e] = sy; }
        }
        __builtin_amdgcn_wave_barrier();
        // append the survivors of the old state in order
        uint32_t base = m;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint8_t sy = s_state[cur][r * 64 + l];
            const bool keep = !s_inp[sy];
            const uint64_t bal = __ballot(keep);
            if (keep) s_state[cur ^ 1][base + mbcnt(bal)] = sy;
            base += (uint32_t)__popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
    }
}

// --- 3. encode ---------------------------------------------------------------
// list entry e lives in lane e/4, byte e%4 of `v`.
template <bool WITH_HIST>
__global__ __launch_bounds__(MTF_WAVES * 64) void k_mtf_encode(const uint8_t *__restrict__ in,
                                                              size_t in_stride, uint32_t n,
                                                              const uint8_t *__restrict__ lists,
                                                              uint32_t max_chunks,
                                                              uint8_t *__restrict__ out, size_t out_stride,
                                                              uint32_t *__restrict__ sub_hist)
{
    __shared__ uint32_t s_hist[WITH_HIST ? MTF_WAVES : 1][256];
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const uint32_t chunk = blockIdx.x * MTF_WAVES + w;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    if (chunk >= nchunks) return;
    const uint32_t lo = chunk * MTF_CHUNK, hi = min(n, lo + MTF_CHUNK);
    const uint8_t *src = in + (size_t)b * in_stride;
    uint8_t *dst = out + (size_t)b * out_stride;
    uint32_t v = reinterpret_cast<const uint32_t *>(lists + ((size_t)b * max_chunks + chunk) * 256)[l];
    if (WITH_HIST) {
        for (int i = l; i < 256; i += 64) s_hist[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t p0 = lo; p0 < hi; p0 += 64) {
        const uint32_t cntv = min(64u, hi - p0);
        const uint32_t inb = (p0 + l < hi) ? src[p0 + l] : 0u;
        uint32_t outb = 0;
        for (uint32_t j = 0; j < cntv; j++) {
            const uint32_t x = __builtin_amdgcn_readlane(inb, j);            // uniform
            const uint32_t z = v ^ (x * 0x01010101u);
            const uint32_t hz = (z - 0x01010101u) & ~z & 0x80808080u;         // 0x80 in every zero byte
            const uint64_t bal = __ballot(hz != 0);
            const uint32_t L = (uint32_t)__builtin_ctzll(bal);                // lane holding x
            const uint32_t hzl = __builtin_amdgcn_readlane(hz, L);
            const uint32_t bidx = (uint32_t)__builtin_ctz(hzl) >> 3;          // byte within that lane
            const uint32_t pos = 4 * L + bidx;
            outb = (l == j) ? pos : outb;
            // entries [0, pos) move up by one, x goes to the front.  Branch-free: for
            // pos == 0 the masked merge rewrites byte 0 of lane 0 with x itself.
            // carry-in = top byte of the previous lane (DPP wave_shr:1, a VALU op -- no LDS
            // round trip on the per-byte critical path); lane 0 keeps `old` = x.
            const uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)(v >> 24), 0x138, 0xf, 0xf, false);
            const uint32_t shifted = (v << 8) | carry;
            const uint32_t m2 = (bidx == 3) ? 0xFFFFFFFFu : ((1u << (8 * (bidx + 1))) - 1u);   // uniform
            const uint32_t mask = (l < L) ? 0xFFFFFFFFu : ((l == L) ? m2 : 0u);
            v = (v & ~mask) | (shifted & mask);
        }
        if (p0 + l < hi) {
            dst[p0 + l] = (uint8_t)outb;
            if (WITH_HIST) atomicAdd(&s_hist[w][outb], 1u);     // once per 64 bytes, off the serial chain
        }
    }
    if (WITH_HIST) {
        __builtin_amdgcn_wave_barrier();
        uint32_t *H = sub_hist + ((size_t)b * max_chunks + chunk) * 256;
        for (int i = l; i < 256; i += 64) H[i] = s_hist[w][i];
    }
}

// ---------------------------------------------------------------------------
#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t mtf_scratch_alloc(MtfScratch &s, uint32_t nmax, uint32_t rows)
{
    s.nmax = nmax; s.rows = rows; s.max_chunks = (nmax + MTF_CHUNK - 1) / MTF_CHUNK;
    size_t a = (size_t)rows * s.max_chunks * 256, c = (size_t)rows * s.max_chunks * sizeof(uint16_t);
    GLC_TRY(hipMalloc((void **)&s.lists, a));
    GLC_TRY(hipMalloc((void **)&s.lens, c));
    s.bytes = a + c;
    return hipSuccess;
}

void mtf_scratch_free(MtfScratch &s)
{
    if (s.lists) (void)hipFree(s.lists);
    if (s.lens) (void)hipFree(s.lens);
    s = MtfScratch();
}

hipError_t mtf_forward(hipStream_t st, const uint8_t *in, size_t in_stride, uint32_t n, uint32_t nblk,
                       uint8_t *out, size_t out_stride, MtfScratch &s, uint32_t *sub_hist)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    dim3 g((nchunks + MTF_WAVES - 1) / MTF_WAVES, nblk), t(MTF_WAVES * 64);
    hipLaunchKernelGGL(k_mtf_chunk_lists, g, t, 0, st, in, in_stride, n, s.lists, s.lens, s.max_chunks);
    hipLaunchKernelGGL(k_mtf_scan_lists, dim3(nblk), dim3(64), 0, st, s.lists, s.lens, n, s.max_chunks);
    if (sub_hist)
        hipLaunchKernelGGL(k_mtf_encode<true>, g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist);
    else
        hipLaunchKernelGGL(k_mtf_encode<false>, g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist);
    return hipGetLastError();
}

} // namespace glc
