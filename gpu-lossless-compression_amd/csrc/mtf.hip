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
    uint32_t sym_next = src[hi - 1 - (l < hi - lo ? l : 0u)];
    for (uint32_t o0 = 0; o0 < hi - lo && len < 256; o0 += 64) {
        const uint32_t o = o0 + l;
        const bool valid = o < hi - lo;
        const uint32_t sym = valid ? sym_next : 0u;
        sym_next = src[hi - 1 - (o + 64 < hi - lo ? o + 64 : 0u)];               // in flight during this batch
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
    uint32_t p4n = reinterpret_cast<const uint32_t *>(lists + (size_t)b * max_chunks * 256)[l];
    uint32_t mn = lens[(size_t)b * max_chunks];
    for (uint32_t c = 0; c < nchunks; c++) {
        uint8_t *L = lists + ((size_t)b * max_chunks + c) * 256;
        const uint32_t m = mn;
        // P = chunk-local list (registers), then publish the current state as the start list
        const uint32_t p4 = p4n;                                       // entries 4l..4l+3 of P
        if (c + 1 < nchunks) {                                         // next chunk's list: in flight during this fold
            p4n = reinterpret_cast<const uint32_t *>(L + 256)[l];
            mn = lens[(size_t)b * max_chunks + c + 1];
        }
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
// MTF as dominance counting.  Give every position i of the chunk the index
// P[i] of the previous occurrence of its symbol (virtual times -256..-1 for the
// symbols of the start list: the symbol at list position q "occurred" at -1-q).
// The P values are distinct, and
//     mtf[i] = #{ j in (P[i], i) : P[j] < P[i] }
// (the symbols whose first occurrence after P[i] lies before i).  A wave evaluates
// 64 positions at once:
//   * j in earlier batches:  P[j] marks a "killed" timestamp; a 4352-bit bitmap of
//     killed timestamps + per-word prefix counts in LDS answers
//     #{j < base : P[j] < P[i]} with two LDS reads and a popcount; the j <= P[i]
//     part of it is exactly P[i]+1.
//   * j in the same batch:   T = #{k < lane : P[k] < P[lane]} by a 64-step
//     readlane / compare / add-with-carry loop (3 VALU per step for 64 outputs).
// No per-byte serial dependency chain and almost no scalar-unit work: the previous
// kernel (one list rotation per input byte) was bound by instruction issue.
template <bool WITH_HIST>
__global__ __launch_bounds__(MTF_WAVES * 64) void k_mtf_encode(const uint8_t *__restrict__ in,
                                                              size_t in_stride, uint32_t n,
                                                              const uint8_t *__restrict__ lists,
                                                              uint32_t max_chunks,
                                                              uint8_t *__restrict__ out, size_t out_stride,
                                                              uint32_t *__restrict__ sub_hist)
{
    constexpr int NW = (MTF_CHUNK + 256) / 64;                 // 68 bitmap words
    __shared__ uint32_t s_hist[WITH_HIST ? MTF_WAVES : 1][256];
    __shared__ int s_last[MTF_WAVES][256];
    __shared__ unsigned long long s_bm[MTF_WAVES][NW + 4];
    __shared__ uint32_t s_cum[MTF_WAVES][NW + 4];
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const uint32_t chunk = blockIdx.x * MTF_WAVES + w;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    if (chunk >= nchunks) return;
    const uint32_t lo = chunk * MTF_CHUNK, C = min(n, lo + MTF_CHUNK) - lo;
    const uint8_t *src = in + (size_t)b * in_stride + lo;
    uint8_t *dst = out + (size_t)b * out_stride + lo;
    int *last = s_last[w];
    unsigned long long *bm = s_bm[w];
    uint32_t *cum = s_cum[w];
    {
        const uint32_t lw = reinterpret_cast<const uint32_t *>(lists + ((size_t)b * max_chunks + chunk) * 256)[l];
#pragma unroll
        for (int j = 0; j < 4; j++) last[(lw >> (8 * j)) & 0xFF] = -1 - (int)(4 * l + j);
        bm[l] = 0; cum[l] = 0;
        if (l < NW + 4 - 64) { bm[64 + l] = 0; cum[64 + l] = 0; }
        if (WITH_HIST) for (int i = l; i < 256; i += 64) s_hist[w][i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    const uint64_t lt_mask = (1ull << l) - 1ull;
    uint32_t sym_next = src[l < C ? l : 0u];
    for (uint32_t base = 0; base < C; base += 64) {
        const uint32_t i = base + l;
        const bool valid = i < C;
        const uint32_t sym = valid ? sym_next : 0u;
        sym_next = src[i + 64 < C ? i + 64 : 0u];                                // in flight during this batch
        // lanes of this batch holding the same symbol
        const uint64_t peers = wave_match<8>(sym, __ballot(valid));
        const uint64_t before = peers & lt_mask;
        const bool hasprev = before != 0;
        const int p = 63 - __builtin_clzll(before | 1ull);                       // previous lane with my symbol
        const bool last_in_batch = (peers >> l) == 1ull;
        const int P = hasprev ? (int)base + p : last[sym];
        // T = #{k < l : P[k] < P[l]}
        // the (biased, strictly positive) P values slide up one lane per step (DPP wave_shr:1,
        // lanes with no source read 0), so lane l meets P[l-1], P[l-2], ... P[0]; counting the
        // LARGER ones lets the zero fill drop out: T = #{k < l : P[k] < P[l]} = l - G
        const uint32_t Pb = (uint32_t)(P + 257);
        uint32_t G = 0, slide = Pb;
#pragma unroll
        for (int k = 0; k < 63; k++) {
            slide = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)slide, 0x138, 0xf, 0xf, true);
            G += (slide > Pb) ? 1u : 0u;
        }
        const uint32_t T = l - G;
        uint32_t o;
        if (hasprev) o = T - (uint32_t)(p + 1);
        else {
            const uint32_t bitx = (uint32_t)(P + 256), wd = bitx >> 6, r = bitx & 63;
            const uint32_t kb = cum[wd] + (uint32_t)__popcll(bm[wd] & ((1ull << r) - 1ull));
            o = T + kb - (uint32_t)(P + 1);           // signed: virtual P adds the -1-P start-list symbols ahead of x
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            dst[i] = (uint8_t)o;
            if (WITH_HIST) atomicAdd(&s_hist[w][o], 1u);
            const uint32_t bitx = (uint32_t)(P + 256);
            atomicOr(&bm[bitx >> 6], 1ull << (bitx & 63));                       // timestamp P is killed by i
            if (last_in_batch) last[sym] = (int)i;
        }
        __builtin_amdgcn_wave_barrier();
        // prefix counts of the killed-timestamp bitmap
        {
            const uint32_t c = (uint32_t)__popcll(bm[l]);
            const uint32_t inc = wave_incl_add(c);
            cum[l] = inc - c;
            const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
            if (l == 0) {
                uint32_t run = tot;
                for (int q = 64; q < NW; q++) { cum[q] = run; run += (uint32_t)__popcll(bm[q]); }
            }
        }
        __builtin_amdgcn_wave_barrier();
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
    const double units = (double)n * nblk;
    int pi = s.prof ? s.prof->begin(PROF_MTF_LISTS, st) : -1;
    hipLaunchKernelGGL(k_mtf_chunk_lists, g, t, 0, st, in, in_stride, n, s.lists, s.lens, s.max_chunks);
    hipLaunchKernelGGL(k_mtf_scan_lists, dim3(nblk), dim3(64), 0, st, s.lists, s.lens, n, s.max_chunks);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_MTF_ENCODE, st) : -1;
    if (sub_hist)
        hipLaunchKernelGGL(k_mtf_encode<true>, g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist);
    else
        hipLaunchKernelGGL(k_mtf_encode<false>, g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist);
    if (pi >= 0) s.prof->end(pi, units, st);
    return hipGetLastError();
}

} // namespace glc
