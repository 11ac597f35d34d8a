// huffman.hip -- histogram merge, Huffman tree + codes, block offsets and the
// bit packer for the cudppCompress stream.  gfx950 / wave64.
//
// Bit-exact replacement of (cudpp-inpar/src/cudpp):
//   huffman_build_histogram_kernel   kernel/compress_kernel.cuh:2037-2121
//   huffman_build_tree_kernel        kernel/compress_kernel.cuh:2199-2512
//     + FindMinimumCount             cta/compress_cta.cuh:550-571
//   huffman_kernel_en                kernel/compress_kernel.cuh:2524-2708
//   huffman_datapack_kernel          kernel/compress_kernel.cuh:2716-2750
//   host glue huffmanEncoding        app/compress_app.cu:65-117
//
// What must be reproduced exactly is the TREE SHAPE: leaves are the present
// symbols (+EOF with count 1) in ascending order in slots 0..n-1; repeatedly the
// two minima by (count, level, slot) are merged, the first minimum is relocated
// to the next free slot >= n and becomes the LEFT child, the second stays and is
// the RIGHT child, the composite takes the first minimum's slot; codes are
// root-to-leaf paths with left = 0.  The stream is, per 4096 symbols, the codes
// concatenated MSB-first into 32-bit words, preceded by the word count.
//
// MI355X design differences: the per-4096-symbol histograms come out of the MTF
// kernel (no separate histogram pass); the tree search is a wave64 arg-min over
// packed (count,level,slot) keys instead of one thread scanning 257 slots; the
// size of every block is known from (sub-histogram . code lengths) BEFORE packing,
// so the packer writes straight into its final place -- no staging array, no
// O(B^2) offset loop, no device->host copy in the middle of the pipeline
// (compress_app.cu:106).
#include "glc_device.h"
#include "glc_internal.h"
#include "huff_tree.h"

namespace glc {

// One workgroup per 1 MiB block.  The kernel is a chain of latencies (two passes over the block's 256 partial
// histograms, a serial tree build in one wave), not a throughput problem: what matters in a batch is that EVERY block
// of the launch is resident at once.  512 threads = 4 workgroups per CU = 1024 blocks on the chip (1024 threads: two
// rounds of 512 blocks, 0.22 ms per 1024 blocks against 0.16; 256 threads: the same 0.16 with longer passes); up to
// 512 blocks fit with 1024 threads, whose passes are half as long.
#ifdef GLC_HB_TIMING
__device__ unsigned long long g_hb_stamp[8];
#define HB_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_hb_stamp[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int glcDebugHuffStamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hb_stamp), sizeof(g_hb_stamp)); }
#else
#define HB_STAMP(i) do { } while (0)
#endif

template <int HB_NT>
__global__ __launch_bounds__(HB_NT) void k_huff_build(const uint32_t *__restrict__ sub_hist, uint32_t max_sub,
                                                    uint32_t n, uint32_t *__restrict__ d_hist,
                                                    uint32_t *__restrict__ codes_out,
                                                    uint32_t *__restrict__ lens_out,
                                                    uint32_t *__restrict__ d_offsets, size_t offset_stride,
                                                    uint32_t *__restrict__ d_size, uint64_t capacity_words,
                                                    uint32_t *__restrict__ d_status,
                                                    const uint32_t *__restrict__ redo_flag,
                                                    const uint32_t *__restrict__ only)
{
    static_assert(HB_NT == 256 || HB_NT == 512 || HB_NT == 1024, "k_huff_build: 256, 512 or 1024 threads");
    constexpr int HB_PARTS = HB_NT / 256, HB_NW = HB_NT / 64;
    __shared__ uint32_t s_hist[257];
    __shared__ uint32_t s_part[HB_PARTS][256];
    __shared__ HuffTreeLds T;
    __shared__ uint32_t s_code[257], s_len[257];
    __shared__ uint32_t s_words[256];
    __shared__ uint32_t s_tmp[HB_NT / 64 + 1];

    const uint32_t b = blockIdx.x, tid = threadIdx.x, l = tid & 63;
    if (only && !only[b]) return;                              // (second pass over the blocks a later sorter tier rewrote)
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    const uint32_t *SH = sub_hist + (size_t)b * max_sub * 256;
    HB_STAMP(0);

    // ---- total histogram (huffman_build_tree_kernel merges partial histograms,
    //      compress_kernel.cuh:2284-2299; EOF gets count 1, :2250) ----
    {
        // thread = (symbol, part of the sub-blocks), 16 loads in flight
        const uint32_t sym = tid & 255, qt = tid >> 8;
        uint32_t c = 0;
        for (uint32_t s = qt * 16; s < nsub; s += 16 * HB_PARTS) {
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = s + k < nsub ? SH[(size_t)(s + k) * 256 + sym] : 0u;
#pragma unroll
            for (int k = 0; k < 16; k++) c += v[k];
        }
        s_part[qt][sym] = c;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < HB_PARTS; q++) c += s_part[q][tid];
        s_hist[tid] = c;
        d_hist[(size_t)b * 256 + tid] = c;
        s_code[tid] = 0; s_len[tid] = 0;
        if (tid == 0) { s_hist[256] = 1; s_code[256] = 0; s_len[256] = 0; }
    }
    __syncthreads();

    HB_STAMP(1);
    huff_tree_build<HB_NT>(T, s_hist, tid);
    __syncthreads();
    HB_STAMP(2);

    // ---- codes: every leaf walks to the root; the k-th step up supplies bit k
    //      (left = 0, right = 1: compress_kernel.cuh:2416-2496) ----
    {
        const int nl = T.nl, used = 2 * nl - 1;
        for (int s = (int)tid; s < used; s += HB_NT) {
            if (T.left[s] < 0) {
                uint32_t code = 0, len = 0;
                int node = s, p = T.parent[s];
                while (p >= 0) {
                    if (T.right[p] == node) code |= 1u << len;
                    len++;
                    node = p; p = T.parent[p];
                }
                const int sym = T.value[s];
                s_code[sym] = code; s_len[sym] = len;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 257; i += HB_NT) {
        codes_out[(size_t)b * 257 + i] = s_code[i];
        lens_out[(size_t)b * 257 + i] = s_len[i];
    }

    HB_STAMP(3);
    // ---- words per 4096-symbol block = ceil(sub_hist . len / 32) ----
    {
        uint32_t ln[4];
#pragma unroll
        for (int r = 0; r < 4; r++) ln[r] = s_len[r * 64 + l];
        for (uint32_t s0 = w; s0 < 256; s0 += 4 * HB_NW) {     // four sub-blocks of this wave at a time: 16 loads in flight
            uint32_t h[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sb = s0 + HB_NW * k;
#pragma unroll
                for (int r = 0; r < 4; r++) h[k][r] = sb < nsub ? SH[(size_t)sb * 256 + r * 64 + l] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sb = s0 + HB_NW * k;
                uint32_t bits = h[k][0] * ln[0] + h[k][1] * ln[1] + h[k][2] * ln[2] + h[k][3] * ln[3];
                bits = wave_sum(bits);
                if (l == 0) s_words[sb] = (sb < nsub) ? (bits + 31) / 32 : 0u;
            }
        }
    }
    __syncthreads();
    HB_STAMP(4);
    {
        const uint32_t wd = tid < 256 ? s_words[tid] : 0u;
        const uint32_t item = (tid < nsub) ? 1 + wd : 0;
        uint32_t total = 0;
        const uint32_t off = block_excl_add<HB_NT>(item, s_tmp, &total);
        if (tid < nsub) {
            d_offsets[(size_t)b * offset_stride + tid] = off;
            if (wd > HUFF_MAX_WORDS && !(redo_flag && redo_flag[b])) atomicOr(d_status, ST_BLOCK_OVERFLOW);
        }
        if (tid == 0) {
            d_size[b] = total;
            if ((uint64_t)total > capacity_words && !(redo_flag && redo_flag[b])) atomicOr(d_status, ST_CAPACITY);
        }
    }
    HB_STAMP(5);
}

// grid (sub-blocks, blocks); 256 threads x 16 symbols
// HP_SUBS consecutive 4096-symbol sub-blocks of a block per workgroup: the {code, length} table of the block arrives once,
// and only the words a sub-block fills are cleared (its bit total is known from the scan before anything is ORed: a Zipf
// sub-block fills ~910 of the 3600 words the worst case needs -- clearing all of them for every 4096 symbols, plus 2 KB of
// table per 4 KB of symbols, was most of what the kernel did).
constexpr uint32_t HP_SUBS = 4;

__global__ __launch_bounds__(256) void k_huff_pack(const uint8_t *__restrict__ mtf, size_t mtf_stride, uint32_t n,
                                                   const uint32_t *__restrict__ codes,
                                                   const uint32_t *__restrict__ lens,
                                                   const uint32_t *__restrict__ d_offsets, size_t offset_stride,
                                                   uint32_t *__restrict__ d_comp, size_t comp_stride,
                                                   uint64_t capacity_words, const uint32_t *__restrict__ only,
                                                   const unsigned long long *__restrict__ block_off)
{
    // block_off (compact layout): block b's words start at d_comp + block_off[b] and capacity_words bounds the whole
    // array; otherwise at d_comp + b * comp_stride with capacity_words per block
    constexpr int SPT = HUFF_BLOCK / 256;                     // 16 symbols per thread
    constexpr int MAXW = HUFF_BLOCK * 28 / 32 + 16;            // code length <= 28 for <= 2^20+1 total count
    __shared__ uint2 s_cl[257];                                // {code, length}: one 8-byte LDS read per symbol
    __shared__ __attribute__((aligned(16))) uint32_t s_words[MAXW];
    __shared__ uint32_t s_tmp[8];
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    if (only && !only[b]) return;
    if (blockIdx.x * HP_SUBS * HUFF_BLOCK >= n) return;
    bool mylong = false;
    for (uint32_t i = tid; i < 257; i += 256) {
        const uint2 cl1 = make_uint2(codes[(size_t)b * 257 + i], lens[(size_t)b * 257 + i]);
        s_cl[i] = cl1;
        mylong |= cl1.y > 14;
    }
    // No code of the block longer than 14 bits (Zipf bytes: 5 .. 13): two symbols' codes make ONE of up to 28 bits before the merge,
    // which then takes 8 steps instead of 16 (GLC_HP_NO_PAIRS: A/B)
#ifdef GLC_HP_NO_PAIRS
    const bool pairs = false;
    (void)mylong;
#else
    const bool pairs = __syncthreads_or((int)mylong) == 0;     // (uniform)
#endif
    const uint64_t base = block_off ? block_off[b] : 0ull;
    for (uint32_t sub = blockIdx.x * HP_SUBS; sub < (blockIdx.x + 1) * HP_SUBS; sub++) {
        const uint32_t lo = sub * HUFF_BLOCK;
        if (lo >= n) break;
        const uint32_t cntb = min((uint32_t)HUFF_BLOCK, n - lo);
        __syncthreads();                                       // the table is in place / the previous sub-block's words have been read

        const uint8_t *src = mtf + (size_t)b * mtf_stride + lo;
        uint8_t sym[SPT];
        const uint32_t i0 = tid * SPT;
        if (i0 + SPT <= cntb && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + i0);
            const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < SPT; j++) sym[j] = (uint8_t)(qq[j >> 2] >> (8 * (j & 3)));
        } else {
#pragma unroll
            for (int j = 0; j < SPT; j++) sym[j] = (i0 + j < cntb) ? src[i0 + j] : 0;
        }
        uint2 cl[SPT];                                         // read once, used by the bit count and by the merge
        uint32_t mybits = 0;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            cl[j] = s_cl[sym[j]];
            if (i0 + j >= cntb) cl[j] = make_uint2(0u, 0u);
            mybits += cl[j].y;
        }
        uint32_t total = 0;
        const uint32_t start = block_excl_add<256>(mybits, s_tmp, &total);
        const uint32_t nwords = (total + 31) / 32;
        for (uint32_t i = tid; 4 * i < nwords + 1; i += 256) reinterpret_cast<uint4 *>(s_words)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();

        // merge, branch-free: `hi` is the word being filled (MSB first), `fill` its used bits.  A code of ln <= 28 bits goes
        // in as the 64-bit value code << (64 - fill - ln): its upper half lands in `hi`, its lower half is the start of the
        // next word, which becomes `hi` when the word is full.  (Pending bits kept right-aligned in a 64-bit accumulator
        // needed a variable-length mask and a branch per symbol: ~25 VALU per symbol against ~12.  Every symbol ORing its own
        // code into the two words it spans -- no "full" test at all -- measured 0.72 against 0.71 ms per GiB: not this.)
        uint32_t wi = start >> 5, fill = start & 31, hi = 0;
        if (pairs) {
#pragma unroll
            for (int j = 0; j < SPT; j += 2) {
                const uint32_t ln = cl[j].y + cl[j + 1].y;         // <= 28
                const uint32_t cd = (cl[j].x << cl[j + 1].y) | cl[j + 1].x;
                const uint64_t V = (uint64_t)cd << ((64u - fill - ln) & 63u);
                hi |= (uint32_t)(V >> 32);
                const uint32_t nf = fill + ln;
                const bool full = nf >= 32;
                if (full) atomicOr(&s_words[wi], hi);
                wi += full ? 1u : 0u;
                hi = full ? (uint32_t)V : hi;
                fill = nf & 31u;
            }
        } else {
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const uint32_t ln = cl[j].y;                       // 0 (and code 0) past the end of the block: a no-op
            const uint64_t V = (uint64_t)cl[j].x << ((64u - fill - ln) & 63u);
            hi |= (uint32_t)(V >> 32);
            const uint32_t nf = fill + ln;
            const bool full = nf >= 32;
            if (full) atomicOr(&s_words[wi], hi);
            wi += full ? 1u : 0u;
            hi = full ? (uint32_t)V : hi;
            fill = nf & 31u;
        }
        }
        if (fill > 0 && mybits > 0) atomicOr(&s_words[wi], hi);
        __syncthreads();

        const uint32_t off = d_offsets[(size_t)b * offset_stride + sub];
        if (base + off + 1 + nwords > capacity_words) continue;          // flagged by k_huff_build / k_compact_offsets
        uint32_t *dst = d_comp + (block_off ? (size_t)base : (size_t)b * comp_stride) + off;
        if (tid == 0) dst[0] = nwords;
        for (uint32_t i = tid; i < nwords; i += 256) dst[1 + i] = s_words[i];
    }
}

// ---------------------------------------------------------------------------
// stream compaction for result collection: block b's words move from the strided
// layout to out[off[b] .. off[b]+size[b]); off has nblk+1 entries (last = total).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_compact_offsets(const uint32_t *__restrict__ sizes, uint32_t nblk,
                                                          unsigned long long *__restrict__ off,
                                                          const unsigned long long *__restrict__ start,
                                                          unsigned long long capacity, uint32_t *__restrict__ d_status)
{
    // start (optional): device word offset the first block begins at (the end of the batch before); capacity
    // (with d_status): the array's size in words -- streams that would pass it are reported and not written
    __shared__ uint32_t s_tmp[20];
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = start ? *start : 0ull;
    __syncthreads();
    for (uint32_t base = 0; base < nblk; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t sz = i < nblk ? sizes[i] : 0;
        uint32_t tot = 0;
        const uint32_t ex = block_excl_add<1024>(sz, s_tmp, &tot);
        const unsigned long long run = s_run;
        if (i < nblk) off[i] = run + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[nblk] = s_run;
        if (d_status && s_run > capacity) atomicOr(d_status, ST_CAPACITY);
    }
}

__global__ __launch_bounds__(256) void k_compact_copy(const uint32_t *__restrict__ comp, size_t stride,
                                                      const uint32_t *__restrict__ sizes,
                                                      const unsigned long long *__restrict__ off,
                                                      uint32_t *__restrict__ out)
{
    const uint32_t b = blockIdx.y;
    const uint32_t sz = sizes[b];
    const uint32_t *src = comp + (size_t)b * stride;
    uint32_t *dst = out + off[b];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < sz; i += gridDim.x * 256) dst[i] = src[i];
}

hipError_t compact_streams(hipStream_t st, const uint32_t *d_comp, size_t stride, const uint32_t *d_sizes,
                           uint32_t nblk, uint32_t *d_out, unsigned long long *d_off)
{
    hipLaunchKernelGGL(k_compact_offsets, dim3(1), dim3(1024), 0, st, d_sizes, nblk, d_off, (const unsigned long long *)nullptr,
                       0ull, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_compact_copy, dim3(32, nblk), dim3(256), 0, st, d_comp, stride, d_sizes, d_off, d_out);
    return hipGetLastError();
}

// the mirror (decode side of the exchange): block b's words move from in[off[b] .. off[b+1]) back to the strided
// layout the decoder reads; a block longer than its stride is cut and reported
__global__ __launch_bounds__(256) void k_expand_copy(const uint32_t *__restrict__ in,
                                                     const unsigned long long *__restrict__ off, size_t stride,
                                                     uint32_t *__restrict__ comp, uint32_t *__restrict__ sizes,
                                                     uint32_t *__restrict__ d_status, uint32_t nblk)
{
    const uint32_t b = blockIdx.y;
    // the offsets come from another process (the exchange): they must ascend and stay inside the total the caller's
    // buffer holds (off[nblk]); a block that does not is reported and nothing of it is read
    const unsigned long long lo = off[b], hi = off[b + 1], total = off[nblk];
    unsigned long long sz = hi - lo;
    if (hi < lo || hi > total) {
        sz = 0;
        if (d_status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(d_status, ST_CORRUPT);
    }
    if (sz > stride) { sz = stride; if (d_status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(d_status, ST_CAPACITY); }
    const uint32_t *src = in + lo;
    uint32_t *dst = comp + (size_t)b * stride;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < (uint32_t)sz; i += gridDim.x * 256) dst[i] = src[i];
    if (sizes && blockIdx.x == 0 && threadIdx.x == 0) sizes[b] = (uint32_t)sz;
}

hipError_t expand_streams(hipStream_t st, const uint32_t *d_in, const unsigned long long *d_off, uint32_t nblk,
                          uint32_t *d_comp, size_t stride, uint32_t *d_sizes, uint32_t *d_status)
{
    hipLaunchKernelGGL(k_expand_copy, dim3(32, nblk), dim3(256), 0, st, d_in, d_off, stride, d_comp, d_sizes, d_status, nblk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t huff_scratch_alloc(HuffScratch &s, uint32_t nmax, uint32_t rows)
{
    s.nmax = nmax; s.rows = rows; s.max_sub = (nmax + HUFF_BLOCK - 1) / HUFF_BLOCK;
    size_t a = (size_t)rows * s.max_sub * 256 * 4, c = (size_t)rows * 257 * 4;
    GLC_TRY(hipMalloc((void **)&s.sub_hist, a));
    GLC_TRY(hipMalloc((void **)&s.codes, c));
    GLC_TRY(hipMalloc((void **)&s.lens, c));
    s.bytes = a + 2 * c;
    return hipSuccess;
}

void huff_scratch_free(HuffScratch &s)
{
    if (s.sub_hist) (void)hipFree(s.sub_hist);
    if (s.codes) (void)hipFree(s.codes);
    if (s.lens) (void)hipFree(s.lens);
    s = HuffScratch();
}

// histograms of the 4096-symbol sub-blocks of caller-supplied symbols (the compress pipeline gets them from the
// MTF kernel; this is for the stand-alone Huffman entry point): one wave per sub-block, 4 copies against
// same-symbol serialisation of the LDS atomics
__global__ __launch_bounds__(256) void k_sub_hist(const uint8_t *__restrict__ sym, size_t stride, uint32_t n,
                                                  uint32_t max_sub, uint32_t *__restrict__ sub_hist)
{
    __shared__ uint32_t s_h[4][4 * 257];
    const uint32_t b = blockIdx.y, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK, sub = blockIdx.x * 4 + w;
    for (uint32_t i = l; i < 4 * 257; i += 64) s_h[w][i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (sub >= nsub) return;
    const uint32_t lo = sub * HUFF_BLOCK, hi = min(n, lo + HUFF_BLOCK);
    const uint8_t *S = sym + (size_t)b * stride;
    uint32_t *H = s_h[w] + (l & 3) * 257;
    for (uint32_t i = lo + l; i < hi; i += 64) atomicAdd(&H[S[i]], 1u);
    __builtin_amdgcn_wave_barrier();
    uint32_t *O = sub_hist + ((size_t)b * max_sub + sub) * 256;
    for (uint32_t i = l; i < 256; i += 64) O[i] = s_h[w][i] + s_h[w][257 + i] + s_h[w][514 + i] + s_h[w][771 + i];
}

hipError_t huff_histogram(hipStream_t st, const uint8_t *sym, size_t stride, uint32_t n, uint32_t nblk, HuffScratch &s)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    hipLaunchKernelGGL(k_sub_hist, dim3((nsub + 3) / 4, nblk), dim3(256), 0, st, sym, stride, n, s.max_sub, s.sub_hist);
    return hipGetLastError();
}

hipError_t huff_build(hipStream_t st, uint32_t n, uint32_t nblk, HuffScratch &s, uint32_t *d_hist,
                      uint32_t *d_offsets, size_t offset_stride, uint32_t *d_size, size_t capacity_words,
                      uint32_t *d_status, const uint32_t *redo_flag, const uint32_t *only)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const int pi = s.prof ? s.prof->begin(PROF_HUFF_BUILD, st) : -1;
    if (nblk <= 512)                                           // resident either way: more threads, shorter passes
        hipLaunchKernelGGL(k_huff_build<1024>, dim3(nblk), dim3(1024), 0, st, s.sub_hist, s.max_sub, n, d_hist, s.codes,
                           s.lens, d_offsets, offset_stride, d_size, (uint64_t)capacity_words, d_status, redo_flag, only);
    else
        hipLaunchKernelGGL(k_huff_build<512>, dim3(nblk), dim3(512), 0, st, s.sub_hist, s.max_sub, n, d_hist, s.codes,
                           s.lens, d_offsets, offset_stride, d_size, (uint64_t)capacity_words, d_status, redo_flag, only);
    if (pi >= 0) s.prof->end(pi, (double)n * nblk, st);
    return hipGetLastError();
}

hipError_t huff_block_offsets(hipStream_t st, const uint32_t *d_sizes, uint32_t nblk, unsigned long long *d_off,
                              const unsigned long long *d_start, size_t capacity_words, uint32_t *d_status)
{
    hipLaunchKernelGGL(k_compact_offsets, dim3(1), dim3(1024), 0, st, d_sizes, nblk, d_off, d_start,
                       (unsigned long long)capacity_words, d_status);
    return hipGetLastError();
}

hipError_t huff_pack(hipStream_t st, const uint8_t *mtf, size_t mtf_stride, uint32_t n, uint32_t nblk,
                     HuffScratch &s, const uint32_t *d_offsets, size_t offset_stride, uint32_t *d_compressed,
                     size_t comp_stride_words, const uint32_t *only, const unsigned long long *d_block_off,
                     size_t capacity_words)
{
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    const int pi = s.prof ? s.prof->begin(PROF_HUFF_PACK, st) : -1;
    hipLaunchKernelGGL(k_huff_pack, dim3((nsub + HP_SUBS - 1) / HP_SUBS, nblk), dim3(256), 0, st, mtf, mtf_stride, n, s.codes, s.lens,
                       d_offsets, offset_stride, d_compressed, comp_stride_words,
                       (uint64_t)(d_block_off ? capacity_words : comp_stride_words), only, d_block_off);
    if (pi >= 0) s.prof->end(pi, (double)n * nblk, st);
    return hipGetLastError();
}

} // namespace glc
