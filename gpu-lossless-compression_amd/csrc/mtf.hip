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
//   3. k_mtf_encode       MTF as dominance counting over previous-occurrence
//      timestamps (see the kernel): four chunks per wave, one per 16-lane DPP row;
//      no serial list, no per-byte dependency chain.
#include "glc_device.h"
#include "glc_internal.h"

namespace glc {

constexpr int MTF_WAVES = 4;                       // waves per workgroup
constexpr int MTF_INP = 256 + 8;                   // membership bytes of mtf_fold + its spill slot

// --- 1. chunk-local recency lists ------------------------------------------
// list = distinct symbols of the chunk, most recent first = the symbols sorted by the ORDER INDEX o (0 = last byte
// of the chunk) of their most recent occurrence.  That minimum does not depend on the order the bytes are looked at,
// so the wave does not walk the chunk batch by batch behind one 64-byte load at a time (a memory latency per half
// cache line: 0.30 ms per 256 MiB, 0.9 TB/s): every lane loads its 4 x 16 bytes up front and lowers
// first[sym] with an LDS atomicMin only when a plain read says it would (most recent bytes first, so after the
// first few dozen bytes almost nothing passes the test); the symbols are then ranked by first[] through a 4096-bit
// occupancy bitmap + prefix popcounts.
__global__ __launch_bounds__(MTF_WAVES * 64) void k_mtf_chunk_lists(const uint8_t *__restrict__ in,
                                                                   size_t in_stride, uint32_t n,
                                                                   uint8_t *__restrict__ lists,
                                                                   uint16_t *__restrict__ lens,
                                                                   uint32_t max_chunks, const uint32_t *__restrict__ only)
{
    __shared__ uint32_t s_first[MTF_WAVES][256];
    __shared__ unsigned long long s_bm[MTF_WAVES][64];
    __shared__ uint32_t s_cum[MTF_WAVES][64];
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63;
    if (only && !only[b]) return;                              // (second pass over the blocks a later sorter tier rewrote)
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const uint32_t chunk = blockIdx.x * MTF_WAVES + w;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    if (chunk >= nchunks) return;                           // whole wave exits together
    const uint32_t lo = chunk * MTF_CHUNK, hi = min(n, lo + MTF_CHUNK), C = hi - lo;
    const uint8_t *src = in + (size_t)b * in_stride;
    uint8_t *L = lists + ((size_t)b * max_chunks + chunk) * 256;
    uint32_t *first = s_first[w];
    for (int i = l; i < 256; i += 64) first[i] = 0xFFFFFFFFu;
    s_bm[w][l] = 0;
    __builtin_amdgcn_wave_barrier();
    if (C == MTF_CHUNK && (reinterpret_cast<uintptr_t>(src + lo) & 15) == 0) {
        // lane l of segment g holds order indices [1024 g + 16 l, +16): the 16 bytes at lo + 4096 - 1024 g - 16 (l + 1)
        uint4 q[4];
#pragma unroll
        for (int g = 0; g < 4; g++)
            q[g] = *reinterpret_cast<const uint4 *>(src + lo + MTF_CHUNK - 1024 * g - 16 * (l + 1));
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t d[4] = {q[g].x, q[g].y, q[g].z, q[g].w};
#pragma unroll
            for (int k = 15; k >= 0; k--) {                   // byte k sits at order index 1024 g + 16 l + 15 - k
                const uint32_t sym = (d[k >> 2] >> (8 * (k & 3))) & 0xFFu, o = 1024u * g + 16u * l + 15u - k;
                if (o < first[sym]) atomicMin(&first[sym], o);
            }
        }
    } else {
        for (uint32_t o = l; o < C; o += 64) {
            const uint32_t sym = src[hi - 1 - o];
            if (o < first[sym]) atomicMin(&first[sym], o);
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t f[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f[j] = first[4 * l + j];
        if (f[j] != 0xFFFFFFFFu) atomicOr(&s_bm[w][f[j] >> 6], 1ull << (f[j] & 63));
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t c = (uint32_t)__popcll(s_bm[w][l]);
    const uint32_t inc = wave_incl_add(c);
    s_cum[w][l] = inc - c;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (f[j] != 0xFFFFFFFFu) {
            const uint32_t wd = f[j] >> 6, r = f[j] & 63;
            L[s_cum[w][wd] + (uint32_t)__popcll(s_bm[w][wd] & ((1ull << r) - 1ull))] = (uint8_t)(4 * l + j);
        }
    }
    if (l == 0) lens[(size_t)b * max_chunks + chunk] = (uint16_t)__builtin_amdgcn_readlane((int)inc, 63);
}

// --- 2. exclusive scan of the lists (in place: lists[c] becomes the MTF list
//        in force at the start of chunk c) -----------------------------------
// The operator  S' = P ++ (S \ P)  is associative, so the 256 folds of a 1 MiB block need not be one chain
// (one wave per block, 0.15 ms with the rest of the machine idle): W waves per block, groups of G = 256 / W chunks
//   A  each wave folds the lists of its G chunks into the group's combined list,
//   B  wave 0 folds the W combined lists into the state at the start of every group,
//   C  each wave walks its G chunks again from that state, publishing the start list of every chunk.


// next[0 .. m) = P (entries 4l .. 4l+3 in p4), then the entries of cur[0 .. ls) not in P, in order.  Returns the new length.
// Every lane filters the four CONSECUTIVE entries 4l .. 4l+3 of cur: one dword read, four membership reads in flight
// together, one wave scan of the kept counts.  (A lane per entry and four rounds of 64 -- read the entry, read its
// membership byte, ballot, write -- is eight LDS round trips one after the other, and a fold is one link of a chain
// of 48: 0.74 us per fold, 35 us per block.)
__device__ __forceinline__ uint32_t mtf_fold(const uint8_t *cur, uint8_t *next, uint8_t *inp, uint32_t p4, uint32_t m,
                                             uint32_t ls, uint32_t l, uint32_t *cur4 = nullptr)
{
    // inp: 256 membership bytes + a slot (256 + lane's byte) that takes the writes of the entries of p4 past m.  The
    // dword of P goes to next whole: what lies past m is overwritten by the kept entries or is past the new length.
    const uint32_t c4 = reinterpret_cast<const uint32_t *>(cur)[l];
    reinterpret_cast<uint32_t *>(inp)[l] = 0;
    __builtin_amdgcn_wave_barrier();
    reinterpret_cast<uint32_t *>(next)[l] = p4;
#pragma unroll
    for (int j = 0; j < 4; j++) inp[4 * l + j < m ? (p4 >> (8 * j)) & 0xFFu : 256u + j] = 1;
    __builtin_amdgcn_wave_barrier();
    if (cur4) *cur4 = c4;
    uint32_t in[4];
#pragma unroll
    for (int j = 0; j < 4; j++) in[j] = inp[(c4 >> (8 * j)) & 0xFFu];
    uint32_t keep[4], cnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { keep[j] = (4 * l + j < ls && !in[j]) ? 1u : 0u; cnt += keep[j]; }
    const uint32_t inc = wave_incl_add(cnt);
    uint32_t pos = m + inc - cnt;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (keep[j]) next[pos] = (uint8_t)(c4 >> (8 * j));
        pos += keep[j];
    }
    __builtin_amdgcn_wave_barrier();
    return m + (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
}

template <int MSC_WAVES>
__global__ __launch_bounds__(MSC_WAVES * 64) void k_mtf_scan_lists(uint8_t *__restrict__ lists,
                                                                  const uint16_t *__restrict__ lens, uint32_t n,
                                                                  uint32_t max_chunks, const uint32_t *__restrict__ only)
{
    constexpr int MSC_GROUP = 256 / MSC_WAVES;
    __shared__ __attribute__((aligned(16))) uint8_t s_state[MSC_WAVES][2][256];
    __shared__ __attribute__((aligned(16))) uint8_t s_inp[MSC_WAVES][MTF_INP];
    __shared__ __attribute__((aligned(16))) uint8_t s_comb[MSC_WAVES][256];      // combined list of every group
    __shared__ uint32_t s_clen[MSC_WAVES];
    __shared__ __attribute__((aligned(16))) uint8_t s_start[MSC_WAVES + 1][256]; // state at the start of every group (and after the last)
    __shared__ __attribute__((aligned(16))) uint8_t s_carry[256];                // state at the start of the round
    const uint32_t b = blockIdx.x, l = threadIdx.x & 63;
    if (only && !only[b]) return;
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    uint8_t *LB = lists + (size_t)b * max_chunks * 256;
    const uint16_t *NB = lens + (size_t)b * max_chunks;
    if (w == 0) for (int i = l; i < 256; i += 64) s_carry[i] = (uint8_t)i;      // identity list before the first chunk
    // 256 chunks (16 groups of 16) per round; cudppCompress blocks (n <= 2^20) need one round, MTF plans may be longer
    for (uint32_t r0 = 0; r0 < nchunks; r0 += MSC_WAVES * MSC_GROUP) {
        const uint32_t nch = min(nchunks - r0, (uint32_t)(MSC_WAVES * MSC_GROUP));
        const uint32_t ngroups = (nch + MSC_GROUP - 1) / MSC_GROUP;
        const uint32_t c0 = r0 + w * MSC_GROUP, c1 = min(r0 + nch, c0 + MSC_GROUP);
        // A: combined list of the group (most recent first), built from an empty state
        uint32_t p4[MSC_GROUP], mm[MSC_GROUP];
#pragma unroll
        for (int k = 0; k < MSC_GROUP; k++) {                  // all loads of the group in flight
            const bool in = c0 + k < c1;
            p4[k] = in ? reinterpret_cast<const uint32_t *>(LB + (size_t)(c0 + k) * 256)[l] : 0u;
            mm[k] = in ? NB[c0 + k] : 0u;
        }
        if (w < ngroups) {
            int cur = 0;
            uint32_t ls = 0;
#pragma unroll
            for (int k = 0; k < MSC_GROUP; k++)
                if (c0 + k < c1) { ls = mtf_fold(s_state[w][cur], s_state[w][cur ^ 1], s_inp[w], p4[k], mm[k], ls, l); cur ^= 1; }
            reinterpret_cast<uint32_t *>(s_comb[w])[l] = reinterpret_cast<const uint32_t *>(s_state[w][cur])[l];
            if (l == 0) s_clen[w] = ls;
        }
        __syncthreads();
        // B: state at the start of every group (s_start[g + 1] = s_start[g] folded with group g), and the state the next
        //    round starts from
        if (w == 0) {
            reinterpret_cast<uint32_t *>(s_start[0])[l] = reinterpret_cast<const uint32_t *>(s_carry)[l];
            uint32_t cp[MSC_WAVES], cl[MSC_WAVES];
#pragma unroll
            for (int g = 0; g < MSC_WAVES; g++) { cp[g] = reinterpret_cast<const uint32_t *>(s_comb[g])[l]; cl[g] = s_clen[g]; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < MSC_WAVES; g++)
                if ((uint32_t)g < ngroups) (void)mtf_fold(s_start[g], s_start[g + 1], s_inp[0], cp[g], cl[g], 256, l);
            reinterpret_cast<uint32_t *>(s_carry)[l] = reinterpret_cast<const uint32_t *>(s_start[ngroups])[l];
        }
        __syncthreads();
        // C: start list of every chunk of the group (the dword of the state a fold reads is the one to publish)
        if (w < ngroups) {
            int cur = 0;
            reinterpret_cast<uint32_t *>(s_state[w][0])[l] = reinterpret_cast<const uint32_t *>(s_start[w])[l];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < MSC_GROUP; k++) {
                if (c0 + k < c1) {
                    uint32_t st4;
                    if (c0 + k + 1 < c1) { (void)mtf_fold(s_state[w][cur], s_state[w][cur ^ 1], s_inp[w], p4[k], mm[k], 256, l, &st4); cur ^= 1; }
                    else st4 = reinterpret_cast<const uint32_t *>(s_state[w][cur])[l];
                    reinterpret_cast<uint32_t *>(LB + (size_t)(c0 + k) * 256)[l] = st4;
                }
            }
        }
        __syncthreads();
    }
}

// --- 3. encode ---------------------------------------------------------------
// MTF as dominance counting.  Give every position i of the chunk the index
// P[i] of the previous occurrence of its symbol (virtual times -256..-1 for the
// symbols of the start list: the symbol at list position q "occurred" at -1-q).
// The P values are distinct, and
//     mtf[i] = #{ j in (P[i], i) : P[j] < P[i] }
// (the symbols whose first occurrence after P[i] lies before i).  Positions are
// evaluated a batch at a time:
//   * j in earlier batches:  P[j] marks a "killed" timestamp; a 4352-bit bitmap of
//     killed timestamps + per-word prefix counts in LDS answers
//     #{j < base : P[j] < P[i]} with two LDS reads and a popcount; the j <= P[i]
//     part of it is exactly P[i]+1.
//   * j in the same batch:   T = #{k < lane : P[k] < P[lane]} by meeting the P values of
//     the lanes below through DPP shifts and counting -- the cost that sets the batch
//     size: a 64-lane batch needs 63 steps.  A batch is therefore a ROW of 16 lanes: every
//     wave works on FOUR chunks at once, one per DPP row, 15 steps of row_shr:k (which
//     never crosses a row) for the same 64 symbols -- each step one v_sub_co_u32_dpp whose
//     borrow a v_addc accumulates -- and each row keeps its own tables.
//   * previous occurrence:   one table entry per symbol holds its last position (high half)
//     and the lanes of the current batch that carry it (low half): every lane ORs its bit
//     in, reads the entry back, and the last lane of a symbol stores the new position with
//     the lane bits cleared.  (Finding the lanes with a ballot per symbol bit costs ~50 VALU
//     per batch; the kernel runs at >90 % of both the VALU issue rate and the LDS
//     instruction rate, so every instruction of either kind shows.)
constexpr int MTF_ROWS = 4;                                 // chunks per wave (one per 16-lane DPP row)
// The killed-timestamp bitmap covers ONE SEGMENT of MTF_SEG positions (+ the 256 virtual timestamps of the list the segment
// starts from): 1280 bits = 20 words, two per lane of the row.  At a segment's end the live timestamps -- one per symbol -- are
// RE-BASED: a symbol's timestamp becomes the number of live timestamps below it (its place from the back of the MTF list), the
// bitmap starts empty again.  (Until round 5 the bitmap covered the whole chunk, 4352 bits, five words per lane: its per-batch
// recount was the largest single piece of the kernel, 0.65 of 3.48 ms per GiB with the recount simply left out.)
constexpr int MTF_NWORDS = 32;                              // bitmap words per row: 20 used, 2 per lane
constexpr uint32_t MTF_SEG = 1024;                          // positions per segment
#ifndef GLC_MTF_QUARTERS_MAX
#define GLC_MTF_QUARTERS_MAX 2048
#endif
constexpr uint32_t MTF_QUARTERS_MAX_CHUNKS = GLC_MTF_QUARTERS_MAX;   // up to this many chunks in a launch: one wave per chunk

//
// QUARTERS (small batches): the rows of a wave are the four QUARTERS of one chunk instead of four chunks, so a block on
// its own is 256 waves of 64 batches, not 64 waves of 256 (0.145 -> ~0.05 ms with the rest of the machine idle; the
// batch loop is a chain of LDS round trips, ~1400 cycles per batch for a wave alone on its SIMD).  The start lists of
// quarters 1..3 are made here: recency list of the quarter before (as k_mtf_chunk_lists does for a chunk) folded into
// its start list (the operator of k_mtf_scan_lists), three times, by the whole wave.
template <bool WITH_HIST, bool QUARTERS, bool ZEROS = false>
__global__ __launch_bounds__(MTF_WAVES * 64) void k_mtf_encode(const uint8_t *__restrict__ in,
                                                              size_t in_stride, uint32_t n,
                                                              const uint8_t *__restrict__ lists,
                                                              uint32_t max_chunks,
                                                              uint8_t *__restrict__ out, size_t out_stride,
                                                              uint32_t *__restrict__ sub_hist, const uint32_t *__restrict__ only)
{
    // slot strides are padded so that the four rows of a wave, which run in lockstep and favour the same symbols, ranks and
    // bitmap words, do not meet in the same LDS banks
    __shared__ uint32_t s_hist[WITH_HIST ? MTF_WAVES * MTF_ROWS : 1][128 + 8];   // 16-bit counters: rank r in word r & 127, half r >> 7
    __shared__ uint32_t s_tab[MTF_WAVES * MTF_ROWS][256 + 8];                    // per symbol: last occurrence + 256 << 16 | lanes of the batch holding it
    __shared__ __attribute__((aligned(16))) unsigned long long s_bm[MTF_WAVES * MTF_ROWS][MTF_NWORDS + 2];
    // prefix counts of bitmap word 5 q + k at entry 8 q + k: a lane's five counts are one aligned 16-byte store (packed
    // ten bytes apart they were an unaligned 8-byte store, which alone cost a quarter of the kernel)
    // prefix counts: entry w = killed timestamps in the words below word w (a lane's two entries are one dword store)
    __shared__ __attribute__((aligned(16))) uint16_t s_cum[MTF_WAVES * MTF_ROWS][MTF_NWORDS + 8];
    // QUARTERS: start lists of the four quarters, and the scratch of the recency list + fold that make them
    __shared__ __attribute__((aligned(16))) uint8_t s_qstart[QUARTERS ? MTF_WAVES : 1][MTF_ROWS][256];
    __shared__ __attribute__((aligned(16))) uint8_t s_qlist[QUARTERS ? MTF_WAVES : 1][256], s_qinp[QUARTERS ? MTF_WAVES : 1][MTF_INP];
    __shared__ uint32_t s_qfirst[QUARTERS ? MTF_WAVES : 1][256], s_qcum[QUARTERS ? MTF_WAVES : 1][16];
    __shared__ unsigned long long s_qbm[QUARTERS ? MTF_WAVES : 1][16];
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63, lr = l & 15, row = l >> 4;
    if (only && !only[b]) return;
    const uint32_t w = threadIdx.x >> 6, slot = w * MTF_ROWS + row;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    constexpr uint32_t QLEN = MTF_CHUNK / MTF_ROWS;                          // 1024
    // QUARTERS: chunk0 = the wave's chunk, row = quarter.  Otherwise chunk0 = first of the wave's four chunks.
    const uint32_t chunk0 = QUARTERS ? blockIdx.x * MTF_WAVES + w : (blockIdx.x * MTF_WAVES + w) * MTF_ROWS;
    if (chunk0 >= nchunks) return;                                           // whole wave exits together
    const uint32_t chunk = QUARTERS ? chunk0 : chunk0 + row;
    const uint32_t Cc0 = min(n, chunk0 * MTF_CHUNK + MTF_CHUNK) - chunk0 * MTF_CHUNK;   // length of the wave's first chunk
    bool live;
    uint32_t lo, C, Cmax;
    if (QUARTERS) {
        live = Cc0 > QLEN * row;
        lo = chunk0 * MTF_CHUNK + (live ? QLEN * row : 0u);
        C = live ? min(Cc0 - QLEN * row, QLEN) : 0u;
        Cmax = min(Cc0, QLEN);                                               // the first quarter is the longest
    } else {
        live = chunk < nchunks;
        lo = (live ? chunk : chunk0) * MTF_CHUNK;
        C = live ? min(n, lo + MTF_CHUNK) - lo : 0u;
        Cmax = Cc0;                                                          // the wave's first chunk is its longest
    }
    const uint8_t *src = in + (size_t)b * in_stride + lo;
    uint8_t *dst = out + (size_t)b * out_stride + lo;
    uint32_t *tab = s_tab[slot];
    unsigned long long *bm = s_bm[slot];
    uint16_t *cum = s_cum[slot];
    uint4 lw;
    if (QUARTERS) {
        uint8_t (*S)[256] = s_qstart[w];
        reinterpret_cast<uint32_t *>(S[0])[l] = reinterpret_cast<const uint32_t *>(lists + ((size_t)b * max_chunks + chunk0) * 256)[l];
        const uint8_t *cbase = in + (size_t)b * in_stride + chunk0 * MTF_CHUNK;
        const bool vec = (reinterpret_cast<uintptr_t>(cbase) & 15) == 0;
        for (uint32_t q = 0; q + 1 < MTF_ROWS && Cc0 > QLEN * (q + 1); q++) {    // quarter q is full whenever q + 1 exists
            uint32_t *first = s_qfirst[w];
            for (int i = l; i < 256; i += 64) first[i] = 0xFFFFFFFFu;
            if (l < 16) s_qbm[w][l] = 0;
            __builtin_amdgcn_wave_barrier();
            const uint8_t *qs = cbase + QLEN * q;
            if (vec) {
                // lane l holds order indices [16 l, 16 l + 16) (0 = the quarter's last byte): byte k sits at 16 l + 15 - k
                const uint4 v = *reinterpret_cast<const uint4 *>(qs + QLEN - 16 * (l + 1));
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const uint32_t sym = (d[k >> 2] >> (8 * (k & 3))) & 0xFFu, o = 16u * l + 15u - k;
                    if (o < first[sym]) atomicMin(&first[sym], o);
                }
            } else {
                for (uint32_t o = l; o < QLEN; o += 64) {
                    const uint32_t sym = qs[QLEN - 1 - o];
                    if (o < first[sym]) atomicMin(&first[sym], o);
                }
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f[j] = first[4 * l + j];
                if (f[j] != 0xFFFFFFFFu) atomicOr(&s_qbm[w][f[j] >> 6], 1ull << (f[j] & 63));
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t c = l < 16 ? (uint32_t)__popcll(s_qbm[w][l]) : 0u;
            const uint32_t inc = wave_incl_add(c);
            if (l < 16) s_qcum[w][l] = inc - c;
            __builtin_amdgcn_wave_barrier();
            uint8_t *L = s_qlist[w];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (f[j] != 0xFFFFFFFFu) {
                    const uint32_t wd = f[j] >> 6, r = f[j] & 63;
                    L[s_qcum[w][wd] + (uint32_t)__popcll(s_qbm[w][wd] & ((1ull << r) - 1ull))] = (uint8_t)(4 * l + j);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            (void)mtf_fold(S[q], S[q + 1], s_qinp[w], reinterpret_cast<const uint32_t *>(L)[l], m, 256, l);
        }
        __builtin_amdgcn_wave_barrier();
        lw = reinterpret_cast<const uint4 *>(S[live ? row : 0])[lr];
    } else {
        lw = reinterpret_cast<const uint4 *>(lists + ((size_t)b * max_chunks + (live ? chunk : chunk0)) * 256)[lr];
    }
    {
        const uint32_t q[4] = {lw.x, lw.y, lw.z, lw.w};
#pragma unroll
        for (int j = 0; j < 16; j++) tab[(q[j >> 2] >> (8 * (j & 3))) & 0xFF] = (255u - (16 * lr + j)) << 16;   // time -1-q, biased by 256
        bm[2 * lr] = 0; bm[2 * lr + 1] = 0;
        reinterpret_cast<uint32_t *>(cum)[lr] = 0;
        if (WITH_HIST) for (int i = lr; i < 128; i += 16) s_hist[slot][i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t mybit = 1u << lr, below = mybit - 1u;
    uint32_t sym_next = src[lr < C ? lr : 0u];
    // seg0 = first position of the current segment (QUARTERS: a row is one segment long)
    for (uint32_t seg0 = 0; seg0 < Cmax; seg0 += MTF_SEG) {
        if (seg0) {
            // ---- re-base: the live timestamp of symbol c, t, becomes (live timestamps below t) = t - (killed below t) ----
            // (every lane takes 16 symbols of its row's table; the prefix counts are those of the segment's last batch)
#pragma unroll 4
            for (int k = 0; k < 16; k++) {
                const uint32_t c = 16u * (uint32_t)k + lr;         // (lane-consecutive symbols: consecutive banks)
                const uint32_t t = tab[c] >> 16, wd = t >> 6;
                const uint32_t kb = cum[wd] + (uint32_t)__popcll(bm[wd] << (63u - (t & 63u)));   // (bit t itself is live: not set)
                tab[c] = (t - kb) << 16;
            }
            __builtin_amdgcn_wave_barrier();
            bm[2 * lr] = 0; bm[2 * lr + 1] = 0;
            reinterpret_cast<uint32_t *>(cum)[lr] = 0;
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t seg_end = min(Cmax, seg0 + MTF_SEG);
        for (uint32_t base = seg0; base < seg_end; base += 16) {
            const uint32_t i = base + lr, lb = base - seg0;          // lb: the batch's first position inside the segment
            const bool valid = i < C;
            const uint32_t sym = sym_next;
            sym_next = src[i + 16 < C ? i + 16 : 0u];                // in flight during this batch
            // lanes of this row holding the same symbol: every lane ORs its bit into the symbol's entry, then reads it
            // back (LDS operations of a wave execute in order; OR is commutative, so no lane order is relied on)
            if (valid) atomicOr(&tab[sym], mybit);
            __builtin_amdgcn_wave_barrier();
            const uint32_t e = tab[sym];
            const uint32_t before = e & below;
            const bool hasprev = before != 0;
            const uint32_t p = 31u - (uint32_t)__builtin_clz(before | 1u);   // previous lane of the row with my symbol
            const bool last_in_batch = ((e & 0xFFFFu) >> lr) == 1u;
            // biased timestamp of the previous occurrence: Pb = (bit index in the segment's bitmap) + 1 > 0
            const uint32_t Pb = hasprev ? lb + p + 257u : (e >> 16) + 1u;
            // T = #{k < lr : P[k] < P[lr]}: lane lr meets P[lr-1], ... P[0] through DPP row_shr:1..15; each step is
            // (shifted P) - P with the borrow added up, and the 15 - lr steps that have no source lane read 0 < Pb
            uint32_t G = 0, t0;
#define GLC_SLIDE(K)                                                                                                \
            "v_sub_co_u32_dpp %1, vcc, %2, %2 row_shr:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
            "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
            asm volatile(GLC_SLIDE(1) GLC_SLIDE(2) GLC_SLIDE(3) GLC_SLIDE(4) GLC_SLIDE(5) GLC_SLIDE(6) GLC_SLIDE(7) GLC_SLIDE(8)
                         GLC_SLIDE(9) GLC_SLIDE(10) GLC_SLIDE(11) GLC_SLIDE(12) GLC_SLIDE(13) GLC_SLIDE(14) GLC_SLIDE(15)
                         : "+v"(G), "=&v"(t0) : "v"(Pb) : "vcc");
#undef GLC_SLIDE
            const uint32_t T = G + lr - 15u;
            const uint32_t bitx = Pb - 1u;                           // index into the killed-timestamp bitmap
            uint32_t o;
            if (hasprev) o = T - (p + 1u);
            else {
                // killed timestamps below bit r of word wd: bit r itself (the timestamp of MY previous occurrence) is still
                // alive -- this position kills it further down -- so "bits 0 .. r" counts the same and is one shift
                const uint32_t wd = bitx >> 6, r = bitx & 63;
                const uint32_t kb = cum[wd] + (uint32_t)__popcll(bm[wd] << (63u - r));
                o = T + kb + 255u - bitx;             // T + kb - (P + 1): the virtual timestamps add the 256 list entries
            }
            __builtin_amdgcn_wave_barrier();
            {   // timestamps of this batch killed inside the batch = the lanes that are not the last of their symbol: one
                // 16-bit store per row (nothing else can have touched that field yet) instead of same-word atomics
                const uint64_t nl = __ballot(valid && !last_in_batch);
                const uint32_t half = (row & 2) ? (uint32_t)(nl >> 32) : (uint32_t)nl;
                if (lr == 0) reinterpret_cast<uint16_t *>(bm)[(lb + 256u) >> 4] = (uint16_t)(half >> (16 * (row & 1)));
            }
            if (WITH_HIST && ZEROS) {
                // ZEROS (the blocks that come back from the other sorter tiers: text, logs): rank 0 is half and more of the output
                // behind the BWT of text, and sixteen lanes adding to ONE counter are sixteen passes of the LDS atomic unit -- the
                // row's zeros are counted with a ballot and added by one lane: 256 text blocks 12.25 -> 12.09 ms.  Not for the
                // bucket sorter's blocks: on Zipf bytes rank 0 is one output in nine, and the ballot made the kernel 6 % slower.
                const uint64_t zb = __ballot(valid && o == 0);
                const uint32_t zrow = (uint32_t)(zb >> (16u * row)) & 0xFFFFu;
                if (lr == 0 && zrow) atomicAdd(&s_hist[slot][0], (uint32_t)__builtin_popcount(zrow));
            }
            if (valid) {
                dst[i] = (uint8_t)o;
                // (ZEROS: rank 0 is counted per row with a ballot, above -- see the template parameter)
                if (WITH_HIST && !(ZEROS && o == 0)) atomicAdd(&s_hist[slot][o & 127], 1u << ((o >> 3) & 16));
                // timestamp P is killed by i: a DWORD atomic (a 64-bit ds_or with its 64-bit shift: 2.72 -> 2.54 ms per GiB without it;
                // prefix counts per dword instead of per 64-bit word, measured too, were slower: 2.66)
                if (!hasprev) atomicOr(&reinterpret_cast<uint32_t *>(bm)[bitx >> 5], 1u << (bitx & 31));
                if (last_in_batch) tab[sym] = (lb + lr + 256u) << 16;        // new last occurrence, lane bits cleared
            }
            __builtin_amdgcn_wave_barrier();
            // prefix counts of the killed-timestamp bitmap: 2 words per lane, scan across the row
            {
                const uint4 v = *reinterpret_cast<const uint4 *>(bm + 2 * lr);
                const uint32_t c0 = (uint32_t)__builtin_popcount(v.x) + (uint32_t)__builtin_popcount(v.y);
                const uint32_t sum = c0 + (uint32_t)__builtin_popcount(v.z) + (uint32_t)__builtin_popcount(v.w);
                uint32_t inc = sum;
                inc += GLC_DPP(inc, 0x111, 0xf);
                inc += GLC_DPP(inc, 0x112, 0xf);
                inc += GLC_DPP(inc, 0x114, 0xf);
                inc += GLC_DPP(inc, 0x118, 0xf);
                const uint32_t r0 = inc - sum;
                reinterpret_cast<uint32_t *>(cum)[lr] = r0 | ((r0 + c0) << 16);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (WITH_HIST && QUARTERS) {
        __builtin_amdgcn_wave_barrier();
        uint32_t *H = sub_hist + ((size_t)b * max_chunks + chunk0) * 256;
        for (int i = l; i < 256; i += 64) {
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < MTF_ROWS; r++) c += (s_hist[w * MTF_ROWS + r][i & 127] >> (16 * (i >> 7))) & 0xFFFFu;
            H[i] = c;
        }
    } else if (WITH_HIST && live) {
        __builtin_amdgcn_wave_barrier();
        uint32_t *H = sub_hist + ((size_t)b * max_chunks + chunk) * 256;
        for (int i = lr; i < 256; i += 16) H[i] = (s_hist[slot][i & 127] >> (16 * (i >> 7))) & 0xFFFFu;
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
                       uint8_t *out, size_t out_stride, MtfScratch &s, uint32_t *sub_hist, const uint32_t *only, bool skewed)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const uint32_t nchunks = (n + MTF_CHUNK - 1) / MTF_CHUNK;
    dim3 g((nchunks + MTF_WAVES - 1) / MTF_WAVES, nblk), t(MTF_WAVES * 64);
    dim3 ge((nchunks + MTF_WAVES * MTF_ROWS - 1) / (MTF_WAVES * MTF_ROWS), nblk);      // encode: 4 chunks per wave
    const double units = (double)n * nblk;
    int pi = s.prof ? s.prof->begin(PROF_MTF_LISTS, st) : -1;
    hipLaunchKernelGGL(k_mtf_chunk_lists, g, t, 0, st, in, in_stride, n, s.lists, s.lens, s.max_chunks, only);
    // 16 waves of 16 chunks: the shortest chain of folds (48) for a block on its own; 8 waves of 32 (72 folds): four
    // workgroups per CU, so a batch of 1024 blocks is resident at once instead of in two rounds (0.156 -> 0.126 ms)
    if (nblk <= 512)
        hipLaunchKernelGGL(k_mtf_scan_lists<16>, dim3(nblk), dim3(16 * 64), 0, st, s.lists, s.lens, n, s.max_chunks, only);
    else
        hipLaunchKernelGGL(k_mtf_scan_lists<8>, dim3(nblk), dim3(8 * 64), 0, st, s.lists, s.lens, n, s.max_chunks, only);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_MTF_ENCODE, st) : -1;
    // few chunks in all (a cudppCompress call, small batches): a wave per chunk, its rows the chunk's quarters
    const bool quarters = (uint64_t)nchunks * nblk <= MTF_QUARTERS_MAX_CHUNKS;
    if (quarters && sub_hist)
        hipLaunchKernelGGL((k_mtf_encode<true, true>), g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist, only);
    else if (quarters)
        hipLaunchKernelGGL((k_mtf_encode<false, true>), g, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist, only);
    else if (sub_hist && skewed)
        hipLaunchKernelGGL((k_mtf_encode<true, false, true>), ge, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist, only);
    else if (sub_hist)
        hipLaunchKernelGGL((k_mtf_encode<true, false>), ge, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist, only);
    else
        hipLaunchKernelGGL((k_mtf_encode<false, false>), ge, t, 0, st, in, in_stride, n, s.lists, s.max_chunks, out,
                           out_stride, sub_hist, only);
    if (pi >= 0) s.prof->end(pi, units, st);
    return hipGetLastError();
}

} // namespace glc
