// bwt_bucket.hip -- the fast suffix sorter: ONE bucketing pass over HBM, then every bucket is
// sorted to the end inside LDS.  gfx950 / wave64.
//
// Replaces (result-for-result, same SA / BWT bytes / index) the reference's
//   cudppSuffixArrayDispatch / ComputeSA     (cudpp-inpar/src/cudpp/app/sa_app.cu:125-298,365-391)
//   bwt_compute_final_kernel                 (kernel/compress_kernel.cuh:55-74)
// for data whose suffixes separate within a few dozen symbols (i.i.d. bytes, float data: configs 2 and 4).
// Blocks it cannot finish are flagged and go on to the second tier in this file, a string sample sort for text-like
// data (k_ss_*, further down); what that cannot finish either goes through the general sorter in bwt_sa.hip.
//
// Idea.  A suffix is mapped to X = the arithmetic code of its first FS_DEPTH = 8 symbols under the block's own
// order-0 symbol statistics:
//      y7 = C[s7];   y_d = C[s_d] + floor(p[s_d] * y_{d+1} / 2^32)  (d = 6..1);   X = C[s0] * 2^32 + p[s0] * y1
// (Six symbols until round 5: the word keeps 36 bits of X, ~six Zipf symbols' worth -- but the pairs that share six symbols are
//  the FREQUENT prefixes, whose interval in 36 bits is wide enough for a 7th and an 8th symbol to be told apart: equal codes went
//  from 0.4 % to 0.03 % of a block, and with them most of what k_fs_sort_bwt's rare path and k_fs_ties cost.)
// (C = exclusive cumulative frequency, p = frequency, both scaled to 2^32; symbols past the end of the
// block contribute C = p = 0).  Two properties carry the whole design:
//   * X is monotone: suffix a < suffix b lexicographically  =>  X(a) <= X(b)   (each step maps the
//     sub-intervals of smaller symbols below those of larger ones, and floor() is monotone), so
//     DIFFERENT codes are always in the right order and only EQUAL codes need a look at the text;
//   * for i.i.d. data X is uniformly distributed whatever the symbol distribution is (Zipf included),
//     so "top 9 bits of X" cuts a 1 MiB block into 512 buckets of 2048 +- 7 % suffixes WITHOUT any
//     histogram of the buckets, and "next 12 bits" cuts a bucket into bins of ~0.5 suffixes.
// One word per suffix = [top 36 bits of X | suffix index : 20 | T[i-1] : 8]: the BWT byte rides along, so
// nothing is gathered afterwards.
//
//   k_fs_hist / k_fs_tables   symbol histogram of the block -> {C, p} table                 1 B read / suffix
//   k_fs_part                 text tile -> words -> bucketed in LDS -> appended to the bucket's slot
//                             (space reserved with one global atomic per (tile, bucket))      1 B R + 8 B W
//   k_fs_scan                 bucket fill -> rank base of every bucket, overflow -> flag
//   k_fs_sort                 one workgroup per bucket: counting sort on 12 more bits with LDS atomics
//                             (no stability needed: the index is not a tie-breaker), direct ranking inside
//                             the tiny bins; writes BWT bytes (+ SA) straight to their final rows;
//                             runs of equal codes go to a work list                             8 B R + 1 B W
//   k_fs_ties                 one thread per member of such a run: rank by comparing the suffixes in the
//                             text (0.4 % of the suffixes of a Zipf block)
// = 19 B of HBM traffic per input byte (23 with the suffix array) where the 5-pass LSD sorter moves ~105.
#include "glc_device.h"
#include "glc_internal.h"
#include "huff_tree.h"                                         // wave_min_u64
#include <stdlib.h>

namespace glc {

#ifndef GLC_FS_DEPTH
#define GLC_FS_DEPTH 8
#endif
constexpr int      FS_DEPTH  = GLC_FS_DEPTH;        // symbols the arithmetic code of a suffix is made of (see fs_code_at)
#ifndef GLC_SS_DEPTH
#define GLC_SS_DEPTH GLC_FS_DEPTH
#endif
constexpr int      SS_DEPTH  = GLC_SS_DEPTH;        // ... in the sample sorter's own words (k_ss_sample's samples and k_fs_part<true> must agree; the tiers need not)
static_assert(FS_DEPTH >= 2 && FS_DEPTH <= 8 && SS_DEPTH >= 2 && SS_DEPTH <= 8, "a thread's 8 codes take their symbols from its 16 staged bytes");
constexpr int      FSP_NT    = 512;                 // k_fs_part: threads
constexpr int      FSP_ITEMS = 8;
constexpr int      FSP_TILE  = FSP_NT * FSP_ITEMS;  // suffixes per tile
constexpr int      FSS_NT    = 512;                 // k_fs_sort: threads
constexpr int      FSS_ITEMS = FS_CAP / FSS_NT;     // slots per thread
#ifndef GLC_FS_BIN_BITS
#define GLC_FS_BIN_BITS 12
#endif
constexpr uint32_t FS_BIN_BITS = GLC_FS_BIN_BITS, FS_BINS = 1u << FS_BIN_BITS;
#ifndef GLC_FSS_LOOK
#define GLC_FSS_LOOK 4
#endif
constexpr uint32_t FSS_LOOK = GLC_FSS_LOOK;         // k_fs_sort_bwt's rank step: words from a bin's start compared in straight-line code
constexpr uint32_t FS_MAX_GROUP = 512;              // longest run of equal codes ranked by direct count
constexpr uint64_t FS_LOW_MASK = (1ull << 28) - 1;  // [index : 20 | bwt : 8]
constexpr uint32_t SS_CELLS = 4096;                 // sample tier: cells of the code space (leading 12 bits) that index the splitters

// at least 16 buckets: k_fs_sort compares bits 28..59 of the words, so the four bits above must be bucket number
uint32_t fs_bucket_log2(uint32_t n)
{
    uint32_t l = 4;
    while (((uint64_t)FS_AVG << l) < n && l < FS_MAXNB_LOG2) l++;
    return l;
}

// ---------------------------------------------------------------------------
// symbol histogram: 8 copies per workgroup (copy = lane & 7, stride 257 words so equal symbols of
// different copies sit in different banks): equal symbols inside a wave serialise on an LDS atomic,
// and Zipf data puts 10 lanes of 64 on the same symbol
// ---------------------------------------------------------------------------
constexpr uint32_t FSH_SLICE = 32768;
// Text-likeness probe, free of charge inside the histogram pass: two of a thread's eight 16-byte vectors give a
// 6-gram each (512 samples per 32 KB slice); a sample whose 6-gram another sample of the slice has already put into a
// 2048-slot LDS table is a REPEAT.  I.i.d. bytes (Zipf(1.0), float data, random) repeat ~0 times per 1 MiB block, text and
// log lines hundreds to thousands of times: exactly the blocks whose order-0 code is lumpy (a frequent 6-gram = one code
// shared by thousands of suffixes), which the bucket sorter would flag after a wasted attempt.  k_fs_tables flags a block
// with FS_DUP_FLAG repeats or more up front: its tiles leave k_fs_part / k_fs_sort / k_fs_ties at once and the sample sorter
// takes it.  A wrong guess costs time, never correctness (the other way round, the attempt flags the block as before).
constexpr uint32_t FSH_SLOTS = 2048, FS_DUP_FLAG = 48;

// {C, p} scaled to 2^32.  floor() on both keeps C[s] + p[s] <= C[s+1], which is what makes the code monotone.
constexpr uint32_t FS_DONE = 8u;                    // flag value: the block is finished (a constant block: k_fs_tables wrote its rows)

__global__ __launch_bounds__(256) void k_fs_tables(const uint32_t *__restrict__ hist, uint32_t n,
                                                   uint2 *__restrict__ tab, const uint32_t *__restrict__ dup,
                                                   uint32_t *__restrict__ flag, const uint8_t *__restrict__ text, size_t stride,
                                                   uint8_t *__restrict__ bwt_out, size_t bwt_stride, int *__restrict__ d_index,
                                                   uint32_t *__restrict__ sa_out, size_t sa_stride, uint32_t step)
{
    __shared__ uint32_t s_tmp[5];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t hraw = hist[(size_t)b * 256 + tid];
    // step > 1: the counts are those of every step-th 32 KB slice of the block (k_fs_hist).  The code is monotone whatever the
    // table is as long as C[s] + p[s] <= C[s + 1] holds for every symbol THAT OCCURS; a SAMPLE'S statistics cut the block into
    // buckets as evenly as the block's own, within the sampling noise.  Every symbol gets one count on top of the sample's
    // (0.1 % of the code space): a symbol the sample missed keeps a positive width and its place in the order -- with width 0
    // the symbols above the sample's last one would sit at C = 2^32, clamped BELOW the end of that last symbol's interval
    // (fuzz seed 2: a block whose sampled slices held two symbols, 79 196 bytes of 254 others in between).
    const uint32_t h = hraw + (step > 1 ? 1u : 0u);
    uint32_t ns = 0;
    const uint32_t c = block_excl_add<256>(h, s_tmp, &ns);
    // A block of ONE symbol (zero pages, padding) has nothing to sort: SA = n-1 .. 0, every BWT byte is the symbol, the
    // index row is n - 1.  Left to the tiers it is their worst case -- every suffix ties with every other for the whole
    // block: bucket overflow, the sample sorter's depth cap, then ~20 prefix-doubling rounds of the general sorter (1.3 ms
    // where a Zipf block takes 0.012).  FS_DONE overrides whatever the flag was (text-likeness, a caller's "sample sorter
    // first"): every tier skips a flagged block, and k_fs_finish does not list this one for anybody.  Its rows are written
    // right here, by this workgroup (a kernel of its own was one more launch in every call's chain).
    bool constant = __syncthreads_or((int)(hraw == ns - (step > 1 ? 256u : 0u))) != 0;
    if (constant && step > 1) {                                // (uniform) one symbol in the SAMPLE: look at the whole block
        const uint8_t *T = text + (size_t)b * stride;
        const uint32_t sym = T[0];
        bool other = false;
        for (uint32_t i = tid; i < n && !other; i += 256) other = T[i] != sym;
        constant = __syncthreads_or((int)other) == 0;
    }
    if (tid == 0) {
        if (constant) flag[b] = FS_DONE;
        else if (dup[b] >= (FS_DUP_FLAG + step - 1) / step) flag[b] = 1u;   // text-like (see k_fs_hist): straight to the sample sorter
    }
    if (constant) {
        const uint32_t sym = text[(size_t)b * stride];
        const uint32_t v = sym * 0x01010101u;
        if (bwt_out) {
            uint8_t *O = bwt_out + (size_t)b * bwt_stride;
            const uint32_t head = min(n, (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(O) & 15u)) & 15u));
            const uint32_t nvec = (n - head) / 16;
            for (uint32_t i = tid; i < head; i += 256) O[i] = (uint8_t)sym;
            for (uint32_t i = tid; i < nvec; i += 256) reinterpret_cast<uint4 *>(O + head)[i] = make_uint4(v, v, v, v);
            for (uint32_t i = head + nvec * 16 + tid; i < n; i += 256) O[i] = (uint8_t)sym;
        }
        if (sa_out) for (uint32_t i = tid; i < n; i += 256) sa_out[(size_t)b * sa_stride + i] = n - 1 - i;
        if (d_index && tid == 0) d_index[b] = (int)(n - 1);
        return;                                                // (uniform; the table of a flagged block is never read)
    }
    const uint64_t C32 = ((uint64_t)c << 32) / ns, P32 = ((uint64_t)h << 32) / ns;
    tab[(size_t)b * 256 + tid] = make_uint2((uint32_t)(C32 > 0xFFFFFFFFull ? 0xFFFFFFFFull : C32),
                                            (uint32_t)(P32 > 0xFFFFFFFFull ? 0xFFFFFFFFull : P32));
}

#ifndef GLC_FSH_COPIES
#define GLC_FSH_COPIES 16
#endif
constexpr int FSH_COPIES = GLC_FSH_COPIES;             // LDS copies of the histogram (same-symbol atomics of a wave spread over them)
__global__ __launch_bounds__(256) void k_fs_hist(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                 uint32_t *__restrict__ hist, uint32_t *__restrict__ dup, uint32_t step)
{
    __shared__ uint32_t s_h[FSH_COPIES * 257];
    __shared__ uint32_t s_fp[FSH_SLOTS];
    __shared__ uint32_t s_dup;
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    const uint32_t lo = blockIdx.x * step * FSH_SLICE;        // (step > 1: every step-th slice -- see k_fs_tables)
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + FSH_SLICE);
    for (uint32_t i = tid; i < FSH_COPIES * 257; i += 256) s_h[i] = 0;
    for (uint32_t i = tid; i < FSH_SLOTS; i += 256) s_fp[i] = 0;
    if (tid == 0) s_dup = 0;
    __syncthreads();
    const uint8_t *T = text + (size_t)b * stride;
    uint32_t *H = s_h + (tid & (FSH_COPIES - 1)) * 257;
    uint32_t done = lo;
    if ((reinterpret_cast<uintptr_t>(T + lo) & 15) == 0) {
        const uint32_t nvec = (hi - lo) / 16;                       // <= 2048 = 8 per thread
        const uint4 *V = reinterpret_cast<const uint4 *>(T + lo);
        uint4 q[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = r * 256 + tid;
            q[r] = i < nvec ? V[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 8; r += 4) {
            if (r * 256 + tid < nvec) {
                // fingerprint of the vector's first six bytes (never 0); equal 6-grams give equal fingerprints and slots
                const uint32_t fp = (q[r].x * 0x9E3779B1u + (q[r].y & 0xFFFFu) * 0x85EBCA6Bu) | 1u;
                const uint32_t old = atomicCAS(&s_fp[(fp * 0xC2B2AE35u) >> 21], 0u, fp);
                if (old == fp) atomicAdd(&s_dup, 1u);
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (r * 256 + tid < nvec) {
                const uint32_t wd[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    atomicAdd(&H[wd[k] & 0xFF], 1u);
                    atomicAdd(&H[(wd[k] >> 8) & 0xFF], 1u);
                    atomicAdd(&H[(wd[k] >> 16) & 0xFF], 1u);
                    atomicAdd(&H[wd[k] >> 24], 1u);
                }
            }
        }
        done = lo + nvec * 16;
    }
    for (uint32_t i = done + tid; i < hi; i += 256) atomicAdd(&H[T[i]], 1u);
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < FSH_COPIES; k++) c += s_h[k * 257 + tid];
    if (c) atomicAdd(&hist[(size_t)b * 256 + tid], c);
    if (tid == 0 && s_dup) atomicAdd(&dup[b], s_dup);
}


// ---------------------------------------------------------------------------
// suffix comparison in the text (runs of equal codes; the sample tier's splitters)
// ---------------------------------------------------------------------------
// (FS_LCP_CAP, glc_internal.h: a longer common prefix flags the block as deep)

// 8 bytes at any address as a big-endian number: ONE unaligned 8-byte load (global memory takes any alignment on
// gfx9+; built from aligned dwords it is three scattered loads per lane, and the gathers of the refinement
// rounds are bound by the number of addresses the texture path takes per clock)
__device__ __forceinline__ uint64_t fs_load_be64(const uint8_t *p)
{
    uint64_t x;
    __builtin_memcpy(&x, p, 8);
    return __builtin_bswap64(x);
}

// 16 bytes at any address as two big-endian numbers: ONE unaligned 16-byte load
__device__ __forceinline__ void fs_load_be128(const uint8_t *p, uint64_t &hi, uint64_t &lo)
{
    uint64_t x[2];
    __builtin_memcpy(x, p, 16);
    hi = __builtin_bswap64(x[0]);
    lo = __builtin_bswap64(x[1]);
}

// suffix a < suffix b ?  (a != b; the shorter of two suffixes that agree to the end of one is the smaller)
// tol (the sample sorter's second form, for blocks with repeats deeper than its cap: see ss_build): two suffixes that
// agree in their first SS_TOL_CAP + 8 bytes are ordered by their POSITIONS -- a total order that every comparison of the
// pass agrees on, and a (SS_TOL_CAP)-order of the suffixes, which is all the prefix-doubling rounds behind it need.
template <bool W16 = false>
__device__ __forceinline__ bool fs_suffix_less(const uint8_t *T, uint32_t n, uint32_t a, uint32_t b, bool *deep, uint32_t k = 0,
                                               bool tol = false)
{
    for (;;) {
        const uint32_t m = max(a, b) + k;
#ifndef GLC_TOL_STEP8
        if (W16 && m + 20 <= n) {                              // (every tolerant comparison starts at k = 0 or 16: all of them end their ties at k = 144)
#else
        if (W16 && !tol && m + 20 <= n) {
#endif
            const uint64_t va = fs_load_be64(T + a + k), vb = fs_load_be64(T + b + k);
            const uint64_t va2 = fs_load_be64(T + a + k + 8), vb2 = fs_load_be64(T + b + k + 8);
            if (va != vb) return va < vb;
            if (va2 != vb2) return va2 < vb2;
            k += 16;
        } else if (m + 12 <= n) {
            const uint64_t va = fs_load_be64(T + a + k), vb = fs_load_be64(T + b + k);
            if (va != vb) return va < vb;
            k += 8;
        } else {
            if (a + k >= n || b + k >= n) return a > b;        // one (or both: called with k > 0) ended: the shorter suffix is the smaller
            const uint32_t ca = T[a + k], cb = T[b + k];
            if (ca != cb) return ca < cb;
            k++;
        }
        if (tol && k > SS_TOL_CAP) { *deep = true; return a < b; }   // (*deep: "agreed up to the cap" -- a tie, not a give-up, for these callers)
        if (k > FS_LCP_CAP) { *deep = true; return false; }
    }
}

// ---------------------------------------------------------------------------
// bucketing pass
// ---------------------------------------------------------------------------
// SPLIT = the sample tier's form: blocks come from a list, and the bucket of a word is found among the block's
// splitter suffixes (code first, text on equal codes) instead of in the top bits of the code.
// sp8[k] / w8: the first 8 text bytes of splitter k / of the word's suffix (big-endian, 0 past the end of the text): on
// equal codes they decide most comparisons without going to the text (text-like blocks are exactly those with thousands
// of suffixes under one code).  sp16[k] / wtxt[8 .. 16): the NEXT 8 bytes, the word's still in the staged tile -- log lines
// share more than 8 bytes with the splitters around them all the time ("2026-09-28T12:3", " host-17 svc-"), and every such
// tie was a walk through the text by one lane with its wave waiting: 2.1 of the kernel's 3.3 ms per 256 log blocks.
// `pending` (k_fs_part<true>'s first go over a thread's words): where the search would have to WALK through the text -- word and
// splitter agree in code and in 16 text bytes: one lane in twenty on text, and its wave waits a chain of scattered loads for it,
// eight times per tile -- it stops instead and hands back its interval (*pending = 1 << 31 | hi << 10 | lo); the walks of a
// tile are then taken together, a lane each (ss_search from that interval on, pending = nullptr).
__device__ __forceinline__ uint32_t ss_search(const uint64_t *sp, const uint64_t *sp8, const uint64_t *sp16, uint32_t lo, uint32_t hi,
                                              uint64_t w, uint64_t w8, const uint8_t *wtxt, const uint8_t *T, uint32_t n,
                                              bool *deep, bool tol, uint32_t *pending)
{
    const uint64_t cw = w >> 28;
    const uint32_t iw = (uint32_t)(w >> 8) & 0xFFFFFu;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t sw = sp[mid], cs = sw >> 28;
        bool le;                                               // splitter[mid] <= w ?
        if (cs != cw) le = cs < cw;
        else {
            const uint32_t is = (uint32_t)(sw >> 8) & 0xFFFFFu;
            const uint64_t s8 = sp8[mid];
            if (s8 != w8) le = s8 < w8;
            else if (is == iw) le = true;
            else {
                uint64_t w16 = 0;
#pragma unroll
                for (int t = 8; t < 16; t++) w16 = (w16 << 8) | wtxt[t];
                const uint64_t s16 = sp16[mid];
                if (s16 != w16) le = s16 < w16;
                else {
                    if (pending) { *pending = 0x80000000u | (hi << 10) | lo; return lo; }
                    le = !fs_suffix_less<true>(T, n, iw, is, deep, 16, tol);
                }
            }
        }
        if (le) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t ss_bucket(const uint64_t *sp, const uint64_t *sp8, const uint64_t *sp16, const uint16_t *cell,
                                              uint64_t w, uint64_t w8, const uint8_t *wtxt, const uint8_t *T, uint32_t n,
                                              bool *deep, bool tol, uint32_t *pending = nullptr)
{
    // the splitters whose code starts with the same 12 bits are the only ones to look at (cell[x] = first splitter,
    // counted from 1, whose leading 12 code bits are >= x): mostly none or one
    const uint32_t x = (uint32_t)(w >> 52);
    const uint32_t lo = (uint32_t)cell[x] - 1u, hi = cell[x + 1];    // the answer is in [lo, hi): splitter[lo] <= w < splitter[hi]
    return ss_search(sp, sp8, sp16, lo, hi, w, w8, wtxt, T, n, deep, tol, pending);
}

template <bool SPLIT>
__global__ __launch_bounds__(FSP_NT) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_fs_part(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                    uint32_t nbl, const uint2 *__restrict__ tab,
                                                    uint64_t *__restrict__ keys, size_t kstride,
                                                    uint32_t *__restrict__ fill, uint32_t *__restrict__ flag,
                                                    const uint32_t *__restrict__ list, const uint64_t *__restrict__ split,
                                                    const uint16_t *__restrict__ cell, const uint64_t *__restrict__ split8,
                                                    uint32_t *__restrict__ zero_bucket, bool tol,
                                                    const uint64_t *__restrict__ split16 = nullptr)
{
    __shared__ uint32_t s_cnt[FS_MAXNB], s_start[FS_MAXNB], s_gbase[FS_MAXNB];
    __shared__ uint64_t s_w[FSP_TILE];
    __shared__ uint32_t s_tmp[FSP_NT / 64 + 1];
    __shared__ uint16_t s_bk[SPLIT ? FSP_TILE : 1];            // (SPLIT) bucket of the word at a position: not in the word's top bits there
    constexpr uint32_t NDEF = SPLIT ? 512 : 1;                 // (SPLIT) searches of a tile that stopped before a walk through the text
    __shared__ uint64_t s_dw[NDEF];                            // ... their words
    __shared__ uint32_t s_dl[NDEF];                            // ... tile position : 12 | hi : 10 | lo : 10, then bucket << 16 | rank
    __shared__ uint32_t s_dn;
    __shared__ uint32_t s_flagged;
    // the symbol table and the staged text are dead before the first word is bucketed: they live inside s_w
    // (38 KB instead of 44 KB of LDS: 4 workgroups per CU instead of 3)
    uint2 *s_tab = reinterpret_cast<uint2 *>(s_w);
    uint8_t *s_txt = reinterpret_cast<uint8_t *>(s_w) + 256 * sizeof(uint2);   // s_txt[k] = T[base - 1 + k]; 16-byte aligned
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, so physical workgroup p runs on XCD p & 7.  The
    // tiles of a block are made consecutive on ONE XCD (logical index = first of the XCD's share + p / 8): their appends to a bucket's
    // slot are neighbours in that XCD's L2 and leave it as whole lines, and the block's fill counters stay in one L2.
    uint32_t bx, by;
    {
        const uint32_t tiles = gridDim.x, total = tiles * gridDim.y, p = blockIdx.y * tiles + blockIdx.x;
        const uint32_t q = total >> 3, r = total & 7u, x = p & 7u;             // XCD x takes q + (x < r) tiles, in order
        const uint32_t lg = x * q + min(x, r) + (p >> 3);
        by = lg / tiles; bx = lg % tiles;
    }
    const uint32_t b = SPLIT ? list[by] : by, tid = threadIdx.x, base = bx * FSP_TILE;
    if (base >= n) return;
    // SPLIT: given up while sampling, no splitters to search; otherwise: flagged up front as text-like (k_fs_tables), or
    // by a tile of this launch whose bucket overflowed -- the block is another sorter's either way.  One read by one
    // thread (other tiles of this launch may flag the block meanwhile), looked at behind the staging barrier below.
    if (tid == 0) { s_flagged = flag[b]; s_dn = 0; }
    const uint8_t *T = text + (size_t)b * stride;
    uint64_t *s_split = s_w + 1024;                            // (SPLIT) behind the table and the staged text, dead with them
    uint16_t *s_cell = reinterpret_cast<uint16_t *>(s_w + 1024 + FS_MAXNB);
    uint64_t *s_split8 = s_w + 1024 + FS_MAXNB + (SS_CELLS + 2 + 3) / 4 + 1;    // behind the cell table (8196 bytes)
    uint64_t *s_split16 = s_split8 + FS_MAXNB;                 // (3586 of s_w's 4096 words in all)
    if (SPLIT) {
        for (uint32_t i = tid; i < (1u << nbl); i += FSP_NT) {
            s_split[i] = split[(size_t)b * FS_MAXNB + i];
            s_split8[i] = split8[(size_t)b * FS_MAXNB + i];
            s_split16[i] = split16[(size_t)b * FS_MAXNB + i];
        }
        for (uint32_t i = tid; i < SS_CELLS + 2; i += FSP_NT) s_cell[i] = cell[(size_t)b * (SS_CELLS + 2) + i];
    }
    if (tid < 256) s_tab[tid] = tab[(size_t)b * 256 + tid];
    if (tid < FS_MAXNB) s_cnt[tid] = 0;
    constexpr uint32_t STG = FSP_TILE + (SPLIT ? 24 : 16);     // staged bytes: T[base - 1 ..]; SPLIT looks 16 bytes into a suffix
    const bool edge = base + STG > n;
    if (base > 0 && !edge && (reinterpret_cast<uintptr_t>(T) & 3) == 0) {
        // aligned dwords of T[base - 4 ...], shifted by 3 bytes on the way into LDS
        const uint32_t *D = reinterpret_cast<const uint32_t *>(T + base - 4);
        uint32_t lo[3], hi[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint32_t q = r * FSP_NT + tid;
            const bool in = q < STG / 4;
            lo[r] = in ? D[q] : 0u; hi[r] = in ? D[q + 1] : 0u;
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint32_t q = r * FSP_NT + tid;
            if (q < STG / 4)
                reinterpret_cast<uint32_t *>(s_txt)[q] = __builtin_amdgcn_alignbyte(hi[r], lo[r], 3);
        }
    } else {
        for (uint32_t k = tid; k < STG; k += FSP_NT) {
            const int64_t g = (int64_t)base - 1 + k;
            s_txt[k] = g < 0 ? T[n - 1] : (g < (int64_t)n ? T[g] : (uint8_t)0);
        }
    }
    __syncthreads();
    if (s_flagged) return;
    // thread = 8 consecutive suffixes gi0 .. gi0+7; byte j of its 16 staged bytes is T[gi0 - 1 + j]
    const uint32_t k0 = tid * FSP_ITEMS, gi0 = base + k0;
    const uint2 qa = *reinterpret_cast<const uint2 *>(s_txt + k0), qb = *reinterpret_cast<const uint2 *>(s_txt + k0 + 8);
    const uint32_t by4[4] = {qa.x, qa.y, qb.x, qb.y};
#define FS_BYTE(j) ((by4[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu)
    constexpr int DEPTH = SPLIT ? SS_DEPTH : FS_DEPTH;
    uint2 e[FSP_ITEMS + DEPTH - 1];                            // table entries of the symbols the 8 codes share
#pragma unroll
    for (int k = 0; k < FSP_ITEMS + DEPTH - 1; k++) {
        const uint2 t = s_tab[FS_BYTE(1 + k)];
        e[k] = (edge && gi0 + k >= n) ? make_uint2(0u, 0u) : t;
    }
    uint64_t w[FSP_ITEMS];
    uint32_t br[FSP_ITEMS];                                    // bucket << 16 | rank inside (tile, bucket)
#pragma unroll
    for (int j = 0; j < FSP_ITEMS; j++) {
        uint32_t y = e[j + DEPTH - 1].x;
#pragma unroll
        for (int d = DEPTH - 2; d >= 1; d--) y = e[j + d].x + __umulhi(e[j + d].y, y);
        const uint64_t X = ((uint64_t)e[j].x << 32) + (uint64_t)e[j].y * y;
        const uint32_t gi = gi0 + j;
        w[j] = (X & ~FS_LOW_MASK) | ((uint64_t)gi << 8) | FS_BYTE(j);
        uint32_t bk;
        if (SPLIT) {
            bool deep = false;
            // bytes j + 1 .. j + 8 of the 16 staged ones = the suffix's first 8 text bytes
            const uint32_t o0 = j + 1, o1 = j + 5;
            const uint32_t d0 = (o0 & 3) ? __builtin_amdgcn_alignbyte(by4[(o0 >> 2) + 1], by4[o0 >> 2], o0 & 3) : by4[o0 >> 2];
            const uint32_t d1 = (o1 & 3) ? __builtin_amdgcn_alignbyte(by4[min((o1 >> 2) + 1, 3u)], by4[o1 >> 2], o1 & 3) : by4[o1 >> 2];
            const uint64_t w8 = ((uint64_t)__builtin_bswap32(d0) << 32) | __builtin_bswap32(d1);
            uint32_t pend = 0;
            bk = gi < n ? ss_bucket(s_split, s_split8, s_split16, s_cell, w[j], w8, s_txt + k0 + j + 1, T, n, &deep, tol, &pend) : 0u;
            if (deep && !tol) atomicOr(&flag[b], 2u);
            if (pend) {
                const uint32_t at = atomicAdd(&s_dn, 1u);
                if (at < NDEF) {
                    s_dw[at] = w[j];
                    s_dl[at] = ((k0 + j) << 20) | (pend & 0xFFFFFu);
                    br[j] = 0x80000000u | at;                  // (the bucket and the rank come with the tile's walks, below)
                    continue;
                }
                // (no room on the list: walked here and now)
                bk = ss_search(s_split, s_split8, s_split16, pend & 0x3FFu, (pend >> 10) & 0x3FFu, w[j], w8, s_txt + k0 + j + 1, T, n, &deep, tol, nullptr);
                if (deep && !tol) atomicOr(&flag[b], 2u);
            }
        } else bk = nbl ? (uint32_t)(X >> (64 - nbl)) : 0u;
        br[j] = (bk << 16) | (gi < n ? atomicAdd(&s_cnt[bk], 1u) : 0u);
        if (!SPLIT && j == 0 && gi == 0) zero_bucket[b] = bk;  // where the word of suffix 0 goes: k_fs_sort_bwt looks for the BWT index there only
    }
#undef FS_BYTE
    if (SPLIT) {
        // the tile's walks, a lane each
        __syncthreads();
        const uint32_t nd = min(s_dn, NDEF);
        for (uint32_t e = tid; e < nd; e += FSP_NT) {
            const uint32_t dl = s_dl[e], k = dl >> 20;
            const uint64_t ww = s_dw[e];
            const uint8_t *wt = s_txt + k + 1;                 // the suffix's first 16 text bytes, still staged
            uint64_t w8 = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) w8 = (w8 << 8) | wt[t];
            bool deep = false;
            const uint32_t bk = ss_search(s_split, s_split8, s_split16, dl & 0x3FFu, (dl >> 10) & 0x3FFu, ww, w8, wt, T, n, &deep, tol, nullptr);
            if (deep && !tol) atomicOr(&flag[b], 2u);
            s_dl[e] = (bk << 16) | atomicAdd(&s_cnt[bk], 1u);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FSP_ITEMS; j++)
            if (br[j] & 0x80000000u) br[j] = s_dl[br[j] & 0x7FFFFFFFu];
    }
    __syncthreads();
    {
        const uint32_t c = tid < FS_MAXNB ? s_cnt[tid] : 0u;
        const uint32_t start = block_excl_add<FSP_NT>(c, s_tmp);
        uint32_t g = 0;
        if (c) {
            g = atomicAdd(&fill[(size_t)b * FS_MAXNB + tid], c);
            if (g + c > FS_FILLMAX) atomicOr(&flag[b], 1u);
        }
        if (tid < FS_MAXNB) { s_start[tid] = start; s_gbase[tid] = g; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FSP_ITEMS; j++)
        if (gi0 + j < n) {
            const uint32_t q = s_start[br[j] >> 16] + (br[j] & 0xFFFFu);
            s_w[q] = w[j];
            if (SPLIT) s_bk[q] = (uint16_t)(br[j] >> 16);
        }
    __syncthreads();
    const uint32_t tile_n = min((uint32_t)FSP_TILE, n - base);
    uint64_t *K = keys + (size_t)b * kstride;
#pragma unroll
    for (int r = 0; r < FSP_ITEMS; r++) {
        const uint32_t p = r * FSP_NT + tid;
        if (p < tile_n) {
            const uint64_t ww = s_w[p];
            uint32_t d;
            if (SPLIT) {
                d = s_bk[p];
            } else d = nbl ? (uint32_t)(ww >> (64 - nbl)) : 0u;
            const uint32_t off = s_gbase[d] + (p - s_start[d]);
            if (off < FS_CAP) K[(size_t)d * FS_CAP + off] = ww;
        }
    }
}

// k_fs_part for the bucket sorter's own pass (no splitters), FSP2_T consecutive tiles of a block per workgroup.  The counters
// put k_fs_part at 0.45 of the VALU issue rate, 0.19 of the scalar one and the LDS 0.38 busy -- nothing is saturated: a
// tile's 12.7 us are a chain  flag + table + text from memory -> codes -> scan -> 512 global atomics with return -> scatter
// -> stores, and four workgroups per CU do not cover its waits.  The loop over tiles lets
//   * the symbol table and the block's flag arrive once per workgroup;
//   * the text of tile t + 1 be requested before tile t is touched (six dwords per thread in registers), so it has the
//     whole of tile t to arrive;
//   * the global atomic of a (tile, bucket) be ISSUED before the words are scattered in LDS and CONSUMED after (the
//     scatter needs the tile-local ranks only), with LDS-only barriers in between so that nobody waits for it early;
//   * the stores of tile t retire under tile t + 1 (the prefetched text is taken out of its registers before they are issued:
//     loads and stores share one counter, in order).
#ifndef GLC_FSP2_T
#define GLC_FSP2_T 4
#endif
constexpr int FSP2_T = GLC_FSP2_T;
#ifndef GLC_FSP2_WAVES
#define GLC_FSP2_WAVES 5
#endif
#ifndef GLC_FSP2_NT
#define GLC_FSP2_NT 512
#endif
#ifndef GLC_FSP2_TILE
#define GLC_FSP2_TILE 4096
#endif
constexpr int FSP2_TILE = GLC_FSP2_TILE;                       // suffixes per tile of k_fs_part2
constexpr int FSP2_NT = GLC_FSP2_NT;                           // threads per workgroup; a thread takes 4096 / FSP2_NT consecutive suffixes

__device__ __forceinline__ void lds_only_barrier()
{
    // __syncthreads() carries a workgroup-scope fence: every wave would wait for ALL its outstanding memory operations,
    // the prefetched text and the atomic in flight included
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NT, int ITEMS, int WPE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_fs_part2(const uint8_t *__restrict__ text, size_t stride, uint32_t n, uint32_t nbl,
                                                     const uint2 *__restrict__ tab, uint64_t *__restrict__ keys, size_t kstride,
                                                     uint32_t *__restrict__ fill, uint32_t *__restrict__ flag,
                                                     uint32_t *__restrict__ zero_bucket, uint32_t tiles_per_wg)
{
    constexpr uint32_t TILE = NT * ITEMS;                      // suffixes per tile
    __shared__ uint32_t s_cnt[FS_MAXNB];
    __shared__ uint16_t s_start[FS_MAXNB], s_gbase[FS_MAXNB];
    __shared__ uint64_t s_w[NT * ITEMS];
    __shared__ uint2 s_tab[256];
    __shared__ uint32_t s_tmp[NT / 64 + 1];
    __shared__ uint32_t s_flagged;
    uint8_t *s_txt = reinterpret_cast<uint8_t *>(s_w);         // s_txt[k] = T[base - 1 + k]: dead before the first word is bucketed
    uint32_t bx, by;
    xcd_order(bx, by);                                         // a block's tiles on ONE XCD, back to back
    const uint32_t b = by, tid = threadIdx.x;
    const uint32_t ntiles = (n + TILE - 1) / TILE, tile0 = bx * tiles_per_wg;
    if (tile0 >= ntiles) return;
    const uint8_t *T = text + (size_t)b * stride;
    if (tid == 0) s_flagged = flag[b];                         // flagged up front as text-like, or by a tile that ran before
    if (tid < 256) s_tab[tid] = tab[(size_t)b * 256 + tid];
    // text of a tile as the dwords of T[base - 1 ...] (unaligned 4-byte loads: global memory takes any alignment): dword
    // q = r NT + tid of the 4112 staged bytes.  Only for inner tiles; the first and the last tile of a block take the byte loop.
    constexpr int NSTG = ((TILE + 16) / 4 + NT - 1) / NT;      // staged dwords per thread
    uint32_t stg[NSTG] = {};
    auto inner = [&](uint32_t tile) { const uint32_t base = tile * TILE; return base > 0 && base + TILE + 16 <= n; };
    auto request = [&](uint32_t tile) {
        const uint8_t *D = T + (size_t)tile * TILE - 1;
#pragma unroll
        for (int r = 0; r < NSTG; r++) {
            const uint32_t q = r * NT + tid;
            const uint32_t qq = q < (TILE + 16) / 4 ? q : 0u;   // (every load is issued: a conditional one may sink to its use)
            uint32_t v;
            __builtin_memcpy(&v, D + 4 * (size_t)qq, 4);
            stg[r] = v;
        }
    };
    const uint32_t tend = min(ntiles, tile0 + tiles_per_wg);
    bool have = false;                                         // stg holds the text of the tile about to be processed
    if (inner(tile0)) { request(tile0); have = true; }
    uint64_t *K = keys + (size_t)b * kstride;
#pragma clang loop unroll(disable)
    for (uint32_t tile = tile0; tile < tend; tile++) {
        const uint32_t base = tile * TILE;
        const bool edge = base + TILE + 16 > n;
        if (tid < FS_MAXNB) s_cnt[tid] = 0;
        if (have) {
#pragma unroll
            for (int r = 0; r < NSTG; r++) {
                const uint32_t q = r * NT + tid;
                if (q < (TILE + 16) / 4) reinterpret_cast<uint32_t *>(s_txt)[q] = stg[r];
            }
        } else {
            for (uint32_t k = tid; k < TILE + 16; k += NT) {
                const int64_t g = (int64_t)base - 1 + k;
                s_txt[k] = g < 0 ? T[n - 1] : (g < (int64_t)n ? T[g] : (uint8_t)0);
            }
        }
        lds_only_barrier();                                    // (stg is in LDS: its registers take the next tile's request)
        const bool next_inner = tile + 1 < tend && inner(tile + 1);
        if (next_inner) request(tile + 1);                     // in flight until this tile's words are in LDS
        if (s_flagged) return;
        // thread = 8 consecutive suffixes gi0 .. gi0+7; byte j of its 16 staged bytes is T[gi0 - 1 + j]
        const uint32_t k0 = tid * ITEMS, gi0 = base + k0;
        constexpr int NBY = (ITEMS + FS_DEPTH + 3) / 4;       // dwords that hold the thread's ITEMS + FS_DEPTH staged bytes (k0 is a multiple of ITEMS)
        uint32_t by4[NBY];
        if (ITEMS % 8 == 0) {
#pragma unroll
            for (int q = 0; q < NBY; q += 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(s_txt + k0 + 4 * q);
                by4[q] = v.x;
                if (q + 1 < NBY) by4[q + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NBY; q++) by4[q] = *reinterpret_cast<const uint32_t *>(s_txt + k0 + 4 * q);
        }
#define FS_BYTE(j) ((by4[(j) >> 2] >> (8 * ((j) & 3))) & 0xFFu)
        uint2 e[ITEMS + FS_DEPTH - 1];                     // table entries of the symbols the 8 codes share
#pragma unroll
        for (int k = 0; k < ITEMS + FS_DEPTH - 1; k++) {
            const uint2 t = s_tab[FS_BYTE(1 + k)];
            e[k] = (edge && gi0 + k >= n) ? make_uint2(0u, 0u) : t;
        }
        uint64_t w[ITEMS];
        uint32_t br[ITEMS];                                // bucket << 16 | rank inside (tile, bucket)
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            uint32_t y = e[j + FS_DEPTH - 1].x;
#pragma unroll
            for (int d = FS_DEPTH - 2; d >= 1; d--) y = e[j + d].x + __umulhi(e[j + d].y, y);
            const uint64_t X = ((uint64_t)e[j].x << 32) + (uint64_t)e[j].y * y;
            const uint32_t gi = gi0 + j;
            w[j] = (X & ~FS_LOW_MASK) | ((uint64_t)gi << 8) | FS_BYTE(j);
            const uint32_t bk = nbl ? (uint32_t)(X >> (64 - nbl)) : 0u;
            br[j] = (bk << 16) | (gi < n ? atomicAdd(&s_cnt[bk], 1u) : 0u);
            if (j == 0 && gi == 0) zero_bucket[b] = bk;        // where the word of suffix 0 goes: k_fs_sort_bwt looks for the BWT index there only
        }
#undef FS_BYTE
        lds_only_barrier();
        uint32_t g = 0, c = 0;
        {
            c = tid < FS_MAXNB ? s_cnt[tid] : 0u;
            const uint32_t start = block_excl_add_lds<NT>(c, s_tmp);
            if (c) g = atomicAdd(&fill[(size_t)b * FS_MAXNB + tid], c);     // issued here, looked at behind the scatter
            if (tid < FS_MAXNB) s_start[tid] = (uint16_t)start;
        }
        lds_only_barrier();                                    // (the staged text and the table reads are done: s_w takes the words)
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
            if (gi0 + j < n) s_w[s_start[br[j] >> 16] + (br[j] & 0xFFFFu)] = w[j];
        if (c && g + c > FS_FILLMAX) { atomicOr(&flag[b], 1u); s_flagged = 1; }
        if (tid < FS_MAXNB) s_gbase[tid] = (uint16_t)(g < FS_CAP ? g : FS_CAP);
        if (next_inner) {                                      // the next tile's text has arrived before this tile's stores are issued
#pragma unroll                                                 // (loads and stores share one in-order counter)
            for (int r = 0; r < NSTG; r++) asm volatile("" : "+v"(stg[r]));
        }
        have = next_inner;
        lds_only_barrier();
        const uint32_t tile_n = min((uint32_t)TILE, n - base);
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t p = r * NT + tid;
            if (p < tile_n) {
                const uint64_t ww = s_w[p];
                const uint32_t d = nbl ? (uint32_t)(ww >> (64 - nbl)) : 0u;
                const uint32_t off = (uint32_t)s_gbase[d] + (p - (uint32_t)s_start[d]);
                if (off < FS_CAP) K[(size_t)d * FS_CAP + off] = ww;
            }
        }
        lds_only_barrier();                                    // s_w is free for the next tile's text
    }
}

// rank base of every bucket (exclusive scan of the fills); a bucket past its slot flags the block
__global__ __launch_bounds__(FS_MAXNB) void k_fs_scan(const uint32_t *__restrict__ fill, uint32_t *__restrict__ fbase,
                                                      uint32_t *__restrict__ flag, const uint32_t *__restrict__ list)
{
    __shared__ uint32_t s_tmp[FS_MAXNB / 64 + 1];
    const uint32_t b = list ? list[blockIdx.x] : blockIdx.x, tid = threadIdx.x;
    const uint32_t f = fill[(size_t)b * FS_MAXNB + tid];
    if (f > FS_FILLMAX) atomicOr(&flag[b], 1u);
    fbase[(size_t)b * FS_MAXNB + tid] = block_excl_add<FS_MAXNB>(f, s_tmp);
}

// ---------------------------------------------------------------------------
// one workgroup sorts one bucket in LDS and writes its rows of the result.  Runs of equal codes (a few
// per bucket on Zipf data) are not resolved here -- that needs the text, and a global-memory round trip on
// the critical path of every workgroup cost more than the whole sort -- but appended to a work list that
// k_fs_ties resolves with one thread per member afterwards.
// ---------------------------------------------------------------------------
// LDS diet (40.7 KB: FOUR workgroups per CU, 8 waves per SIMD, where 52 KB allowed three): the bin counters are
// 16-bit pairs (a bucket holds < 4096 words), the sorted words are capped at FS_FILLMAX (a fuller bucket flags its
// block), and the BWT bytes are staged in the counter array once the bin starts are dead.
__global__ __launch_bounds__(FSS_NT) void k_fs_sort(uint32_t n, uint32_t nbl, const uint64_t *__restrict__ keys,
                                                    size_t kstride, const uint32_t *__restrict__ fill,
                                                    const uint32_t *__restrict__ fbase, uint32_t *__restrict__ flag,
                                                    uint8_t *__restrict__ bwt_out, size_t bwt_stride,
                                                    int *__restrict__ d_index, uint32_t *__restrict__ sa_out,
                                                    size_t sa_stride, uint4 *__restrict__ wl, uint32_t wl_cap,
                                                    uint32_t *__restrict__ wl_count)
{
    __shared__ uint64_t s_w[FS_FILLMAX + 4];                   // + 4 sentinels behind the last word
    __shared__ uint32_t s_cp[FS_BINS / 2 + 1];                 // bin counters, then bin starts (two 16-bit values per word), then BWT bytes
    __shared__ uint32_t s_tmp[FSS_NT / 64 + 1];
    __shared__ uint32_t s_deep, s_wl;
    uint16_t *s16 = reinterpret_cast<uint16_t *>(s_cp);
    uint8_t *s_cb = reinterpret_cast<uint8_t *>(s_cp);
    uint32_t gx, gy;
    xcd_order(gx, gy);
    const uint32_t b = gy, bk = gx, tid = threadIdx.x;
    if (tid == 0) { s_deep = flag[b]; s_wl = 0; }              // (one read: another bucket may flag the block meanwhile)
    for (uint32_t i = tid; i < FS_BINS / 2; i += FSS_NT) s_cp[i] = 0;
    const uint32_t c = fill[(size_t)b * FS_MAXNB + bk];
    const uint32_t R0 = fbase[(size_t)b * FS_MAXNB + bk];
    const uint64_t *K = keys + (size_t)b * kstride + (size_t)bk * FS_CAP;
    uint64_t w[FSS_ITEMS];
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        const uint32_t i = r * FSS_NT + tid;
        w[r] = ~0ull;
        if (r * FSS_NT < c && c <= FS_FILLMAX && i < c) w[r] = K[i];
    }
    if (tid < 4 && c <= FS_FILLMAX) s_w[c + tid] = ~0ull;      // what the rank step reads past the last bin compares as larger
    __syncthreads();
    if (s_deep || c == 0) return;                              // flagged: the block goes through the general sorter
    // 1. counting sort on the 12 bits below the bucket number (arrival order inside a bin: any order will do)
    const uint32_t bshift = 64 - nbl - FS_BIN_BITS;
    uint32_t rk[FSS_ITEMS];
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        rk[r] = 0;
        if (r * FSS_NT >= c) continue;                         // (uniform: a bucket fills half of the slots on average)
        const uint32_t i = r * FSS_NT + tid;
        const uint32_t bin = (uint32_t)(w[r] >> bshift) & (FS_BINS - 1), sh = 16 * (bin & 1);
        if (i < c) rk[r] = (atomicAdd(&s_cp[bin >> 1], 1u << sh) >> sh) & 0xFFFFu;
    }
    __syncthreads();
    {
        constexpr int PW = FS_BINS / 2 / FSS_NT;               // packed words per thread
        uint32_t v[PW], sum = 0;
#pragma unroll
        for (int k = 0; k < PW; k++) { v[k] = s_cp[tid * PW + k]; sum += (v[k] & 0xFFFFu) + (v[k] >> 16); }
        uint32_t run = block_excl_add<FSS_NT>(sum, s_tmp);
#pragma unroll
        for (int k = 0; k < PW; k++) {
            const uint32_t lo = run, hi = run + (v[k] & 0xFFFFu);
            s_cp[tid * PW + k] = lo | (hi << 16);
            run = hi + (v[k] >> 16);
        }
        if (tid == FSS_NT - 1) s_cp[FS_BINS / 2] = c;          // end of the last bin
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        if (r * FSS_NT >= c) continue;
        const uint32_t i = r * FSS_NT + tid;
        if (i < c) s_w[s16[(uint32_t)(w[r] >> bshift) & (FS_BINS - 1)] + rk[r]] = w[r];
    }
    __syncthreads();
    // 2. final position = bin start + number of smaller codes in the bin (thread = the words at positions r NT + tid
    //    of the bin-sorted array).  A bin holds ~1.5 words seen from one of them: the four words from the bin start
    //    are compared in straight-line code with no bounds at all -- what lies behind the bin's end is a larger code
    //    or a sentinel -- and only a bin of more than four walks the rest in a loop.  Words with EQUAL codes form a
    //    group [gp, gp + gs) whose internal order is not known yet: they take a second, exact look (rare).
    uint32_t *SAo = sa_out ? sa_out + (size_t)b * sa_stride + R0 : nullptr;
    uint32_t pos[FSS_ITEMS], grp[FSS_ITEMS];                   // grp = group start << 16 | group size (0: not tied)
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        const uint32_t p = r * FSS_NT + tid;
        pos[r] = 0xFFFFFFFFu; grp[r] = 0;
        if (r * FSS_NT >= c) continue;
        if (p < c) {
            const uint64_t wv = s_w[p];
            w[r] = wv;
            const uint32_t key = (uint32_t)(wv >> 28);         // inside a bin only the low 32 bits of the code can differ
            const uint32_t bin = (uint32_t)(wv >> bshift) & (FS_BINS - 1);
            const uint32_t gs = s16[bin], ge = s16[bin + 1];
            if (ge - gs > FS_MAX_GROUP) { s_deep = 1; }
            else {
                uint32_t less = 0, eqt = 0;
                const uint2 *B = reinterpret_cast<const uint2 *>(s_w) + gs;
#pragma unroll
                for (uint32_t t = 0; t < 4; t++) {
                    const uint2 wq = B[t];
                    const uint32_t kq = __builtin_amdgcn_alignbit(wq.y, wq.x, 28);
                    less += kq < key ? 1u : 0u;
                    eqt += kq == key ? 1u : 0u;
                }
                if (ge - gs > 4) {
#pragma clang loop unroll(disable)
                    for (uint32_t q = gs + 4; q < ge; q++) {
                        const uint2 wq = reinterpret_cast<const uint2 *>(s_w)[q];
                        const uint32_t kq = __builtin_amdgcn_alignbit(wq.y, wq.x, 28);
                        less += kq < key; eqt += kq == key;
                    }
                }
                const uint32_t idx = (uint32_t)(wv >> 8) & 0xFFFFFu;
                uint32_t at = gs + less;
                if (eqt > 1) {                                 // exact: members of my group inside the bin, and those before me
                    uint32_t eqb = 0; eqt = 0;
#pragma clang loop unroll(disable)
                    for (uint32_t q = gs; q < ge; q++) {
                        const uint2 wq = reinterpret_cast<const uint2 *>(s_w)[q];
                        const bool e = __builtin_amdgcn_alignbit(wq.y, wq.x, 28) == key;
                        eqt += e; eqb += e & (q < p);
                    }
                    if (eqt > 1) { grp[r] = (at << 16) | eqt; at += eqb; }
                }
                pos[r] = at;
                if (!grp[r]) {
                    if (SAo) SAo[at] = idx;
                    if (idx == 0 && d_index) d_index[b] = (int)(R0 + at);
                }
            }
        }
    }
    __syncthreads();                                           // s_cp (bin starts) is dead from here: it takes the BWT bytes
    if (s_deep) { if (tid == 0) atomicOr(&flag[b], 2u); return; }
    // rows R0 .. R0 + c of the block's BWT are staged so that aligned dwords of LDS are aligned dwords of the output
    uint8_t *O = bwt_out ? bwt_out + (size_t)b * bwt_stride + R0 : nullptr;
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(O) & 3u);
    // 3. tied groups -> the block's work list.  Entries are reserved with ONE global atomic per workgroup
    //    (an atomic per group on a shared counter serialised the whole kernel: +2.4 ms per 256 blocks).
    bool any = false;
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        if (pos[r] != 0xFFFFFFFFu) s_cb[shift + pos[r]] = (uint8_t)w[r];      // (rows of tied groups are rewritten by k_fs_ties)
        if (grp[r]) {
            any = true;
            const uint32_t gp = grp[r] >> 16, gs = grp[r] & 0xFFFFu;
            if (pos[r] == gp) reinterpret_cast<uint32_t *>(s_w)[2 * gp + 1] = atomicAdd(&s_wl, gs);   // (the codes are dead)
        }
    }
    if (__syncthreads_or((int)any)) {
        if (tid == 0) {
            const uint32_t tot = s_wl;
            uint32_t base = atomicAdd(&wl_count[b], tot);
            if (base + tot > wl_cap) { atomicOr(&flag[b], 4u); base = 0xFFFFFFFFu; }   // list full: block flagged
            s_deep = base;
        }
        __syncthreads();
        const uint32_t base = s_deep;
        uint4 *WL = wl + (size_t)b * wl_cap;
        if (base != 0xFFFFFFFFu) {
#pragma unroll
            for (int r = 0; r < FSS_ITEMS; r++) {
                if (grp[r]) {
                    const uint32_t gp = grp[r] >> 16, gs = grp[r] & 0xFFFFu, slot0 = base + reinterpret_cast<const uint32_t *>(s_w)[2 * gp + 1];
                    WL[slot0 + (pos[r] - gp)] = make_uint4((uint32_t)(w[r] & FS_LOW_MASK), R0 + gp, slot0, gs);
                }
            }
        }
    }
    // 4. the rows
    if (O) {
        const uint32_t end = shift + c;                        // staged bytes [shift, end)
        for (uint32_t q = tid; 4 * q < end; q += FSS_NT) {
            const uint32_t v = s_cp[q];
            if (4 * q >= shift && 4 * q + 4 <= end) *reinterpret_cast<uint32_t *>(O - shift + 4 * q) = v;
            else {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (4 * q + k >= shift && 4 * q + k < end) O[4 * q + k - shift] = (uint8_t)(v >> (8 * k));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_fs_sort_bwt: k_fs_sort for the BWT-only path (no suffix array asked for), on a scalar-instruction diet.  The counters
// put k_fs_sort at 109 VALU + 93.5 SALU + 14.7 LDS instructions per suffix with the scalar pipe 0.66 busy, and nearly
// all of the scalar work is exec-mask bookkeeping of per-item conditions, eight items per thread.  Here
//   * the ONE partial round of a bucket (c mod 512 words) is round 0 and carries the only per-lane predicate; rounds
//     1 .. c / 512 are full and run under a uniform branch alone (k_fs_sort: `i < c` in every round of every loop);
//   * the rare cases of the rank step -- a bin of more than four words, another word with the same code -- are found
//     per wave with ONE ballot and handled behind a wave-uniform branch (k_fs_sort: three nested divergent branches per
//     item, executed or not);
//   * the row of suffix 0 (the BWT index) is looked for in the ONE bucket k_fs_part saw it go to (zero_bucket), not in
//     every word of every bucket.
// Same LDS layout and access pattern (a word is ranked AT ITS POSITION of the bin-sorted array: bin bounds, neighbours and
// the staged row of neighbouring lanes share banks -- the owner-ranked form measured in round 4 lost more to LDS bank
// conflicts than it saved in instructions, tools/exp/attic), same work list, entry for entry.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FSS_NT) void k_fs_sort_bwt(uint32_t nbl, const uint64_t *__restrict__ keys, size_t kstride,
                                                        const uint32_t *__restrict__ fill, const uint32_t *__restrict__ fbase,
                                                        uint32_t *__restrict__ flag, uint8_t *__restrict__ bwt_out,
                                                        size_t bwt_stride, int *__restrict__ d_index, uint4 *__restrict__ wl,
                                                        uint32_t wl_cap, uint32_t *__restrict__ wl_count,
                                                        const uint32_t *__restrict__ zero_bucket)
{
    __shared__ uint64_t s_w[FS_FILLMAX + FSS_LOOK];            // + sentinels behind the last word
    __shared__ uint32_t s_cp[FS_BINS / 2 + 1];                 // bin counters, then bin starts (two 16-bit values per word), then BWT bytes
    __shared__ uint32_t s_tmp[FSS_NT / 64 + 1];
    __shared__ uint32_t s_deep, s_wl;
    uint16_t *s16 = reinterpret_cast<uint16_t *>(s_cp);
    uint8_t *s_cb = reinterpret_cast<uint8_t *>(s_cp);
    uint32_t gx, gy;
    xcd_order(gx, gy);
    const uint32_t b = gy, bk = gx, tid = threadIdx.x;
    // A block flagged BEFORE this kernel started -- text-like (k_fs_tables), a bucket past its slot (k_fs_part2), constant -- is
    // another sorter's: its 512 workgroups leave on a scalar load, before the table of counters is cleared and three rounds of
    // words are asked for (0.25 ms per batch of 256 text blocks were spent leaving).  Bits 1 and 8 only: nobody sets those while
    // this kernel runs, so every wave of the workgroup sees the same (what k_fs_sort_bwt itself sets, 2 and 4, goes through s_deep)
    if (scalar_load_u32(flag + b) & (1u | FS_DONE)) return;
    if (tid == 0) { s_deep = flag[b]; s_wl = 0; }              // (one read: another bucket may flag the block meanwhile)
    for (uint32_t i = tid; i < FS_BINS / 2; i += FSS_NT) s_cp[i] = 0;
    const uint64_t *K = keys + (size_t)b * kstride + (size_t)bk * FS_CAP;
    uint64_t w[FSS_ITEMS];
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) w[r] = ~0ull;
    // The first FSS_SPEC full rounds are requested BEFORE the bucket's fill is known: a bucket of an i.i.d.-like block holds
    // 2048 +- 7 % words, so words 0 .. 1535 are (nearly) always there, and fill -> words was two memory latencies in a row at
    // the head of every workgroup's chain.  A slot has FS_CAP words whatever its fill: reading past the fill is harmless, and
    // such a round is never looked at (r <= full guards every use).
    constexpr int FSS_SPEC = 3;
#pragma unroll
    for (int r = 1; r <= FSS_SPEC; r++) w[r] = K[(r - 1) * FSS_NT + tid];
    const uint32_t c = fill[(size_t)b * FS_MAXNB + bk];
    const uint32_t R0 = fbase[(size_t)b * FS_MAXNB + bk];
    const uint32_t cc = c <= FS_FILLMAX ? c : 0u;              // (a fuller bucket has flagged its block in k_fs_scan)
    // item r of a thread: r = 0 -> word full * 512 + tid of the partial round (lanes tid < part), r >= 1 -> word
    // (r - 1) * 512 + tid of a full round (all lanes, r <= full)
    const uint32_t full = cc / FSS_NT, part = cc % FSS_NT;
    const bool v0 = tid < part;
    const uint32_t i0 = full * FSS_NT + tid;
    if (v0) w[0] = K[i0];
#pragma unroll
    for (int r = FSS_SPEC + 1; r < FSS_ITEMS; r++)
        if ((uint32_t)r <= full) w[r] = K[(r - 1) * FSS_NT + tid];
    if (tid < FSS_LOOK && cc) s_w[cc + tid] = ~0ull;                  // what the rank step reads past the last bin compares as larger
    __syncthreads();
    if (s_deep || cc == 0) return;                             // flagged: the block is another sorter's
    // 1. counting sort on the 12 bits below the bucket number (arrival order inside a bin: any order will do)
    const uint32_t bshift = 64 - nbl - FS_BIN_BITS;
    uint32_t rk[FSS_ITEMS];
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) rk[r] = 0;
    {
        auto count = [&](int r) {
            const uint32_t bin = (uint32_t)(w[r] >> bshift) & (FS_BINS - 1), sh = 16 * (bin & 1);
            rk[r] = (atomicAdd(&s_cp[bin >> 1], 1u << sh) >> sh) & 0xFFFFu;
        };
        if (v0) count(0);
#pragma unroll
        for (int r = 1; r < FSS_ITEMS; r++)
            if ((uint32_t)r <= full) count(r);
    }
    __syncthreads();
    {
        constexpr int PW = FS_BINS / 2 / FSS_NT;               // packed words per thread
        uint32_t v[PW], sum = 0;
#pragma unroll
        for (int k = 0; k < PW; k++) { v[k] = s_cp[tid * PW + k]; sum += (v[k] & 0xFFFFu) + (v[k] >> 16); }
        uint32_t run = block_excl_add<FSS_NT>(sum, s_tmp);
#pragma unroll
        for (int k = 0; k < PW; k++) {
            const uint32_t lo = run, hi = run + (v[k] & 0xFFFFu);
            s_cp[tid * PW + k] = lo | (hi << 16);
            run = hi + (v[k] >> 16);
        }
        if (tid == FSS_NT - 1) s_cp[FS_BINS / 2] = c;          // end of the last bin
    }
    __syncthreads();
    {
        // in LDS a word is [code bits 28..59 : 32 | the word's low dword (code bits 28..31, index, BWT byte) : 32]: the rank step
        // compares HIGH DWORDS only -- four 4-byte reads where the whole words were four 8-byte reads and a funnel shift each
        auto scatter = [&](int r) {
            const uint32_t lo = (uint32_t)w[r], hi = (uint32_t)(w[r] >> 32);
            s_w[s16[(uint32_t)(w[r] >> bshift) & (FS_BINS - 1)] + rk[r]] =
                ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 28) << 32) | lo;
        };
        if (v0) scatter(0);
#pragma unroll
        for (int r = 1; r < FSS_ITEMS; r++)
            if ((uint32_t)r <= full) scatter(r);
    }
    __syncthreads();
    // 2. final position = bin start + number of smaller codes in the bin (thread = the words at positions i0 / (r - 1) NT +
    //    tid of the bin-sorted array).  The four words from the bin start are compared in straight-line code with no bounds
    //    at all -- what lies behind the bin's end is a larger code or a sentinel; a bin of more than four, or a second word
    //    with my code among the four, is the rare case: detected per wave, redone exactly over the whole bin.
    const bool zb = zero_bucket[b] == bk;                       // (uniform) suffix 0 is one of this bucket's words
    uint32_t pos[FSS_ITEMS], grp[FSS_ITEMS];                   // grp = group start << 16 | group size (0: not tied)
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) { pos[r] = 0xFFFFFFFFu; grp[r] = 0; }
    {
        auto rank = [&](int r, uint32_t p) {
            const uint64_t wv = s_w[p];
            w[r] = wv;
            const uint32_t key = (uint32_t)(wv >> 32);         // inside a bin only the low 32 bits of the code can differ
            const uint32_t bin = (key >> (bshift - 28)) & (FS_BINS - 1);
            const uint32_t gs = s16[bin], ge = s16[bin + 1];
            uint32_t less = 0, eqt = 0;
            const uint32_t *B = reinterpret_cast<const uint32_t *>(s_w) + 2 * gs + 1;
#pragma unroll
            for (uint32_t t = 0; t < FSS_LOOK; t++) {
                const uint32_t kq = B[2 * t];
                less += kq < key ? 1u : 0u;
                eqt += kq == key ? 1u : 0u;
            }
            uint32_t at = gs + less;
            const bool rare = (ge - gs > FSS_LOOK) | (eqt > 1);
            if (__builtin_amdgcn_ballot_w64(rare) != 0) {      // (wave-uniform)
                if (rare) {
                    if (ge - gs > FS_MAX_GROUP) { s_deep = 1; at = 0xFFFFFFFFu; }
                    else {
                        uint32_t ls = 0, eq = 0, eqb = 0;
#pragma clang loop unroll(disable)
                        for (uint32_t q = gs; q < ge; q++) {
                            const uint32_t kq = reinterpret_cast<const uint32_t *>(s_w)[2 * q + 1];
                            ls += kq < key; eq += kq == key; eqb += (kq == key) & (q < p);
                        }
                        at = gs + ls;
                        if (eq > 1) { grp[r] = (at << 16) | eq; at += eqb; }
                    }
                }
            }
            pos[r] = at;
        };
        if (v0) rank(0, i0);
#pragma unroll
        for (int r = 1; r < FSS_ITEMS; r++)
            if ((uint32_t)r <= full) rank(r, (r - 1) * FSS_NT + tid);
        if (zb) {
#pragma unroll
            for (int r = 0; r < FSS_ITEMS; r++)
                if (pos[r] != 0xFFFFFFFFu && ((uint32_t)w[r] & 0x0FFFFF00u) == 0 && !grp[r]) d_index[b] = (int)(R0 + pos[r]);
        }
    }
    __syncthreads();                                           // s_cp (bin starts) is dead from here: it takes the BWT bytes
    if (s_deep) { if (tid == 0) atomicOr(&flag[b], 2u); return; }
    // rows R0 .. R0 + c of the block's BWT are staged so that aligned dwords of LDS are aligned dwords of the output
    uint8_t *O = bwt_out + (size_t)b * bwt_stride + R0;
    const uint32_t shift = (uint32_t)(reinterpret_cast<uintptr_t>(O) & 3u);
    // 3. tied groups -> the block's work list.  Entries are reserved with ONE global atomic per workgroup.  (Round 5: the
    //    ~1.5 us a workgroup sits out for its return are NOT on the kernel's critical path -- with entries of the bucket's own
    //    and no atomic at all the kernel ran 3.50 against 3.47 ms per GiB, and k_fs_ties, which then has to walk 512 short
    //    lists per block, 0.32 against 0.18.)
    bool any = false;
#pragma unroll
    for (int r = 0; r < FSS_ITEMS; r++) {
        if (pos[r] != 0xFFFFFFFFu) s_cb[shift + pos[r]] = (uint8_t)w[r];      // (rows of tied groups are rewritten by k_fs_ties)
        if (grp[r]) {
            any = true;
            const uint32_t gp = grp[r] >> 16, gs = grp[r] & 0xFFFFu;
            if (pos[r] == gp) reinterpret_cast<uint32_t *>(s_w)[2 * gp + 1] = atomicAdd(&s_wl, gs);   // (the codes are dead)
        }
    }
    if (__syncthreads_or((int)any)) {
        if (tid == 0) {
            const uint32_t tot = s_wl;
            uint32_t base = atomicAdd(&wl_count[b], tot);
            if (base + tot > wl_cap) { atomicOr(&flag[b], 4u); base = 0xFFFFFFFFu; }   // list full: block flagged
            s_deep = base;
        }
        __syncthreads();
        const uint32_t base = s_deep;
        uint4 *WL = wl + (size_t)b * wl_cap;
        if (base != 0xFFFFFFFFu) {
#pragma unroll
            for (int r = 0; r < FSS_ITEMS; r++) {
                if (grp[r]) {
                    const uint32_t gp = grp[r] >> 16, gs = grp[r] & 0xFFFFu, slot0 = base + reinterpret_cast<const uint32_t *>(s_w)[2 * gp + 1];
                    WL[slot0 + (pos[r] - gp)] = make_uint4((uint32_t)(w[r] & FS_LOW_MASK), R0 + gp, slot0, gs);
                }
            }
        }
    }
    // 4. the rows
    {
        const uint32_t end = shift + c;                        // staged bytes [shift, end)
        for (uint32_t q = tid; 4 * q < end; q += FSS_NT) {
            const uint32_t v = s_cp[q];
            if (4 * q >= shift && 4 * q + 4 <= end) *reinterpret_cast<uint32_t *>(O - shift + 4 * q) = v;
            else {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++)
                    if (4 * q + k >= shift && 4 * q + k < end) O[4 * q + k - shift] = (uint8_t)(v >> (8 * k));
            }
        }
    }
}

// ---------------------------------------------------------------------------
// groups of equal codes: one thread per member counts the members whose suffix is smaller (text comparison,
// 8 bytes at a time) and writes its row.  Equal codes do not certify equal symbols: the comparison starts at
// the first symbol.
// ---------------------------------------------------------------------------
#ifndef GLC_FST_W
#define GLC_FST_W 4
#endif
constexpr int FST_W = GLC_FST_W;                              // k_fs_ties: members of a run met at a time
__global__ __launch_bounds__(256) void k_fs_ties(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                 const uint4 *__restrict__ wl, uint32_t wl_cap,
                                                 const uint32_t *__restrict__ wl_count, uint32_t *__restrict__ flag,
                                                 uint8_t *__restrict__ bwt_out, size_t bwt_stride,
                                                 int *__restrict__ d_index, uint32_t *__restrict__ sa_out, size_t sa_stride)
{
    uint32_t gx, gy;
    xcd_order(gx, gy);                                         // a block's text in one L2 for the comparisons
    const uint32_t b = gy;
    const uint32_t total = wl_count[b];
    if (total == 0 || total > wl_cap || flag[b]) return;       // (flags of this pass are all set before it starts)
    const uint4 *WL = wl + (size_t)b * wl_cap;
    const uint8_t *T = text + (size_t)b * stride;
    for (uint32_t e = gx * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const uint4 me = WL[e];
        const uint32_t gs = me.w, idx = me.x >> 8;
        uint32_t rank = 0;
        bool deep = false;
        // four members at a time: their list entries, then their first 8 text bytes, are loaded together (a member of a
        // run of 100 equal codes would otherwise wait for 200 memory round trips one after the other); only a pair
        // that agrees in those 8 bytes, or sits at the end of the block, takes the byte-exact loop
        const bool fast_me = idx + 12 <= n;
        const uint64_t mine = fast_me ? fs_load_be64(T + idx) : 0ull;
        for (uint32_t f0 = me.z; f0 < me.z + gs && !deep; f0 += FST_W) {
            uint32_t oi[FST_W];
            uint64_t ov[FST_W];
#pragma unroll
            for (int k = 0; k < FST_W; k++) oi[k] = f0 + k < me.z + gs ? WL[f0 + k].x >> 8 : idx;
#pragma unroll
            for (int k = 0; k < FST_W; k++) ov[k] = (oi[k] != idx && fast_me && oi[k] + 12 <= n) ? fs_load_be64(T + oi[k]) : mine;
#pragma unroll
            for (int k = 0; k < FST_W; k++) {
                if (oi[k] == idx) continue;                    // myself, or past the end of the run
                if (fast_me && oi[k] + 12 <= n && ov[k] != mine) rank += ov[k] < mine ? 1u : 0u;
                else rank += fs_suffix_less(T, n, oi[k], idx, &deep) ? 1u : 0u;
            }
        }
        if (deep) { atomicOr(&flag[b], 2u); continue; }
        const uint32_t row = me.y + rank;
        if (bwt_out) bwt_out[(size_t)b * bwt_stride + row] = (uint8_t)me.x;
        if (sa_out) sa_out[(size_t)b * sa_stride + row] = idx;
        if (idx == 0 && d_index) d_index[b] = (int)row;
    }
}

// blocks the fast path gave up on -> live counts for the general sorter
// (`redo` is a per-call copy of the flags for the stages queued speculatively behind the sort: under stage
//  pipelining the next call clears `flag` while they may still be reading)
// nflag[0] = flagged blocks; nflag[3] = ticket of this launch's workgroups: the last one stores the count where the host reads
// it (h_nflag: pinned, device-mapped) -- a copy command behind the pass was one more launch in every call
// nflag[4] = blocks the text-likeness probe did NOT call text-like (constant blocks apart): h_nflag[1].  The host's streak of
// "every block of the call was text-like" (sa_build_begin: small calls skip the bucket sorter's attempt while it lasts) ends
// with the first such block, also in a call that skipped.
__global__ void k_fs_finish(const uint32_t *__restrict__ flag, uint32_t n, uint32_t nblk, uint32_t *__restrict__ lcnt,
                            uint32_t *__restrict__ nflag, uint32_t *__restrict__ redo, uint32_t *__restrict__ keep,
                            uint32_t *__restrict__ list, uint32_t *__restrict__ h_nflag, const uint32_t *__restrict__ dup,
                            uint32_t dup_flag)
{
    __shared__ uint32_t s_last;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) {
        const uint32_t f = (flag[b] && flag[b] != FS_DONE) ? n : 0u;    // (FS_DONE: a constant block, finished by k_fs_tables)
        lcnt[b] = f;
        redo[b] = f;
        keep[b] = f ? 0u : 1u;
        if (f) list[atomicAdd(nflag, 1u)] = b;                 // (any order)
        if (flag[b] != FS_DONE && dup[b] < dup_flag) atomicAdd(nflag + 4, 1u);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&nflag[3], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        reinterpret_cast<volatile uint32_t *>(h_nflag)[1] = atomicAdd(nflag + 4, 0u);
        *reinterpret_cast<volatile uint32_t *>(h_nflag) = atomicAdd(nflag, 0u);
        __threadfence_system();
    }
}

// everything the pass accumulates into, cleared by ONE launch (six hipMemsetAsync calls were six dispatches of ~2 us with
// ~8 us between them: 90 us of a 1 MiB call that takes 400)
__global__ __launch_bounds__(256) void k_fs_clear(uint32_t nblk, uint32_t *__restrict__ hist, uint32_t *__restrict__ fill,
                                                  uint32_t *__restrict__ flag, uint32_t flag_value, uint32_t *__restrict__ wlcnt,
                                                  uint32_t *__restrict__ dup, uint32_t *__restrict__ nflag)
{
    const uint32_t nh = nblk * 256u, nf = nblk * FS_MAXNB, total = nh + nf + 3u * nblk + 8u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        if (i < nh) hist[i] = 0;
        else if (i < nh + nf) fill[i - nh] = 0;
        else {
            const uint32_t j = i - nh - nf;
            if (j < nblk) flag[j] = flag_value;
            else if (j < 2 * nblk) wlcnt[j - nblk] = 0;
            else if (j < 3 * nblk) dup[j - 2 * nblk] = 0;
            else nflag[j - 3 * nblk] = 0;                       // [0] flagged blocks, [1] [2] the sample sorter's, [3] k_fs_finish's ticket, [4] blocks the probe did not call text-like
        }
    }
}

// ===========================================================================
// second tier: string sample sort of the blocks the bucket sorter flagged
// ===========================================================================
// Text, logs and other data with context make the order-0 code lumpy -- a frequent 6-gram puts thousands of suffixes
// on ONE code, so fixed bucket boundaries overflow and runs of equal codes are long.  For those blocks the buckets
// are cut at SPLITTER SUFFIXES instead: 32 samples per bucket, sorted exactly (code, then text), every 32nd is the
// first suffix of a bucket, so buckets hold ~2048 +- 20 % suffixes whatever the distribution -- a code shared by
// 10000 suffixes is simply spread over five buckets, cut by text comparison.  A bucket is then sorted in LDS in
// rounds of 5 symbols read from the text, starting behind the common prefix of the bucket's two splitters
// (everything between two suffixes shares their common prefix); see k_ss_cut / k_ss_windows.
// Same words, same slots, same outputs as the first tier; only very deep repeats are left to the general sorter.
#ifdef GLC_SS_CLOCKS
// experiment builds only (tools/exp/ss_clocks.py): s_memrealtime ticks (100 MHz) per phase, summed over thread 0 of every
// workgroup of k_ss_cut ([0, 8)) and of k_ss_sample ([32, 40)) and over every wave of k_ss_windows ([16, 24)); [8] / [24] / [40] count them
__device__ unsigned long long g_ss_clk[256][48];           // 256 copies: the adds of a million workgroups do not queue on 32 addresses
#define SS_CLK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); clk_[k] += t_ - clk_t_; clk_t_ = t_; } } while (0)
#define SS_CLK_BEGIN() unsigned long long clk_[8] = {}, clk_t_ = __builtin_amdgcn_s_memrealtime()
#define SS_COUNT(k) do { if ((threadIdx.x & 63u) == 0) atomicAdd(&g_ss_clk[blockIdx.x & 255u][k], 1ull); } while (0)
#define SS_MAX(k, i) do { if (threadIdx.x == 0) atomicMax(&g_ss_clk[0][k], clk_[i]); } while (0)
#define SS_CLK_END(base) do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) if (clk_[k_]) atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][(base) + k_], clk_[k_]); atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][(base) + 8], 1ull); } } while (0)
#else
#define SS_CLK(k) do { } while (0)
#define SS_CLK_BEGIN() do { } while (0)
#define SS_COUNT(k) do { } while (0)
#define SS_MAX(k, i) do { } while (0)
#define SS_CLK_END(base) do { } while (0)
#endif
constexpr int SSA_NT = 1024;                                   // k_ss_sample: threads
constexpr uint32_t SS_PER_BUCKET = 32, SS_MAXS = FS_MAXNB * SS_PER_BUCKET;
constexpr uint32_t SS_L0_CAP = 1024;                           // longest splitter prefix skipped at once
#ifndef GLC_SSA_SLOTS
#define GLC_SSA_SLOTS 4
#endif
constexpr int SSA_SLOTS = GLC_SSA_SLOTS;                       // k_ss_sample: members of a long run of equal codes a lane holds
constexpr uint32_t SSA_LONG_CAP = 64u * SSA_SLOTS;             // ... the longest run ordered that way
constexpr uint32_t SSA_SLAB = 32, SSA_WIN = 120;               // ... runs that start in SSA_SLAB places and end within SSA_WIN are ordered together

__device__ __forceinline__ uint64_t fs_code_at(const uint2 *tab, const uint8_t *T, uint32_t n, uint32_t i)
{
    uint2 e[SS_DEPTH];
    if (SS_DEPTH <= 8 && i + 8 <= n) {                         // (one load for the symbols)
        uint64_t x;
        __builtin_memcpy(&x, T + i, 8);
#pragma unroll
        for (int k = 0; k < SS_DEPTH; k++) e[k] = tab[(x >> (8 * k)) & 0xFFu];
    } else {
#pragma unroll
        for (int k = 0; k < SS_DEPTH; k++) e[k] = i + k < n ? tab[T[i + k]] : make_uint2(0u, 0u);
    }
    uint32_t y = e[SS_DEPTH - 1].x;
#pragma unroll
    for (int d = SS_DEPTH - 2; d >= 1; d--) y = e[d].x + __umulhi(e[d].y, y);
    return ((uint64_t)e[0].x << 32) + (uint64_t)e[0].y * y;
}

// a < b for sample words [code : 36 | index : 20 | 0 : 8]; ~0 = padding, larger than everything
__device__ __forceinline__ bool ss_word_less(uint64_t a, uint64_t b, const uint8_t *T, uint32_t n, bool *deep, bool tol)
{
    if (a == ~0ull) return false;
    if (b == ~0ull) return true;
    const uint64_t ca = a >> 28, cb = b >> 28;
    if (ca != cb) return ca < cb;
    const uint32_t ia = (uint32_t)(a >> 8) & 0xFFFFFu, ib = (uint32_t)(b >> 8) & 0xFFFFFu;
    if (ia == ib) return false;
    return fs_suffix_less<true>(T, n, ia, ib, deep, 0, tol);
}

// k_ss_sample's integer sort, up to four stages of the bitonic network per trip through LDS.  The stages of merge level L (runs
// of k = 2^L) pair words at distances j = k/2 .. 1; the ones with j = 2^SH .. 2^(SH+3) only pair words whose indices differ in
// bits [SH, SH + 4), so a thread that holds the 16 words base + (a << SH), a = 0 .. 15, does them all in registers.  Windows of
// index bits: [10, 14), [6, 10), [2, 6) and the two lowest bits (NBITS = 2, on 16 consecutive words); levels 1 .. 4 are one trip
// on 16 consecutive words (FIRST).  33 trips for 16384 words instead of 105.  Words live at ssa_phys(index) during the sort --
// low nibble XORed with bits 5 .. 8 -- so that lanes 128 bytes (16 consecutive words) or 512 bytes apart spread over the banks.
__device__ __forceinline__ uint32_t ssa_phys(uint32_t e) { return e ^ ((e >> 5) & 15u); }

template <int SH, int NBITS, bool FIRST>
__device__ __forceinline__ void ssa_sort_pass(uint64_t *s, uint32_t tid, uint32_t klevel)
{
    asm volatile("" : "+v"(tid));                              // (the 16 addresses are made here, trip by trip: hoisted out of the level loop they spill)
    const uint32_t base = (tid & ((1u << SH) - 1u)) | ((tid >> SH) << (SH + 4));
    uint64_t v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = s[ssa_phys(base + ((uint32_t)a << SH))];
#pragma unroll
    for (int lk = (FIRST ? 1 : 0); lk <= (FIRST ? 4 : 0); lk++) {
        const uint32_t k = FIRST ? (1u << lk) : klevel;
#pragma unroll
        for (int bit = NBITS - 1; bit >= 0; bit--) {
            const uint32_t j = 1u << (SH + bit);
            if (j >= k) continue;                              // (a stage of this window that the level does not have)
#pragma unroll
            for (int a = 0; a < 16; a++) {
                if (a & (1 << bit)) continue;
                const int c = a | (1 << bit);
                const bool up = ((base + ((uint32_t)a << SH)) & k) == 0;
                const uint64_t x = v[a], y = v[c];
                const bool sw = (y < x) == up;                 // (equal words: the padding; swapped or not, the same)
                v[a] = sw ? y : x;
                v[c] = sw ? x : y;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 16; a++) s[ssa_phys(base + ((uint32_t)a << SH))] = v[a];
    __syncthreads();
}

template <bool TOL>
__global__ __launch_bounds__(SSA_NT) void k_ss_sample(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                      uint32_t nbl, const uint2 *__restrict__ tab,
                                                      const uint32_t *__restrict__ list, uint64_t *__restrict__ split,
                                                      uint16_t *__restrict__ cell, uint32_t *__restrict__ flag,
                                                      uint32_t *__restrict__ l0_out, uint64_t *__restrict__ split8, uint32_t seed,
                                                      uint64_t *__restrict__ split16)
{
    constexpr bool tol = TOL;
    __shared__ uint64_t s_s[SS_MAXS];                          // 128 KB: one workgroup per CU
    __shared__ ulonglong2 s_k[SSA_NT / 64][SSA_WIN];           // step (b): a wave's keys; before that, the code table
    uint2 *s_tab = reinterpret_cast<uint2 *>(&s_k[0][0]);
    __shared__ uint32_t s_deep, s_ties, s_work, s_big;
    const uint32_t b = list[blockIdx.x], tid = threadIdx.x, nb = 1u << nbl;
    if (flag[b]) return;                                       // (uniform) flagged before the attempt: per_probe, a mostly periodic block
    const uint8_t *T = text + (size_t)b * stride;
    const uint32_t S = min(nb * SS_PER_BUCKET, n);
    uint32_t S2 = 1;
    while (S2 < S) S2 <<= 1;
    if (tid < 256) s_tab[tid] = tab[(size_t)b * 256 + tid];
    if (tid == 0) { s_deep = 0; s_ties = 0; }
    SS_CLK_BEGIN();
    __syncthreads();
    for (uint32_t j = tid; j < S2; j += SSA_NT) {
        uint64_t w = ~0ull;
        if (j < S) {
            // one sample per stride of n / S positions, at a hashed offset inside it: evenly spaced samples (every 64th
            // suffix of a 1 MiB block) would only ever see one phase of data with a period, e.g. byte 0 of every float
            // (The offset is a full avalanche hash of j.  Round 3's `(j * 2654435761) >> 12` is a Weyl sequence: the offset
            // advances by 55 mod 64 from one stride to the next, the samples sit on a near-lattice of 119 / 55 bytes, and log
            // lines of ~88 bytes beat against it -- whole classes of suffixes under-sampled, a bucket of > 4032 words and the
            // block handed to the general sorter: 2 of 256 log blocks, max LCP 51, found with distinct blocks in bench.py.)
            // (j n / S: S is a power of two, or n itself -- no 64-bit divisions)
            const uint32_t lo = S == n ? j : (uint32_t)(((uint64_t)j * n) >> (nbl + 5)), hi = S == n ? j + 1 : (uint32_t)(((uint64_t)(j + 1) * n) >> (nbl + 5));
            uint32_t h = (j + 1u + seed * SS_MAXS) * 0x9E3779B1u;       // (seed: a second attempt draws other samples)
            h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
            const uint32_t i = lo + h % (hi - lo);
            w = (fs_code_at(s_tab, T, n, i) & ~FS_LOW_MASK) | ((uint64_t)i << 8);
        }
        s_s[j] = w;
    }
    __syncthreads();
    SS_CLK(0);                                                 // samples drawn
#ifndef GLC_SSA_NETWORK_ONLY
    // (a) the words as plain integers: (code, position).  No text is read.  A full sample (16384 words, 16 per thread) goes
    // through the bitonic network up to four stages at a time: a thread takes the 16 words whose indices differ in four given
    // bits and does every stage that pairs words across those bits in registers -- 33 trips through LDS instead of 105
    // (ssa_sort_pass): 157 -> 78 us.
    if (S2 == SS_MAXS) {
#ifndef GLC_SSA_PLAIN_STAGES
        uint64_t v[16];
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = s_s[a * SSA_NT + tid];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; a++) s_s[ssa_phys(a * SSA_NT + tid)] = v[a];
        __syncthreads();
        ssa_sort_pass<0, 4, true>(s_s, tid, 0);                // levels 1 .. 4
        for (int L = 5; L <= 14; L++) {
            const uint32_t k = 1u << L;
            if (L >= 11) ssa_sort_pass<10, 4, false>(s_s, tid, k);
            if (L >= 7) ssa_sort_pass<6, 4, false>(s_s, tid, k);
            ssa_sort_pass<2, 4, false>(s_s, tid, k);
            ssa_sort_pass<0, 2, false>(s_s, tid, k);
        }
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = s_s[ssa_phys(a * SSA_NT + tid)];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; a++) s_s[a * SSA_NT + tid] = v[a];
        __syncthreads();
#endif
    }
    for (uint32_t k = 2; k <= S2; k <<= 1) {
#ifndef GLC_SSA_PLAIN_STAGES
        if (S2 == SS_MAXS) break;
#endif
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < S2 / 2; t += SSA_NT) {
                const uint32_t low = t & (j - 1), i = ((t - low) << 1) + low, q = i + j;
                const uint64_t a = s_s[i], c = s_s[q];
                if (((i & k) == 0) ? c < a : a < c) { s_s[i] = c; s_s[q] = a; }
            }
            __syncthreads();
        }
    }
    SS_CLK(1);                                                 // integer sort
    // (b) runs of equal codes are ordered by the text, wave by wave.  A wave takes the runs that START in a slab of 32 places, all
    // of them at once (a run at a time was a chain of ~150 memory round trips per wave), as one window of up to 120 places; at the
    // start the sub-runs [a, b) are the runs and every member stands at its place.  A round gathers 16 text bytes per member still
    // tied, puts the keys at the members' places in LDS (16 B x 120 per wave: what is left beside the samples), and every member
    // walks over the other places of its sub-run -- all members at once, as many steps as the longest sub-run of the window is
    // long: keys below it, keys equal, equal ones standing before it -> its new sub-run and place, where it then moves (its
    // position goes through the same LDS).  These are fs_suffix_less's 16-byte steps, so the order is the network's order; a
    // run with a tied member whose next 16 bytes come within 4 of the end of the text (where fs_suffix_less changes its step)
    // leaves the rounds and is ranked pair by pair with fs_suffix_less itself.  A run that does not end inside its window is ordered on its own if it
    // has up to 256 members (they stay in their lanes, four to a lane, and are counted class by class -- the members of one sub-run
    // with one key together: key broadcast with v_readlane, two ballots per slot); a longer one (one code on 1.5 % of the samples: not text) sends the block to the
    // network with the text comparisons in it, which works from any order -- decided before any of this work is done.
    if (tid == 0) { s_work = 0; s_big = 0; }
    __syncthreads();
    {
        bool big = false;
        for (uint32_t j = tid; j + SSA_LONG_CAP < S; j += SSA_NT) big |= (s_s[j] >> 28) == (s_s[j + SSA_LONG_CAP] >> 28);
        if (big) s_big = 1;
    }
    __syncthreads();
    if (tid == 0 && s_big) SS_COUNT(43);
    if (s_big == 0) {
        const uint32_t lane = tid & 63u;
        const uint64_t upto = ~0ull >> (63u - lane);           // bits <= lane
        volatile uint32_t *vdeep = &s_deep;
        ulonglong2 *s_kw = s_k[tid >> 6];                      // this wave's keys, at their members' places
        uint2 *s_xw = reinterpret_cast<uint2 *>(s_kw);         // ... and, between two rounds, {position, sub-run} on the move
        bool anydeep = false;
        for (;;) {
            uint32_t slab = 0;
            if (lane == 0) slab = atomicAdd(&s_work, 1u);
            slab = (uint32_t)__builtin_amdgcn_readfirstlane((int)slab);
            const uint32_t w0 = slab * SSA_SLAB;
            if (w0 >= S || (!tol && *vdeep)) break;
            uint64_t wm[2], bd[2];
            uint32_t idx[2], ab[2], run0[2];
            bool in[2], unf[2];
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const uint32_t t = 64u * sl + lane, q = w0 + t;
                const bool valid = t < SSA_WIN;
                wm[sl] = valid && q < S ? s_s[q] : ~0ull;
                const uint64_t before = valid && q > 0 && q <= S ? s_s[q - 1] >> 28 : ~0ull;
                bd[sl] = __ballot(valid && (q >= S || q == 0 || before != (wm[sl] >> 28)));
                idx[sl] = (uint32_t)(wm[sl] >> 8) & 0xFFFFFu;
            }
            const uint32_t qe = w0 + SSA_WIN;                  // the place behind the window: does a run go on there?
            const bool open_end = qe < S && (s_s[qe] >> 28) == (s_s[qe - 1] >> 28);
            uint32_t long_head = 0xFFFFu;
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const uint32_t t = 64u * sl + lane, q = w0 + t;
                uint32_t head = 0xFFFFu, end = 0xFFFFu;        // (places in the window; 0xFFFF: outside it)
                const uint64_t hb = bd[sl] & upto, eb = bd[sl] & ~upto;
                if (hb) head = 64u * sl + 63u - (uint32_t)__builtin_clzll(hb);
                else if (sl == 1 && bd[0]) head = 63u - (uint32_t)__builtin_clzll(bd[0]);
                if (eb) end = 64u * sl + (uint32_t)__builtin_ctzll(eb);
                else if (sl == 0 && bd[1]) end = 64u + (uint32_t)__builtin_ctzll(bd[1]);
                else if (!open_end) end = SSA_WIN;
                const bool mine = t < SSA_WIN && q < S && head < SSA_SLAB;     // its run starts in this slab
                const uint64_t lost = __ballot(mine && end == 0xFFFFu);
                if (lost) long_head = (uint32_t)__builtin_amdgcn_readlane((int)head, __builtin_ctzll(lost));
                in[sl] = mine && end != 0xFFFFu && end - head > 1;
                unf[sl] = in[sl];
                ab[sl] = head | (end << 16);
                run0[sl] = ab[sl];
            }
            if (__any(in[0] || in[1])) {
                uint32_t d = 0;
                bool pairwise = false, pw[2] = {false, false};     // places of the runs set aside for the pair-by-pair ranking
                for (;;) {
                    if (!__any(unf[0] || unf[1])) break;
                    // a tied member whose next 16 bytes come within 4 of the end of the text: ITS run leaves the rounds (the
                    // window's other runs go on) and is ranked pair by pair below -- ranking the whole window that way was a chain
                    // of 120 x 2 walks, 0.25 ms, and the slowest block is what a kernel with one workgroup per block takes
#pragma unroll
                    for (int ss = 0; ss < 2; ss++) {
                        uint64_t shm = __ballot(unf[ss] && idx[ss] + d + 20 > n);
                        while (shm) {
                            const uint32_t R = (uint32_t)__builtin_amdgcn_readlane((int)run0[ss], __builtin_ctzll(shm));
                            shm &= shm - 1;
#pragma unroll
                            for (int sl = 0; sl < 2; sl++) if (in[sl] && run0[sl] == R) { pw[sl] = true; unf[sl] = false; }
                            pairwise = true;
                        }
                    }
                    if (!__any(unf[0] || unf[1])) break;
                    if (!tol && (d > FS_LCP_CAP || *vdeep)) { anydeep = true; break; }
                    const bool by_place = tol && d > SS_TOL_CAP;   // tied up to the cap: by position (fs_suffix_less's rule)
                    uint64_t kh[2], kl[2];
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) {
                        kh[sl] = 0; kl[sl] = 0;
                        if (by_place) kl[sl] = idx[sl];
                        else if (unf[sl]) fs_load_be128(T + idx[sl] + d, kh[sl], kl[sl]);
                        if (unf[sl]) s_kw[64u * sl + lane] = make_ulonglong2(kh[sl], kl[sl]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    uint32_t cl[2] = {0, 0}, ce[2] = {0, 0}, ct[2] = {0, 0};
                    // (a round in which every sub-run's keys are all equal -- the inside of a deep repeat, round after round up to
                    //  the cap -- changes nothing: seen from one look at each member's successor)
                    bool differs = false;
#pragma unroll
                    for (int sl = 0; sl < 2; sl++)
                        if (unf[sl]) {
                            const uint32_t t = 64u * sl + lane, A = ab[sl] & 0xFFFFu, B = ab[sl] >> 16;
                            const ulonglong2 k = s_kw[t + 1 < B ? t + 1 : A];
                            differs |= k.x != kh[sl] || k.y != kl[sl];
                        }
                    if (!__any(differs)) { if (by_place) break; d += 16; continue; }
                    for (uint32_t dl = 1;; dl++) {
                        bool act[2];
#pragma unroll
                        for (int sl = 0; sl < 2; sl++) act[sl] = unf[sl] && dl < (ab[sl] >> 16) - (ab[sl] & 0xFFFFu);
                        if (!__any(act[0] || act[1])) break;
#pragma unroll
                        for (int sl = 0; sl < 2; sl++) {
                            if (!act[sl]) continue;
                            const uint32_t t = 64u * sl + lane, A = ab[sl] & 0xFFFFu, B = ab[sl] >> 16;
                            uint32_t peer = t + dl;
                            if (peer >= B) peer -= B - A;
                            const ulonglong2 k = s_kw[peer];
                            const bool eq = k.x == kh[sl] && k.y == kl[sl];
                            cl[sl] += (k.x < kh[sl] || (k.x == kh[sl] && k.y < kl[sl])) ? 1u : 0u;
                            ce[sl] += eq ? 1u : 0u;
                            ct[sl] += (eq && peer < t) ? 1u : 0u;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sl = 0; sl < 2; sl++)
                        if (unf[sl]) {
                            const uint32_t A = (ab[sl] & 0xFFFFu) + cl[sl];
                            s_xw[A + ct[sl]] = make_uint2(idx[sl], A | ((A + ce[sl] + 1u) << 16));
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sl = 0; sl < 2; sl++)
                        if (unf[sl]) {
                            const uint2 x = s_xw[64u * sl + lane];
                            idx[sl] = x.x; ab[sl] = x.y;
                            unf[sl] = (x.y >> 16) - (x.y & 0xFFFFu) > 1;
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (by_place) break;
                    d += 16;
                }
                if (pairwise) {
                    SS_COUNT(41);
                    uint32_t cnt[2] = {0, 0};
                    for (uint32_t j = 0; j < SSA_WIN; j++) {
                        const int lj = (int)(j & 63u);
                        const uint32_t ixj = (uint32_t)(j < 64 ? __builtin_amdgcn_readlane((int)idx[0], lj) : __builtin_amdgcn_readlane((int)idx[1], lj));
                        const uint32_t rj = (uint32_t)(j < 64 ? __builtin_amdgcn_readlane((int)run0[0], lj) : __builtin_amdgcn_readlane((int)run0[1], lj));
                        const uint32_t pj = (uint32_t)(j < 64 ? __builtin_amdgcn_readlane((int)(pw[0] ? 1u : 0u), lj) : __builtin_amdgcn_readlane((int)(pw[1] ? 1u : 0u), lj));
                        if (!pj) continue;
#pragma unroll 1
                        for (int sl = 0; sl < 2; sl++) {
                            const uint32_t mine = sl ? idx[1] : idx[0], myrun = sl ? run0[1] : run0[0];
                            const bool have = sl ? pw[1] : pw[0];
                            if (!have || myrun != rj || mine == ixj) continue;
                            bool dp = false;
                            const bool lt = fs_suffix_less<true>(T, n, ixj, mine, &dp, 0, tol);
                            anydeep |= dp;
                            if (sl) cnt[1] += lt ? 1u : 0u; else cnt[0] += lt ? 1u : 0u;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) if (pw[sl]) s_xw[min((run0[sl] & 0xFFFFu) + cnt[sl], SSA_WIN - 1u)] = make_uint2(idx[sl], 0u);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) if (pw[sl]) idx[sl] = s_xw[64u * sl + lane].x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (!tol && anydeep) *vdeep = 1;
#pragma unroll
                for (int sl = 0; sl < 2; sl++)
                    if (in[sl]) s_s[w0 + 64u * sl + lane] = (wm[sl] & ~(uint64_t)0x0FFFFF00u) | ((uint64_t)idx[sl] << 8);
            }
            if (long_head == 0xFFFFu || (!tol && *vdeep)) continue;
            // the run that leaves the window: up to SSA_LONG_CAP members, in their lanes for good
            {
                SS_COUNT(42);
                const uint32_t rs = w0 + long_head;
                const uint64_t c = s_s[rs] >> 28;
                uint64_t wl[SSA_SLOTS];
                uint32_t ix[SSA_SLOTS], lab[SSA_SLOTS], pos[SSA_SLOTS];
                bool lin[SSA_SLOTS], lunf[SSA_SLOTS];
                uint32_t r = 0;
#pragma unroll
                for (int sl = 0; sl < SSA_SLOTS; sl++) {
                    const uint32_t q = rs + 64u * sl + lane;
                    wl[sl] = q < S ? s_s[q] : ~0ull;
                    lin[sl] = q < S && (wl[sl] >> 28) == c;
                    r += (uint32_t)__popcll(__ballot(lin[sl]));
                    ix[sl] = (uint32_t)(wl[sl] >> 8) & 0xFFFFFu;
                }
                const int ns = (int)((r + 63u) >> 6);
#pragma unroll
                for (int sl = 0; sl < SSA_SLOTS; sl++) { lunf[sl] = lin[sl]; lab[sl] = r << 16; pos[sl] = lin[sl] ? 64u * sl + lane : 0xFFFFu; }
                uint32_t d = 0;
                bool pairwise = false;
                for (;;) {
                    bool sh = false, any = false;
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) { sh |= lunf[sl] && ix[sl] + d + 20 > n; any |= lunf[sl]; }
                    if (!__any(any)) break;
                    if (__any(sh)) { pairwise = true; break; }
                    if (!tol && (d > FS_LCP_CAP || *vdeep)) { anydeep = true; break; }
                    const bool by_place = tol && d > SS_TOL_CAP;
                    uint64_t kh[SSA_SLOTS], kl[SSA_SLOTS];
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) {
                        kh[sl] = 0; kl[sl] = 0;
                        if (by_place) kl[sl] = ix[sl];
                        else if (lunf[sl]) fs_load_be128(T + ix[sl] + d, kh[sl], kl[sl]);
                    }
                    // counted class by class: the members of one sub-run with one key get their new sub-run and places together (the
                    // key broadcast with v_readlane, two ballots per slot; equal keys keep their order, which inside a sub-run is the
                    // order of the lanes: places are handed out that way from the start).  A deep repeat is ONE class round after
                    // round, and two in the round in which a member reaches the end of the repeat -- member by member that round
                    // cost 20 us for 140 members.
                    uint32_t nab[SSA_SLOTS], npos[SSA_SLOTS];
                    bool todo[SSA_SLOTS];
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) { nab[sl] = lab[sl]; npos[sl] = pos[sl]; todo[sl] = lunf[sl]; }
                    for (;;) {
                        bool found = false;
                        uint64_t KH = 0, KL = 0;
                        uint32_t AB = 0;
#pragma unroll
                        for (int sl = 0; sl < SSA_SLOTS; sl++) {
                            const uint64_t mk = __ballot(todo[sl]);
                            if (!found && mk) {
                                const int li = __builtin_ctzll(mk);
                                KH = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(kh[sl] >> 32), li) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kh[sl], li);
                                KL = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(kl[sl] >> 32), li) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kl[sl], li);
                                AB = (uint32_t)__builtin_amdgcn_readlane((int)lab[sl], li);
                                found = true;
                            }
                        }
                        if (!found) break;
                        const uint32_t A = AB & 0xFFFFu;
                        uint32_t cl = 0, ce = 0;
                        uint64_t eqm[SSA_SLOTS];
                        bool eq[SSA_SLOTS];
#pragma unroll
                        for (int sl = 0; sl < SSA_SLOTS; sl++) {
                            eqm[sl] = 0; eq[sl] = false;
                            if (sl >= ns) continue;
                            const bool inr = lunf[sl] && lab[sl] == AB;
                            eq[sl] = inr && kh[sl] == KH && kl[sl] == KL;
                            cl += (uint32_t)__popcll(__ballot(inr && (kh[sl] < KH || (kh[sl] == KH && kl[sl] < KL))));
                            eqm[sl] = __ballot(eq[sl]);
                            ce += (uint32_t)__popcll(eqm[sl]);
                        }
                        uint32_t before = 0;
#pragma unroll
                        for (int sl = 0; sl < SSA_SLOTS; sl++) {
                            if (eq[sl]) {
                                nab[sl] = (A + cl) | ((A + cl + ce) << 16);
                                npos[sl] = A + cl + before + (uint32_t)__popcll(eqm[sl] & ((1ull << lane) - 1ull));
                                todo[sl] = false;
                            }
                            before += (uint32_t)__popcll(eqm[sl]);
                        }
                    }
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) { lab[sl] = nab[sl]; pos[sl] = npos[sl]; lunf[sl] = lunf[sl] && (lab[sl] >> 16) - (lab[sl] & 0xFFFFu) > 1; }
                    if (by_place) break;
                    d += 16;
                }
                if (pairwise) {
                    uint32_t cnt[SSA_SLOTS];
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) cnt[sl] = 0;
                    for (uint32_t j = 0; j < r; j++) {
                        uint32_t ixj = 0;
#pragma unroll
                        for (int sl = 0; sl < SSA_SLOTS; sl++) if ((int)(j >> 6) == sl) ixj = (uint32_t)__builtin_amdgcn_readlane((int)ix[sl], (int)(j & 63u));
#pragma unroll 1
                        for (int sl = 0; sl < ns; sl++) {
                            uint32_t mine = 0;
                            bool have = false;
#pragma unroll
                            for (int v = 0; v < SSA_SLOTS; v++) if (v == sl) { mine = ix[v]; have = lin[v]; }
                            if (!have || mine == ixj) continue;
                            bool dp = false;
                            const bool lt = fs_suffix_less<true>(T, n, ixj, mine, &dp, 0, tol);
                            anydeep |= dp;
#pragma unroll
                            for (int v = 0; v < SSA_SLOTS; v++) if (v == sl) cnt[v] += lt ? 1u : 0u;
                        }
                    }
#pragma unroll
                    for (int sl = 0; sl < SSA_SLOTS; sl++) pos[sl] = min(cnt[sl], r - 1u);
                }
                if (!tol && anydeep) *vdeep = 1;
#pragma unroll
                for (int sl = 0; sl < SSA_SLOTS; sl++) if (lin[sl]) s_s[rs + pos[sl]] = wl[sl];
            }
        }
        if (!tol && anydeep) s_deep = 1;
    }
    __syncthreads();
    const bool ranked = s_big == 0;
    SS_CLK(3);                                                 // runs ordered
    SS_MAX(tol ? 45 : 44, 3);
#else
    const bool ranked = false;
#endif
    uint32_t ties_before = 0;                                  // (tolerant form) ties counted up to the last stage
    bool many = false;
    for (uint32_t k = 2; !ranked && k <= S2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < S2 / 2; t += SSA_NT) {
                const uint32_t low = t & (j - 1), i = ((t - low) << 1) + low, q = i + j;
                const uint64_t a = s_s[i], c = s_s[q];
                bool deep = false;
                const bool up = (i & k) == 0;                  // ascending run?
                const bool swap = up ? ss_word_less(c, a, T, n, &deep, tol) : ss_word_less(a, c, T, n, &deep, tol);
                if (deep) { if (tol) atomicAdd(&s_ties, 1u); else s_deep = 1; }
                if (swap) { s_s[i] = c; s_s[q] = a; }
            }
            __syncthreads();
            // tolerant form: a comparison that ran into the cap is a tie, not a give-up -- but a block where they are the
            // rule (periodic data, a block made of copies: a quarter or more of a stage's comparisons walk to the cap) is
            // no business of this tier, and is given up at the first such stage (a duplicated 20 KB makes ~600 ties per
            // stage of 8192 comparisons, a repeated page all of them)
            if (tol) { const uint32_t now = s_ties; many = now - ties_before > S2 / 8; ties_before = now; }
            if (s_deep || many) break;                         // (uniform: read after the barrier, written before it)
        }
        if (s_deep || many) break;
    }
    SS_CLK(5);                                                 // the network with text comparisons (when it runs)
    if (s_deep || many) { if (tid == 0) atomicOr(&flag[b], 2u); SS_CLK_END(32); return; }
    if (tol) {
        // the tolerant form is for blocks with deep repeats INSIDE otherwise ordinary data.  Where a quarter of the
        // neighbouring samples agree beyond the cap (periodic data, a block made of copies) nearly every suffix would be left
        // to the doubling rounds anyway, and every round of this tier on the way there is wasted: such a block is given up
        // here and takes the general sorter from scratch (repeated 4 KiB page, 64 blocks: 148 ms resumed, 116 from scratch)
        uint32_t ties = 0;
        for (uint32_t j = tid + 1; j < S; j += SSA_NT) {
            const uint64_t a = s_s[j - 1], c = s_s[j];
            if ((a >> 28) == (c >> 28)) {
                const uint32_t ia = (uint32_t)(a >> 8) & 0xFFFFFu, ic = (uint32_t)(c >> 8) & 0xFFFFFu;
                bool same = max(ia, ic) + SS_TOL_CAP + 8 <= n;
                for (uint32_t k = 0; same && k < SS_TOL_CAP + 8; k += 8) same = fs_load_be64(T + ia + k) == fs_load_be64(T + ic + k);
                ties += same ? 1u : 0u;
            }
        }
        __syncthreads();
        if (tid == 0) s_ties = 0;
        __syncthreads();
        if (ties) atomicAdd(&s_ties, ties);
        __syncthreads();
        if (s_ties * 4u > S) { if (tid == 0) atomicOr(&flag[b], 2u); return; }
    }
    for (uint32_t k = tid; k < nb; k += SSA_NT) {
        const uint64_t sw = k ? s_s[(uint32_t)(((uint64_t)k * S) / nb)] : 0ull;
        split[(size_t)b * FS_MAXNB + k] = sw;
        const uint32_t is = (uint32_t)(sw >> 8) & 0xFFFFFu;  // its first 8 text bytes, 0 past the end (k_fs_part<true>)
        uint64_t f8 = 0;
        if (is + 8 <= n) f8 = fs_load_be64(T + is);
        else for (uint32_t t = 0; t < 8; t++) f8 = (f8 << 8) | (is + t < n ? (uint64_t)T[is + t] : 0ull);
        split8[(size_t)b * FS_MAXNB + k] = f8;
        uint64_t f16 = 0;                                      // ... and the 8 bytes behind them
        if (is + 16 <= n) f16 = fs_load_be64(T + is + 8);
        else for (uint32_t t = 8; t < 16; t++) f16 = (f16 << 8) | (is + t < n ? (uint64_t)T[is + t] : 0ull);
        split16[(size_t)b * FS_MAXNB + k] = f16;
    }
    // every suffix of a bucket lies between its two splitters and shares their common prefix: l0 of bucket k, here
    // for all buckets at once (in k_ss_cut it was three dependent memory round trips of ONE thread, with the other
    // 1023 of the workgroup waiting at the first barrier)
    for (uint32_t k = tid; k < nb; k += SSA_NT) {
        uint32_t l0 = 0;
        if (k >= 1 && k + 1 < nb) {
            const uint32_t ia = (uint32_t)(s_s[(uint32_t)(((uint64_t)k * S) / nb)] >> 8) & 0xFFFFFu;
            const uint32_t ib = (uint32_t)(s_s[(uint32_t)(((uint64_t)(k + 1) * S) / nb)] >> 8) & 0xFFFFFu;
            if (ia != ib) {
                const uint32_t m = max(ia, ib);
                while (l0 < SS_L0_CAP && m + l0 + 12 <= n) {
                    const uint64_t x = fs_load_be64(T + ia + l0) ^ fs_load_be64(T + ib + l0);
                    if (x) { l0 += (uint32_t)__builtin_clzll(x) >> 3; break; }
                    l0 += 8;
                }
            }
        }
        l0_out[(size_t)b * FS_MAXNB + k] = l0;
    }
    SS_CLK(6);                                                 // tolerant check, splitters, l0
    // cell[x] = first splitter (counted from 1) whose leading 12 code bits are >= x; nb if there is none
    for (uint32_t x = tid; x < SS_CELLS + 2; x += SSA_NT) {
        uint32_t lo = 1, hi = nb;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)(s_s[(uint32_t)(((uint64_t)mid * S) / nb)] >> 52) < x) lo = mid + 1; else hi = mid;
        }
        cell[(size_t)b * (SS_CELLS + 2) + x] = (uint16_t)lo;
    }
    SS_CLK(7);                                                 // cells
    SS_CLK_END(32);
}

// 7 symbols from position i as 9-bit digits (symbol + 1; 0 behind the end of the block: the shorter suffix is smaller).
// Two steps so that the loads of all of a thread's suffixes are in flight together: ss_sym_load returns the 8 bytes
// at T + i (big-endian), or the symbols already as digits when the suffix ends within 12 bytes (bit 63 marks that).
constexpr uint32_t SS_STEP = 7;

__device__ __forceinline__ uint64_t ss_sym_load(const uint8_t *T, uint32_t n, uint32_t i)
{
    if (i + 12 <= n) return fs_load_be64(T + i) >> 1;          // (bit 63 clear; the dropped bit belongs to the 8th byte)
    uint64_t k = 0;
#pragma unroll
    for (int j = 0; j < (int)SS_STEP; j++) k = (k << 9) | (i + j < n ? (uint64_t)T[i + j] + 1u : 0ull);
    return k | (1ull << 63);
}

__device__ __forceinline__ uint64_t ss_sym_key(uint64_t raw)
{
    if (raw >> 63) return raw & ~(1ull << 63);
    uint64_t k = 0;
#pragma unroll
    for (int j = 0; j < (int)SS_STEP; j++) k = (k << 9) | (((raw >> (55 - 8 * j)) & 0xFFu) + 1u);
    return k;
}

#ifndef GLC_SSS_NT
#define GLC_SSS_NT 512
#endif
constexpr int SSS_NT = GLC_SSS_NT;                              // k_ss_cut: threads
constexpr uint32_t SS_NPIV = 64, SS_NBIN = 2 * SS_NPIV + 1;
#ifndef GLC_SS_NPL
#define GLC_SS_NPL 4
#endif
constexpr uint32_t SS_NPL = GLC_SS_NPL, SS_NPIV0 = 64 * SS_NPL; // first cut: 256 pivots
// shares of a bucket's positions handed out to the waves (one per wave).  A share is what a wave cuts into windows, so it
// should hold several full windows: with 64 shares of ~32 positions each window filled an eighth of the wave's 256
// slots and k_ss_windows took 8.0 ms per 256 text blocks; 32 / 16 / 8 / 4 shares: 6.6 / 5.8 / 5.4 / 5.3 ms (7-byte rounds).
// With the 14-byte rounds (shares x waves per bucket, whole text256 encode): 16x4 13.84 ms, 8x4 13.43, 8x2 13.51, 8x8 13.50,
// 4x4 13.28, 4x2 13.27, 2x2 13.20, 3x3 13.16 (profiles/r05_dissect.md).
#ifndef GLC_SS_SHARES
#define GLC_SS_SHARES 3
#endif
constexpr uint32_t SS_SHARES = GLC_SS_SHARES;
constexpr uint32_t SS_WIN = 256;                               // positions a wave finishes at a time (4 per lane)
constexpr uint32_t SSL_SMALL = GLC_SSL_SMALL;                           // k_ss_long: members of a "small" long bin
// (SS_LONG, glc_internal.h: runs longer than that are cut with pivots by k_ss_long; shorter ones are counted out in the windows)
// k_ss_windows' form of a round's key: the SS_STEP = 7 text bytes themselves in the top 56 bits (0 past the end of
// the text) and the window slot in the low 8, so that no two keys of a window are equal: a position's new place is
// ONE count (keys below it) and the start of its new run a second one (keys below the key with slot 0).  Bytes cannot
// tell a suffix that ENDS from one that goes on with zero bytes, so a round with a member whose 7 bytes reach the end of
// the text (i + 8 > n: the last suffixes of a block) is done with the 9-bit digits above instead.
__device__ __forceinline__ uint64_t ss_raw7(const uint8_t *T, uint32_t n, uint32_t i)
{
    (void)n;
    return fs_load_be64(T + i) & ~0xFFull;                     // (callers: i + 8 <= n)
}
constexpr uint32_t SS_MAXSTEP = FS_LCP_CAP / SS_STEP + 1;      // rounds of a run before the block is given up as deep
// ... or, in the tolerant form (ss_build), SS_TOL_MAXSTEP rounds before the run is left as it is: its members agree in more
// than SS_TOL_CAP + 8 bytes, the rounds of prefix doubling behind the sample sorter order them.  A capped run's step count:
constexpr uint32_t SS_TOL_MAXSTEP = SS_TOL_CAP / SS_STEP + 1;
constexpr uint32_t SS_CAPPED = 255;
static_assert(SS_MAXSTEP + 2 < SS_CAPPED, "the step field of a run descriptor holds the rounds and the marker");
// the resumed doubling starts at depth SS_TOL_CAP: every place the tolerant form stops at (fs_suffix_less: k > SS_TOL_CAP in steps
// of 8; the runs: SS_TOL_MAXSTEP rounds of SS_STEP symbols) must lie at or beyond it
static_assert(SS_TOL_CAP % 8 == 0 && SS_TOL_CAP <= FS_LCP_CAP && SS_TOL_MAXSTEP * SS_STEP >= SS_TOL_CAP,
              "GLC_SS_TOL_CAP: a multiple of 8, not beyond the give-up cap, reached by the tolerant rounds");

// run descriptor of a position: start : 12 | end : 12 | rounds done : 8   (a decided position: end = start + 1)
__device__ __forceinline__ uint32_t ss_run(uint32_t ss, uint32_t se, uint32_t st) { return ss | (se << 12) | (st << 24); }

// sorts one key per lane across the wave (ascending by lane)
__device__ __forceinline__ uint64_t wave_sort_u64(uint64_t k, uint32_t lane)
{
#pragma unroll
    for (uint32_t kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)k, (int)j), hi = (uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), (int)j);
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            const bool up = (lane & kk) == 0, lower = (lane & j) == 0;
            const uint64_t mn = o < k ? o : k, mx = o < k ? k : o;
            k = (up == lower) ? mn : mx;
        }
    }
    return k;
}

// bin of a key among 64 sorted pivots held one per lane: 2 i = between pivot i-1 and pivot i, 2 i + 1 = equal to pivot i
// (the first of equal pivots).  All lanes must call it together.
__device__ __forceinline__ uint32_t ss_pivot_bin(uint64_t piv, uint64_t key)
{
    uint32_t lo = 0, hi = SS_NPIV;                             // first pivot >= key
#pragma unroll
    for (int it = 0; it < 7; it++) {
        const uint32_t mid = (lo + hi) >> 1, m = mid < SS_NPIV ? mid : SS_NPIV - 1;
        const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)piv, (int)m), phi = (uint32_t)__shfl((int)(uint32_t)(piv >> 32), (int)m);
        const uint64_t pm = ((uint64_t)phi << 32) | plo;
        if (lo < hi) { if (pm < key) lo = mid + 1; else hi = mid; }
    }
    const uint32_t l = lo < SS_NPIV ? lo : SS_NPIV - 1;
    const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)piv, (int)l), phi = (uint32_t)__shfl((int)(uint32_t)(piv >> 32), (int)l);
    const uint64_t pl = ((uint64_t)phi << 32) | plo;
    return 2 * lo + ((lo < SS_NPIV && pl == key) ? 1u : 0u);
}

// The sort of a bucket, in two kernels.  The suffixes are ordered in rounds of 7 symbols read from the text.  A RUN is
// a range of positions whose suffixes agree in everything looked at so far (each run carries its own depth); the
// whole bucket is the first run.
//   k_ss_cut     (one workgroup per bucket, everything in LDS) cuts LONG runs with 64 PIVOTS -- keys of 64 of the
//                run's members, sorted by one wave -- into the bins "between two pivots" (runs at the same depth, ~1/65
//                of the size whatever the key distribution is: text is anything but uniform) and "equal to a pivot"
//                (runs one round deeper; a key that hundreds of members share is almost surely a pivot).  Members
//                count into the bins with LDS atomics.  The first cut is made by the whole workgroup, later ones
//                (runs still longer than a window) by single waves.  The words go back to their slot as
//                [run descriptor : 32 | index : 20 | bwt : 8 ...].
//   k_ss_windows (four waves per bucket, each on its own) finishes WINDOWS of up to 256 positions in registers, 4 per
//                lane: gather 8 bytes, write the key, count the smaller keys of the run, move; equal keys = a run of
//                the next round.  Waves take shares of the bucket from a counter and never wait for each other:
//                when this was the tail of the cutting kernel, the slowest of 16 waves took 3x the mean and the
//                other 15 sat on 73 KB of LDS meanwhile.
// Nothing here depends on the symbol statistics.
// sorts one 128-bit key {hi, lo} per lane across the wave (ascending by lane)
__device__ __forceinline__ void wave_sort_u128(uint64_t &h, uint64_t &l, uint32_t lane)
{
#pragma unroll
    for (uint32_t kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            const uint64_t oh = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(h >> 32), (int)j) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)h, (int)j);
            const uint64_t ol = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(l >> 32), (int)j) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)l, (int)j);
            const bool up = (lane & kk) == 0, lower = (lane & j) == 0;
            const bool oless = (oh < h) | ((oh == h) & (ol < l));
            const bool take = (up == lower) ? oless : !oless;  // keep the smaller of the pair in the lower lane of an ascending half
            const bool same = (oh == h) & (ol == l);
            if (take && !same) { h = oh; l = ol; }
        }
    }
}

__device__ __forceinline__ bool ss_less128(const ulonglong2 a, uint64_t h, uint64_t l) { return (a.x < h) | ((a.x == h) & (a.y < l)); }

#ifndef GLC_SSC_SIDE
#define GLC_SSC_SIDE 2                                      // k_ss_cut: pivot searches of a thread that run side by side
#endif
#ifndef GLC_SSC_WPE
#define GLC_SSC_WPE 8
#endif
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(GLC_SSC_WPE, 8))) void k_ss_cut(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                               uint32_t nbl, uint64_t *__restrict__ keys, size_t kstride,
                                               const uint32_t *__restrict__ fill,
                                               uint32_t *__restrict__ flag, const uint32_t *__restrict__ list,
                                               const uint32_t *__restrict__ l0_in, uint2 *__restrict__ long_list,
                                               size_t long_cap, unsigned long long *__restrict__ long_count)
{
    // The first cut of a bucket, nothing else: keys and words stay in registers, the pivots are gathered on their own
    // (256 threads read the word and the text of one sample each, beside the loads of everybody's positions), and LDS
    // only stages the bucket on its way back to the slot -- 35 KB, four workgroups per CU, where the cut-until-done form
    // (keys, words and runs of 4032 positions: 73 KB) had two, each a chain of memory and LDS round trips between barriers.
    // Round 6: the key is FOURTEEN text bytes from ONE 16-byte gather ({bytes 0..7, bytes 8..13 << 16}: two steps of the run
    // descriptors' unit, the windows' form of a round).  A gather of 16 bytes costs what one of 8 does, and a bin "equal to a
    // pivot" is then two steps deeper and a fraction of the size: log lines share "2026-09-28T12:3" and " host-17 svc-" --
    // on 7 bytes a bucket's first cut left ~800 bins of more than a window per block for k_ss_long (3.1 ms per 256 blocks),
    // text ~150.  The pivot lists and the merged pivots live where the bucket is staged afterwards.
    constexpr int ITEMS = FS_CAP / NT;
    static_assert(NT >= (int)SS_NPIV0 && FS_CAP % NT == 0, "one pivot sample per thread of the first four waves");
    static_assert(2 * SS_NPIV0 * sizeof(ulonglong2) <= FS_FILLMAX * sizeof(uint64_t), "pivot lists + merged pivots inside the staging array");
    __shared__ __attribute__((aligned(16))) uint64_t s_out[FS_FILLMAX];   // the bucket in its new order: [run : 32 | index : 20 | bwt : 8 ...]
    __shared__ uint32_t s_cnt[2 * SS_NPIV0 + 4];               // bin counters, then bin starts (+ end)
    __shared__ uint32_t s_nlong, s_bound[64];
    __shared__ unsigned long long s_at;
    ulonglong2 *s_pl = reinterpret_cast<ulonglong2 *>(s_out);  // [SS_NPL][64] the sorted lists the pivots are merged from ...
    ulonglong2 *s_piv0 = s_pl + SS_NPIV0;                      // ... and the pivots, sorted (both dead before the bucket is staged)
    uint32_t gx, gy;
    xcd_order(gx, gy);
    const uint32_t b = list[gy], bk = gx, tid = threadIdx.x;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint8_t *T = text + (size_t)b * stride;
    const uint32_t c = fill[(size_t)b * FS_MAXNB + bk];
    if (flag[b] || c == 0 || c > FS_FILLMAX) return;           // (uniform; the flags are set by earlier kernels only)
    uint64_t *K = keys + (size_t)b * kstride + (size_t)bk * FS_CAP;
    SS_CLK_BEGIN();
    if (c == 1) { if (tid == 0) K[0] = (K[0] & FS_LOW_MASK) | ((uint64_t)ss_run(0, 1, 0) << 32); return; }
    const uint32_t l0 = l0_in[(size_t)b * FS_MAXNB + bk];      // common prefix of the bucket's two splitters (k_ss_sample)
    uint32_t vv[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t p = r * NT + tid;
        vv[r] = p < c ? (uint32_t)(K[p] & FS_LOW_MASK) : 0u;
    }
    uint32_t sv = 0;                                           // pivot sample of this thread: position tid c / 256
    if (tid < SS_NPIV0) sv = (uint32_t)(K[(uint32_t)(((uint64_t)tid * c) / SS_NPIV0)] & FS_LOW_MASK);
    for (uint32_t i = tid; i < 2 * SS_NPIV0 + 4; i += NT) s_cnt[i] = 0;
    if (tid == 0) s_nlong = 0;
    SS_CLK(0);                                                 // words loaded
    // a member whose 16 bytes reach the end of the text: the whole cut takes ONE step with the 9-bit digits that tell "ended"
    // from a zero byte (63 bits in the key's high half, the low half 0)
    bool tl = false;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) tl |= r * NT + tid < c && (vv[r] >> 8) + l0 + 20 > n;
    const bool digits = __syncthreads_or(tl) != 0;
    const uint32_t deeper = digits ? 1u : 2u;                  // steps a bin "equal to a pivot" is deeper than its run
    uint64_t kh[ITEMS], kl[ITEMS], sh = 0, sl = 0;
    if (tid < SS_NPIV0) {
        if (digits) sh = ss_sym_load(T, n, (sv >> 8) + l0);
        else fs_load_be128(T + (sv >> 8) + l0, sh, sl);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t p = r * NT + tid;
        kh[r] = 0; kl[r] = 0;
        if (p < c) {
            if (digits) kh[r] = ss_sym_load(T, n, (vv[r] >> 8) + l0);
            else fs_load_be128(T + (vv[r] >> 8) + l0, kh[r], kl[r]);
        }
    }
    // 256 pivots (bins of ~c / 513: the windows count inside runs directly, quadratic in their length): four waves sort 64
    // sampled keys each, every pivot then finds its place among the other three lists
    if (tid < SS_NPIV0) {
        if (digits) { sh = ss_sym_key(sh); sl = 0; } else sl &= ~0xFFFFull;
        wave_sort_u128(sh, sl, lane);
        s_pl[wv * 64 + lane] = make_ulonglong2(sh, sl);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (digits) kh[r] = ss_sym_key(kh[r]);
        else kl[r] &= ~0xFFFFull;
    }
    __syncthreads();
    SS_CLK(1);
    if (tid < SS_NPIV0) {
        const uint32_t w = tid >> 6;
        const ulonglong2 kv = s_pl[w * 64 + lane];
        uint32_t rank = lane;
#pragma unroll
        for (uint32_t ow = 0; ow < SS_NPL; ow++) {
            if (ow == w) continue;
            uint32_t lo = 0, hi = 64;                          // elements of list ow that come before kv (ties: the lower list first)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const ulonglong2 x = s_pl[ow * 64 + mid];
                if (ss_less128(x, kv.x, kv.y) || (x.x == kv.x && x.y == kv.y && ow < w)) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        s_piv0[rank] = kv;
    }
    __syncthreads();
    SS_CLK(2);                                                 // pivots
    // first pivot >= key: eight branch-free steps over the first 255 pivots (the searches of a thread's eight members side by
    // side: their LDS reads overlap), then a look at the pivot found -- or at the 256th
    uint32_t br[ITEMS];                                        // bin : 10 | arrival rank in the bin : 12
    {
        uint32_t at[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) at[r] = 0;
#pragma unroll
        for (int r0 = 0; r0 < ITEMS; r0 += GLC_SSC_SIDE) {
#pragma unroll
            for (uint32_t step = SS_NPIV0 / 2; step >= 1; step >>= 1) {
#pragma unroll
                for (int r = r0; r < r0 + GLC_SSC_SIDE; r++)
                    if (ss_less128(s_piv0[at[r] + step - 1], kh[r], kl[r])) at[r] += step;
            }
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t p = r * NT + tid;
            br[r] = 0;
            if (p < c) {
                const ulonglong2 x = s_piv0[at[r]];            // (at <= 255)
                const bool past = at[r] == SS_NPIV0 - 1 && ss_less128(x, kh[r], kl[r]);
                const uint32_t bn = past ? 2 * SS_NPIV0 : 2 * at[r] + ((x.x == kh[r] && x.y == kl[r]) ? 1u : 0u);
                br[r] = bn | (atomicAdd(&s_cnt[bn], 1u) << 10);
            }
        }
    }
    __syncthreads();
    SS_CLK(3);                                                 // binned (the pivots are dead: s_out takes the bucket)
    if (wv == 0) {
        constexpr int PER = (2 * SS_NPIV0 + 2 + 63) / 64;      // 514 starts + the end
        uint32_t cc[PER], tot = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) { const uint32_t i = PER * lane + k; cc[k] = i < 2 * SS_NPIV0 + 1 ? s_cnt[i] : 0u; tot += cc[k]; }
        uint32_t run = wave_incl_add(tot) - tot;
#pragma unroll
        for (int k = 0; k < PER; k++) { const uint32_t i = PER * lane + k; if (i <= 2 * SS_NPIV0 + 1) s_cnt[i] = run; run += cc[k]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t p = r * NT + tid;
        if (p < c) {
            const uint32_t bn = br[r] & 0x3FFu, gs = s_cnt[bn], ge = s_cnt[bn + 1];
            s_out[gs + (br[r] >> 10)] = (uint64_t)vv[r] | ((uint64_t)ss_run(gs, ge, (bn & 1) ? deeper : 0u) << 32);   // a pivot's bin: all keys equal
        }
    }
    // bins still longer than a window (a key shared by hundreds of suffixes, or an unlucky gap between pivots) go on a
    // list: k_ss_long cuts them again, a workgroup per bin (when that was the tail of this kernel, one or two waves worked
    // and the other fourteen sat on 73 KB of LDS: 40 % of a text bucket's time here, 60 % of a log bucket's).  Two size
    // classes, counted in the halves of one 64-bit counter: up to SSL_SMALL members, and more.
    for (uint32_t i = tid; i < 2 * SS_NPIV0 + 1; i += NT) {
        const uint32_t gs = s_cnt[i], ge = s_cnt[i + 1];
        if (ge - gs > SS_LONG) s_bound[atomicAdd(&s_nlong, 1u)] = gs | (ge << 16);
    }
    __syncthreads();
    SS_CLK(4);                                                 // scanned, scattered, long bins listed
    const uint32_t nlong = s_nlong;
    if (nlong && tid < 64) {                                   // (nlong <= SSL_PER_BUCKET < 64)
        uint32_t mine = 0, big = 0;
        if (tid < nlong) { mine = s_bound[tid]; big = (mine >> 16) - (mine & 0xFFFFu) > SSL_SMALL ? 1u : 0u; }
        const unsigned long long bigs = __ballot(tid < nlong && big), smalls = __ballot(tid < nlong && !big);
        if (tid == 0) s_at = atomicAdd(long_count, (unsigned long long)__popcll(smalls) | ((unsigned long long)__popcll(bigs) << 32));
        __builtin_amdgcn_wave_barrier();
        if (tid < nlong) {
            const unsigned long long below = (1ull << tid) - 1ull;
            const size_t at = big ? long_cap + (size_t)(s_at >> 32) + __popcll(bigs & below)
                                  : (size_t)(uint32_t)s_at + __popcll(smalls & below);
            long_list[at] = make_uint2(b | (bk << 20), mine);
        }
    }
    SS_CLK(5);
    for (uint32_t p = tid; p < c; p += NT) K[p] = s_out[p];    // back to the slot in run order
    SS_CLK(7);
    SS_CLK_END(0);
}

// the long bins k_ss_cut listed (runs of more than a window: a key that hundreds of suffixes of a bucket share), each cut
// by ONE workgroup with 64 pivots of its own members until no run in it is longer than a window.  The bin lives in LDS
// meanwhile (16 bytes per member), so there are two instances: bins of up to SSL_SMALL members, a wave each (nine per
// CU), and the others, four waves each (two workgroups per CU).  Workgroups take the list's entries i, i + G, ... -- no
// tickets.  A cut: the members' next 7 bytes (8 loads in flight per lane); all equal -> the run is one round deeper and
// nothing moves; else 64 of the keys, sorted by a wave, are the pivots the members search (in LDS, four searches per
// lane at a time) for their bin -- "between two pivots": a run at the same depth, "equal to a pivot": one round deeper.
template <uint32_t CAP, uint32_t NT, bool BIG>
__global__ __launch_bounds__(NT) void k_ss_long(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                uint64_t *__restrict__ keys, size_t kstride, uint32_t *__restrict__ flag,
                                                const uint32_t *__restrict__ l0_in, const uint2 *__restrict__ long_list,
                                                size_t long_cap, const unsigned long long *__restrict__ long_count, bool tol)
{
    __shared__ uint64_t s_kl[CAP];
    __shared__ uint32_t s_vl[CAP], s_segl[CAP];
    __shared__ uint32_t cnt[SS_NBIN + 3];
    __shared__ uint64_t s_piv[SS_NPIV + 1];
    __shared__ uint32_t s_any[2];                              // [0]: a member's 8 bytes reach the end of the text; [1]: a key differs from the first
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    auto sync = [] { if (NT == 64) __builtin_amdgcn_wave_barrier(); else __syncthreads(); };
    const unsigned long long both = *long_count;
    const uint32_t count = BIG ? (uint32_t)(both >> 32) : (uint32_t)both;
    const uint2 *LST = long_list + (BIG ? long_cap : 0);
    for (uint32_t i = tid; i < SS_NBIN + 3; i += NT) cnt[i] = 0;
    if (tid < 2) s_any[tid] = 0;
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const uint2 ent = LST[e];
        const uint32_t b = ent.x & 0xFFFFFu, bk = ent.x >> 20, A = ent.y & 0xFFFFu, B = ent.y >> 16;
        // ONE read of the flag per entry, by one thread: other workgroups of this launch raise it (deep runs), and waves that
        // read it for themselves could disagree -- one leaving for the next entry while the others wait at a barrier
        uint32_t fl;
        if (NT == 64) fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)flag[b]);
        else {
            sync();
            if (tid == 0) s_flag = flag[b];
            sync();
            fl = s_flag;
        }
        if (fl || B - A > CAP) continue;                       // (given up on already; the second cannot happen)
        const uint8_t *T = text + (size_t)b * stride;
        uint64_t *K = keys + (size_t)b * kstride + (size_t)bk * FS_CAP;
        const uint32_t l0 = l0_in[(size_t)b * FS_MAXNB + bk];
        // positions are the bucket's; the LDS arrays hold [A, B)
        uint64_t *s_k = s_kl - A;
        uint32_t *s_v = s_vl - A, *s_seg = s_segl - A;
        sync();
        for (uint32_t p = A + tid; p < B; p += NT) { const uint64_t x = K[p]; s_v[p] = (uint32_t)x; s_seg[p] = (uint32_t)(x >> 32); }
        sync();
        bool deep = false;
        uint32_t pos = A;
        while (pos < B) {
            // window [pos, W): the runs that start in it and end within SS_WIN positions (every wave works it out for itself)
            const uint32_t lim = min(B, pos + SS_LONG);
            uint32_t W = lim, g0 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + lane + 64 * j;
                if (p < lim) {
                    const uint32_t g = s_seg[p], ss = g & 0xFFFu, se = (g >> 12) & 0xFFFu;
                    if (j == 0) g0 = g;
                    if (se > lim) W = min(W, ss);              // a run that runs out of the window: the window ends before it
                }
            }
            W = (uint32_t)wave_min_u64((uint64_t)W);
            if (W != pos) { pos = W; continue; }               // (windows are finished by k_ss_windows)
            // ---- the run at pos is longer than a window: cut it ----
            const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g0);
            const uint32_t ss = g & 0xFFFu, se = (g >> 12) & 0xFFFu, st = g >> 24, gsz = se - ss;
            if (st == SS_CAPPED) { pos = se; continue; }         // (tolerant form: left as it is)
            if (st > (tol ? SS_TOL_MAXSTEP : SS_MAXSTEP)) {
                if (!tol) { deep = true; break; }
                const uint32_t rc = ss_run(ss, se, SS_CAPPED);
                for (uint32_t p = ss + tid; p < se; p += NT) s_seg[p] = rc;
                sync();
                pos = se;
                continue;
            }
            const uint32_t off = l0 + SS_STEP * st;
            {   // does a member's load reach the end of the text?  Then the whole cut uses the 9-bit digits (ss_sym_key)
                bool t = false;
                for (uint32_t p = ss + tid; p < se; p += NT) t |= (s_v[p] >> 8) + off + 12 > n;
                if (__ballot(t) != 0 && lane == 0) s_any[0] = 1;
            }
            sync();
            const bool digits = s_any[0] != 0;
            const uint64_t k0 = 0;
            for (uint32_t p0 = ss; p0 < se; p0 += NT * 8) {    // keys of the members, 8 loads in flight per lane
                uint64_t raw[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t p = p0 + tid + NT * j;
                    raw[j] = p < se ? (digits ? ss_sym_load(T, n, (s_v[p] >> 8) + off) : fs_load_be64(T + (s_v[p] >> 8) + off) >> 8) : 0ull;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) { const uint32_t p = p0 + tid + NT * j; if (p < se) s_k[p] = digits ? ss_sym_key(raw[j]) : raw[j]; }
            }
            (void)k0;
            sync();
            {   // all keys equal: the run is one round deeper, nothing moves
                const uint64_t first = s_k[ss];
                bool d = false;
                for (uint32_t p = ss + tid; p < se; p += NT) d |= s_k[p] != first;
                if (__ballot(d) != 0 && lane == 0) s_any[1] = 1;
            }
            sync();
            const bool differ = s_any[1] != 0;
            sync();
            if (tid < 2) s_any[tid] = 0;
            if (!differ) {
                const uint32_t r1 = ss_run(ss, se, st + 1);
                for (uint32_t p = ss + tid; p < se; p += NT) s_seg[p] = r1;
                sync();
                continue;
            }
            if (tid < 64) s_piv[lane] = wave_sort_u64(s_k[ss + (lane * gsz) / SS_NPIV], lane);
            sync();
            for (uint32_t p0 = ss; p0 < se; p0 += NT * 4) {    // bin and arrival rank of every member -> s_seg (the run's descriptor is in g)
                uint64_t key[4];
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const uint32_t p = p0 + tid + NT * j; key[j] = p < se ? s_k[p] : 0ull; lo[j] = 0; hi[j] = SS_NPIV; }
#pragma unroll
                for (int it = 0; it < 7; it++) {               // first pivot >= key
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t mid = (lo[j] + hi[j]) >> 1;
                        const uint64_t pm = s_piv[mid < SS_NPIV ? mid : SS_NPIV - 1];
                        if (lo[j] < hi[j]) { if (pm < key[j]) lo[j] = mid + 1; else hi[j] = mid; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t p = p0 + tid + NT * j;
                    if (p < se) {
                        const uint32_t bn = 2 * lo[j] + ((lo[j] < SS_NPIV && s_piv[lo[j]] == key[j]) ? 1u : 0u);
                        s_seg[p] = bn | (atomicAdd(&cnt[bn], 1u) << 8);
                    }
                }
            }
            sync();
            if (tid < 64) {
                uint32_t c3[3], tot = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { const uint32_t i = 3 * lane + k; c3[k] = i < SS_NBIN ? cnt[i] : 0u; tot += c3[k]; }
                uint32_t run = ss + wave_incl_add(tot) - tot;
#pragma unroll
                for (int k = 0; k < 3; k++) { const uint32_t i = 3 * lane + k; if (i <= SS_NBIN) cnt[i] = run; run += c3[k]; }
            }
            sync();
            for (uint32_t p = ss + tid; p < se; p += NT) {     // to the bins, through s_k (the keys are used up)
                const uint32_t x = s_seg[p], bn = x & 0xFFu, gs = cnt[bn], ge = cnt[bn + 1];
                s_k[gs + (x >> 8)] = (uint64_t)s_v[p] | ((uint64_t)ss_run(gs, ge, st + (bn & 1)) << 32);
            }
            sync();
            for (uint32_t p = ss + tid; p < se; p += NT) { const uint64_t x = s_k[p]; s_v[p] = (uint32_t)x; s_seg[p] = (uint32_t)(x >> 32); }
            for (uint32_t i = tid; i < SS_NBIN + 3; i += NT) cnt[i] = 0;
            sync();                                            // look at pos again: the runs there are shorter or deeper now
        }
        sync();
        if (deep) { if (tid == 0) atomicOr(&flag[b], 2u); continue; }
        for (uint32_t p = A + tid; p < B; p += NT) K[p] = (uint64_t)s_v[p] | ((uint64_t)s_seg[p] << 32);
    }
}

#ifndef GLC_SSW_PER_BUCKET
#define GLC_SSW_PER_BUCKET 3
#endif
constexpr int SSW_PER_BUCKET = GLC_SSW_PER_BUCKET;             // one-wave workgroups per bucket; wave w takes shares w, w + SSW_PER_BUCKET, ...

// a run descriptor whose positions still have to be ordered: more than one member, and not left as it is (SS_CAPPED)
template <bool TOL>
__device__ __forceinline__ bool ss_undecided_t(uint32_t g) { return ((g >> 12) & 0xFFFu) - (g & 0xFFFu) > 1 && (!TOL || (g >> 24) != SS_CAPPED); }

// A round here takes 14 text bytes (two steps of the run descriptors' unit) from ONE 16-byte gather per member: the kernel is
// bound by the number of scattered accesses a CU takes (~6 cycles per lane access out of L2: 12 rounds x 62 members per wave
// of the 7-byte form account for two thirds of its time), and a gather of 16 bytes costs what one of 8 does.  The key is
// [bytes 0 .. 7 | bytes 8 .. 13, 0, window slot]: unique, so a member's new place is ONE count of smaller keys (a 128-bit
// comparison per key read, where the 7-byte form made two 64-bit ones), and the runs of the next round are found AFTER the
// count: every place learns which slot's member comes to it (a byte per place), a place whose member's 14 bytes differ from
// its left neighbour's starts a run, and four ballots of those flags give every place its run's first and last place -- no
// counters, no scan.  Keys and words stay at their slots in LDS for the round (5.4 KB per wave); the window's words and run
// descriptors live in registers, four places per lane.  A round with a member whose 16 bytes reach the end of the text takes
// ONE step with the 9-bit digits that tell "ended" from a zero byte (63 bits + the slot).
#ifndef GLC_SSW_WAVES
#define GLC_SSW_WAVES 7
#endif
// SHARES one-wave workgroups per bucket, a share each (batches: 3 -- more shares are more, smaller windows at the shares' ends; a call
// of a few blocks: 12 -- a lone block's 1536 waves left most of the chip idle behind chains of three windows each)
template <bool TOL, int SHARES = (int)SS_SHARES>
__global__ __launch_bounds__(64, GLC_SSW_WAVES) void k_ss_windows(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                      const uint64_t *__restrict__ keys, size_t kstride,
                                                      const uint32_t *__restrict__ fill, const uint32_t *__restrict__ fbase,
                                                      uint32_t *__restrict__ flag, const uint32_t *__restrict__ list,
                                                      const uint32_t *__restrict__ l0_in, uint8_t *__restrict__ bwt_out,
                                                      size_t bwt_stride, int *__restrict__ d_index,
                                                      uint32_t *__restrict__ sa_out, size_t sa_stride)
{
    constexpr bool tol = TOL;
    auto ss_undecided = [](uint32_t g) { return ss_undecided_t<TOL>(g); };
    __shared__ ulonglong2 s_kw[SS_WIN];                        // keys of the window's members {hi, lo}, at their slots
    __shared__ uint32_t s_vw[SS_WIN];                          // ... and their words (index << 8 | BWT byte)
    __shared__ uint8_t s_inv[SS_WIN];                          // the slot whose member comes to a place
    __shared__ uint32_t s_bound[SHARES + 1];
    uint32_t gx, gy;
    xcd_order(gx, gy);
    constexpr uint32_t PER = SHARES == (int)SS_SHARES ? (uint32_t)SSW_PER_BUCKET : (uint32_t)SHARES;   // (the small-call form: a wave per share)
    const uint32_t b = list[gy], bk = gx / PER, w0 = gx % PER;
    const uint32_t lane = threadIdx.x;
    const uint8_t *T = text + (size_t)b * stride;
    const uint32_t c = fill[(size_t)b * FS_MAXNB + bk];
    const uint32_t R0 = fbase[(size_t)b * FS_MAXNB + bk];
    const uint64_t *K = keys + (size_t)b * kstride + (size_t)bk * FS_CAP;
    bool deep = flag[b] != 0;                                  // (set by earlier kernels only, or by other waves: then it does not matter what this one does)
    if (deep || c == 0 || c > FS_FILLMAX) return;
    const uint32_t l0 = l0_in[(size_t)b * FS_MAXNB + bk];
    SS_CLK_BEGIN();
    // shares [A, B) of the positions; a share ends where a run ends
    for (uint32_t t = lane; t <= (uint32_t)SHARES; t += 64) {
        uint32_t A = (uint32_t)(((uint64_t)c * t) / (uint32_t)SHARES);
        if (A > 0 && A < c) { const uint32_t g = (uint32_t)(K[A] >> 32); if ((g & 0xFFFu) < A) A = (g >> 12) & 0xFFFu; }
        s_bound[t] = A;
    }
    __builtin_amdgcn_wave_barrier();
    SS_CLK(0);                                                 // prologue + share bounds
    uint8_t *O = bwt_out ? bwt_out + (size_t)b * bwt_stride + R0 : nullptr;
    uint32_t *SAo = sa_out ? sa_out + (size_t)b * sa_stride + R0 : nullptr;
    for (uint32_t ch = w0; ch < (uint32_t)SHARES; ch += PER) {
        const uint32_t A = s_bound[ch], B = s_bound[ch + 1];
        uint32_t pos = A;
        while (pos < B) {
            if (deep) break;
            // window [pos, W): the runs that start in it and end within SS_WIN positions
            const uint32_t lim = min(B, pos + SS_WIN);
            uint32_t g4[4], x4[4], W = lim;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + lane + 64 * j;
                const uint64_t x = p < lim ? K[p] : 0ull;
                g4[j] = (uint32_t)(x >> 32); x4[j] = (uint32_t)x & (uint32_t)FS_LOW_MASK;
                if (p < lim) {
                    const uint32_t ss = g4[j] & 0xFFFu, se = (g4[j] >> 12) & 0xFFFu;
                    if (se > lim) W = min(W, ss);              // a run that runs out of the window: the window ends before it
                }
            }
            W = (uint32_t)wave_min_u64((uint64_t)W);
            SS_CLK(1);                                         // window's words loaded
            if (W == pos) {
                // a run longer than a window: k_ss_long leaves none -- except, in the tolerant form, the runs it capped.
                // Their rows are written as they are (any order: the doubling rounds behind this kernel order them).
                const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g4[0]);
                if (!tol || (g >> 24) != SS_CAPPED) { deep = true; break; }
                const uint32_t se = (g >> 12) & 0xFFFu;
                for (uint32_t p = pos + lane; p < se; p += 64) {
                    const uint32_t v = (uint32_t)K[p] & (uint32_t)FS_LOW_MASK, idx = v >> 8;
                    if (O) O[p] = (uint8_t)v;
                    if (SAo) SAo[p] = idx | (p == pos ? SA_CAND : GRP_SAME);   // (see the rows written below)
                    if (idx == 0 && d_index) d_index[b] = (int)(R0 + p);
                }
                pos = se;
                continue;
            }
            bool und = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + lane + 64 * j;
                if (p >= W) g4[j] = ss_run(0, 1, 0);                               // not of this window: a place on its own
                und |= p < W && ss_undecided(g4[j]);
            }
            // ---- window [pos, W): rounds in registers until every position is decided ----
            uint32_t rounds_here = 0;
            const uint64_t le = (2ull << lane) - 1ull;         // lanes 0 .. lane
            while (__ballot(und) != 0) {
                uint32_t at[4];
                bool dp = false, tail = false, mv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t p = pos + lane + 64 * j;
                    at[j] = 0; mv[j] = false;
                    if (p < W) {
                        const uint32_t ss = g4[j] & 0xFFFu, se = (g4[j] >> 12) & 0xFFFu, st = g4[j] >> 24;
                        if (ss_undecided(g4[j]) && st > (tol ? SS_TOL_MAXSTEP : SS_MAXSTEP)) {
                            if (tol) g4[j] = ss_run(ss, se, SS_CAPPED);          // left as it is (all its members do this)
                            else dp = true;
                        }
                        if (ss_undecided(g4[j])) {
                            mv[j] = true;
                            at[j] = (x4[j] >> 8) + l0 + SS_STEP * st;
                            tail |= at[j] + 16 > n;                // the 16-byte load reaches the end of the text
                        }
                    }
                }
                if (__ballot(dp) != 0) { deep = true; break; }
                // (a window this deep: another wave may have given the block up meanwhile -- a run 75 steps deep costs
                //  ~0.4 ms and a block with a duplicated region has thousands of them; without this look every wave went
                //  through its own before the kernel ended, 6 ms per 64 such blocks)
                if (!tol && (++rounds_here & 7u) == 0 && __hip_atomic_load(&flag[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                const bool digits = __ballot(tail) != 0;       // (wave-uniform) one step of 9-bit digits instead of two of bytes
                const uint32_t step = digits ? 1u : 2u;
                SS_CLK(2);                                     // round set up
#ifdef GLC_SS_CLOCKS
                {
                    uint32_t tr = 0, und_n = 0;
                    for (int j = 0; j < 4; j++) {
                        const uint32_t L = mv[j] ? ((g4[j] >> 12) & 0xFFFu) - (g4[j] & 0xFFFu) : 0u;
                        tr += wave_max(L > 1 ? L : 0u);
                        und_n += (uint32_t)__popcll(__ballot(L > 1));
                    }
                    {   // members of this round by the size of their run; rounds with few undecided members / small runs only
                        uint32_t c2 = 0, c4 = 0, c16 = 0, c64 = 0, cbig = 0, maxL = 0;
                        for (int j = 0; j < 4; j++) {
                            const uint32_t L = mv[j] ? ((g4[j] >> 12) & 0xFFFu) - (g4[j] & 0xFFFu) : 0u;
                            c2 += (uint32_t)__popcll(__ballot(L == 2)); c4 += (uint32_t)__popcll(__ballot(L == 3 || L == 4));
                            c16 += (uint32_t)__popcll(__ballot(L > 4 && L <= 16)); c64 += (uint32_t)__popcll(__ballot(L > 16 && L <= 64));
                            cbig += (uint32_t)__popcll(__ballot(L > 64)); maxL = max(maxL, wave_max(L));
                        }
                        if (lane == 0) {
                            unsigned long long *G = g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u];
                            atomicAdd(&G[9], (unsigned long long)c2); atomicAdd(&G[10], (unsigned long long)c4); atomicAdd(&G[11], (unsigned long long)c16);
                            atomicAdd(&G[12], (unsigned long long)c64); atomicAdd(&G[13], (unsigned long long)cbig);
                            if (und_n <= 8) atomicAdd(&G[14], 1ull);
                            if (maxL <= 4) atomicAdd(&G[15], 1ull);
                            if (maxL <= 2) atomicAdd(&G[29], 1ull);
                            if (maxL <= 16) atomicAdd(&G[30], 1ull);
                        }
                    }
                    if (lane == 0) { atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][25], 1ull); atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][26], (unsigned long long)tr);
                                     atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][27], (unsigned long long)und_n); atomicAdd(&g_ss_clk[(blockIdx.x * 7u + blockIdx.y) & 255u][28], (unsigned long long)(W - pos)); }
                }
#endif
                // keys and words of the members, at their slots
                {
                    uint64_t kh[4], kl[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        kh[j] = 0; kl[j] = 0;
                        if (mv[j]) {
                            if (!digits) fs_load_be128(T + at[j], kh[j], kl[j]);
                            else kh[j] = ss_sym_key(ss_sym_load(T, n, at[j]));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t sl = lane + 64 * j;
                        if (mv[j]) {
                            s_kw[sl] = make_ulonglong2(kh[j], digits ? (uint64_t)sl : ((kl[j] & ~0xFFFFull) | sl));
                            s_vw[sl] = x4[j];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                SS_CLK(3);                                     // text gathered
                if (tol) {
                    // the inside of a deep repeat: every member's key equals its run's first -- places, words and runs stay as they
                    // are, only the labels advance (the count below would find exactly that, quadratically, round after round up to
                    // the cap: a 2000-byte phrase 64 times in a block is 2000 runs of 64 that stay whole for ten rounds).  Taken when
                    // no run of the window changes; telling the runs apart (a byte per run in LDS, the count skipped member by member)
                    // measured 1.69 against 1.71 ms per 64 partly deep blocks, not worth its two extra barriers per round.
                    bool differs = false;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (mv[j]) {
                            const ulonglong2 k = s_kw[lane + 64 * j], k0 = s_kw[(g4[j] & 0xFFFu) - pos];
                            differs |= (k.x != k0.x) | (((k.y ^ k0.y) >> 16) != 0);
                        }
                    if (__ballot(differs) == 0) {
                        und = false;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (mv[j]) {
                                g4[j] = ss_run(g4[j] & 0xFFFu, (g4[j] >> 12) & 0xFFFu, (g4[j] >> 24) + step);
                                und |= ss_undecided(g4[j]);
                            }
                        __builtin_amdgcn_wave_barrier();
                        continue;
                    }
                }
                // a member's new place = the smaller keys of its run; the place learns which slot comes to it
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (mv[j]) {
                        const uint32_t sl = lane + 64 * j, ss = g4[j] & 0xFFFu, se = (g4[j] >> 12) & 0xFFFu;
                        const ulonglong2 k = s_kw[sl];
                        uint32_t below = 0;
                        // (three 64-bit compares + two scalar mask operations per key; as the borrow of a 128-bit subtraction -- four
                        //  32-bit steps and the add, no scalar work -- the batch of 256 text blocks ran 14.3 against 13.5 ms)
#pragma unroll 4
                        for (uint32_t q = ss; q < se; q++) {
                            const ulonglong2 kq = s_kw[q - pos];
                            below += ((kq.x < k.x) | ((kq.x == k.x) & (kq.y < k.y))) ? 1u : 0u;
                        }
                        s_inv[ss + below - pos] = (uint8_t)sl;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                SS_CLK(4);                                     // counted
                // a place starts a run if it is the old run's first or its 14 bytes (its digits) differ from its left neighbour's
                uint64_t hb[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t sl = lane + 64 * j;
                    bool head = true;
                    if (mv[j]) {
                        const uint32_t i1 = s_inv[sl];
                        x4[j] = s_vw[i1];
                        if (pos + sl != (g4[j] & 0xFFFu)) {
                            const ulonglong2 k1 = s_kw[i1], k0 = s_kw[s_inv[sl - 1]];
                            head = (k1.x != k0.x) | (((k1.y ^ k0.y) >> 8) != 0);
                        }
                    }
                    hb[j] = __ballot(head);
                }
                // run of a place: from the last head at or below it to the next head above it (uniform where a 64-lane group has none)
                uint32_t prevh[4], nexth[4];
                prevh[0] = 0;
#pragma unroll
                for (int j = 1; j < 4; j++) prevh[j] = hb[j - 1] ? 64u * (j - 1) + 63u - (uint32_t)__builtin_clzll(hb[j - 1]) : prevh[j - 1];
                nexth[3] = SS_WIN;
#pragma unroll
                for (int j = 2; j >= 0; j--) nexth[j] = hb[j + 1] ? 64u * (j + 1) + (uint32_t)__builtin_ctzll(hb[j + 1]) : nexth[j + 1];
                und = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (mv[j]) {
                        const uint64_t lo_m = hb[j] & le, hi_m = hb[j] & ~le;
                        const uint32_t rs = lo_m ? 64u * j + 63u - (uint32_t)__builtin_clzll(lo_m) : prevh[j];
                        const uint32_t re = hi_m ? 64u * j + (uint32_t)__builtin_ctzll(hi_m) : nexth[j];
                        g4[j] = ss_run(pos + rs, pos + re, (g4[j] >> 24) + step);
                        und |= ss_undecided(g4[j]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                SS_CLK(5);                                     // moved, runs found
            }
            if (deep) break;
            // rows R0 + pos .. R0 + W
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t p = pos + lane + 64 * j;
                if (p < W) {
                    const uint32_t v = x4[j], idx = v >> 8;
                    if (O) O[p] = (uint8_t)v;
                    // (tolerant form: the rows k_grp_flags has to look at -- a bucket's first row, whose suffix may tie with the last
                    //  of the bucket before, and every row that may still have been in a run at depth SS_TOL_CAP: members of a run left
                    //  as it is, but also suffixes told apart beyond the cap -- they may tie with a SPLITTER, and those ties were placed
                    //  by position, on either side of it.  A place decided under the label `st` parted from its neighbours before depth
                    //  l0 + 7 (st + 1): a round here advances the label with the decision, but a member alone in a bin BETWEEN two pivots
                    //  of k_ss_cut / k_ss_long keeps the run's label and differs somewhere in the NEXT seven symbols -- the hunt that
                    //  found it: two suffixes 131 and 132 symbols from the end of a block, cut apart at label 18 in some runs.)
                    //  A member of a run left as it is agrees with the member before it in SS_TOL_MAXSTEP x 7 >= SS_TOL_CAP symbols: it
                    //  continues that row's group for sure (GRP_SAME), and k_grp_flags need not read 2 x 128 bytes of text to find that
                    //  out -- those full-length comparisons were most of its 0.72 ms per 64 partly deep blocks.
                    uint32_t mark = 0;
                    if (tol) {
                        if ((g4[j] >> 24) == SS_CAPPED && p > (g4[j] & 0xFFFu)) mark = GRP_SAME;
                        else if (p == 0 || l0 + SS_STEP * ((g4[j] >> 24) + 1) >= SS_TOL_CAP) mark = SA_CAND;
                    }
                    if (SAo) SAo[p] = idx | mark;
                    if (idx == 0 && d_index) d_index[b] = (int)(R0 + p);
                }
            }
            SS_CLK(6);                                         // rows written
            pos = W;
        }
    }
    SS_CLK_END(16);
    if (deep && lane == 0) atomicOr(&flag[b], 2u);
}

// second attempt: the blocks of `list` whose ONLY trouble was a bucket past its slot (flag == 1: the samples' luck --
// bucket populations of text-like blocks have a heavier tail than 32 samples per bucket suggest, ~1 % of log blocks end up
// with a bucket of 4033-4200 words) are listed again, their flags and fills cleared, for a pass with other samples
// (want = 2: the blocks whose only trouble was a repeat deeper than the cap -- listed for the tolerant form)
__global__ void k_ss_retry_list(uint32_t *__restrict__ flag, const uint32_t *__restrict__ list, uint32_t nflag,
                                uint32_t *__restrict__ list2, uint32_t *__restrict__ count, uint32_t *__restrict__ fill,
                                uint32_t want, uint32_t mutate)
{
    const uint32_t j = blockIdx.x;
    if (j >= nflag) return;
    const uint32_t b = list[j];
    if (flag[b] != want) return;                               // (uniform per workgroup)
    __shared__ uint32_t s_at;
    if (threadIdx.x == 0) s_at = atomicAdd(count, 1u);
    if (!mutate) return;                                       // (count only: flags and fills stay what the diagnostics report)
    for (uint32_t i = threadIdx.x; i < FS_MAXNB; i += blockDim.x) fill[(size_t)b * FS_MAXNB + i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { list2[s_at] = b; flag[b] = 0; }
}

// what this tier gave up on keeps its live count for the general sorter
__global__ void k_ss_finish(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ list, uint32_t nflag,
                            uint32_t n, uint32_t *__restrict__ lcnt, uint32_t *__restrict__ nleft)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nflag) {
        const uint32_t b = list[j], f = flag[b] ? n : 0u;
        lcnt[b] = f;
        if (f) atomicAdd(nleft, 1u);
    }
}

__global__ void k_ss_split_masks(const uint32_t *__restrict__ redo, const uint32_t *__restrict__ lcnt, uint32_t nblk,
                                 uint32_t *__restrict__ done, uint32_t *__restrict__ open)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) { const uint32_t l = lcnt[b]; done[b] = (redo[b] && !l) ? 1u : 0u; open[b] = l; }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t fs_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk, SaScratch &s,
                    uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *sa_out)
{
    const uint32_t nbl = fs_bucket_log2(n), nb = 1u << nbl;
    {
        const uint32_t words = nblk * (256u + FS_MAXNB + 3u) + 8u, g = (words + 1023) / 1024;
        // (skip_tier1: every block starts flagged -- no attempt, the sample sorter takes them all)
        hipLaunchKernelGGL(k_fs_clear, dim3(g < 2048 ? g : 2048), dim3(256), 0, st, nblk, s.fs_hist, s.fs_fill, s.fs_flag,
                           s.skip_tier1 ? 1u : 0u, s.fs_wlcnt, s.fs_dup, s.fs_nflag);
    }
    uint32_t *h_nflag = s.h_max_cnt + 4;                       // pinned, device-mapped: written by the kernel that finishes the pass
    const double units = (double)n * nblk;
    int pi = s.prof ? s.prof->begin(PROF_FS_HIST, st) : -1;
    // statistics from every 4th 32 KB slice of a block of 512 KiB or more (the pass reads a quarter of the input: 0.27 -> 0.07 ms
    // per GiB; GLC_FSH_STEP=1: all of it)
    static const int step_env = getenv("GLC_FSH_STEP") ? atoi(getenv("GLC_FSH_STEP")) : 0;
    const uint32_t nslices = (n + FSH_SLICE - 1) / FSH_SLICE;
    const uint32_t hstep = step_env > 0 ? (uint32_t)step_env : (nslices >= 16 ? 4u : 1u);
    hipLaunchKernelGGL(k_fs_hist, dim3((nslices + hstep - 1) / hstep, nblk), dim3(256), 0, st, text, text_stride, n,
                       s.fs_hist, s.fs_dup, hstep);
    if (pi >= 0) s.prof->end(pi, units, st);
    hipLaunchKernelGGL(k_fs_tables, dim3(nblk), dim3(256), 0, st, s.fs_hist, n, s.fs_tab, s.fs_dup, s.fs_flag, text, text_stride,
                       bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax, hstep);
    if (s.skip_tier1) {
        // sorter mode 4, or a small call behind a streak of all-text-like calls: no attempt, every block goes to the sample sorter
        hipLaunchKernelGGL(k_fs_finish, dim3((nblk + 255) / 256), dim3(256), 0, st, s.fs_flag, n, nblk, s.fs_lcnt, s.fs_nflag,
                           s.fs_redo[s.parity & 1], s.fs_keep[s.parity & 1], s.ss_list, h_nflag, s.fs_dup, (FS_DUP_FLAG + hstep - 1) / hstep);
        return hipGetLastError();
    }
    // Sub-waves (GLC_FS_SUBWAVE = blocks per sub-wave, 0 = the whole call at once): the 8-byte suffix words of a block make
    // one round trip through memory between k_fs_part and k_fs_sort -- 16 of the encoder's 26 bytes of HBM traffic per input
    // byte when a call's words (8 MiB per block) are far more than the 256 MB Infinity Cache holds.  Bucketing and sorting
    // 16 blocks at a time keeps the words of a sub-wave (128 MiB) inside that cache: the bucket sort reads what the
    // bucketing pass has just written.
    static const uint32_t subwave = getenv("GLC_FS_SUBWAVE") ? (uint32_t)atoi(getenv("GLC_FS_SUBWAVE")) : 0u;
    const uint32_t step = subwave && subwave < nblk ? subwave : nblk;
    for (uint32_t b0 = 0; b0 < nblk; b0 += step) {
        const uint32_t nbk = nblk - b0 < step ? nblk - b0 : step;
        const double u = (double)n * nbk;
        pi = s.prof ? s.prof->begin(PROF_FS_PART, st) : -1;
        static const bool old_part = getenv("GLC_FS_PART_OLD") != nullptr;      // A/B: one tile per workgroup
        if (!old_part) {
            // tiles of 4096 suffixes, per workgroup: 8 for batches, 4 for a few blocks, 1 for one to three (more workgroups for a block on its own).  Round 5, bench.py
            // `value` on one box, 1024-block batches, stage overlap on: 4 / 8 / 12 / 16 / 24 / 32 / 64 tiles -> 93.1 / 95.1 / 94.7 /
            // 95.1 / 95.6 / 93.8 / 87-94 GB/s (16 and 24 fall into two modes from run to run: 94.3-96.5); the kernel itself 3.0 ->
            // 2.9 ms per GiB.  Tiles of 8192 suffixes (1024 threads x 8, 64 KB of LDS, one workgroup per CU: runs of ~128 bytes, half
            // the global atomics) run 2.7 ms -- and give the same `value` as 4 tiles of 4096: a 1024-thread workgroup needs half a CU
            // free AT ONCE, which the MTF and Huffman kernels of the batch before, sharing the chip under stage overlap, rarely leave.
            static const int per_env = getenv("GLC_FSP2_PER") ? atoi(getenv("GLC_FSP2_PER")) : 0;    // A/B: tiles per workgroup
            static const bool small_tiles = getenv("GLC_FSP2_SMALL") != nullptr;                     // A/B: 4096-suffix tiles for batches too
            // Batches of 16 blocks or more: tiles of 8192 suffixes, 1024 threads (runs of ~128 bytes leave for a bucket's slot, half the
            // global atomics; 64 KB of LDS, two workgroups per CU), 16 tiles per workgroup.  Until the MTF kernel was re-based this
            // shape ran 2.7 ms and gave the same `value` (the stages of the batch before rarely left half a CU free at once); since:
            // alternating on one box, 6 steps: 100.2 / 101.7 / 102.0 GB/s as it was, 102.3 / 102.9 / 103.0 so (8 tiles per workgroup:
            // 102.3 / 102.5 / 102.7); stages back to back 97.8-99.8 -> 101.2-101.7; the kernel 2.91-3.10 -> 2.67-2.73 ms per GiB.
            // (512 threads x 16 suffixes for the same tile: 3.3 ms.)
            if (nbk >= 16 && !small_tiles) {
#ifndef GLC_FSP2_BT
#define GLC_FSP2_BT 8192
#endif
                constexpr int BT = GLC_FSP2_BT, BN = 1024;
                const uint32_t tiles = (n + BT - 1) / BT;
                const uint32_t per = per_env > 0 ? (uint32_t)per_env : 16u;
                hipLaunchKernelGGL((k_fs_part2<BN, BT / BN, 4>), dim3((tiles + per - 1) / per, nbk), dim3(BN), 0, st,
                                   text + (size_t)b0 * text_stride, text_stride, n, nbl, s.fs_tab + (size_t)b0 * 256,
                                   s.keyA + (size_t)b0 * s.fs_kstride, s.fs_kstride, s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_flag + b0,
                                   s.fs_zero + b0, per);
            } else {
                const uint32_t tiles = (n + FSP2_TILE - 1) / FSP2_TILE;
                // (a call of one to three blocks: ONE tile per workgroup -- 256 workgroups for a block on its own: 17.3 -> 12.0 us of
                //  a single cudppCompress call's chain, 0.190 -> 0.182 ms per call)
                const uint32_t per = per_env > 0 ? (uint32_t)per_env : (nbk >= 16 ? 2 * FSP2_T : (nbk >= 4 ? FSP2_T : 1u));
                hipLaunchKernelGGL((k_fs_part2<FSP2_NT, FSP2_TILE / FSP2_NT, GLC_FSP2_WAVES>), dim3((tiles + per - 1) / per, nbk), dim3(FSP2_NT), 0, st,
                                   text + (size_t)b0 * text_stride, text_stride, n, nbl, s.fs_tab + (size_t)b0 * 256,
                                   s.keyA + (size_t)b0 * s.fs_kstride, s.fs_kstride, s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_flag + b0,
                                   s.fs_zero + b0, per);
            }
        }
        else
        hipLaunchKernelGGL(k_fs_part<false>, dim3((n + FSP_TILE - 1) / FSP_TILE, nbk), dim3(FSP_NT), 0, st, text + (size_t)b0 * text_stride,
                           text_stride, n, nbl, s.fs_tab + (size_t)b0 * 256, s.keyA + (size_t)b0 * s.fs_kstride, s.fs_kstride,
                           s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_flag + b0, (const uint32_t *)nullptr,
                           (const uint64_t *)nullptr, (const uint16_t *)nullptr, (const uint64_t *)nullptr, s.fs_zero + b0, false);
        if (pi >= 0) s.prof->end(pi, u, st);
        hipLaunchKernelGGL(k_fs_scan, dim3(nbk), dim3(FS_MAXNB), 0, st, s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_base + (size_t)b0 * FS_MAXNB,
                           s.fs_flag + b0, (const uint32_t *)nullptr);
        pi = s.prof ? s.prof->begin(PROF_FS_SORT, st) : -1;
        static const bool old_sort = getenv("GLC_FS_SORT_OLD") != nullptr;      // A/B: round 3's kernel for the BWT path too
        if (sa_out || !bwt_out || !d_index || old_sort)                          // the suffix array itself is asked for: the kernel that writes it
            hipLaunchKernelGGL(k_fs_sort, dim3(nb, nbk), dim3(FSS_NT), 0, st, n, nbl, s.keyA + (size_t)b0 * s.fs_kstride, s.fs_kstride,
                               s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_base + (size_t)b0 * FS_MAXNB, s.fs_flag + b0,
                               bwt_out ? bwt_out + (size_t)b0 * bwt_stride : nullptr, bwt_stride, d_index ? d_index + b0 : nullptr,
                               sa_out ? sa_out + (size_t)b0 * s.nmax : nullptr, (size_t)s.nmax, s.fs_wl + (size_t)b0 * s.fs_wl_cap, s.fs_wl_cap,
                               s.fs_wlcnt + b0);
        else
            hipLaunchKernelGGL(k_fs_sort_bwt, dim3(nb, nbk), dim3(FSS_NT), 0, st, nbl, s.keyA + (size_t)b0 * s.fs_kstride, s.fs_kstride,
                               s.fs_fill + (size_t)b0 * FS_MAXNB, s.fs_base + (size_t)b0 * FS_MAXNB, s.fs_flag + b0,
                               bwt_out + (size_t)b0 * bwt_stride, bwt_stride, d_index + b0, s.fs_wl + (size_t)b0 * s.fs_wl_cap, s.fs_wl_cap,
                               s.fs_wlcnt + b0, s.fs_zero + b0);
        if (pi >= 0) s.prof->end(pi, u, st);
    }
    static const int tg_env = getenv("GLC_FST_GRID") ? atoi(getenv("GLC_FST_GRID")) : 0;     // A/B: workgroups per block of k_fs_ties
    hipLaunchKernelGGL(k_fs_ties, dim3(tg_env > 0 ? tg_env : 24, nblk), dim3(256), 0, st, text, text_stride, n, s.fs_wl, s.fs_wl_cap, s.fs_wlcnt,
                       s.fs_flag, bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax);
    hipLaunchKernelGGL(k_fs_finish, dim3((nblk + 255) / 256), dim3(256), 0, st, s.fs_flag, n, nblk, s.fs_lcnt, s.fs_nflag,
                       s.fs_redo[s.parity & 1], s.fs_keep[s.parity & 1], s.ss_list, h_nflag, s.fs_dup, (FS_DUP_FLAG + hstep - 1) / hstep);
    return hipGetLastError();
}

hipError_t ss_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nflag, SaScratch &s,
                    uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *sa_out, uint32_t attempt)
{
    const uint32_t nbl = fs_bucket_log2(n), nb = 1u << nbl;
    // attempt 0: the blocks k_fs_finish listed in ss_list; attempt 1: the ones k_ss_retry_list listed behind them (a bucket
    // past its slot: other samples); attempt 2: the ones listed behind those (a repeat deeper than the cap), in the TOLERANT
    // form -- suffixes that agree in more than SS_TOL_CAP + 8 bytes are left in the order of their positions (comparisons)
    // or as they are (runs), nothing is given up on for depth: the result is the suffixes ordered by their first
    // FS_LCP_CAP symbols, for the prefix-doubling rounds to finish (sa_build_finish)
    const uint32_t *list = s.ss_list + (size_t)attempt * s.rows;
    const bool tol = attempt == 2;
    if (attempt == 0) {
        GLC_TRY(hipMemsetAsync(s.ss_flag, 0, (size_t)s.rows * 4, st));
        GLC_TRY(hipMemsetAsync(s.fs_fill, 0, (size_t)s.rows * FS_MAXNB * 4, st));
        GLC_TRY(per_probe(st, text, text_stride, n, nflag, s));   // (blocks that are mostly one periodic stretch: not this sorter's)
    }
    if (tol)
        hipLaunchKernelGGL(k_ss_sample<true>, dim3(nflag), dim3(SSA_NT), 0, st, text, text_stride, n, nbl, s.fs_tab, list,
                           s.ss_split, s.ss_cell, s.ss_flag, s.ss_l0, s.ss_split + (size_t)s.rows * FS_MAXNB, attempt,
                           s.ss_split + 2 * (size_t)s.rows * FS_MAXNB);
    else
        hipLaunchKernelGGL(k_ss_sample<false>, dim3(nflag), dim3(SSA_NT), 0, st, text, text_stride, n, nbl, s.fs_tab, list,
                           s.ss_split, s.ss_cell, s.ss_flag, s.ss_l0, s.ss_split + (size_t)s.rows * FS_MAXNB, attempt,
                           s.ss_split + 2 * (size_t)s.rows * FS_MAXNB);
    hipLaunchKernelGGL(k_fs_part<true>, dim3((n + FSP_TILE - 1) / FSP_TILE, nflag), dim3(FSP_NT), 0, st, text, text_stride,
                       n, nbl, s.fs_tab, s.keyA, s.fs_kstride, s.fs_fill, s.ss_flag, list, s.ss_split, s.ss_cell,
                       s.ss_split + (size_t)s.rows * FS_MAXNB, (uint32_t *)nullptr, tol, s.ss_split + 2 * (size_t)s.rows * FS_MAXNB);
    hipLaunchKernelGGL(k_fs_scan, dim3(nflag), dim3(FS_MAXNB), 0, st, s.fs_fill, s.fs_base, s.ss_flag, list);
    GLC_TRY(hipMemsetAsync(s.ss_long_count, 0, 8, st));
    const size_t long_cap = (size_t)s.rows * FS_MAXNB * SSL_PER_BUCKET;
    hipLaunchKernelGGL(k_ss_cut<SSS_NT>, dim3(nb, nflag), dim3(SSS_NT), 0, st, text, text_stride, n, nbl, s.keyA, s.fs_kstride,
                       s.fs_fill, s.ss_flag, list, s.ss_l0, s.ss_long, long_cap, s.ss_long_count);
    // the long bins: as many workgroups as fit the GPU (LDS: 17 KB / 65.5 KB each), the list's entries strided over them
    hipLaunchKernelGGL((k_ss_long<SSL_SMALL, 64, false>), dim3(256 * (163840 / (16 * SSL_SMALL + 1200))), dim3(64), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                       s.ss_flag, s.ss_l0, s.ss_long, long_cap, s.ss_long_count, tol);
    hipLaunchKernelGGL((k_ss_long<FS_FILLMAX, 256, true>), dim3(256 * 2), dim3(256), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                       s.ss_flag, s.ss_l0, s.ss_long, long_cap, s.ss_long_count, tol);
    constexpr int FEW = 12;                                    // shares (= waves) per bucket for a call of up to four blocks
    if (tol && nflag <= 4)
        hipLaunchKernelGGL((k_ss_windows<true, FEW>), dim3(nb * FEW, nflag), dim3(64), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                           s.fs_fill, s.fs_base, s.ss_flag, list, s.ss_l0, bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax);
    else if (tol)
        hipLaunchKernelGGL(k_ss_windows<true>, dim3(nb * SSW_PER_BUCKET, nflag), dim3(64), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                           s.fs_fill, s.fs_base, s.ss_flag, list, s.ss_l0, bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax);
    else if (nflag <= 4)
        hipLaunchKernelGGL((k_ss_windows<false, FEW>), dim3(nb * FEW, nflag), dim3(64), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                           s.fs_fill, s.fs_base, s.ss_flag, list, s.ss_l0, bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax);
    else
        hipLaunchKernelGGL(k_ss_windows<false>, dim3(nb * SSW_PER_BUCKET, nflag), dim3(64), 0, st, text, text_stride, n, s.keyA, s.fs_kstride,
                           s.fs_fill, s.fs_base, s.ss_flag, list, s.ss_l0, bwt_out, bwt_stride, d_index, sa_out, (size_t)s.nmax);
    hipLaunchKernelGGL(k_ss_finish, dim3((nflag + 255) / 256), dim3(256), 0, st, s.ss_flag, list, nflag, n, s.fs_lcnt,
                       s.fs_nflag + 1);
    return hipGetLastError();
}

// lists the blocks of the first attempt (`from` = 0) that deserve another one, behind the list of attempt `to` - 1: to = 1,
// flag == 1 (a bucket past its slot); to = 2, flag == 2 (a repeat deeper than the cap).  Their number -> s.fs_nflag[2]
hipError_t ss_retry_prepare(hipStream_t st, uint32_t nflag, SaScratch &s, uint32_t to, bool count_only)
{
    GLC_TRY(hipMemsetAsync(s.fs_nflag + 2, 0, 4, st));
    hipLaunchKernelGGL(k_ss_retry_list, dim3(nflag), dim3(256), 0, st, s.ss_flag, s.ss_list, nflag, s.ss_list + (size_t)to * s.rows,
                       s.fs_nflag + 2, s.fs_fill, to, count_only ? 0u : 1u);
    return hipGetLastError();
}

hipError_t ss_split_masks(hipStream_t st, uint32_t nblk, SaScratch &s)
{
    hipLaunchKernelGGL(k_ss_split_masks, dim3((nblk + 255) / 256), dim3(256), 0, st, s.fs_redo[s.parity & 1], s.fs_lcnt, nblk, s.ss_mask[0], s.ss_mask[1]);
    return hipGetLastError();
}

} // namespace glc

#ifdef GLC_SS_CLOCKS
extern "C" int glcSsClocks(unsigned long long *out48, int reset)
{
    static unsigned long long h[256][48];
    if (out48) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(glc::g_ss_clk), sizeof(glc::g_ss_clk)) != hipSuccess) return 0;
        for (int k = 0; k < 48; k++) { out48[k] = 0; for (int c = 0; c < 256; c++) out48[k] += h[c][k]; }
    }
    if (reset) { for (auto &r : h) for (auto &x : r) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(glc::g_ss_clk), h, sizeof h) != hipSuccess) return 0; }
    return 1;
}
#endif
