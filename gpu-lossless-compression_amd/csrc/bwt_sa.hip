// bwt_sa.hip -- the GENERAL suffix sorter (any data, any LCP depth): suffix array + BWT for blocks of
// <= 2^20 bytes, many blocks per launch (blockIdx.y = block).  gfx950 / wave64.  sa_build() at the end of this
// file runs the bucket sorter of bwt_bucket.hip first and sends only the blocks that one flags (deep common
// prefixes: text, logs, long repeats) through the kernels below.
//
// Replaces, result-for-result, the reference's
//   cudppSuffixArrayDispatch / ComputeSA     (cudpp-inpar/src/cudpp/app/sa_app.cu:125-298,365-391)
//   strConstruct / resultConstruct           (kernel/sa_kernel.cuh:47-82)
//   bwt_compute_final_kernel                 (kernel/compress_kernel.cuh:55-74)
// The reference builds the SA with a recursive skew/DC3 on cub + moderngpu.  The
// SA of (in[i]+1)$ with a unique minimal sentinel is unique, so this file uses a
// different, MI355X-shaped algorithm and still produces identical bytes:
//
//   round 0 : one 64-bit word per suffix = [41-bit mixed-radix code of the first
//             5 symbols (symbol = byte+1, 0 past the end) | 20-bit suffix index],
//             LSD radix sort on the 41 key bits, LDS-staged digit buckets.
//   round r : refinement of the *unresolved* suffixes only, results scattered back to
//             their SA slots, singletons dropped.  First by the next 3 text symbols
//             (word = [group:19 | code:25 | i:20], no rank array: enough for i.i.d.
//             bytes / float data), then, for deep-LCP data, by prefix doubling
//             (Larsson-Sadakane: word = [rank(i):21 | rank(i+h):21 | i:20]).
//   Every sort is Onesweep-style: one histogram read for all digits, then one
//   stable scatter per digit with a decoupled look-back across tiles.
//
// One array of 8-byte words is the only thing the sort moves (key and payload are
// the same word), so a radix pass costs 8 B read (histogram) + 8 B read + 8 B
// written per live suffix.
#include "glc_device.h"
#include <vector>
#include "glc_internal.h"

namespace glc {

constexpr int      SA_THREADS = 256;
constexpr int      SA_ITEMS   = 8;        // 16 measured slower on MI355X (BWT 22.5 vs 17.6 ms / 256 MiB)
constexpr int      SA_TILE    = SA_THREADS * SA_ITEMS;      // suffixes per workgroup
constexpr int      SA_MAXRADIX = 512;
constexpr uint32_t VAL_BITS   = 20;
constexpr uint64_t VAL_MASK   = (1ull << VAL_BITS) - 1;
constexpr uint32_t R1_SHIFT   = 41;                         // rank(i) field, rounds >= 1
constexpr uint32_t R2_SHIFT   = 20;                         // rank(i+h) field

__device__ __forceinline__ uint32_t live_count(const uint32_t *cnt, uint32_t nfixed, uint32_t b)
{
    return cnt ? cnt[b] : nfixed;
}

// ---------------------------------------------------------------------------
// The radix kernels use their own tile: RS_NT threads x 8 words.
// ---------------------------------------------------------------------------
constexpr int RS_NT    = 512;                    // threads per radix workgroup
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE  = RS_NT * RS_ITEMS;       // words per radix tile (longer per-digit write runs than 2048)
constexpr int RS_WAVES = RS_NT / 64;

// ---------------------------------------------------------------------------
// per block, turn [tile][digit] histograms into per-(tile,digit) exclusive prefixes and the
// per-digit base.  One 512-thread workgroup / block.  (Used by the decoder's LF construction;
// the suffix sorter's own passes are the Onesweep kernels below.)
// ---------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(512) void k_rs_scan(uint32_t *__restrict__ tile_hist,
                                                 const uint32_t *__restrict__ cnt, uint32_t nfixed,
                                                 uint32_t *__restrict__ digit_base, uint32_t max_tiles,
                                                 uint32_t tile_elems)
{
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t s_tmp[16];
    const uint32_t b = blockIdx.x, d = threadIdx.x;
    const uint32_t m = live_count(cnt, nfixed, b);
    const uint32_t ntiles = (m + tile_elems - 1) / tile_elems;
    uint32_t run = 0;
    if (d < RADIX) {
        uint32_t *h = tile_hist + (size_t)b * max_tiles * SA_MAXRADIX + d;
        for (uint32_t t0 = 0; t0 < ntiles; t0 += 8) {
            uint32_t x[8];
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = (t0 + j < ntiles) ? h[(size_t)(t0 + j) * SA_MAXRADIX] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (t0 + j < ntiles) { h[(size_t)(t0 + j) * SA_MAXRADIX] = run; run += x[j]; }
        }
    }
    uint32_t ex = block_excl_add<512>(run, s_tmp);
    if (d < RADIX) digit_base[(size_t)b * SA_MAXRADIX + d] = ex;
}

// ---------------------------------------------------------------------------
// Onesweep form of the radix pass: no per-pass histogram/scan kernels.
//   k_rs_prehist   ONE read of the words gives the per-block digit totals of every pass
//                  of the sort (LSD totals do not depend on the order of the words).
//   k_rs_digitbase exclusive scan of those totals.
//   k_rs_onesweep  the stable scatter above, but the tile's per-digit offset comes from a
//                  decoupled look-back over the earlier tiles of the block: thread d owns
//                  digit d and chains 4-byte granules {epoch:8 | flag:2 | count:22}
//                  (agent-scope relaxed atomics; the granule is its own flag).  Tiles take
//                  tickets in arrival order, so a tile only ever waits for tiles that are
//                  already running.  The epoch tag makes stale granules of earlier launches
//                  invisible, so the state array is cleared only once per 255 launches.
// ---------------------------------------------------------------------------
struct PassPlan { uint32_t npass; uint32_t shift[5]; uint32_t bits[5]; };
constexpr int RS_MAXPASS = 5;

// Round 0 never materialises its input: the first radix pass and the histogram kernel read the TEXT
// (1 byte per suffix instead of an 8-byte word written by one kernel and read by two) and build the
// round-0 words [5 symbols base 257 : 41 | index : 20] in registers from a tile staged in LDS.
struct TextSrc { const uint8_t *text; size_t stride; uint32_t n; };

// stage text[base, base + RS_TILE + 4) of the block into s_txt (bytes past n read as 0 and are masked below)
__device__ __forceinline__ void stage_text_tile(const TextSrc ts, uint32_t b, uint32_t base, uint8_t *s_txt)
{
    const uint8_t *T = ts.text + (size_t)b * ts.stride;
    const uint32_t tid = threadIdx.x;
    if ((reinterpret_cast<size_t>(T + base) & 3) == 0 && base + RS_TILE + 4 <= ts.n) {
        const uint32_t *T4 = reinterpret_cast<const uint32_t *>(T + base);
        uint32_t q[RS_TILE / 4 / RS_NT];
#pragma unroll
        for (int r = 0; r < RS_TILE / 4 / RS_NT; r++) q[r] = T4[r * RS_NT + tid];
        const uint32_t halo = (tid == 0) ? T4[RS_TILE / 4] : 0u;
#pragma unroll
        for (int r = 0; r < RS_TILE / 4 / RS_NT; r++) reinterpret_cast<uint32_t *>(s_txt)[r * RS_NT + tid] = q[r];
        if (tid == 0) reinterpret_cast<uint32_t *>(s_txt)[RS_TILE / 4] = halo;
    } else {
        uint8_t q[RS_TILE / RS_NT + 1];
#pragma unroll
        for (int r = 0; r <= RS_TILE / RS_NT; r++) {
            const uint32_t i = r * RS_NT + tid, gi = base + i;
            q[r] = T[gi < ts.n ? gi : 0u];
        }
#pragma unroll
        for (int r = 0; r <= RS_TILE / RS_NT; r++) {
            const uint32_t i = r * RS_NT + tid;
            if (i < RS_TILE + 4) s_txt[i] = q[r];
        }
    }
}

// word of suffix base + li from the staged tile
__device__ __forceinline__ uint64_t text_word(const uint8_t *s_txt, uint32_t li, uint32_t gi, uint32_t n)
{
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) c = c * 257 + ((gi + j < n) ? (uint32_t)s_txt[li + j] + 1u : 0u);
    return (c << VAL_BITS) | gi;
}

template <bool FROM_TEXT>
__global__ __launch_bounds__(RS_NT) void k_rs_prehist(const uint64_t *__restrict__ key,
                                                      const uint32_t *__restrict__ cnt, uint32_t nfixed,
                                                      PassPlan pp, uint32_t *__restrict__ ghist, uint32_t nmax,
                                                      TextSrc ts)
{
    __shared__ uint32_t s_h[RS_MAXPASS][SA_MAXRADIX];
    __shared__ __attribute__((aligned(16))) uint8_t s_txt[FROM_TEXT ? RS_TILE + 8 : 8];
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    const uint32_t m = live_count(cnt, nfixed, b);
    if (blockIdx.x * RS_TILE >= m) return;
    for (uint32_t i = tid; i < RS_MAXPASS * SA_MAXRADIX; i += RS_NT) (&s_h[0][0])[i] = 0;
    __syncthreads();
    const uint64_t *K = key + (size_t)b * nmax;
    for (uint32_t base = blockIdx.x * RS_TILE; base < m; base += gridDim.x * RS_TILE) {
        uint64_t kq[RS_ITEMS];
        if (FROM_TEXT) {
            __syncthreads();                                  // previous tile's readers are done
            stage_text_tile(ts, b, base, s_txt);
            __syncthreads();
            // blocked: 8 consecutive suffixes per thread share their symbols -- 12 bytes (one b64 + one b32
            // LDS read) and a rolling base-257 code instead of 40 byte reads (which suffix a thread counts
            // does not matter for a histogram)
            const uint32_t li0 = tid * RS_ITEMS, gi0 = base + li0;
            const uint2 lo = *reinterpret_cast<const uint2 *>(s_txt + li0);
            const uint32_t hi = *reinterpret_cast<const uint32_t *>(s_txt + li0 + 8);
            uint32_t sy[RS_ITEMS + 4];
#pragma unroll
            for (int j = 0; j < RS_ITEMS + 4; j++) {
                const uint32_t wd = j < 4 ? lo.x : (j < 8 ? lo.y : hi);
                sy[j] = (gi0 + j < ts.n) ? ((wd >> (8 * (j & 3))) & 0xFFu) + 1u : 0u;
            }
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) c = c * 257 + sy[j];
#pragma unroll
            for (int r = 0; r < RS_ITEMS; r++) {
                kq[r] = c << VAL_BITS;                        // the index bits are not part of any digit
                c = (c - (uint64_t)sy[r] * 4362470401ull) * 257 + sy[r + 5 < RS_ITEMS + 4 ? r + 5 : 0];
            }
        } else {
#pragma unroll
            for (int r = 0; r < RS_ITEMS; r++) {              // loads first (LDS atomics would serialise them)
                const uint32_t i = base + r * RS_NT + tid;
                kq[r] = (i < m) ? K[i] : 0ull;
            }
        }
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            const uint32_t i = FROM_TEXT ? base + tid * RS_ITEMS + r : base + r * RS_NT + tid;
            if (i < m) {
                const uint64_t k = kq[r];
                for (uint32_t p = 0; p < pp.npass; p++)
                    atomicAdd(&s_h[p][(uint32_t)(k >> pp.shift[p]) & ((1u << pp.bits[p]) - 1u)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *G = ghist + (size_t)b * RS_MAXPASS * SA_MAXRADIX;
    for (uint32_t i = tid; i < pp.npass * SA_MAXRADIX; i += RS_NT) {
        const uint32_t c = (&s_h[0][0])[i];
        if (c) atomicAdd(&G[i], c);
    }
}

__global__ __launch_bounds__(512) void k_rs_digitbase(const uint32_t *__restrict__ ghist,
                                                      uint32_t *__restrict__ digit_base)
{
    __shared__ uint32_t s_tmp[16];
    const size_t o = ((size_t)blockIdx.x * RS_MAXPASS + blockIdx.y) * SA_MAXRADIX + threadIdx.x;
    const uint32_t c = ghist[o];
    digit_base[o] = block_excl_add<512>(c, s_tmp);
}

constexpr uint32_t OS_AGG = 1u << 22, OS_PFX = 2u << 22, OS_FLAGS = 3u << 22, OS_CNT = (1u << 22) - 1;

template <int BITS, bool FROM_TEXT = false>
__global__ __launch_bounds__(RS_NT, 8) void k_rs_onesweep(const uint64_t *__restrict__ key_in,
                                                       uint64_t *__restrict__ key_out,
                                                       const uint32_t *__restrict__ cnt, uint32_t nfixed,
                                                       uint32_t shift, uint32_t *__restrict__ state,
                                                       uint32_t *__restrict__ ticket, uint32_t epoch,
                                                       const uint32_t *__restrict__ digit_base, uint32_t db_stride,
                                                       uint32_t nmax, uint32_t max_tiles,
                                                       uint32_t *__restrict__ d_err, TextSrc ts = TextSrc{})
{
    constexpr int RADIX = 1 << BITS;
    static_assert(RADIX <= RS_NT, "one digit per thread in the look-back");
    __shared__ uint16_t s_wc[RS_WAVES][RADIX];         // 16-bit: counts <= 512 per wave, starts < RS_TILE (4 workgroups / CU)
    __shared__ uint32_t s_gbase[RADIX];
    __shared__ uint64_t s_key[RS_TILE];
    __shared__ uint32_t s_tmp[RS_WAVES + 1];
    __shared__ uint32_t s_tile;
    // (Measured and rejected on MI355X: persistent workgroups, 4 per block, prefetching the next
    //  tile's ticket and words while ranking -- 1.49-1.95 ms per full pass against 1.27 ms: the
    //  second set of words costs the 64-VGPR budget and sibling workgroups look back in lock step.
    //  A global ticket over (tile, block) pairs -- no lock step -- at 6 waves/SIMD: 1.40-1.54 ms.)
    // blocks are the FAST grid dimension: consecutive workgroups take tiles of different blocks, so the
    // predecessors of a tile (same block) were dispatched a whole row of workgroups earlier and its
    // look-back finds a finished prefix at the first probe (tiles-fast dispatch: 33 % of the pass was
    // spent walking back over in-flight predecessors)
    const uint32_t b = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const uint32_t m = live_count(cnt, nfixed, b);
    if (tid == 0) s_tile = atomicAdd(&ticket[b], 1u);
    for (uint32_t i = tid; i < RS_WAVES * RADIX / 2; i += RS_NT) reinterpret_cast<uint32_t *>(&s_wc[0][0])[i] = 0;
    __syncthreads();
    const uint32_t t = s_tile, base = t * RS_TILE;
    if (base >= m) return;
    const uint32_t tile_n = min((uint32_t)RS_TILE, m - base);
    const uint64_t *K = key_in + (size_t)b * nmax + base;
    uint64_t *KO = key_out + (size_t)b * nmax;

    const uint32_t dbase = (tid < RADIX) ? digit_base[(size_t)b * db_stride + tid] : 0u;   // off the critical path
    uint64_t k[RS_ITEMS];
    uint32_t rk[RS_ITEMS];
    // all loads first: the wave barriers in the ranking loop pin memory operations, and a load
    // issued inside it is waited for before the next one starts (8 serial HBM latencies per tile)
    if (FROM_TEXT) {                                          // s_key is free until the bucketing phase
        uint8_t *s_txt = reinterpret_cast<uint8_t *>(s_key);
        stage_text_tile(ts, b, base, s_txt);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            const uint32_t i = w * (RS_TILE / RS_WAVES) + r * 64 + l;
            k[r] = (i < tile_n) ? text_word(s_txt, i, base + i, ts.n) : 0ull;
        }
    } else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            const uint32_t i = w * (RS_TILE / RS_WAVES) + r * 64 + l;
            k[r] = (i < tile_n) ? K[i] : 0ull;
        }
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const uint32_t i = w * (RS_TILE / RS_WAVES) + r * 64 + l;
        const bool valid = i < tile_n;
        const uint32_t d = (uint32_t)(k[r] >> shift) & (RADIX - 1);
        const uint64_t peers = wave_match<BITS>(d, __ballot(valid));
        const uint32_t pre = mbcnt(peers), tot = (uint32_t)__popcll(peers);
        const uint32_t old = s_wc[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && pre == 0) s_wc[w][d] = (uint16_t)(old + tot);
        __builtin_amdgcn_wave_barrier();
        rk[r] = old + pre;
    }
    __syncthreads();

    uint32_t c[RS_WAVES], tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int q = 0; q < RS_WAVES; q++) { c[q] = s_wc[q][tid]; tot += c[q]; }
    }
    // publish this tile's count of digit `tid` as soon as it is known, then look back
    uint32_t excl = 0;
    if (tid < RADIX) {
        uint32_t *ST = state + ((size_t)b * max_tiles) * SA_MAXRADIX + tid;
        const uint32_t tag = epoch << 24;
        if (t == 0) {
            __hip_atomic_store(&ST[0], tag | OS_PFX | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // Probe the predecessor FIRST.  With blocks as the fast grid dimension it is a whole row
            // of workgroups ahead and has usually finished (measured: 1.13 probes per tile, 0.03 of them
            // finding an unpublished granule), so the common case is one load round trip followed by
            // the inclusive-prefix store -- no aggregate store to wait for (agent-scope stores write
            // through the per-XCD L2 and a wait placed after one covers its acknowledgement too).
            // Only when the predecessor is still in flight is the aggregate published and the walk
            // continued; a successor that probes us before either store just polls again.
            int look = (int)t - 1;
            uint32_t g = __hip_atomic_load(&ST[(size_t)look * SA_MAXRADIX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!((g >> 24) == epoch && (g & OS_FLAGS) == OS_PFX)) {
                __hip_atomic_store(&ST[(size_t)t * SA_MAXRADIX], tag | OS_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t spins = 0;
                for (;;) {
                    if ((g >> 24) == epoch && (g & OS_FLAGS) != 0) {
                        excl += g & OS_CNT;
                        if ((g & OS_FLAGS) == OS_PFX) break;
                        look--;
                    } else {                                              // not published in this launch yet
                        if (++spins > (1u << 22)) { atomicOr(d_err, 8u); break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    g = __hip_atomic_load(&ST[(size_t)look * SA_MAXRADIX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                excl = g & OS_CNT;
            }
            __hip_atomic_store(&ST[(size_t)t * SA_MAXRADIX], tag | OS_PFX | (excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    uint32_t run = block_excl_add<RS_NT>(tot, s_tmp);
    if (tid < RADIX) {
        s_gbase[tid] = dbase + excl - run;
#pragma unroll
        for (int q = 0; q < RS_WAVES; q++) { s_wc[q][tid] = (uint16_t)run; run += c[q]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const uint32_t i = w * (RS_TILE / RS_WAVES) + r * 64 + l;
        if (i < tile_n) {
            const uint32_t d = (uint32_t)(k[r] >> shift) & (RADIX - 1);
            s_key[s_wc[w][d] + rk[r]] = k[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const uint32_t p = r * RS_NT + tid;
        if (p < tile_n) {
            const uint64_t kk = s_key[p];
            const uint32_t d = (uint32_t)(kk >> shift) & (RADIX - 1);
            KO[s_gbase[d] + p] = kk;
        }
    }
}

// ---------------------------------------------------------------------------
// After a sort: group heads, ranks, write-back, compaction of unresolved.
// A suffix is resolved when its group (equal sort key) is a singleton.
//
// Two refinement modes for the NEXT round's words:
//   MODE_TEXT  [dense group number:19 | code of symbols d..d+2 (base 257):25 | i:20]
//              -- no rank array at all: the next 3 symbols come straight from the
//              text.  Shallow-LCP data (i.i.d. bytes, float mantissas) is finished by
//              1-2 such rounds and never pays the random ISA scatter.
//   MODE_ISA   [rank(i):21 | rank(i+h):21 | i:20]  classic prefix doubling; ISA is
//              scattered for every suffix of the list.
// ---------------------------------------------------------------------------
constexpr int MODE_ISA = 0, MODE_TEXT = 1;
constexpr uint32_t TXT_GRP_SHIFT = 45, TXT_CODE_SHIFT = 20;

// ---------------------------------------------------------------------------
// One kernel per round.  The sorted words are staged through LDS once; every tile gets its exclusive prefix
// (last head, #unresolved, #unresolved groups) from k_rank_pre + k_rank_scan below, so tiles are static: no
// tickets, no look-back, no spinning.  The BWT byte T[SA-1] is gathered here, so there is no separate gather
// kernel and (unless the caller wants it) no second pass over the suffix array.
// ---------------------------------------------------------------------------
constexpr uint64_t LB_AGG = 1ull << 62, LB_PFX = 2ull << 62, LB_FLAGS = 3ull << 62;

__device__ __forceinline__ uint64_t lb_pack(uint32_t head, uint32_t uc, uint32_t uh)
{
    return ((uint64_t)head << 41) | ((uint64_t)uc << 20) | (uint64_t)uh;
}
__device__ __forceinline__ uint64_t lb_combine(uint64_t earlier, uint64_t later)
{   // flags stripped; head = max, counts add (fields cannot overflow: uc <= 2^20, uh <= 2^19)
    const uint64_t he = earlier >> 41, hl = later >> 41;
    const uint64_t cnts = (earlier & ((1ull << 41) - 1)) + (later & ((1ull << 41) - 1));
    return ((he > hl ? he : hl) << 41) | cnts;
}

// Tile aggregates ahead of the rank kernel.  A decoupled look-back inside k_sa_rank1 (all 512 tiles of a
// block are in flight together, so every tile waits for up to 511 predecessors) cost 1.2 of its 2.9 ms;
// reading the sorted words once more (0.45 ms) and scanning the 512 aggregates of a block in one
// workgroup gives every tile its exclusive prefix before the rank kernel starts -- no tickets, no spinning.
__global__ __launch_bounds__(SA_THREADS) void k_rank_pre(const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                         uint32_t nfixed, unsigned long long *__restrict__ tile_agg,
                                                         uint32_t nmax, uint32_t max_tiles)
{
    __shared__ uint32_t s_r[3][SA_THREADS / 64];
    const uint32_t tid = threadIdx.x, b = blockIdx.y, t = blockIdx.x;
    const uint32_t m = live_count(cnt, nfixed, b), base = t * SA_TILE;
    if (base >= m) return;
    const uint64_t *K = key + (size_t)b * nmax;
    // coalesced: row r of the tile is 256 consecutive elements, one per thread; the neighbours come from
    // the adjacent lanes (DPP wave_shr:1 / wave_shl:1), the two edge lanes of a wave load theirs
    const uint32_t l = tid & 63;
    uint64_t c[SA_ITEMS], edge[SA_ITEMS];
#pragma unroll
    for (int r = 0; r < SA_ITEMS; r++) {
        const uint32_t e = base + r * SA_THREADS + tid;
        c[r] = K[e < m ? e : 0u];
    }
#pragma unroll
    for (int r = 0; r < SA_ITEMS; r++) {
        const uint32_t e = base + r * SA_THREADS + tid;
        const uint32_t ei = (l == 0) ? e - 1 : e + 1;           // (e = 0 wraps to 0xFFFFFFFF: fails the bound test)
        const bool ed = (l == 0 || l == 63) && ei < m;
        edge[r] = K[ed ? ei : (e < m ? e : 0u)];                // unconditional (interior lanes re-read their own word:
                                                                 // an L1 hit) so the loads of all rows stay in flight
    }
    uint32_t lh = 0, uc = 0, uh = 0;
#pragma unroll
    for (int r = 0; r < SA_ITEMS; r++) {
        const uint32_t e = base + r * SA_THREADS + tid;
        const uint32_t ei = (l == 0) ? e - 1 : e + 1;
        const bool ed = (l == 0 || l == 63) && ei < m;
        const uint64_t kc = c[r] >> VAL_BITS, ke = ed ? edge[r] >> VAL_BITS : 0ull;
        const uint32_t clo = (uint32_t)kc, chi = (uint32_t)(kc >> 32);
        uint64_t kp = ((uint64_t)GLC_DPP(chi, 0x138, 0xf) << 32) | GLC_DPP(clo, 0x138, 0xf);      // lane - 1
        uint64_t kn = ((uint64_t)GLC_DPP(chi, 0x130, 0xf) << 32) | GLC_DPP(clo, 0x130, 0xf);      // lane + 1
        if (l == 0) kp = ke;
        if (l == 63) kn = ke;
        if (e < m) {
            const bool head = (e == 0) || (kc != kp);
            const bool nhead = (e + 1 >= m) || (kn != kc);
            if (head) lh = max(lh, e);
            if (!(head && nhead)) { uc++; if (head) uh++; }
        }
    }
    lh = wave_max(lh); uc = wave_sum(uc); uh = wave_sum(uh);
    if ((tid & 63) == 0) { s_r[0][tid >> 6] = lh; s_r[1][tid >> 6] = uc; s_r[2][tid >> 6] = uh; }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0, c = 0, h = 0;
#pragma unroll
        for (int q = 0; q < SA_THREADS / 64; q++) { a = max(a, s_r[0][q]); c += s_r[1][q]; h += s_r[2][q]; }
        tile_agg[(size_t)b * max_tiles + t] = lb_pack(a, c, h);
    }
}

// exclusive scan of the tile aggregates of a block (in place) + the block's unresolved total
__global__ __launch_bounds__(256) void k_rank_scan(unsigned long long *__restrict__ tile_agg,
                                                   const uint32_t *__restrict__ cnt, uint32_t nfixed,
                                                   uint32_t max_tiles, uint32_t *__restrict__ cnt_next,
                                                   uint32_t *__restrict__ d_max_cnt)
{
    __shared__ uint32_t s_tmp[8];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t m = live_count(cnt, nfixed, b);
    const uint32_t ntiles = (m + SA_TILE - 1) / SA_TILE;
    unsigned long long *A = tile_agg + (size_t)b * max_tiles;
    uint32_t run_h = 0, run_c = 0, run_u = 0;                   // carried over 256-tile rounds
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 256) {
        const uint32_t t = t0 + tid;
        const uint64_t a = t < ntiles ? A[t] : 0ull;
        const uint32_t h = (uint32_t)(a >> 41), c = (uint32_t)((a >> 20) & 0x1FFFFF), u = (uint32_t)(a & 0xFFFFF);
        uint32_t tc = 0, tu = 0;
        const uint32_t eh = block_excl_max<256>(h, s_tmp);
        const uint32_t ec = block_excl_add<256>(c, s_tmp, &tc);
        const uint32_t eu = block_excl_add<256>(u, s_tmp, &tu);
        uint32_t th = wave_max(h);
        __syncthreads();
        if ((tid & 63) == 0) s_tmp[tid >> 6] = th;
        __syncthreads();
        th = max(max(s_tmp[0], s_tmp[1]), max(s_tmp[2], s_tmp[3]));
        __syncthreads();
        if (t < ntiles) A[t] = lb_pack(max(run_h, eh), run_c + ec, run_u + eu);
        run_h = max(run_h, th); run_c += tc; run_u += tu;
    }
    if (tid == 0) {
        cnt_next[b] = run_c;
        if (run_c) { atomicMax(d_max_cnt, run_c); atomicAdd(d_max_cnt + 1, run_c); }
    }
}

template <bool ROUND0>
__global__ __launch_bounds__(SA_THREADS) void k_sa_rank1(const uint64_t *__restrict__ key,
                                                         const uint32_t *__restrict__ pos,
                                                         const uint32_t *__restrict__ cnt, uint32_t nfixed,
                                                         unsigned long long *__restrict__ tile_state,
                                                         uint32_t *__restrict__ isa, uint32_t *__restrict__ sa,
                                                         uint64_t *__restrict__ key_next,
                                                         uint32_t *__restrict__ pos_next,
                                                         uint32_t *__restrict__ hd_next,
                                                         uint32_t *__restrict__ cnt_next,
                                                         uint32_t *__restrict__ d_max_cnt,
                                                         uint32_t nmax, uint32_t max_tiles, int mode,
                                                         const uint8_t *__restrict__ text, size_t text_stride,
                                                         uint32_t n, uint32_t depth,
                                                         uint8_t *__restrict__ bwt_out, size_t bwt_stride,
                                                         int *__restrict__ d_index, uint32_t *__restrict__ d_err)
{
    // LK(1 + i) = word of tile element i; one pad slot per 8 words so the blocked (8 per thread) reads
    // are bank-conflict free (lane stride 72 B instead of 64 B = 16-way conflict on ds_read_b64)
#define LK(j) s_k[(j) + ((j) >> 3)]
    __shared__ uint64_t s_k[SA_TILE + 2 + (SA_TILE + 2) / 8 + 1];
    __shared__ uint32_t s_tmp[12];
    const uint32_t tid = threadIdx.x;
    // (an XCD-aware remap -- all tiles of a block on one XCD so its text stays in that L2 -- was
    //  measured slower on MI355X: 17.5 vs 16.3 ms per 256 blocks; plain dispatch order is kept)
    const uint32_t b = blockIdx.y;
    const uint32_t m = live_count(cnt, nfixed, b);
    const uint32_t t = blockIdx.x, base = t * SA_TILE;
    if (base >= m) return;
    const uint64_t excl = tile_state[(size_t)b * max_tiles + t];   // exclusive prefix from k_rank_pre + k_rank_scan
    const uint64_t *K = key + (size_t)b * nmax;
    const uint32_t *P = pos ? pos + (size_t)b * nmax : nullptr;
    {   // SA_ITEMS + 1 loads per thread, all in flight before the first LDS store
        uint64_t kq[SA_ITEMS + 1];
#pragma unroll
        for (int r = 0; r <= SA_ITEMS; r++) {
            const uint32_t i = r * SA_THREADS + tid;
            const int64_t g = (int64_t)base - 1 + i;
            kq[r] = (i < SA_TILE + 2 && g >= 0 && g < (int64_t)m) ? K[g] : 0ull;
        }
#pragma unroll
        for (int r = 0; r <= SA_ITEMS; r++) {
            const uint32_t i = r * SA_THREADS + tid;
            if (i < SA_TILE + 2) LK(i) = kq[r];
        }
    }
    __syncthreads();
    const uint32_t l0 = tid * SA_ITEMS, e0 = base + l0;
    uint32_t headm = 0, unresm = 0, lh = 0, uc = 0, uh = 0;
#pragma unroll
    for (int i = 0; i < SA_ITEMS; i++) {
        const uint32_t e = e0 + i;
        if (e < m) {
            const uint64_t kp = LK(l0 + i) >> VAL_BITS, kc = LK(l0 + i + 1) >> VAL_BITS, kn = LK(l0 + i + 2) >> VAL_BITS;
            const bool head = (e == 0) || (kc != kp);
            const bool nhead = (e + 1 >= m) || (kn != kc);
            if (head) { headm |= 1u << i; lh = e; }
            if (!(head && nhead)) { unresm |= 1u << i; uc++; if (head) uh++; }
        }
    }
    // thread-exclusive prefixes within the tile + tile aggregate
    const uint32_t carry_t = block_excl_max<SA_THREADS>(lh, s_tmp);
    uint32_t tile_uc = 0, tile_uh = 0;
    const uint32_t off_t = block_excl_add<SA_THREADS>(uc, s_tmp, &tile_uc);
    const uint32_t goff_t = block_excl_add<SA_THREADS>(uh, s_tmp, &tile_uh);
    const uint32_t carry = max(carry_t, (uint32_t)(excl >> 41));
    uint32_t off = (uint32_t)((excl >> 20) & 0x1FFFFF) + off_t;
    uint32_t gcount = (uint32_t)(excl & 0xFFFFF) + goff_t;

    uint32_t *ISA = isa + (size_t)b * nmax, *SAo = sa + (size_t)b * nmax;
    uint64_t *KN = key_next + (size_t)b * nmax;
    uint32_t *PN = pos_next + (size_t)b * nmax;
    uint32_t *HN = hd_next + (size_t)b * nmax;
    const uint8_t *T = text + (size_t)b * text_stride;
    uint8_t *O = bwt_out ? bwt_out + (size_t)b * bwt_stride : nullptr;
    // Phase 1 (registers / LDS only): suffix index, group head position of every element
    uint32_t v[SA_ITEMS], hdpos[SA_ITEMS];
    {
        uint32_t running = carry;
#pragma unroll
        for (int i = 0; i < SA_ITEMS; i++) {
            if (headm & (1u << i)) running = e0 + i;
            hdpos[i] = running;
            v[i] = (uint32_t)(LK(l0 + i + 1) & VAL_MASK);
        }
    }
    // Phase 2: every gather of the tile issued back to back (a gather inside the store loop is
    // waited for before the next one starts).  ROUND0: slot == e and the head's slot is its own
    // position, so nothing is gathered but the BWT byte.
    uint32_t slot[ROUND0 ? 1 : SA_ITEMS], grp[ROUND0 ? 1 : SA_ITEMS];
    if (!ROUND0) {
#pragma unroll
        for (int i = 0; i < SA_ITEMS; i++) {
            const uint32_t e = e0 + i;
            slot[i] = P[e < m ? e : 0u]; grp[i] = P[e < m ? hdpos[i] : 0u];
        }
    }
#define SLOT(i) (ROUND0 ? e0 + (i) : slot[ROUND0 ? 0 : (i)])
#define GRP(i)  (ROUND0 ? hdpos[i] : grp[ROUND0 ? 0 : (i)])
    uint32_t ch[SA_ITEMS];
    if (O) {
#pragma unroll
        for (int i = 0; i < SA_ITEMS; i++) ch[i] = T[(e0 + i < m) ? (v[i] ? v[i] - 1 : n - 1) : 0u];
    }
    if (ROUND0) {                                             // SA[e] = v, coalesced straight from the staged words
        for (uint32_t i = tid; i < SA_TILE; i += SA_THREADS)
            if (base + i < m) SAo[base + i] = (uint32_t)(LK(i + 1) & VAL_MASK);
    }
    __syncthreads();                                          // the staged words are dead: s_k becomes the
    uint64_t *s_un = s_k;                                     // tile's list of unresolved {group, suffix}
    const uint32_t tile_off0 = (uint32_t)((excl >> 20) & 0x1FFFFF);
    // Phase 3: stores
    uint32_t packed[SA_ITEMS / 4] = {};
#pragma unroll
    for (int i = 0; i < SA_ITEMS; i++) {
        const uint32_t e = e0 + i;
        if (e < m) {
            const bool head = headm & (1u << i), unres = unresm & (1u << i);
            if (!ROUND0) SAo[SLOT(i)] = v[i];
            if (O) {
                if (v[i] == 0) d_index[b] = (int)SLOT(i);
                if (!ROUND0) O[SLOT(i)] = (uint8_t)ch[i];
                else packed[i >> 2] |= ch[i] << (8 * (i & 3));
            }
            if (mode == MODE_ISA) {
                ISA[v[i]] = GRP(i) + 1;
                if (unres) {
                    PN[off] = SLOT(i);
                    KN[off] = ((uint64_t)(GRP(i) + 1) << R1_SHIFT) | v[i];
                    off++;
                }
            } else {
                if (head && unres) gcount++;
                if (unres) {
                    PN[off] = SLOT(i);
                    HN[off] = GRP(i);
                    s_un[off - tile_off0] = ((uint64_t)(gcount - 1) << TXT_GRP_SHIFT) | v[i];
                    off++;
                }
            }
        }
    }
#undef SLOT
#undef GRP
    if (mode != MODE_ISA) {
        // next round's words for the unresolved suffixes, densely: one thread per list entry reads the
        // 3 symbols that follow the `depth` already compared (the same loads issued per tile ELEMENT
        // touch 24 mostly-idle wave instructions when 4 % of the elements are unresolved)
        __syncthreads();
        for (uint32_t j = tid; j < tile_uc; j += SA_THREADS) {
            const uint64_t ent = s_un[j];
            const uint32_t p0 = (uint32_t)(ent & VAL_MASK) + depth;
            const uint32_t x0 = T[p0 < n ? p0 : 0u], x1 = T[p0 + 1 < n ? p0 + 1 : 0u], x2 = T[p0 + 2 < n ? p0 + 2 : 0u];
            const uint32_t c0 = p0 < n ? x0 + 1 : 0u, c1 = p0 + 1 < n ? x1 + 1 : 0u, c2 = p0 + 2 < n ? x2 + 1 : 0u;
            const uint64_t code = ((uint64_t)c0 * 257 + c1) * 257 + c2;
            KN[tile_off0 + j] = ent | (code << TXT_CODE_SHIFT);
        }
    }
    if (O && ROUND0) {                                        // round 0: slot == e, 8 consecutive bytes per thread
        const bool al = ((reinterpret_cast<uintptr_t>(O) & 3) == 0);
#pragma unroll
        for (int q = 0; q < SA_ITEMS / 4; q++) {
            const uint32_t e = e0 + 4 * q;
            if (e + 3 < m && al) *reinterpret_cast<uint32_t *>(O + e) = packed[q];
            else for (int j = 0; j < 4; j++) if (e + j < m) O[e + j] = (uint8_t)(packed[q] >> (8 * j));
        }
    }
}

// ---------------------------------------------------------------------------
// Tile-local refinement.  The unresolved list of a round is in SA order, so the members of a group
// (equal bits >= gshift) are contiguous, and a refinement round only has to order each group by its
// new key bits [20, gshift).  Groups are tiny (2-10 suffixes on Zipf data, at most a few hundred), so
// instead of five global radix passes over the list every entry ranks itself inside its group by a
// direct count -- one read and one write of the list.  A tile owns the groups that START in it and
// stages RL_HALO more entries for their tails; a group longer than RL_HALO flags its block, and
// flagged blocks (only those) go through the global radix sort afterwards.
// ---------------------------------------------------------------------------
constexpr uint32_t RL_T = 2048, RL_HALO = 2048, RL_NT = 256;
#ifndef GLC_RL_MAXG
#define GLC_RL_MAXG 128
#endif
constexpr uint32_t RL_MAXG = GLC_RL_MAXG;                       // longest group ranked by direct count (quadratic in the group: 1024 until
                                                                // round 6 -- a 256 KiB periodic stretch inside Zipf data spent a third of its 17.8 ms
                                                                // per 32 blocks counting inside groups of hundreds; 128: 12.8 ms, 32: the same, and
                                                                // bench.py's partly_deep batch 9.4 / 9.2 / 10.5 ms)
constexpr uint32_t RL_E = (RL_T + RL_HALO + 1 + RL_NT - 1) / RL_NT;   // staged entries per thread (17)

__global__ __launch_bounds__(RL_NT) void k_refine_local(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                        const uint32_t *__restrict__ cnt, uint32_t nmax,
                                                        uint32_t gshift, uint32_t *__restrict__ flag)
{
    __shared__ uint64_t s_w[RL_NT * RL_E];
    __shared__ uint16_t s_gs[RL_NT * RL_E], s_end[RL_NT * RL_E];
    __shared__ uint32_t s_tmp[RL_NT / 64 + 1];
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    const uint32_t m = cnt[b], t0 = blockIdx.x * RL_T;
    if (t0 >= m) return;
    const uint64_t *I = in + (size_t)b * nmax;
    uint64_t *O = out + (size_t)b * nmax;
    const uint32_t hi = min(m, t0 + RL_T + RL_HALO);            // staged list range [t0 - 1, hi)
    const uint32_t cntl = hi - t0 + 1;                          // local index j <-> list index t0 - 1 + j
    {
        uint64_t q[RL_E];
#pragma unroll
        for (int r = 0; r < (int)RL_E; r++) {                   // all loads in flight, coalesced
            const uint32_t i = r * RL_NT + tid;
            const int64_t g = (int64_t)t0 - 1 + i;
            q[r] = (i < cntl && g >= 0) ? I[g] : ~0ull;        // before the list: a group nobody has
        }
#pragma unroll
        for (int r = 0; r < (int)RL_E; r++) s_w[r * RL_NT + tid] = q[r];
    }
    __syncthreads();
    // group start of every staged entry: max-scan of the head positions (thread = RL_E consecutive entries)
    const uint32_t j0 = tid * RL_E;
    uint32_t lh = 0;
    uint32_t headm = 0;
#pragma unroll
    for (uint32_t i = 0; i < RL_E; i++) {
        const uint32_t j = j0 + i;
        if (j > 0 && j < cntl && (s_w[j] >> gshift) != (s_w[j - 1] >> gshift)) { headm |= 1u << i; lh = j; }
    }
    uint32_t run = block_excl_max<RL_NT>(lh, s_tmp);
#pragma unroll
    for (uint32_t i = 0; i < RL_E; i++) {
        const uint32_t j = j0 + i;
        if (headm & (1u << i)) run = j;
        if (j < cntl) { s_gs[j] = (uint16_t)run; s_end[j] = 0xFFFFu; }
    }
    __syncthreads();
    // group end, stored at the group's start: the next head, or the end of the LIST (not of the staged range)
#pragma unroll
    for (uint32_t i = 0; i < RL_E; i++) {
        const uint32_t j = j0 + i;
        if ((headm & (1u << i)) && j > 0) s_end[s_gs[j - 1]] = (uint16_t)j;
    }
    if (tid == 0 && t0 - 1 + cntl == m) s_end[s_gs[cntl - 1]] = (uint16_t)cntl;
    __syncthreads();
    // owned entries rank themselves inside their group
    for (uint32_t j = 1 + tid; j < cntl; j += RL_NT) {
        const uint32_t gs = s_gs[j];
        if (gs == 0 || gs > RL_T) continue;                     // group started before this tile / starts in the halo
        const uint32_t ge = s_end[gs];
        if (ge == 0xFFFFu || ge - gs > RL_MAXG) {               // tail not staged or too long for a direct count
            if (j == gs) atomicOr(&flag[b], 1u);
            continue;
        }
        const uint64_t w = s_w[j], key = w >> VAL_BITS;
        uint32_t rank = 0;
        for (uint32_t f = gs; f < ge; f++) {
            const uint64_t kf = s_w[f] >> VAL_BITS;
            rank += (kf < key || (kf == key && f < j)) ? 1u : 0u;
        }
        O[t0 - 1 + gs + rank] = w;
    }
}

__global__ void k_refine_counts(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ flag,
                                uint32_t *__restrict__ cnt_flagged, uint32_t nblk, uint32_t *__restrict__ total)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) {
        const uint32_t c = flag[b] ? cnt[b] : 0u;
        cnt_flagged[b] = c;
        if (c) atomicAdd(total, c);
    }
}

// switch from text refinement to prefix doubling: ranks of every suffix in the order
// established so far.  Resolved suffixes: rank = own SA slot + 1 ...
__global__ __launch_bounds__(256) void k_isa_init(const uint32_t *__restrict__ sa, uint32_t *__restrict__ isa,
                                                  uint32_t n, uint32_t nmax, const uint32_t *__restrict__ cnt0)
{
    const uint32_t b = blockIdx.y;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < live_count(cnt0, n, b)) isa[(size_t)b * nmax + sa[(size_t)b * nmax + j]] = j + 1;
}

// ... unresolved suffixes: rank = SA slot of their group head + 1; also rewrite their words
// into the doubling format [rank(i):21 | (rank(i+h) filled by k_sa_fill_rank2) | i:20]
__global__ __launch_bounds__(SA_THREADS) void k_isa_fix(uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                        const uint32_t *__restrict__ hd, uint32_t *__restrict__ isa,
                                                        uint32_t nmax)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    uint64_t *K = key + (size_t)b * nmax;
    const uint32_t *H = hd + (size_t)b * nmax;
    uint32_t *ISA = isa + (size_t)b * nmax;
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        const uint32_t v = (uint32_t)(K[i] & VAL_MASK);
        const uint32_t r1 = H[i] + 1;
        ISA[v] = r1;
        K[i] = ((uint64_t)r1 << R1_SHIFT) | v;
    }
}

// ---------------------------------------------------------------------------
// CHAIN GROUPS (round 6).  A group of the doubling rounds is the set of ALL suffixes that share their first h symbols.  If its
// members, by position, are i_0 < i_1 = i_0 + d < ... < i_{L-1} = i_0 + (L - 1) d and the text is d-periodic over [i_0, i_{L-1})
// (T[j] = T[j + d]), then suffix(i_k) = u . suffix(i_{k+1}) with ONE string u = T[i_0 .. i_0 + d) for every k < L - 1, so the
// comparison of i_k with i_{k+1} is the comparison of i_{k+1} with i_{k+2}, ... down to that of the last two members: the whole
// group is ONE monotone chain, ascending or descending by position, and the direction is decided where suffix(i_{L-1}) and
// suffix(i_{L-2}) differ -- within d + h symbols, because i_{L-1} + d is not a member (a group is a full equivalence class).
// Such groups are what periodic stretches turn into (one residue class of the period per group), and plain doubling needs
// log2(stretch / h) more rounds over ALL of their members to order them: two periodic halves of a 1 MiB block, 17 rounds of 2.8 ms
// per 32 blocks.  Here a group found to be a chain gets its final order in ONE round: its members' second key is their place in
// the chain instead of rank(i + h), the round's sort and rank pass do the rest.
//   k_chain_init    heads of the unresolved list's groups clear the group's record {min position, ~max position} and its info word
//   k_chain_minmax  every member: atomic min / max (one pair per wave where a wave's 64 entries are one group)
//   k_chain_decide  the group's last entry knows L (its SA slot - the head's + 1): d = (max - min) / (L - 1) if that divides and
//                   d <= CHAIN_DMAX -> candidate
//   k_chain_verify  every member: on the lattice min + k d?  and T[v .. v + d) == T[v + d .. v + 2 d) where both are below max
//   k_chain_dir     the last entry: the direction, by the text, from symbol h on (at most d + 16 more)
// The records live in the round's spare word array (free between the rank pass and the sort), the info words in the group-head
// array the text-refinement rounds use (dead in doubling rounds).
// ---------------------------------------------------------------------------
constexpr uint32_t CHAIN_DMAX = 4096;                          // longest stride tried (the verification reads d bytes per member, the direction walk d + 16)
constexpr uint32_t CHAIN_CAND = 1u << 31, CHAIN_OK = 1u << 30, CHAIN_DESC = 1u << 29, CHAIN_D = (1u << 20) - 1;

__global__ __launch_bounds__(SA_THREADS) void k_chain_init(const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                           uint2 *__restrict__ rec, uint32_t *__restrict__ info, uint32_t nmax)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    const uint64_t *K = key + (size_t)b * nmax;
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        const uint32_t g = (uint32_t)(K[i] >> R1_SHIFT);
        if (i == 0 || (uint32_t)(K[i - 1] >> R1_SHIFT) != g) {
            rec[(size_t)b * nmax + g - 1] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            info[(size_t)b * nmax + g - 1] = 0u;
        }
    }
}

__global__ __launch_bounds__(SA_THREADS) void k_chain_minmax(const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                             uint2 *__restrict__ rec, uint32_t nmax)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    const uint64_t *K = key + (size_t)b * nmax;
    uint2 *R = rec + (size_t)b * nmax;
    const uint32_t rounds = (m + gridDim.x * SA_THREADS - 1) / (gridDim.x * SA_THREADS);      // (uniform: every lane of a wave takes the same trips)
    for (uint32_t t = 0; t < rounds; t++) {
        const uint32_t i = (t * gridDim.x + blockIdx.x) * SA_THREADS + threadIdx.x;
        const bool in = i < m;
        const uint64_t k = in ? K[i] : 0ull;
        const uint32_t g = (uint32_t)(k >> R1_SHIFT), v = (uint32_t)(k & VAL_MASK);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        if (__ballot(in && g == g0) == __ballot(true)) {       // the wave's 64 entries are one group: one pair of atomics for all
            const uint32_t mn = ~wave_max(~v), mx = wave_max(v);
            if ((threadIdx.x & 63) == 0) { atomicMin(&R[g - 1].x, mn); atomicMin(&R[g - 1].y, ~mx); }
        } else if (in) { atomicMin(&R[g - 1].x, v); atomicMin(&R[g - 1].y, ~v); }
    }
}

__global__ __launch_bounds__(SA_THREADS) void k_chain_decide(const uint64_t *__restrict__ key, const uint32_t *__restrict__ pos,
                                                             const uint32_t *__restrict__ cnt, const uint2 *__restrict__ rec,
                                                             uint32_t *__restrict__ info, uint32_t nmax)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    const uint64_t *K = key + (size_t)b * nmax;
    const uint32_t *P = pos + (size_t)b * nmax;
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        const uint32_t g = (uint32_t)(K[i] >> R1_SHIFT);
        if (i + 1 < m && (uint32_t)(K[i + 1] >> R1_SHIFT) == g) continue;     // not the group's last entry
        const uint32_t L = P[i] - (g - 1) + 1;                 // the group's entries hold consecutive SA slots from the head's on
        const uint2 r = rec[(size_t)b * nmax + g - 1];
        const uint32_t mn = r.x, mx = ~r.y;
        if (L >= 2 && mx > mn && (mx - mn) % (L - 1) == 0) {
            const uint32_t d = (mx - mn) / (L - 1);
            if (d <= CHAIN_DMAX) info[(size_t)b * nmax + g - 1] = d | CHAIN_CAND;
        }
    }
}

__device__ __forceinline__ bool chain_eq16(const uint8_t *a, const uint8_t *b)
{
    uint4 x, y;
    __builtin_memcpy(&x, a, 16);
    __builtin_memcpy(&y, b, 16);
    return x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
}

// (vcache: one halfword per 16 text bytes, cleared before the pass: the stride d for which T[q .. q + 16) == T[q + d .. q + d + 16) has
//  been found to hold.  The d residue classes of ONE periodic stretch are d groups, and every one of them needs the stretch to be
//  d-periodic: without the cache each 16-byte piece of it was compared d times -- 4.8 ms per 32 blocks of two periodic halves, a third
//  of the call.  Two members that look at a piece at the same time both compare it and both store d: harmless.)
__global__ __launch_bounds__(SA_THREADS) void k_chain_verify(const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                             const uint2 *__restrict__ rec, uint32_t *__restrict__ info,
                                                             uint32_t nmax, const uint8_t *__restrict__ text, size_t text_stride,
                                                             uint16_t *__restrict__ vcache)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    const uint64_t *K = key + (size_t)b * nmax;
    const uint8_t *T = text + (size_t)b * text_stride;
    uint16_t *VC = vcache + (size_t)b * nmax * 2;              // (the spare slot array: 4 bytes per suffix, n / 16 halfwords used)
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        const uint64_t k = K[i];
        const uint32_t g = (uint32_t)(k >> R1_SHIFT), v = (uint32_t)(k & VAL_MASK);
        uint32_t *I = info + (size_t)b * nmax + g - 1;
        const uint32_t inf = *I;
        if (!(inf & CHAIN_CAND)) continue;
        const uint32_t d = inf & CHAIN_D;
        const uint2 r = rec[(size_t)b * nmax + g - 1];
        const uint32_t mn = r.x, mx = ~r.y;
        bool ok = (v - mn) % d == 0;
        if (ok && v + 2 * d <= mx) {                           // u of this member == u of the next one (both strings end below max: inside the text)
            const uint32_t end = v + d, a = min((v + 15u) & ~15u, end), z = max(end & ~15u, a);
            uint32_t q = v;
            for (; ok && q < a; q++) ok = T[q] == T[q + d];
            for (q = a; ok && q < z; q += 16) {
                if (VC[q >> 4] == (uint16_t)d) continue;       // (d <= CHAIN_DMAX < 65536, never 0)
                ok = chain_eq16(T + q, T + q + d);
                if (ok) VC[q >> 4] = (uint16_t)d;
            }
            for (q = z; ok && q < end; q++) ok = T[q] == T[q + d];
        }
        if (!ok) atomicAnd(I, ~CHAIN_CAND);
    }
}

__global__ __launch_bounds__(SA_THREADS) void k_chain_dir(const uint64_t *__restrict__ key, const uint32_t *__restrict__ cnt,
                                                          const uint2 *__restrict__ rec, uint32_t *__restrict__ info, uint32_t nmax,
                                                          const uint8_t *__restrict__ text, size_t text_stride, uint32_t n, uint32_t h)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    const uint64_t *K = key + (size_t)b * nmax;
    const uint8_t *T = text + (size_t)b * text_stride;
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        const uint32_t g = (uint32_t)(K[i] >> R1_SHIFT);
        if (i + 1 < m && (uint32_t)(K[i + 1] >> R1_SHIFT) == g) continue;     // the group's last entry decides
        uint32_t *I = info + (size_t)b * nmax + g - 1;
        const uint32_t inf = *I;
        if (!(inf & CHAIN_CAND)) continue;
        const uint32_t d = inf & CHAIN_D;
        const uint32_t mx = ~rec[(size_t)b * nmax + g - 1].y;
        // suffix(mx) against suffix(mx - d): equal in their first h symbols (one group), different within d + h
        const uint32_t A = mx, B = mx - d;                     // (A > B: A is the shorter suffix)
        uint32_t k = h & ~15u;
        const uint32_t lim = h + d + 16;
        int less = -1;                                         // suffix(A) < suffix(B) ?
        while (less < 0 && k <= lim) {
            if (A + k + 16 <= n) {
                if (chain_eq16(T + A + k, T + B + k)) { k += 16; continue; }
            }
            for (uint32_t t = 0; t < 16 && less < 0; t++) {
                if (A + k + t >= n) less = 1;                  // suffix(A) ends first: the shorter suffix is the smaller
                else if (T[A + k + t] != T[B + k + t]) less = T[A + k + t] < T[B + k + t] ? 1 : 0;
            }
            k += 16;
        }
        if (less < 0) { *I = 0u; continue; }                   // (cannot happen: see above; the group is left to the doubling)
        *I = d | CHAIN_OK | (less ? CHAIN_DESC : 0u);
    }
}

// rounds >= 1: fill in rank(i+h) -- or, for the members of a chain group, their place in the chain
__global__ __launch_bounds__(SA_THREADS) void k_sa_fill_rank2(uint64_t *__restrict__ key,
                                                              const uint32_t *__restrict__ cnt,
                                                              const uint32_t *__restrict__ isa, uint32_t n,
                                                              uint32_t h, uint32_t nmax, const uint2 *__restrict__ rec,
                                                              const uint32_t *__restrict__ info)
{
    const uint32_t b = blockIdx.y, m = cnt[b];
    uint64_t *K = key + (size_t)b * nmax;
    const uint32_t *ISA = isa + (size_t)b * nmax;
    for (uint32_t i = blockIdx.x * SA_THREADS + threadIdx.x; i < m; i += gridDim.x * SA_THREADS) {
        uint64_t k = K[i];
        uint32_t v = (uint32_t)(k & VAL_MASK);
        uint32_t r2;
        const uint32_t inf = info ? info[(size_t)b * nmax + (uint32_t)(k >> R1_SHIFT) - 1] : 0u;
        if (inf & CHAIN_OK) {
            const uint2 r = rec[(size_t)b * nmax + (uint32_t)(k >> R1_SHIFT) - 1];
            const uint32_t d = inf & CHAIN_D, at = (v - r.x) / d, L = (~r.y - r.x) / d + 1;
            r2 = (inf & CHAIN_DESC) ? L - 1 - at : at;         // descending: the member nearest the chain's end is the smallest
        } else r2 = (v + h < n) ? ISA[v + h] : 0u;
        K[i] = k | ((uint64_t)r2 << R2_SHIFT);
    }
}

__global__ void k_sa_export(const uint32_t *__restrict__ sa, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) out[0] = n;
    if (i < n) out[i + 1] = sa[i];
}

// ---------------------------------------------------------------------------
// Resuming from the sample sorter's tolerant form (bwt_bucket.hip ss_build, attempt 2): s.sa holds the suffixes of a block
// ordered by their first SS_TOL_CAP symbols -- exact everywhere except that suffixes which agree in more than that come in
// no particular order.  The GROUPS the doubling rounds have to order are the maximal ranges of neighbouring rows whose
// suffixes share SS_TOL_CAP symbols, found by looking (not taken from the sample sorter's own bookkeeping: its buckets are
// cut at suffixes compared up to the cap, so a group can straddle two of them).  The words the rank kernel expects:
// [group number : 44 | suffix : 20] in row order; everything else -- ISA, the compacted list of the unresolved, BWT
// bytes, index -- comes out of the general sorter's own first round (k_sa_rank1<true>, MODE_ISA) over these words.
// ---------------------------------------------------------------------------
#ifdef GLC_DEBUG_CAND
__device__ uint32_t g_cand_dbg[64];
#endif
constexpr uint32_t GRP_NT = 256;

__global__ __launch_bounds__(GRP_NT) void k_grp_flags(const uint8_t *__restrict__ text, size_t text_stride, uint32_t n,
                                                      uint32_t *__restrict__ sa, uint32_t nmax, const uint32_t *__restrict__ list,
                                                      const uint32_t *__restrict__ flag, uint32_t *__restrict__ tcount,
                                                      uint32_t max_gtiles, uint32_t cap)
{
    __shared__ uint32_t s_c[GRP_NT / 64];
    const uint32_t b = list[blockIdx.y], tid = threadIdx.x, r = blockIdx.x * GRP_NT + tid;
    if (flag[b]) return;                                       // the tolerant form gave this block up too (a bucket past its slot)
    const uint8_t *T = text + (size_t)b * text_stride;
    uint32_t *SA = sa + (size_t)b * nmax;
    bool head = false;
    if (r < n) {
        const uint32_t cur = SA[r];
        bool same = false;
        // Only the rows the sample sorter marked can continue a group (SA_CAND: members of a run it left as it was, and the
        // first row of a bucket); every other row it told from the row before by fewer than `cap` symbols.  (Looking at every
        // row -- two scattered 8-byte loads at least -- was 1.35 ms per 64 blocks, a tenth of the resumed path.)
#ifdef GLC_DEBUG_CAND
        if (r > 0 && !(cur & (SA_CAND | GRP_SAME))) {          // the check: would the full comparison have called this row a continuation?
            const uint32_t a = SA[r - 1] & ~(GRP_SAME | SA_CAND), c = cur & ~(GRP_SAME | SA_CAND);
            if (max(a, c) + cap <= n) {
                bool sm = true;
                for (uint32_t k = 0; k < cap; k += 8) {
                    uint64_t x, y;
                    __builtin_memcpy(&x, T + a + k, 8);
                    __builtin_memcpy(&y, T + c + k, 8);
                    if (x != y) { sm = false; break; }
                }
                if (sm) {
                    const uint32_t at = atomicAdd(&g_cand_dbg[0], 1u);
                    if (at < 15) { g_cand_dbg[1 + 4 * at] = b; g_cand_dbg[2 + 4 * at] = r; g_cand_dbg[3 + 4 * at] = a; g_cand_dbg[4 + 4 * at] = c; }
                }
            }
        }
#endif
#ifdef GLC_DEBUG_CAND
        if (r > 0 && (cur & GRP_SAME)) {                       // the check: does a row marked up front really continue its group?
            const uint32_t a = SA[r - 1] & ~(GRP_SAME | SA_CAND), c = cur & ~(GRP_SAME | SA_CAND);
            bool sm = max(a, c) + cap <= n;
            for (uint32_t k = 0; sm && k < cap; k += 8) {
                uint64_t x, y;
                __builtin_memcpy(&x, T + a + k, 8);
                __builtin_memcpy(&y, T + c + k, 8);
                sm = x == y;
            }
            if (!sm) {
                const uint32_t at = atomicAdd(&g_cand_dbg[0], 1u);
                if (at < 15) { g_cand_dbg[1 + 4 * at] = b | 0x80000000u; g_cand_dbg[2 + 4 * at] = r; g_cand_dbg[3 + 4 * at] = a; g_cand_dbg[4 + 4 * at] = c; }
            }
        }
#endif
        if (r > 0 && (cur & GRP_SAME)) same = true;            // (a member of a run the sample sorter left at the cap: marked there)
        else if (r > 0 && (cur & SA_CAND)) {
            const uint32_t prev = SA[r - 1];                   // (a neighbour may have set the mark in it already)
            const uint32_t a = prev & ~(GRP_SAME | SA_CAND), c = cur & ~(GRP_SAME | SA_CAND);
            if (max(a, c) + cap <= n) {                        // a suffix shorter than the cap shares less than the cap with anybody
                same = true;
                for (uint32_t k = 0; k < cap; k += 8) {
                    uint64_t x, y;
                    __builtin_memcpy(&x, T + a + k, 8);
                    __builtin_memcpy(&y, T + c + k, 8);
                    if (x != y) { same = false; break; }
                }
            }
        }
        head = !same;
        if (same && !(cur & GRP_SAME)) atomicOr(&SA[r], GRP_SAME);                   // (atomic: the row's own thread only ever ORs this bit; readers mask it)
    }
    const uint32_t cw = (uint32_t)__popcll(__ballot(head));
    if ((tid & 63) == 0) s_c[tid >> 6] = cw;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
#pragma unroll
        for (uint32_t w = 0; w < GRP_NT / 64; w++) t += s_c[w];
        tcount[(size_t)b * max_gtiles + blockIdx.x] = t;
    }
}

// exclusive scan of a block's tile counts, in place
__global__ __launch_bounds__(1024) void k_grp_scan(uint32_t *__restrict__ tcount, uint32_t max_gtiles, uint32_t ntiles,
                                                   const uint32_t *__restrict__ list, const uint32_t *__restrict__ flag)
{
    __shared__ uint32_t s_tmp[1024 / 64 + 1];
    const uint32_t b = list[blockIdx.x], tid = threadIdx.x;
    if (flag[b]) return;
    uint32_t *C = tcount + (size_t)b * max_gtiles;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < ntiles; t0 += 1024) {
        const uint32_t t = t0 + tid, c = t < ntiles ? C[t] : 0u;
        uint32_t tot = 0;
        const uint32_t e = block_excl_add<1024>(c, s_tmp, &tot);
        if (t < ntiles) C[t] = carry + e;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(GRP_NT) void k_grp_keys(uint32_t n, uint32_t *__restrict__ sa, uint32_t nmax,
                                                     const uint32_t *__restrict__ list, const uint32_t *__restrict__ flag,
                                                     const uint32_t *__restrict__ tcount, uint32_t max_gtiles,
                                                     uint64_t *__restrict__ key, uint32_t *__restrict__ cnt)
{
    __shared__ uint32_t s_c[GRP_NT / 64];
    const uint32_t b = list[blockIdx.y], tid = threadIdx.x, r = blockIdx.x * GRP_NT + tid;
    if (flag[b]) return;
    uint32_t *SA = sa + (size_t)b * nmax;
    const uint32_t v = r < n ? SA[r] : GRP_SAME;
    const bool head = !(v & GRP_SAME);
    const uint64_t bal = __ballot(head);
    if ((tid & 63) == 0) s_c[tid >> 6] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t gid = tcount[(size_t)b * max_gtiles + blockIdx.x];          // heads in the tiles before this one
    for (uint32_t w = 0; w < (tid >> 6); w++) gid += s_c[w];
    gid += (uint32_t)__popcll(bal & ((2ull << (tid & 63)) - 1ull));       // ... and up to this row: the row's group, counted from 1
    if (r < n) {
        SA[r] = v & ~(GRP_SAME | SA_CAND);
        key[(size_t)b * nmax + r] = ((uint64_t)gid << VAL_BITS) | (uint64_t)(v & ~(GRP_SAME | SA_CAND));
    }
    if (blockIdx.x == 0 && tid == 0) cnt[b] = n;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t sa_scratch_alloc(SaScratch &s, uint32_t nmax, uint32_t rows)
{
    s.nmax = nmax; s.rows = rows; s.max_tiles = (nmax + SA_TILE - 1) / SA_TILE;
    s.rs_tiles = (nmax + RS_TILE - 1) / RS_TILE;
    const size_t ne = (size_t)nmax * rows;
    size_t total = 0;
    auto A = [&](void **p, size_t bytes) -> hipError_t { total += bytes; return hipMalloc(p, bytes); };
    {   // keyA | keyB in one allocation: the bucket sorter uses it as [rows][buckets][FS_CAP]
        s.fs_kstride = ((size_t)1 << fs_bucket_log2(nmax)) * FS_CAP;
        const size_t words = s.fs_kstride * rows > 2 * ne ? s.fs_kstride * rows : 2 * ne;
        GLC_TRY(A((void **)&s.keyA, words * 8));
        s.keyB = s.keyA + ne;
    }
    GLC_TRY(A((void **)&s.fs_hist, (size_t)rows * 256 * 4));
    GLC_TRY(A((void **)&s.fs_tab, (size_t)rows * 256 * 8));
    GLC_TRY(A((void **)&s.fs_fill, (size_t)rows * FS_MAXNB * 4));
    GLC_TRY(A((void **)&s.fs_base, (size_t)rows * FS_MAXNB * 4));
    GLC_TRY(A((void **)&s.fs_flag, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_lcnt, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_redo[0], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_redo[1], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_keep[0], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_keep[1], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.ss_mask[0], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.ss_mask[1], (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_dup, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_zero, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.fs_nflag, 32));
    GLC_TRY(A((void **)&s.ss_list, (size_t)rows * 4 * 3));
    GLC_TRY(A((void **)&s.ss_split, (size_t)rows * FS_MAXNB * 8 * 3));    // words, then the first 8 text bytes of every splitter, then the next 8
    GLC_TRY(A((void **)&s.ss_flag, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.ss_cell, (size_t)rows * 4098 * 2));
    GLC_TRY(A((void **)&s.ss_l0, (size_t)rows * FS_MAXNB * 4));
    GLC_TRY(A((void **)&s.ss_long, (size_t)rows * FS_MAXNB * (SSL_PER_BUCKET + SSL_BIG_PER_BUCKET) * sizeof(uint2)));
    GLC_TRY(A((void **)&s.ss_long_count, 8));
    GLC_TRY(A((void **)&s.ss_gtile, (size_t)rows * ((nmax + GRP_NT - 1) / GRP_NT) * 4));
    GLC_TRY(A((void **)&s.ss_cnt2, (size_t)rows * 4));
    s.fs_wl_cap = nmax / 8 < 1024 ? 1024 : nmax / 8;
    GLC_TRY(A((void **)&s.fs_wl, (size_t)rows * s.fs_wl_cap * 16));
    GLC_TRY(A((void **)&s.fs_wlcnt, (size_t)rows * 4));
    // (posA/B, hdA/B, isa, sa, tile_hist, tile_state -- 24.6 MiB per 1 MiB block, used by the general sorter alone --
    //  are allocated by sa_general_reserve the first time a block gets that far)
    GLC_TRY(A((void **)&s.digit_base, (size_t)rows * RS_MAXPASS * SA_MAXRADIX * 4));
    GLC_TRY(A((void **)&s.ghist, (size_t)rows * RS_MAXPASS * SA_MAXRADIX * 4));
    GLC_TRY(A((void **)&s.ticket, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.cntA, (size_t)rows * 4)); GLC_TRY(A((void **)&s.cntB, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.d_max_cnt, 16));
    GLC_TRY(A((void **)&s.rl_flag, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.rl_cnt, (size_t)rows * 4));
    GLC_TRY(hipHostMalloc((void **)&s.h_max_cnt, 32, hipHostMallocDefault));
    s.bytes = total;
    return hipSuccess;
}

// the general sorter's own arrays (and s.sa, which the other tiers also write when the suffix array itself is the
// result asked for): allocated on first use -- a plan whose data never leaves the bucket / sample sorters (the 4 GiB
// benchmark workload: 3 plans x 1024 rows) does not carry 24.6 MiB per row for a tier it does not run
hipError_t sa_general_reserve(SaScratch &s, bool only_sa)
{
    const size_t ne = (size_t)s.nmax * s.rows;
    auto A = [&](void **p, size_t bytes) -> hipError_t {
        if (*p) return hipSuccess;
        s.bytes += bytes;
        return hipMalloc(p, bytes);
    };
    GLC_TRY(A((void **)&s.sa, ne * 4));
    if (only_sa) return hipSuccess;
    GLC_TRY(A((void **)&s.posA, ne * 4)); GLC_TRY(A((void **)&s.posB, ne * 4));
    GLC_TRY(A((void **)&s.isa, ne * 4));
    GLC_TRY(A((void **)&s.hdA, ne * 4)); GLC_TRY(A((void **)&s.hdB, ne * 4));
    if (!s.tile_hist) s.epoch = 255;                         // fresh look-back granules: force a clear first
    GLC_TRY(A((void **)&s.tile_hist, (size_t)s.rows * s.rs_tiles * SA_MAXRADIX * 4));
    GLC_TRY(A((void **)&s.tile_state, (size_t)s.rows * s.max_tiles * 8));
    return hipSuccess;
}

void sa_scratch_free(SaScratch &s)
{
    void *ps[] = {s.keyA, s.ss_mask[0], s.ss_mask[1], s.ss_long, s.ss_long_count, s.ss_gtile, s.ss_cnt2, s.ss_list, s.ss_split, s.ss_flag, s.ss_cell, s.ss_l0, s.fs_hist, s.fs_tab, s.fs_fill, s.fs_base, s.fs_flag, s.fs_lcnt, s.fs_redo[0], s.fs_redo[1], s.fs_keep[0], s.fs_keep[1], s.fs_dup, s.fs_zero, s.fs_nflag, s.per_info, s.per_list, s.per_ok, s.per_count, s.per_base, s.per_text, s.fs_wl, s.fs_wlcnt, s.posA, s.posB, s.hdA, s.hdB, s.isa, s.sa, s.tile_hist, s.digit_base, s.ghist,
                  s.tile_state, s.ticket, s.cntA, s.cntB, s.d_max_cnt, s.rl_flag, s.rl_cnt};
    for (void *p : ps) if (p) (void)hipFree(p);
    if (s.h_max_cnt) (void)hipHostFree(s.h_max_cnt);
    if (s.ev_flag) (void)hipEventDestroy(s.ev_flag);
    if (s.ev_fork) (void)hipEventDestroy(s.ev_fork);
    if (s.ev_join) (void)hipEventDestroy(s.ev_join);
    if (s.aux) { (void)hipStreamSynchronize(s.aux); (void)hipStreamDestroy(s.aux); }
    s = SaScratch();
}

// one LSD sort = prehist + digitbase + npass onesweep launches; result ends in `*cur`
static hipError_t radix_sort(hipStream_t st, uint64_t *&cur, uint64_t *&alt, const uint32_t *cnt, uint32_t nfixed,
                             const PassPlan &pp, uint32_t tiles, uint32_t nblk, SaScratch &s, double live_total,
                             const TextSrc *src = nullptr, bool profile = true)
{
    // `tiles` counts SA_TILE-word tiles (rank kernel); the radix kernels use RS_TILE
    const uint32_t rs_tiles = (tiles * (uint32_t)SA_TILE + RS_TILE - 1) / RS_TILE;
    GLC_TRY(hipMemsetAsync(s.ghist, 0, (size_t)nblk * RS_MAXPASS * SA_MAXRADIX * 4, st));
    // workgroups per block of the histogram pass: 32 where the blocks of a call fill the chip between them, up to 256 for a
    // call of a few (a lone block's second sort waited 112 us for 32 workgroups to read its 8 MB list)
    const uint32_t ph = nblk >= 8 ? 32u : (nblk >= 2 ? 128u : 256u), phg = rs_tiles < ph ? rs_tiles : ph;
    if (src)
        hipLaunchKernelGGL(k_rs_prehist<true>, dim3(phg, nblk), dim3(RS_NT), 0, st, cur, cnt, nfixed,
                           pp, s.ghist, s.nmax, *src);
    else
        hipLaunchKernelGGL(k_rs_prehist<false>, dim3(phg, nblk), dim3(RS_NT), 0, st, cur, cnt, nfixed,
                           pp, s.ghist, s.nmax, TextSrc{});
    hipLaunchKernelGGL(k_rs_digitbase, dim3(nblk, pp.npass), dim3(512), 0, st, s.ghist, s.digit_base);
    for (uint32_t p = 0; p < pp.npass; p++) {
        if (++s.epoch > 255) {                              // epoch tags wrapped: clear stale granules once
            GLC_TRY(hipMemsetAsync(s.tile_hist, 0, (size_t)s.rows * s.rs_tiles * SA_MAXRADIX * 4, st));
            s.epoch = 1;
        }
        GLC_TRY(hipMemsetAsync(s.ticket, 0, (size_t)nblk * 4, st));
        // profiled kernel = k_rs_onesweep<8,false> (16 algorithmic bytes per live suffix); the text-sourced
        // first pass of round 0 is a different kernel (1 R + 8 W) and is left out
        const int pi = (profile && s.prof && pp.bits[p] == 8 && !(p == 0 && src)) ? s.prof->begin(PROF_RS_ONESWEEP8, st) : -1;
        dim3 g(nblk, rs_tiles);
        if (pp.bits[p] == 8 && p == 0 && src)
            hipLaunchKernelGGL((k_rs_onesweep<8, true>), g, dim3(RS_NT), 0, st, cur, alt, cnt, nfixed, pp.shift[p],
                               s.tile_hist, s.ticket, s.epoch, s.digit_base + p * SA_MAXRADIX,
                               (uint32_t)(RS_MAXPASS * SA_MAXRADIX), s.nmax, s.rs_tiles, s.d_max_cnt + 2, *src);
        else if (pp.bits[p] == 8)
            hipLaunchKernelGGL(k_rs_onesweep<8>, g, dim3(RS_NT), 0, st, cur, alt, cnt, nfixed, pp.shift[p], s.tile_hist,
                               s.ticket, s.epoch, s.digit_base + p * SA_MAXRADIX, (uint32_t)(RS_MAXPASS * SA_MAXRADIX),
                               s.nmax, s.rs_tiles, s.d_max_cnt + 2);
        else
            hipLaunchKernelGGL(k_rs_onesweep<9>, g, dim3(RS_NT), 0, st, cur, alt, cnt, nfixed, pp.shift[p], s.tile_hist,
                               s.ticket, s.epoch, s.digit_base + p * SA_MAXRADIX, (uint32_t)(RS_MAXPASS * SA_MAXRADIX),
                               s.nmax, s.rs_tiles, s.d_max_cnt + 2);
        if (pi >= 0) s.prof->end(pi, live_total, st);
        uint64_t *x = cur; cur = alt; alt = x;
    }
    return hipGetLastError();
}

// order every group of the unresolved list by its new key bits: tile-local for the blocks whose groups
// all fit, global radix sort for the rest.  Result in `cur` either way.
static hipError_t refine_sort(hipStream_t st, uint64_t *&cur, uint64_t *&alt, const uint32_t *cnt, const PassPlan &pp,
                              uint32_t gshift, uint32_t maxc, uint32_t tiles, uint32_t nblk, SaScratch &s,
                              double live_total)
{
    GLC_TRY(hipMemsetAsync(s.rl_flag, 0, (size_t)nblk * 4, st));
    hipLaunchKernelGGL(k_refine_local, dim3((maxc + RL_T - 1) / RL_T, nblk), dim3(RL_NT), 0, st, cur, alt, cnt, s.nmax,
                       gshift, s.rl_flag);
    GLC_TRY(hipMemsetAsync(s.d_max_cnt + 3, 0, 4, st));
    hipLaunchKernelGGL(k_refine_counts, dim3((nblk + 255) / 256), dim3(256), 0, st, cnt, s.rl_flag, s.rl_cnt, nblk,
                       s.d_max_cnt + 3);
    GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 3, s.d_max_cnt + 3, 4, hipMemcpyDeviceToHost, st));
    GLC_TRY(hipStreamSynchronize(st));                         // one more host round trip per refinement round
    (void)live_total;
    const uint32_t flagged_live = s.h_max_cnt[3];
    if (flagged_live == 0) { uint64_t *x = cur; cur = alt; alt = x; return hipSuccess; }   // every group was local
    // npass is odd: the radix sort leaves its result in what is `alt` now -- where the tile-local kernel wrote
    return radix_sort(st, cur, alt, s.rl_cnt, 0, pp, tiles, nblk, s, (double)flagged_live);
}

// the general sorter; cnt0 (optional) = per-block element counts: n for the blocks to sort, 0 for the others
// resume_depth != 0: s.keyA already holds, for the blocks of cnt0, the words [group : 44 | suffix : 20] of an order that is
// exact for the first resume_depth symbols (k_grp_keys): no sort from the text, prefix doubling from that depth on
static hipError_t sa_build_general(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                                   SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, int *rounds_out,
                                   const uint32_t *cnt0, uint32_t nsorted, uint32_t resume_depth = 0)
{
    GLC_TRY(sa_general_reserve(s, false));
    uint32_t tiles = (n + SA_TILE - 1) / SA_TILE;
    uint64_t *cur = s.keyA, *alt = s.keyB;
    double live_total = (double)n * nsorted;
    GLC_TRY(hipMemsetAsync(s.d_max_cnt, 0, 16, st));          // [2] doubles as the device error word of the sort
    if (!resume_depth) {   // 41 key bits at [20, 61): 8+8+8+8+9
        PassPlan pp = {5, {VAL_BITS, VAL_BITS + 8, VAL_BITS + 16, VAL_BITS + 24, VAL_BITS + 32}, {8, 8, 8, 8, 9}};
        const TextSrc src{text, text_stride, n};               // pass 0 and the histograms read the text itself
        GLC_TRY(radix_sort(st, cur, alt, cnt0, n, pp, tiles, nblk, s, live_total, &src));
    }

    const uint32_t *cnt_cur = cnt0;
    uint32_t *cnt_next = s.cntA;
    uint32_t *pos_cur = nullptr, *pos_next = s.posA, *pos_spare = s.posB;
    uint32_t *hd_cur = nullptr, *hd_next = s.hdA, *hd_spare = s.hdB;
    // `depth` = symbols the current order is exact for; text refinement adds 3 per round,
    // prefix doubling doubles it.  Text refinement first: data whose suffixes separate
    // within ~11 symbols (i.i.d. bytes, float mantissas) never builds the rank array.
    int mode = (s.force_isa || resume_depth) ? MODE_ISA : MODE_TEXT;
    uint32_t depth = resume_depth ? resume_depth : 5, live = n, text_rounds = 0, isa_rounds = 0;
    int rounds = 0;
    if (mode == MODE_ISA) {
        // ranks are needed from the first refinement on: k_sa_rank<true> writes them in MODE_ISA
    }
    for (;;) {
        dim3 g(tiles, nblk);
        GLC_TRY(hipMemsetAsync(s.d_max_cnt, 0, 8, st));
        hipLaunchKernelGGL(k_rank_pre, g, dim3(SA_THREADS), 0, st, cur, cnt_cur, live, (unsigned long long *)s.tile_state,
                           s.nmax, s.max_tiles);
        hipLaunchKernelGGL(k_rank_scan, dim3(nblk), dim3(256), 0, st, (unsigned long long *)s.tile_state, cnt_cur, live,
                           s.max_tiles, cnt_next, s.d_max_cnt);
        if (!pos_cur)
            hipLaunchKernelGGL(k_sa_rank1<true>, g, dim3(SA_THREADS), 0, st, cur, pos_cur, cnt_cur, live,
                               (unsigned long long *)s.tile_state, s.isa, s.sa, alt, pos_next, hd_next, cnt_next,
                               s.d_max_cnt, s.nmax, s.max_tiles, mode, text, text_stride, n, depth, bwt_out, bwt_stride,
                               d_index, s.d_max_cnt + 2);
        else
            hipLaunchKernelGGL(k_sa_rank1<false>, g, dim3(SA_THREADS), 0, st, cur, pos_cur, cnt_cur, live,
                               (unsigned long long *)s.tile_state, s.isa, s.sa, alt, pos_next, hd_next, cnt_next,
                               s.d_max_cnt, s.nmax, s.max_tiles, mode, text, text_stride, n, depth, bwt_out, bwt_stride,
                               d_index, s.d_max_cnt + 2);
        GLC_TRY(hipGetLastError());
        GLC_TRY(hipMemcpyAsync(s.h_max_cnt, s.d_max_cnt, 16, hipMemcpyDeviceToHost, st));
        GLC_TRY(hipStreamSynchronize(st));
        rounds++;
        const uint32_t maxc = s.h_max_cnt[0];
        live_total = (double)s.h_max_cnt[1];
        if (s.h_max_cnt[2]) return hipErrorUnknown;               // a look-back spin hit its bound
        if (maxc == 0) break;
        if (depth >= 2u * n + 16u) return hipErrorUnknown;        // cannot happen: depth >= n resolves everything
        // next round works on the compacted list that k_sa_rank<true> wrote into `alt`
        { uint64_t *x = cur; cur = alt; alt = x; }
        cnt_cur = cnt_next; cnt_next = (cnt_next == s.cntA) ? s.cntB : s.cntA;
        if (pos_cur == nullptr) { pos_cur = pos_next; pos_next = pos_spare; }
        else { uint32_t *x = pos_cur; pos_cur = pos_next; pos_next = x; }
        if (hd_cur == nullptr) { hd_cur = hd_next; hd_next = hd_spare; }
        else { uint32_t *x = hd_cur; hd_cur = hd_next; hd_next = x; }
        live = maxc;
        tiles = (maxc + SA_TILE - 1) / SA_TILE;
        const uint32_t fill_blocks = (maxc + SA_THREADS * 4 - 1) / (SA_THREADS * 4);
        if (mode == MODE_TEXT) {
            // deep data (text, long repeats): 3 symbols per round is too slow -> prefix doubling
            const bool deep = (rounds == 1 && live_total > 0.25 * (double)n * nsorted) || text_rounds >= 3;
            if (deep) {
                hipLaunchKernelGGL(k_isa_init, dim3((n + 255) / 256, nblk), dim3(256), 0, st, s.sa, s.isa, n, s.nmax, cnt0);
                hipLaunchKernelGGL(k_isa_fix, dim3(fill_blocks, nblk), dim3(SA_THREADS), 0, st, cur, cnt_cur, hd_cur,
                                   s.isa, s.nmax);
                mode = MODE_ISA;
            }
        }
        if (mode == MODE_TEXT) {
            // 44 key bits at [20, 64): 8+9+9+9+9
            PassPlan pp = {5, {VAL_BITS, VAL_BITS + 8, VAL_BITS + 17, VAL_BITS + 26, VAL_BITS + 35}, {8, 9, 9, 9, 9}};
            GLC_TRY(refine_sort(st, cur, alt, cnt_cur, pp, TXT_GRP_SHIFT, maxc, tiles, nblk, s, live_total));
            depth += 3;
            text_rounds++;
        } else {
            // chain groups (one residue class of a periodic stretch each) get their final order this round; worth its five light
            // passes over the list only where a lot is still live (GLC_CHAIN_MIN: tests lower it, 0 switches it off)
            static const long chain_min = getenv("GLC_CHAIN_MIN") ? atol(getenv("GLC_CHAIN_MIN")) : 16384;
            // ... in the FIRST doubling round (a periodic stretch over a large alphabet is all chains at once), and again in rounds
            // 2 and 4 where more than half of everything is still live: over a small alphabet the period's 5-grams repeat, a group of
            // the first round holds several residue classes and only becomes chains once the depth tells them apart (two periodic
            // halves over {0, 1}: 50.1 ms per 32 blocks with the first round alone, 15.6 with all three; blocks with something deep
            // INSIDE have a few per cent live and are spared the later attempts' launches: 9.2 against 9.6 ms per 64)
            static const long chain_every = getenv("GLC_CHAIN_ROUNDS") ? atol(getenv("GLC_CHAIN_ROUNDS")) : 0x15;   // bit r: try in doubling round r
            const bool mostly_live = live_total >= 0.5 * (double)n * nsorted;
            const bool chains = chain_min > 0 && live_total >= (double)chain_min && pos_cur != nullptr && isa_rounds < 32 &&
                                ((chain_every >> isa_rounds) & 1) && (isa_rounds == 0 || mostly_live);
            isa_rounds++;
            uint2 *rec = reinterpret_cast<uint2 *>(alt);      // (the spare word array: free until the sort below)
            if (chains) {
                dim3 cg(fill_blocks, nblk), ct(SA_THREADS);
                hipLaunchKernelGGL(k_chain_init, cg, ct, 0, st, cur, cnt_cur, rec, s.hdA, s.nmax);
                hipLaunchKernelGGL(k_chain_minmax, cg, ct, 0, st, cur, cnt_cur, rec, s.nmax);
                hipLaunchKernelGGL(k_chain_decide, cg, ct, 0, st, cur, pos_cur, cnt_cur, rec, s.hdA, s.nmax);
                // (the verification's cache, n / 16 halfwords at the head of every block's part of pos_next -- which the NEXT rank pass writes)
                GLC_TRY(hipMemset2DAsync(pos_next, (size_t)s.nmax * 4, 0, ((size_t)n / 16 + 1) * 2, nblk, st));
                hipLaunchKernelGGL(k_chain_verify, cg, ct, 0, st, cur, cnt_cur, rec, s.hdA, s.nmax, text, text_stride,
                                   reinterpret_cast<uint16_t *>(pos_next));
                hipLaunchKernelGGL(k_chain_dir, cg, ct, 0, st, cur, cnt_cur, rec, s.hdA, s.nmax, text, text_stride, n, depth);
            }
            hipLaunchKernelGGL(k_sa_fill_rank2, dim3(fill_blocks, nblk), dim3(SA_THREADS), 0, st, cur, cnt_cur, s.isa,
                               n, depth, s.nmax, rec, chains ? s.hdA : (const uint32_t *)nullptr);
            // 42 key bits at [20, 62): 8+8+8+9+9
            PassPlan pp = {5, {VAL_BITS, VAL_BITS + 8, VAL_BITS + 16, VAL_BITS + 24, VAL_BITS + 33}, {8, 8, 8, 9, 9}};
            GLC_TRY(refine_sort(st, cur, alt, cnt_cur, pp, R1_SHIFT, maxc, tiles, nblk, s, live_total));
            depth *= 2;
        }
    }
    if (rounds_out) *rounds_out = rounds;
    return hipSuccess;
}

// Two-phase form, so that a caller can queue the stages that FOLLOW the sort before the host waits for the sorter's
// one readback (cudpp_api.cpp queues MTF + Huffman speculatively: the GPU never idles behind the wait, and the
// rare batch with flagged blocks re-runs them):
//   sa_build_begin   enqueues the bucket sorter and the readback of the flagged-block count (an event marks it);
//                    with s.sorter != 0 it runs the whole general sort instead (which blocks per round).
//   sa_build_finish  waits for that event only; if blocks were flagged, enqueues the general sorter for them.
//                    *nflagged > 0 tells the caller that bwt_out / d_index of those blocks were rewritten.
hipError_t sa_build_begin(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                          SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index)
{
    if (n == 0 || n > s.nmax || n > MAX_BLOCK_ELEMS || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    s.last_flagged = nblk;
    s.last_general = nblk;
    s.pending = false;
    if (s.sorter == 1 || s.sorter == 2) {
        const bool isa = s.force_isa;
        s.force_isa = isa || s.sorter == 2;
        const hipError_t e = sa_build_general(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, nullptr,
                                              nullptr, nblk);
        s.force_isa = isa;
        return e;
    }
    // bucket sorter first; the suffix array itself is only written when it is the result asked for
    if (!bwt_out) GLC_TRY(sa_general_reserve(s, true));
    s.skip_tier1 = sa_skips_tier1(s, nblk);                  // the caller knows its data is text-like (sorter 4), or the plan's last calls say so
    GLC_TRY(fs_build(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, bwt_out ? nullptr : s.sa));
    // (the flagged-block count is in s.h_max_cnt[4] when the pass's last kernel is through: it writes it there itself --
    //  k_fs_finish / k_fs_ties -- where a copy command behind the pass was one more ~5 us link in a single call's chain)
    if (!s.ev_flag) GLC_TRY(hipEventCreateWithFlags(&s.ev_flag, hipEventDisableTiming));
    GLC_TRY(hipEventRecord(s.ev_flag, st));
    s.pending = true;
    return hipSuccess;
}

static hipError_t sa_build_finish_tiers(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                                        SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *nflagged);

hipError_t sa_build_finish(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                           SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *nflagged)
{
    s.partial_used = false;
    const hipError_t e = sa_build_finish_tiers(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, nflagged);
    // whatever the tiers did, the side stream's stages (if any were queued) are joined into st here
    if (s.partial_used) { const hipError_t j = hipStreamWaitEvent(st, s.ev_join, 0); if (e == hipSuccess && j != hipSuccess) return j; }
    return e;
}

static hipError_t sa_build_finish_tiers(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                                        SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, uint32_t *nflagged)
{
    if (nflagged) *nflagged = 0;
    if (!s.pending) return hipSuccess;                       // general sorter only: nothing was deferred
    s.pending = false;
    GLC_TRY(hipEventSynchronize(s.ev_flag));
    uint32_t nflag = s.h_max_cnt[4];
    // the streak of calls whose every block the probe called text-like (k_fs_finish counts the others, in skipped calls too)
    s.last_skipped = s.skip_tier1;
    s.textlike_streak = (nflag == nblk && s.h_max_cnt[5] == 0) ? (s.textlike_streak < 1000u ? s.textlike_streak + 1 : 1000u) : 0u;
    s.last_flagged = nflag;
    s.last_general = 0;
    s.last_retried = 0;
    s.last_resumed = 0;
    if (nflag == 0) return hipSuccess;
    if (nflagged) *nflagged = nflag;
    if (s.sorter != 3) {
        // second tier: string sample sort of the flagged blocks; what it gives up on (very deep repeats) is counted again
        GLC_TRY(ss_build(st, text, text_stride, n, nflag, s, bwt_out, bwt_stride, d_index, bwt_out ? nullptr : s.sa));
        GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 5, s.fs_nflag + 1, 4, hipMemcpyDeviceToHost, st));
        GLC_TRY(hipEventRecord(s.ev_flag, st));
        GLC_TRY(hipEventSynchronize(s.ev_flag));
        uint32_t left = s.h_max_cnt[5];
        if (left) {
            // some blocks were given up on.  Those whose only trouble was a bucket past its slot get ONE more attempt with
            // other samples (a bucket of 4033-4200 words where 4032 fit: ~1 % of log-style blocks; the general sorter
            // costs ten times the sample sorter, and its rounds hold the host)
            if (s.stage_partial) GLC_TRY(ss_split_masks(st, nblk, s));
            GLC_TRY(ss_retry_prepare(st, nflag, s));
            GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 6, s.fs_nflag + 2, 4, hipMemcpyDeviceToHost, st));
            GLC_TRY(hipEventRecord(s.ev_flag, st));
            GLC_TRY(hipEventSynchronize(s.ev_flag));
            const uint32_t again = s.h_max_cnt[6];
            s.last_retried = again;
            if (again && s.stage_partial && left < nflag) {
                // the stages behind the sort for the blocks the first attempt finished, on a side stream BESIDE the second attempt
                // (ss_mask was written before ss_retry_prepare touched anything: see above)
                if (!s.aux) {
                    // (lowest priority: the side stream's kernels fill the chip, the second attempt's small launches on `st` are a
                    //  chain of latencies -- with equal priorities its bucketing pass took 614 us beside k_mtf_encode instead of 27)
                    int least = 0, greatest = 0;
                    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                    GLC_TRY(hipStreamCreateWithPriority(&s.aux, hipStreamNonBlocking, least));
                    GLC_TRY(hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming));
                    GLC_TRY(hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming));
                }
                GLC_TRY(hipEventRecord(s.ev_fork, st));
                GLC_TRY(hipStreamWaitEvent(s.aux, s.ev_fork, 0));
                GLC_TRY(s.stage_partial(s.aux, s.ss_mask[0]));
                GLC_TRY(hipEventRecord(s.ev_join, s.aux));
                s.partial_used = true;
            }
            if (again) {
                s.h_max_cnt[7] = left - again;                 // the others stay given up on; the second attempt adds its own
                GLC_TRY(hipMemcpyAsync(s.fs_nflag + 1, s.h_max_cnt + 7, 4, hipMemcpyHostToDevice, st));
                GLC_TRY(ss_build(st, text, text_stride, n, again, s, bwt_out, bwt_stride, d_index, bwt_out ? nullptr : s.sa, 1));
                GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 5, s.fs_nflag + 1, 4, hipMemcpyDeviceToHost, st));
                GLC_TRY(hipEventRecord(s.ev_flag, st));
                GLC_TRY(hipEventSynchronize(s.ev_flag));
                left = s.h_max_cnt[5];
            }
        }
        s.last_general = left;                                 // what the sample sorter (both attempts) gave up on
        s.last_resumed = 0;
        s.last_periodic = 0;
        if (left && s.periodic && bwt_out && d_index && n >= 16 * PER_PMAX && s.nmax >= PER_NU) {
            // Blocks that are ONE periodic stretch (a page repeated, a short pattern, one byte up to a different last one):
            // every suffix ties with the one a period further on for nearly the whole block -- ~18 doubling rounds over a
            // million live suffixes.  Their suffix array has a closed form over the sorted rotations of the period and the
            // few suffixes around the break (bwt_periodic.hip); only those -- a text of <= 7 p + 2 bytes per block -- are
            // sorted, by the general sorter, whatever the block's size.
            GLC_TRY(per_reserve(s));
            GLC_TRY(per_detect(st, text, text_stride, n, nflag, s));
            GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 6, s.per_count, 8, hipMemcpyDeviceToHost, st));   // {blocks taken, longest text of representatives}
            GLC_TRY(hipEventRecord(s.ev_flag, st));
            GLC_TRY(hipEventSynchronize(s.ev_flag));
            const uint32_t nper = s.h_max_cnt[6] < PER_TAKE ? s.h_max_cnt[6] : PER_TAKE;   // (k_per_detect takes no more than its scratch holds)
            if (nper) {
                // (the longest text of representatives among the taken blocks -- they are sorted as one batch of equal length -- comes
                //  with the count: round 5 copied every block's info to the host and waited a second time for it: ADVICE r5)
                uint32_t nu = s.h_max_cnt[7] < 64 ? 64 : s.h_max_cnt[7];
                nu = (nu + 15u) & ~15u;
                if (nu > PER_NU) nu = PER_NU;
                GLC_TRY(per_text(st, text, text_stride, n, nper, nu, s));
                GLC_TRY(sa_build_general(st, s.per_text, PER_NU, nu, nper, s, nullptr, 0, nullptr, nullptr, nullptr, nper));
                GLC_TRY(per_expand(st, text, text_stride, n, nper, nu, s, bwt_out, bwt_stride, d_index));
                GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 6, s.per_count + 2, 4, hipMemcpyDeviceToHost, st));
                GLC_TRY(hipEventRecord(s.ev_flag, st));
                GLC_TRY(hipEventSynchronize(s.ev_flag));
                const uint32_t done = s.h_max_cnt[6];
                s.last_periodic = done;
                left -= done < left ? done : left;
                s.h_max_cnt[7] = left;                         // the device-side count of blocks still given up on follows
                GLC_TRY(hipMemcpyAsync(s.fs_nflag + 1, s.h_max_cnt + 7, 4, hipMemcpyHostToDevice, st));
            }
        }
        if (left && s.resume_min) {
            // Blocks whose only trouble was a repeat deeper than the cap (zero pages, a duplicated region, long periodic
            // stretches inside otherwise ordinary data): the sample sorter once more, in its TOLERANT form and writing the
            // suffix array -- everything it can order it orders, suffixes that agree in more than the cap stay as they come
            // -- then prefix doubling from that depth on, over the groups of rows that still share the cap (a few thousand
            // suffixes of such a block for ~11 rounds, where the general sorter from scratch takes a million through ~20).
            // counted first, without touching anything: below resume_min the blocks go on as they are (their flags and fills
            // stay what glcPlanDebugSortFlags / BucketFill report)
            GLC_TRY(ss_retry_prepare(st, nflag, s, 2, true));
            GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 6, s.fs_nflag + 2, 4, hipMemcpyDeviceToHost, st));
            GLC_TRY(hipEventRecord(s.ev_flag, st));
            GLC_TRY(hipEventSynchronize(s.ev_flag));
            const uint32_t deep2 = s.h_max_cnt[6];
            // (a few blocks: the extra pass and its launches cost what the shorter doubling saves -- 2.6 against 2.2-2.7 ms for
            //  one block, 3.5 against 4.4 for eight)
            if (deep2 >= s.resume_min) {
                GLC_TRY(ss_retry_prepare(st, nflag, s, 2));    // listed, flags and fills cleared for the tolerant pass
                GLC_TRY(sa_general_reserve(s, false));
                s.h_max_cnt[7] = left - deep2;                 // the others stay given up on; this attempt adds its own
                GLC_TRY(hipMemcpyAsync(s.fs_nflag + 1, s.h_max_cnt + 7, 4, hipMemcpyHostToDevice, st));
                GLC_TRY(ss_build(st, text, text_stride, n, deep2, s, nullptr, 0, nullptr, s.sa, 2));
                GLC_TRY(hipMemcpyAsync(s.h_max_cnt + 5, s.fs_nflag + 1, 4, hipMemcpyDeviceToHost, st));
                const uint32_t *list3 = s.ss_list + 2 * (size_t)s.rows;
                const uint32_t gt = (n + GRP_NT - 1) / GRP_NT, max_gt = (s.nmax + GRP_NT - 1) / GRP_NT;
                GLC_TRY(hipMemsetAsync(s.ss_cnt2, 0, (size_t)s.rows * 4, st));
                hipLaunchKernelGGL(k_grp_flags, dim3(gt, deep2), dim3(GRP_NT), 0, st, text, text_stride, n, s.sa, s.nmax, list3,
                                   s.ss_flag, s.ss_gtile, max_gt, SS_TOL_CAP);
                hipLaunchKernelGGL(k_grp_scan, dim3(deep2), dim3(1024), 0, st, s.ss_gtile, max_gt, gt, list3, s.ss_flag);
                hipLaunchKernelGGL(k_grp_keys, dim3(gt, deep2), dim3(GRP_NT), 0, st, n, s.sa, s.nmax, list3, s.ss_flag,
                                   s.ss_gtile, max_gt, s.keyA, s.ss_cnt2);
                GLC_TRY(hipEventRecord(s.ev_flag, st));
                GLC_TRY(hipEventSynchronize(s.ev_flag));
                const uint32_t left2 = s.h_max_cnt[5];          // = left - deep2 + what the tolerant form gave up on (a bucket past its slot)
                const uint32_t resumed = left - left2;
                if (resumed)
                    GLC_TRY(sa_build_general(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, nullptr, s.ss_cnt2,
                                             resumed, SS_TOL_CAP));
                s.last_resumed = resumed;
                left = left2;
            }
        }
        nflag = left;
        if (nflag == 0) return hipSuccess;
    } else s.last_general = nflag;
    return sa_build_general(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, nullptr, s.fs_lcnt, nflag);
}

hipError_t sa_build(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nblk,
                    SaScratch &s, uint8_t *bwt_out, size_t bwt_stride, int *d_index, int *rounds_out)
{
    if (rounds_out) *rounds_out = 0;
    GLC_TRY(sa_build_begin(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index));
    return sa_build_finish(st, text, text_stride, n, nblk, s, bwt_out, bwt_stride, d_index, nullptr);
}

// per-block exclusive scan of [tile][512] histograms (used by the decoder's LF construction)
hipError_t tile_hist_scan9(hipStream_t st, uint32_t *tile_hist, uint32_t count, uint32_t *digit_base,
                           uint32_t max_tiles, uint32_t nblk, uint32_t tile_elems)
{
    hipLaunchKernelGGL(k_rs_scan<9>, dim3(nblk), dim3(512), 0, st, tile_hist, (const uint32_t *)nullptr, count,
                       digit_base, max_tiles, tile_elems);
    return hipGetLastError();
}

hipError_t sa_export(hipStream_t st, const uint32_t *sa, uint32_t n, uint32_t *out)
{
    hipLaunchKernelGGL(k_sa_export, dim3((n + 256) / 256), dim3(256), 0, st, sa, n, out);
    return hipGetLastError();
}

} // namespace glc
#ifdef GLC_DEBUG_CAND
extern "C" int glcDebugCand(unsigned int *out64, int reset)
{
    static unsigned int z[64];
    if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(glc::g_cand_dbg), 256) != hipSuccess) return 0;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(glc::g_cand_dbg), z, 256) != hipSuccess) return 0;
    return 1;
}
#endif
