// huff_tree.h -- the reference's Huffman tree shape, built by one wave64.
// Shared by the encoder (huffman.hip) and the decoder (decode.hip): both must
// derive the identical tree from the 256-bin histogram (+EOF with count 1).
//
// Restates huffman_build_tree_kernel's merge loop (cudpp-inpar/src/cudpp/kernel/
// compress_kernel.cuh:2306-2392) with FindMinimumCount's order (cta/compress_cta.cuh:
// 550-571: lowest count, then lowest level, then lowest slot) with two sorted queues -- the
// leaves, ranked once, and the composites in creation order -- whose merged order is consumed
// in batches of pairs by one wave (see huff_tree_build; GLC_HUFF_SERIAL keeps the one-merge-
// at-a-time form with the queues in registers, for A/B).
#pragma once
#include "glc_device.h"

namespace glc {

constexpr int      HUFF_NODES = 2 * 257 - 1;          // 513
constexpr uint64_t HUFF_KEY_NONE = ~0ull;

struct HuffTreeLds {
    uint32_t key[320], sorted[320], rank[320];         // leaf keys (count << 9 | slot) by slot and by rank (nl <= 257)
    uint32_t ones;                                     // leaves of count 1 besides EOF
    int16_t  value[HUFF_NODES];                        // symbol, -1 for a composite node
    int16_t  left[HUFF_NODES], right[HUFF_NODES], parent[HUFF_NODES];
    int      nl, head;
    // the batched merge loop (huff_tree_build): the composite queue in creation order, the children of every composite,
    // the two windows' order keys and the elements of a batch in merged order
    uint32_t qa[264], qb[264], kids[264];              // count << 5 | level, slot << 16 | node; left | right << 16
    unsigned long long lw[64], cw[64];
    uint2    pr[128];
};

// wave-wide minimum of a 64-bit key, broadcast to every lane: inclusive min-scan on DPP (row_shr within
// rows of 16, row_bcast:15 / :31 across rows; lanes without a source see the identity ~0), lane 63 holds
// the result.  (The xor-shuffle butterfly it replaces is 12 ds_bpermute round trips per reduction, and
// the tree build does up to 512 reductions back to back.)
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k)
{
#define GLC_MIN_STEP(ctrl, rowmask)                                                                         \
    {                                                                                                       \
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)k, ctrl, rowmask, 0xf, false);          \
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)(uint32_t)(k >> 32), ctrl, rowmask, 0xf, false);  \
        const uint64_t other = ((uint64_t)hi << 32) | lo;                                                   \
        k = other < k ? other : k;                                                                          \
    }
    GLC_MIN_STEP(0x111, 0xf) GLC_MIN_STEP(0x112, 0xf) GLC_MIN_STEP(0x114, 0xf) GLC_MIN_STEP(0x118, 0xf)
    GLC_MIN_STEP(0x142, 0xa) GLC_MIN_STEP(0x143, 0xc)
#undef GLC_MIN_STEP
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), 63);
    return ((uint64_t)hi << 32) | lo;
}

// wave-wide minimum of a 32-bit value, broadcast to every lane (inclusive min-scan on DPP, lane 63 holds the result)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t k)
{
#define GLC_MIN32_STEP(ctrl, rowmask)                                                                            \
    { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k, ctrl, rowmask, 0xf, false); k = o < k ? o : k; }
    GLC_MIN32_STEP(0x111, 0xf) GLC_MIN32_STEP(0x112, 0xf) GLC_MIN32_STEP(0x114, 0xf) GLC_MIN32_STEP(0x118, 0xf)
    GLC_MIN32_STEP(0x142, 0xa) GLC_MIN32_STEP(0x143, 0xc)
#undef GLC_MIN32_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)k, 63);
}

// Called by ALL NT threads of the workgroup (contains barriers).  hist257[256] must already hold the EOF count 1.
//
// The reference finds the two minima by (count, level, slot) among ALL live nodes for every merge (257-slot scans,
// compress_cta.cuh:550-571).  A wave-wide arg-min over register-held candidates does the same in ~1100 cycles per merge
// -- 24 dependent DPP steps, and a wave on its own issues one instruction every ~4.5 cycles -- 0.11 ms per tree with
// the rest of the CU idle.  Here the same order comes out of two sorted queues and scalar code:
//   * leaves, ranked once by (count, slot) -- their level is 0, so this IS their order among themselves (the ranking
//     is the one step every thread of the workgroup takes part in: a leaf per thread counts the keys below its own);
//   * composites in creation order.  Minima leave in non-decreasing order of count, so a new composite's count is
//     >= that of every composite made before it: it belongs at the TAIL of the composite queue, except among the
//     composites of the SAME count at the tail, where (level, slot) decides -- an insertion among those few (rare:
//     ties between the counts of composites; exact, not a heuristic).
// The minimum of all live nodes is the smaller of the two queue heads; a leaf and a composite never tie (level 0
// against level >= 1), so that is ONE scalar compare of count << 5 | level.  Both queues live in the registers of
// wave 0 and are read / written with v_readlane / v_writelane: a merge is ~70 scalar instructions and a few lane
// reads -- no LDS round trip, no cross-lane reduction.
//
// Node numbering (any numbering gives the same codes; the decoder stores whatever ids it is given): leaves 0 .. nl-1 in
// ascending symbol order -- the reference's slots, which is what the tie-break compares -- composite k is node nl + k,
// the root is 2 nl - 2.  A composite takes the SLOT (tie-break rank) of its first minimum, which becomes its LEFT
// child; the second minimum is the RIGHT child (compress_kernel.cuh:2344-2385).
template <int NT>
__device__ __forceinline__ void huff_tree_build(HuffTreeLds &T, const uint32_t *hist257, unsigned tid)
{
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    const uint32_t l = tid & 63u;
    const bool wave0 = tid < 64;
    // leaves: present symbols in ascending order -> slots 0..nl-1 (compress_kernel.cuh:2310-2321); key = count << 9 | slot
    if (wave0) {
        uint32_t nl = 0;
        for (int r = 0; r < 5; r++) {
            const uint32_t sym = r * 64 + l;
            const uint32_t c = sym < 257 ? hist257[sym] : 0u;
            const uint64_t bal = __ballot(c > 0);
            if (c > 0) {
                const uint32_t slot = nl + mbcnt(bal);
                T.value[slot] = (int16_t)sym; T.left[slot] = -1; T.right[slot] = -1;
                T.key[slot] = (c << 9) | slot;
            }
            nl += (uint32_t)__popcll(bal);
        }
        if (l == 0) { T.nl = (int)nl; T.ones = 0; }
    }
    for (uint32_t i = tid; i < 320; i += NT) T.rank[i] = 0;
    __syncthreads();
    // rank of every leaf among the leaves.  The EOF leaf (count 1, the last slot) is left out of the loops: it sorts
    // behind every other leaf of count 1 and ahead of everything else.
    {
        constexpr uint32_t P = NT / 256;                       // threads per leaf
        const uint32_t nlm = (uint32_t)T.nl - 1u;
        const uint32_t i = tid / P, part = tid % P;
        if (i < nlm) {
            const uint32_t ki = T.key[i];
            const uint32_t per = (nlm + P - 1) / P, j0 = part * per, j1 = j0 + per < nlm ? j0 + per : nlm;
            uint32_t cnt = 0;
            for (uint32_t j = j0; j < j1; j++) cnt += T.key[j] < ki ? 1u : 0u;
            if (part == 0) {
                cnt += (ki >> 9) >= 2u ? 1u : 0u;
                if ((ki >> 9) == 1u) atomicAdd(&T.ones, 1u);
            }
            if (P == 1) T.rank[i] = cnt; else atomicAdd(&T.rank[i], cnt);
        }
    }
    __syncthreads();
    {
        const uint32_t nlm = (uint32_t)T.nl - 1u;
        if (tid < nlm) T.sorted[T.rank[tid]] = T.key[tid];
        if (tid == 0) T.sorted[T.ones] = T.key[nlm];
    }
    __syncthreads();
    if (!wave0) return;

    const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane(T.nl);
#ifndef GLC_HUFF_SERIAL
    // ---- the merges, in BATCHES.  The reference takes the two minima by (count, level, slot) 256 times, one after the
    // other; as two sorted queues and scalar code that is ~680 cycles per merge for a wave on its own, 73 us per tree --
    // a third of a single cudppCompress call.  But merges are far from all dependent on each other: with x0 <= x1 <= ...
    // the live nodes in order (leaves and composites merged), the first new composite is c0 = x0 + x1, every composite
    // made after it is no smaller, and composites only ever join the END of their queue -- so ALL the pairs (x2i, x2i+1)
    // with x2i+1 < c0 are merges the serial loop would make, in that order, whatever they produce.  A batch: the heads
    // of both queues (64 each) ranked in merged order by two binary searches, the elements below the bound paired up,
    // one lane per pair.  A Zipf block's MTF histogram takes 11 batches for its 256 merges (55, 65, 52, 34, ... pairs).
    // The queue must stay sorted by (count, level, slot): new composites come out in non-decreasing COUNT; where
    // counts tie and level or slot do not follow, the live entries are ranked again (rare).
    const uint32_t nm = nl - 1u;                               // merges
    uint32_t lh = 0, ch = 0, k = 0;                            // leaves taken, composites taken, composites made (uniform)
    auto leaf_ord = [](uint32_t key) -> unsigned long long { return ((unsigned long long)(key >> 9) << 14) | (key & 511u); };
    auto comp_ord = [](uint32_t A, uint32_t B) -> unsigned long long {
        return ((unsigned long long)(A >> 5) << 14) | ((unsigned long long)(A & 31u) << 9) | (B >> 16);
    };
    constexpr unsigned long long INF = ~0ull;
    while (k < nm) {
        const uint32_t li = lh + l, ci = ch + l;
        const bool lval = li < nl, cval = ci < k;
        const uint32_t lkey = lval ? T.sorted[li] : 0u;
        const uint32_t qa = cval ? T.qa[ci] : 0u, qb = cval ? T.qb[ci] : 0u;
        const unsigned long long kl = lval ? leaf_ord(lkey) : INF, kc = cval ? comp_ord(qa, qb) : INF;
        T.lw[l] = kl; T.cw[l] = kc;
        __builtin_amdgcn_wave_barrier();
        // merged rank of this lane's leaf / composite: its own index + the elements of the other window below it (a leaf and
        // a composite never tie: level 0 against level >= 1)
        uint32_t lo0 = 0, hi0 = 64, lo1 = 0, hi1 = 64;
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const uint32_t m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
            const unsigned long long c = T.cw[m0 & 63u], f = T.lw[m1 & 63u];
            if (lo0 < hi0) { if (c < kl) lo0 = m0 + 1; else hi0 = m0; }
            if (lo1 < hi1) { if (f < kc) lo1 = m1 + 1; else hi1 = m1; }
        }
        const uint32_t rl_ = l + lo0, rc_ = l + lo1;
        // the two smallest of all: among the first two of each window
        unsigned long long x0, x1;
        {
            const unsigned long long a0 = T.lw[0], a1 = T.lw[1], b0 = T.cw[0], b1 = T.cw[1];
            if (a0 < b0) { x0 = a0; x1 = a1 < b0 ? a1 : b0; } else { x0 = b0; x1 = b1 < a0 ? b1 : a0; }
        }
        const uint32_t lv0 = (uint32_t)(x0 >> 9) & 31u, lv1 = (uint32_t)(x1 >> 9) & 31u;
        const unsigned long long c0 = (((x0 >> 14) + (x1 >> 14)) << 14) | ((unsigned long long)((lv0 > lv1 ? lv0 : lv1) + 1u) << 9) | (x0 & 511u);
        // nothing outside the windows may be smaller than what is paired
        unsigned long long bound = c0;
        if (lh + 64u < nl) { const unsigned long long o = leaf_ord(T.sorted[lh + 64u]); bound = o < bound ? o : bound; }
        if (ch + 64u < k) { const unsigned long long o = comp_ord(T.qa[ch + 64u], T.qb[ch + 64u]); bound = o < bound ? o : bound; }
        const uint32_t nlb = (uint32_t)__popcll(__ballot(kl < bound)), ncb = (uint32_t)__popcll(__ballot(kc < bound));
        uint32_t mm = (nlb + ncb) >> 1;                        // pairs of this batch (>= 1: x0 and x1 are below the bound)
        mm = mm ? mm : 1u;                                     // (the two smallest are always a merge: the loop cannot stall)
        const uint32_t m = mm < nm - k ? mm : nm - k;
        const bool lin = lval && rl_ < 2u * m, cin = cval && rc_ < 2u * m;
        if (lin) T.pr[rl_] = make_uint2((lkey >> 9) << 5, (lkey & 511u) * 65537u);     // level 0; slot << 16 | node, node = slot
        if (cin) T.pr[rc_] = make_uint2(qa, qb);
        const uint32_t nlv = (uint32_t)__popcll(__ballot(lin)), ncv = (uint32_t)__popcll(__ballot(cin));
        __builtin_amdgcn_wave_barrier();
        if (l < m) {
            const uint2 e0 = T.pr[2 * l], e1 = T.pr[2 * l + 1];
            const uint32_t v0 = e0.x & 31u, v1 = e1.x & 31u;
            T.qa[k + l] = (((e0.x >> 5) + (e1.x >> 5)) << 5) | ((v0 > v1 ? v0 : v1) + 1u);
            T.qb[k + l] = (e0.y & 0xFFFF0000u) | (nl + k + l);
            T.kids[k + l] = (e0.y & 0xFFFFu) | (e1.y << 16);   // first minimum = left child, second = right
        }
        __builtin_amdgcn_wave_barrier();
        lh += nlv; ch += ncv;
        // sorted by (count, level, slot)?  Look at every new entry and the live entry before it.
        bool viol = false;
        if (l < m && k + l > ch) viol = comp_ord(T.qa[k + l - 1], T.qb[k + l - 1]) > comp_ord(T.qa[k + l], T.qb[k + l]);
        if (__ballot(viol) != 0) {
            // rank the live entries [ch, k + m) again (at most 128 of them; keys are distinct: the slots of live composites
            // are leaves of disjoint subtrees)
            const uint32_t lo = ch, cnt = k + m - ch;
            uint2 ent[2];
            uint32_t rk[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t e = l + 64 * r;
                rk[r] = 0; ent[r] = make_uint2(0u, 0u);
                if (e < cnt) {
                    ent[r] = make_uint2(T.qa[lo + e], T.qb[lo + e]);
                    const unsigned long long me = comp_ord(ent[r].x, ent[r].y);
                    for (uint32_t f = 0; f < cnt; f++) rk[r] += comp_ord(T.qa[lo + f], T.qb[lo + f]) < me ? 1u : 0u;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (l + 64 * r < cnt) { T.qa[lo + rk[r]] = ent[r].x; T.qb[lo + rk[r]] = ent[r].y; }
            __builtin_amdgcn_wave_barrier();
        }
        k += m;
    }
    // the tree, written by all lanes: composite k = node nl + k with its children
    const uint32_t root = nm ? nl + nm - 1u : 0u;              // 2 nl - 2; node 0 when the EOF leaf is alone
    if (l == 0) T.parent[root] = -1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t kk = r * 64 + l;
        if (kk < nm) {
            const uint32_t id = nl + kk, kd = T.kids[kk], lf = kd & 0xFFFFu, rt = kd >> 16;
            T.left[id] = (int16_t)lf; T.right[id] = (int16_t)rt; T.value[id] = -1;
            T.parent[lf] = (int16_t)id; T.parent[rt] = (int16_t)id;
        }
    }
#else
    // leaf queue: the 64 leaves around the head in one register, refilled from T.sorted every 64 leaves.
    // composite queue: a ring of 128 entries in two registers (entry q = lane q & 63 of register (q >> 6) & 1): live
    // composites are disjoint subtrees of >= 2 leaves, so at most 128 are live, and at most 127 when one is added (a
    // merge out of 128 live composites takes at least one of them) -- entry k never lands on a live entry k - 128.
    uint32_t curL = l < nl ? T.sorted[l] : NONE;
    uint32_t QA[2] = {NONE, NONE}, QB[2] = {0, 0}, CH[4] = {0, 0, 0, 0};   // count << 5 | level, slot << 16 | node; children by composite
    auto rl = [](uint32_t v, uint32_t lane) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); };
    // lane writes as compare + select (this compiler has no builtin for v_writelane_b32, and an asm statement costs more
    // in register copies around it than the two instructions it saves)
    auto wl = [&](uint32_t &reg, uint32_t lane, uint32_t v) { reg = l == lane ? v : reg; };
    auto wl3 = [&](uint32_t &r0, uint32_t &r1, uint32_t &r2, uint32_t lane, uint32_t v0, uint32_t v1, uint32_t v2) {
        const bool me = l == lane;
        r0 = me ? v0 : r0; r1 = me ? v1 : r1; r2 = me ? v2 : r2;
    };
    auto get2 = [&](const uint32_t (&R)[2], uint32_t q) -> uint32_t {
        const uint32_t v0 = rl(R[0], q & 63u), v1 = rl(R[1], q & 63u);
        return (q & 64u) ? v1 : v0;
    };
    auto set2 = [&](uint32_t (&R)[2], uint32_t q, uint32_t v) {                  // (rare path)
        if (q & 64u) wl(R[1], q & 63u, v); else wl(R[0], q & 63u, v);
    };
    const uint32_t nm = nl - 1u;                               // merges
    uint32_t lh = 0, lx = rl(curL, 0);                         // leaves taken, the head leaf (NONE past the last)
    uint32_t ch = 0, cA = NONE, cB = 0;                        // composites taken, the head composite (cA = NONE: queue empty)
    uint32_t tA = 0, tB = 0;                                   // the composite at the tail
#pragma unroll
    for (int seg = 0; seg < 4; seg++) {
        const uint32_t kend = nm < (uint32_t)(seg + 1) * 64u ? nm : (uint32_t)(seg + 1) * 64u;
        for (uint32_t k = (uint32_t)seg * 64u; k < kend; k++) {          // composite k is queue entry k when it is made
            uint32_t a0, b0, a1, b1;
#define GLC_TAKE(a, b)                                                                                              \
            {                                                                                                       \
                const uint32_t lA = (lx >> 9) << 5;            /* NONE >> 9 << 5 is above every real key, below NONE */ \
                if (lA < cA) {                                 /* a leaf (both queues empty cannot happen: k < nm) */ \
                    const uint32_t slot = lx & 511u;                                                                \
                    a = lA; b = slot * 65537u;                 /* slot << 16 | node, node = slot */                \
                    lh++;                                                                                           \
                    if ((lh & 63u) == 0) {                                                                          \
                        asm volatile("" ::: "memory");         /* a real branch: taken four times per tree */      \
                        curL = lh + l < nl ? T.sorted[lh + l] : NONE;                                               \
                    }                                                                                               \
                    lx = rl(curL, lh & 63u);                                                                        \
                } else {                                                                                            \
                    a = cA; b = cB;                                                                                 \
                    ch++;                                                                                           \
                    cA = NONE;                                 /* entries ch .. k-1 are live */                     \
                    if (ch != k) { cA = get2(QA, ch); cB = get2(QB, ch); }                                          \
                }                                                                                                   \
            }
            GLC_TAKE(a0, b0)
            GLC_TAKE(a1, b1)
#undef GLC_TAKE
            const uint32_t l0 = a0 & 31u, l1 = a1 & 31u;
            const uint32_t A = (((a0 >> 5) + (a1 >> 5)) << 5) | ((l0 > l1 ? l0 : l1) + 1u);
            const uint32_t B = (b0 & 0xFFFF0000u) | (nl + k);
            const uint32_t kids = (b0 & 0xFFFFu) | (b1 << 16);
            const uint32_t pl = k & 63u;
            bool reorder = false;
            if (tA == A) {                                     // (rare: the tail composite has this count and level)
                asm volatile("" ::: "memory");
                reorder = k > ch && tB > B;
            }
            if (reorder) {
                // composites of the same count and level at the tail with a larger slot move up by one
                wl(CH[seg], pl, kids);
                uint32_t q = k;
                while (q > ch) {
                    const uint32_t pa = get2(QA, q - 1), pb = get2(QB, q - 1);
                    if (pa != A || pb < B) break;
                    set2(QA, q, pa); set2(QB, q, pb);
                    q--;
                }
                set2(QA, q, A); set2(QB, q, B);
                if (q == ch) { cA = A; cB = B; }
            } else {
                wl3(QA[seg & 1], QB[seg & 1], CH[seg], pl, A, B, kids);
                if (k == ch) { cA = A; cB = B; }
                tA = A; tB = B;
            }
        }
    }
    // the tree, written by all lanes: composite k = node nl + k with the children recorded above
    const uint32_t root = nm ? nl + nm - 1u : 0u;              // 2 nl - 2; node 0 when the EOF leaf is alone
    if (l == 0) T.parent[root] = -1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t k = r * 64 + l;
        if (k < nm) {
            const uint32_t id = nl + k, lf = CH[r] & 0xFFFFu, rt = CH[r] >> 16;
            T.left[id] = (int16_t)lf; T.right[id] = (int16_t)rt; T.value[id] = -1;
            T.parent[lf] = (int16_t)id; T.parent[rt] = (int16_t)id;
        }
    }
#endif
    if (l == 0) T.head = (int)root;
}

} // namespace glc
