// huff_tree.h -- the reference's Huffman tree shape, built by one wave64.
// Shared by the encoder (huffman.hip) and the decoder (decode.hip): both must
// derive the identical tree from the 256-bin histogram (+EOF with count 1).
//
// Restates huffman_build_tree_kernel's merge loop (cudpp-inpar/src/cudpp/kernel/
// compress_kernel.cuh:2306-2392) with FindMinimumCount's order (cta/compress_cta.cuh:
// 550-571: lowest count, then lowest level, then lowest slot) as a wave-wide arg-min
// over packed 64-bit keys  count<<32 | level<<16 | slot.
#pragma once
#include "glc_device.h"

namespace glc {

constexpr int      HUFF_NODES = 2 * 257 - 1;          // 513
constexpr uint64_t HUFF_KEY_NONE = ~0ull;

struct HuffTreeLds {
    uint64_t key[320];                                 // leaf keys on their way into the registers (slots < nl <= 257)
    uint32_t count[HUFF_NODES];
    int16_t  level[HUFF_NODES], value[HUFF_NODES];     // value = symbol, -1 for a composite node
    int16_t  left[HUFF_NODES], right[HUFF_NODES], parent[HUFF_NODES];
    int      nl, head;
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

// Called by ONE full wave (l = lane).  hist257[256] must already hold the EOF count 1.
// The candidates live in REGISTERS: slot s (< 320) is register s >> 6 of lane s & 63 and holds count << 5 | level
// (count <= 2^20 + 1, level < 32; ~0 = no candidate).  The slot itself need not be stored -- it is where the value
// sits -- so FindMinimumCount's order (count, level, slot) is two 32-bit wave minima: the smallest value, then the
// smallest slot among the lanes that hold it.  (64-bit keys in LDS: five reads and a barrier per arg-min, 512 arg-mins
// per block back to back; 64-bit keys in registers: 12 DPP steps of compare-and-select on register pairs.)
__device__ __forceinline__ void huff_tree_build(HuffTreeLds &T, const uint32_t *hist257, unsigned l)
{
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    // leaves: present symbols in ascending order -> slots 0..nl-1 (compress_kernel.cuh:2310-2321)
    uint32_t nl = 0;
    for (int r = 0; r < 5; r++) {
        const uint32_t sym = r * 64 + l;
        const uint32_t c = sym < 257 ? hist257[sym] : 0u;
        const uint64_t bal = __ballot(c > 0);
        if (c > 0) {
            const uint32_t slot = nl + mbcnt(bal);
            T.count[slot] = c; T.level[slot] = 0; T.value[slot] = (int16_t)sym;
            T.left[slot] = -1; T.right[slot] = -1; T.parent[slot] = -1;
            T.key[slot] = (uint64_t)(c << 5);
        }
        nl += (uint32_t)__popcll(bal);
    }
    for (uint32_t s = nl + l; s < 320; s += 64) T.key[s] = NONE;
    __builtin_amdgcn_wave_barrier();
    // Q = value << 9 | slot for values below 2^23 (count < 2^18), else ~0: while the smallest candidate is that small
    // -- all but the last few merges of a 1 MiB block, whose counts add up to 2^20 + 1 -- ONE wave minimum finds it
    uint32_t P[5], Q[5];
    auto packed = [&](uint32_t v, int r) -> uint32_t { return v < (1u << 23) ? (v << 9) | (uint32_t)(r * 64 + (int)l) : NONE; };
#pragma unroll
    for (int r = 0; r < 5; r++) { P[r] = (uint32_t)T.key[r * 64 + l]; Q[r] = packed(P[r], r); }
    __builtin_amdgcn_wave_barrier();

    // (value, slot) of the smallest candidate; value NONE if there is none
    auto arg_min = [&](uint32_t &val) -> uint32_t {
        {
            uint32_t qb = Q[0];
#pragma unroll
            for (int r = 1; r < 5; r++) qb = Q[r] < qb ? Q[r] : qb;
            const uint32_t q = wave_min_u32(qb);
            if (q != NONE) { val = q >> 9; return q & 511u; }
        }
        uint32_t best = P[0], br = 0;
#pragma unroll
        for (int r = 1; r < 5; r++) { const bool lt = P[r] < best; best = lt ? P[r] : best; br = lt ? (uint32_t)r : br; }
        val = wave_min_u32(best);
        return wave_min_u32(best == val ? br * 64 + l : NONE);
    };
    int head = -1;
    for (uint32_t k = 0;; k++) {
        uint32_t v1, v2;
        const uint32_t m1 = arg_min(v1);
        if (v1 == NONE) break;
        const int min1 = (int)m1;
        head = min1;
#pragma unroll
        for (int r = 0; r < 5; r++) if (m1 == r * 64 + l) { P[r] = NONE; Q[r] = NONE; }
        const uint32_t m2 = arg_min(v2);
        if (v2 == NONE) break;
        const int min2 = (int)m2;
        // min1 moves to the next free slot >= nl and becomes the LEFT child; min2 stays and
        // is the RIGHT child; the composite takes min1's slot (compress_kernel.cuh:2344-2385)
        const uint32_t c1 = v1 >> 5, c2 = v2 >> 5;
        const int l1 = (int)(v1 & 31u), l2 = (int)(v2 & 31u);
        const int lv = (l1 > l2 ? l1 : l2) + 1;
        const uint32_t nk = ((c1 + c2) << 5) | (uint32_t)lv;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            if (m1 == r * 64 + l) { P[r] = nk; Q[r] = packed(nk, r); }
            if (m2 == r * 64 + l) { P[r] = NONE; Q[r] = NONE; }
        }
        if (l == 0) {
            const int i = (int)(nl + k);
            const int lf = T.left[min1], rt = T.right[min1];
            T.count[i] = c1; T.level[i] = (int16_t)l1; T.value[i] = T.value[min1];
            T.left[i] = (int16_t)lf; T.right[i] = (int16_t)rt; T.parent[i] = (int16_t)min1;
            if (lf >= 0) T.parent[lf] = (int16_t)i;
            if (rt >= 0) T.parent[rt] = (int16_t)i;
            T.left[min1] = (int16_t)i; T.right[min1] = (int16_t)min2; T.value[min1] = -1;
            T.count[min1] = c1 + c2; T.level[min1] = (int16_t)lv; T.parent[min1] = -1;
            T.parent[min2] = (int16_t)min1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (l == 0) { T.nl = (int)nl; T.head = head; }
}

} // namespace glc
