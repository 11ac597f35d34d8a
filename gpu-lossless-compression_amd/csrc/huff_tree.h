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

// Called by ONE full wave (l = lane).  hist257[256] must already hold the EOF count 1.
// The candidate keys live in REGISTERS: slot s (< 320) is register s >> 6 of lane s & 63, so the two arg-mins of a
// merge are five compares + one DPP reduction each with no LDS round trip between them (with the keys in LDS the
// 512 reductions of a block each waited for five LDS reads and a barrier: 0.24 -> 0.1x ms per 256 blocks).
__device__ __forceinline__ void huff_tree_build(HuffTreeLds &T, const uint32_t *hist257, unsigned l)
{
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
            T.key[slot] = ((uint64_t)c << 32) | slot;
        }
        nl += (uint32_t)__popcll(bal);
    }
    for (uint32_t s = nl + l; s < 320; s += 64) T.key[s] = HUFF_KEY_NONE;
    __builtin_amdgcn_wave_barrier();
    uint64_t key[5];
#pragma unroll
    for (int r = 0; r < 5; r++) key[r] = T.key[r * 64 + l];
    __builtin_amdgcn_wave_barrier();

    int head = -1;
    for (uint32_t k = 0;; k++) {
        uint64_t best = key[0];
#pragma unroll
        for (int r = 1; r < 5; r++) best = key[r] < best ? key[r] : best;
        best = wave_min_u64(best);
        if (best == HUFF_KEY_NONE) break;
        const int min1 = (int)(best & 0xFFFF);
        head = min1;
#pragma unroll
        for (int r = 0; r < 5; r++) if ((unsigned)min1 == r * 64 + l) key[r] = HUFF_KEY_NONE;
        uint64_t best2 = key[0];
#pragma unroll
        for (int r = 1; r < 5; r++) best2 = key[r] < best2 ? key[r] : best2;
        best2 = wave_min_u64(best2);
        if (best2 == HUFF_KEY_NONE) break;
        const int min2 = (int)(best2 & 0xFFFF);
        // min1 moves to the next free slot >= nl and becomes the LEFT child; min2 stays and
        // is the RIGHT child; the composite takes min1's slot (compress_kernel.cuh:2344-2385)
        const uint32_t c1 = (uint32_t)(best >> 32), c2 = (uint32_t)(best2 >> 32);
        const int l1 = (int)((best >> 16) & 0xFFFF), l2 = (int)((best2 >> 16) & 0xFFFF);
        const int lv = (l1 > l2 ? l1 : l2) + 1;
        const uint64_t nk = ((uint64_t)(c1 + c2) << 32) | ((uint64_t)lv << 16) | (uint32_t)min1;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            if ((unsigned)min1 == r * 64 + l) key[r] = nk;
            if ((unsigned)min2 == r * 64 + l) key[r] = HUFF_KEY_NONE;
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
