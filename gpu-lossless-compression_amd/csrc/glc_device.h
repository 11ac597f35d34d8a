// glc_device.h -- wave64 / workgroup helpers shared by the gfx950 kernels.
// CDNA4 only: wavefront = 64 lanes, ballots are 64-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glc {

constexpr int WAVE = 64;

__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ unsigned mbcnt(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Lanes of `valid` whose low BITS bits of d equal this lane's (the multi-split "match").
// Accumulates the MISMATCH mask: per bit, x = 0 / ~0 (v_bfe_i32), ballot(x), and lanes that differ
// from me are ballot ^ x; two bits fold into the accumulator with one v_or3 per half.  5 VALU
// per bit instead of the 8-9 the select form (set ? bal : ~bal) compiles to -- the LSD passes
// are VALU-bound on this loop.
template <int BITS>
__device__ __forceinline__ uint64_t wave_match(uint32_t d, uint64_t valid)
{
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int bit = 0; bit < BITS; bit++) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1);
        const uint64_t bal = __ballot((int)x < 0);
        mlo |= (uint32_t)bal ^ x;
        mhi |= (uint32_t)(bal >> 32) ^ x;
    }
    return valid & ~(((uint64_t)mhi << 32) | mlo);
}

// Wave64 scans on DPP (gfx9 row_shr within rows of 16, then row_bcast:15 / row_bcast:31 across
// rows): 6 VALU instructions, no LDS crossbar.  (__shfl_up compiles to ds_bpermute_b32: an LDS round
// trip per step, 6 dependent steps per scan, and the rank kernel runs three scans per tile.)
#define GLC_DPP(x, ctrl, rowmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rowmask), 0xf, false))

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t x)
{
    x += GLC_DPP(x, 0x111, 0xf);       // row_shr:1  (lanes without a source read `old` = 0)
    x += GLC_DPP(x, 0x112, 0xf);       // row_shr:2
    x += GLC_DPP(x, 0x114, 0xf);       // row_shr:4
    x += GLC_DPP(x, 0x118, 0xf);       // row_shr:8
    x += GLC_DPP(x, 0x142, 0xa);       // row_bcast:15 -> rows 1 and 3
    x += GLC_DPP(x, 0x143, 0xc);       // row_bcast:31 -> rows 2 and 3
    return x;
}

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x)
{
    uint32_t y;
    y = GLC_DPP(x, 0x111, 0xf); x = x > y ? x : y;
    y = GLC_DPP(x, 0x112, 0xf); x = x > y ? x : y;
    y = GLC_DPP(x, 0x114, 0xf); x = x > y ? x : y;
    y = GLC_DPP(x, 0x118, 0xf); x = x > y ? x : y;
    y = GLC_DPP(x, 0x142, 0xa); x = x > y ? x : y;
    y = GLC_DPP(x, 0x143, 0xc); x = x > y ? x : y;
    return x;
}

// value of the previous lane (0 for lane 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t wave_prev(uint32_t x) { return GLC_DPP(x, 0x138, 0xf); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_add(x), 63);
}

__device__ __forceinline__ uint32_t wave_max(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_max(x), 63);
}

// Workgroup exclusive prefix sum; every thread of the NT-thread block calls it.
// s_tmp: NT/64 + 1 words of LDS.  *total (optional) = block sum.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_add(uint32_t x, uint32_t *s_tmp, uint32_t *total = nullptr)
{
    constexpr int NW = NT / WAVE;
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint32_t inc = wave_incl_add(x);
    __syncthreads();                       // protect s_tmp reuse
    if (l == 63) s_tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) { uint32_t v = s_tmp[i]; if ((unsigned)i < w) base += v; tot += v; }
    if (total) *total = tot;
    return base + inc - x;
}

// block_excl_add with ONE barrier that orders LDS only (s_waitcnt lgkmcnt(0); s_barrier): global loads / atomics a
// software-pipelined kernel has in flight stay in flight.  The caller guarantees s_tmp is not in use when it is entered.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_add_lds(uint32_t x, uint32_t *s_tmp)
{
    constexpr int NW = NT / WAVE;
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t inc = wave_incl_add(x);
    if (l == 63) s_tmp[w] = inc;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) { const uint32_t v = s_tmp[i]; if ((unsigned)i < w) base += v; }
    return base + inc - x;
}

// Workgroup exclusive prefix max (identity 0).
template <int NT>
__device__ __forceinline__ uint32_t block_excl_max(uint32_t x, uint32_t *s_tmp)
{
    constexpr int NW = NT / WAVE;
    const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
    uint32_t inc = wave_incl_max(x);
    __syncthreads();
    if (l == 63) s_tmp[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) { uint32_t v = s_tmp[i]; if ((unsigned)i < w) base = base > v ? base : v; }
    const uint32_t prev = wave_prev(inc);
    return base > prev ? base : prev;
}

// XCD-aware order of a (x, y) grid: physical workgroup p runs on XCD p & 7; the logical workgroups are dealt out so that
// every XCD takes one contiguous run of them (everything of a block on one XCD: what its workgroups share stays in one L2)
// a dword every lane of the workgroup wants, through the scalar cache (a fraction of a vector load's latency, and a hit for
// every later workgroup of the CU that asks for the same word).  Only for words nobody writes while this kernel runs: the scalar
// cache is not coherent with stores of other workgroups.
__device__ __forceinline__ uint32_t scalar_load_u32(const uint32_t *p)
{
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(sa) : "memory");
    return v;
}

__device__ __forceinline__ void xcd_order(uint32_t &bx, uint32_t &by)
{
    const uint32_t nx = gridDim.x, total = nx * gridDim.y, p = blockIdx.y * nx + blockIdx.x;
    const uint32_t q = total >> 3, r = total & 7u, x = p & 7u;
    const uint32_t lg = x * q + min(x, r) + (p >> 3);
    by = lg / nx; bx = lg % nx;
}


} // namespace glc
