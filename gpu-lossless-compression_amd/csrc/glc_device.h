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

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t x)
{
    const unsigned l = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t y = __shfl_up(x, o, WAVE);
        if (l >= (unsigned)o) x += y;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x)
{
    const unsigned l = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t y = __shfl_up(x, o, WAVE);
        if (l >= (unsigned)o) x = x > y ? x : y;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, WAVE);
    return x;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t x)
{
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) { uint32_t y = __shfl_xor(x, o, WAVE); x = x > y ? x : y; }
    return x;
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
    uint32_t prev = __shfl_up(inc, 1, WAVE);
    if (l == 0) prev = 0;
    return base > prev ? base : prev;
}

} // namespace glc
