// bwt_periodic.hip -- blocks that are ONE periodic stretch: T[i] = T[i + p] up to a short tail.  gfx950 / wave64.
//
// A page repeated to the end of the block, a two-byte pattern, one byte repeated up to a different last byte: every suffix of
// such a block ties with the suffix one period further on for (nearly) the whole block, which is the worst case of every
// comparison-based tier here (the sample sorter's depth cap, then ~18 prefix-doubling rounds over a million live suffixes:
// 1.3 - 2.0 ms per block where a Zipf block takes 0.012).  The reference's skew / DC3 sorter (cudpp-inpar/src/cudpp/app/
// sa_app.cu:125-298) has no such case -- its cost does not depend on the depth of the repeats -- so this tier exists to take
// the floor out of that cliff for the blocks where the answer has a closed form.  The result is the same unique suffix array.
//
// Let p be the smallest period of the block's beginning, e the first position with T[e] != T[e - p] (e = n if there is none),
// t = n - e the tail, L = the multiple of p that covers max(p, t), Z = PER_Z = 2; taken if p, L <= PER_PMAX and
// e >= Z L + 2 p + 1.  Suffixes are of three kinds:
//   far       i < e - Z L          more than Z L periodic symbols ahead: T[i ..] starts with the rotation R_c^inf, c = i mod p
//   exit      e - Z L <= i < e     (Z L of them) the last periods before the break
//   tail      e <= i < n           (t of them)
// * Two far suffixes of different classes differ within p symbols (p is the smallest period: the rotations are distinct), so
//   their order is the order of the rotations R_c.
// * Two far suffixes i < i' of ONE class agree until the nearer one reaches e; there it reads T[e] (or the end of the text)
//   where the other reads T[e - p]: the nearer one is the smaller iff X := (e == n or T[e] < T[e - p]).  The same for every
//   pair: a class is one monotone chain of positions, descending if X, ascending if not.
// * An exit or tail suffix against a far suffix of class c must be decided INSIDE the far suffix's periodic part, wherever in
//   its chain that suffix sits -- then the comparison is ONE comparison with R_c^inf.  How deep can it go?  An exit suffix i of
//   class c reads T[e] where the rotation reads T[e - p], at depth e - i <= Z L.  An exit suffix of ANOTHER class with p or more
//   periodic symbols left differs from R_c within p.  But one with FEWER than p left (e - i < p) may agree with R_c through
//   all of them and go on agreeing through the tail: depth up to (p - 1) + t + 1 <= 2 L.  That is why the zone is 2 L wide and
//   not L (round 5's form: 'abaa' * k + 'baab' has the exit suffix "ab" + "baab" tie with the far suffix nearest the break for
//   all of that suffix's L + 1 periodic symbols, and the real text then reads T[e] where the model read T[e - p]; a CPU model
//   of this layout against a naive suffix array, tests/periodic_model.py, gives 30 wrong of 60 000 random small blocks with
//   Z = 1 and none with Z = 2).  A tail suffix is no longer than t <= L.
// So the block's suffix array is the sorted order of p + Z L + t REPRESENTATIVES -- p rotations, Z L + t explicit suffixes --
// with every rotation expanded into its class's chain.  The representatives are the suffixes of a small text
//   U = T[0 .. Z L + 2 p + 1)  |  0xFF  |  T[e - Z L .. n)  |  0 0 0 ...     (2 Z L + 2 p + t + 2 bytes and zero padding)
// at positions c < p (rotation c: Z L + p + 2 or more periodic symbols before the separator, enough for every comparison above
// even when an explicit suffix has run into the padding and ties with up to p - 1 zeros of the rotation: 2 p + t - 1 <= 2 L + p)
// and Z L + 2 p + 2 + k (explicit suffix e - Z L + k; it ends where U ends, and the zero padding behind it makes "the shorter
// suffix is the smaller" come out as it does at the end of a text).  U is sorted by the general sorter (a few thousand
// suffixes, whatever n is), and a class's rows of the BWT are one byte repeated: T[i - 1] = T[(c + p - 1) mod p] for every
// member but position 0.
#include "glc_device.h"
#include "glc_internal.h"

namespace glc {

constexpr uint32_t PER_NT = 1024;
// (per_span = L, per_text_len = bytes of U: glc_internal.h)
constexpr uint32_t PER_TRIES = 8;                              // candidate periods looked at per block

__device__ __forceinline__ bool per_eq16(const uint8_t *a, const uint8_t *b)
{
    uint4 x, y;
    __builtin_memcpy(&x, a, 16);
    __builtin_memcpy(&y, b, 16);
    return x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
}

// one workgroup per listed block that the other tiers gave up on: smallest period of its beginning, where it breaks
__global__ __launch_bounds__(PER_NT) void k_per_detect(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                       const uint32_t *__restrict__ list, uint32_t *__restrict__ flag,
                                                       uint4 *__restrict__ info, uint32_t *__restrict__ plist,
                                                       uint32_t *__restrict__ pcount, uint32_t take, bool probe)
{
    __shared__ uint32_t s_cand[PER_PMAX / 32];
    __shared__ uint32_t s_e, s_p;
    const uint32_t b = list[blockIdx.x], tid = threadIdx.x;
    // probe (before the sample sorter's first attempt, every listed block): nothing is taken -- a block whose beginning is periodic
    // for 3/8 of the block or more is flagged (2 | 1) so that the sample sorter leaves it alone: a quarter of its samples would tie
    // beyond any cap, its exact attempt (2.3 ms per 32 blocks of two periodic halves, 0.6 per 64 deep_repeats blocks) and its
    // tolerant one (4.9) can only give up.  The tier proper then looks at the block as at any other flagged one.
    if ((!probe && !flag[b]) || n < 16 * PER_PMAX) return;     // finished by an earlier tier / too small to be worth a tier
    const uint8_t *T = text + (size_t)b * stride;
    for (uint32_t i = tid; i < PER_PMAX / 32; i += PER_NT) s_cand[i] = 0;
    if (tid == 0) s_p = 0;
    __syncthreads();
    for (uint32_t j = tid + 1; j <= PER_PMAX; j += PER_NT)
        if (per_eq16(T, T + j)) atomicOr(&s_cand[(j - 1) >> 5], 1u << ((j - 1) & 31));
    __syncthreads();
    uint32_t from = 0, emax = 0;                               // candidates are looked at in ascending order; furthest break so far
    uint32_t seen_p[PER_TRIES], seen_e[PER_TRIES];             // candidates examined so far and where each broke
    for (uint32_t tries = 0; tries < PER_TRIES;) {
        uint32_t p = 0;
        for (uint32_t w = from >> 5; w < PER_PMAX / 32 && !p; w++) {
            const uint32_t m = s_cand[w] & (w == (from >> 5) ? ~0u << (from & 31) : ~0u);
            if (m) p = w * 32 + (uint32_t)__builtin_ctz(m) + 1;
        }
        if (!p) break;                                         // (uniform: every thread reads the same words)
        from = p;                                              // (bit p - 1 is this candidate: the next search starts behind it)
        // a multiple k q of an examined candidate q whose period held up to position k q or beyond breaks exactly where q broke
        // (T[e] != T[e - q] = T[e - k q]): nothing new to learn, and no scan of the block -- a periodic block offers q, 2 q, 3 q, ...
        // as candidates, and eight scans of it were 0.2 ms of a lone block's call
        bool known = false;
#pragma unroll
        for (uint32_t k = 0; k < PER_TRIES; k++) known |= k < tries && p % seen_p[k] == 0 && seen_e[k] >= p;
        if (known) continue;
        if (tid == 0) s_e = n;
        __syncthreads();
        // first position e >= p with T[e] != T[e - p]
        uint32_t mine = n;
        for (uint32_t i = 16 * tid; i + p < n && mine == n; i += 16 * PER_NT) {
            if (i + p + 16 <= n && per_eq16(T + i, T + i + p)) continue;
            for (uint32_t k = 0; k < 16 && i + k + p < n; k++)
                if (T[i + k] != T[i + k + p]) { mine = i + k + p; break; }
        }
        if (mine < n) atomicMin(&s_e, mine);
        __syncthreads();
        const uint32_t e = s_e;
        __syncthreads();
        seen_p[tries] = p; seen_e[tries] = e;
        tries++;
        // p is the SMALLEST period of T[0 .. e) iff no smaller candidate reached as far (the rotations of a smallest period are
        // distinct, which the closed form rests on: "11111" taken with p = 5 would have five equal classes)
        const uint32_t t = n - e, L = PER_Z * per_span(p, t);    // (L: the explicit zone, Z spans)
        if (probe) { emax = max(emax, e); if (emax >= n / 8 * 3) break; continue; }
        if (e > emax && t <= PER_PMAX && L <= PER_Z * PER_PMAX && e >= L + 2 * p + 1 && per_text_len(p, t) + 16 <= PER_NU) {
            if (tid == 0) {
                const uint32_t slot = atomicAdd(pcount, 1u);   // (past `take` slots: the block stays with the tiers behind this one; the host clamps the count)
                if (slot < take) {
                    atomicMax(pcount + 1, per_text_len(p, t) + 16u);   // the longest text of representatives of the call: they are sorted as one batch
                    plist[slot] = b;
                    info[b] = make_uint4(p, e, (e == n || T[e] < T[e - p]) ? 1u : 0u, slot);
                }
            }
            return;
        }
        emax = max(emax, e);
        // (a candidate that is no period of the whole block: a larger one may still be -- "abab c abab c ...")
    }
    // Not this tier's block -- but if its beginning is periodic for 3/8 of the block or more, it is not the TOLERANT sample
    // sorter's either (a quarter of its samples would tie beyond the cap: that attempt took 4.9 ms per 32 blocks of two periodic
    // halves before it gave up).  The block's flag becomes 2 | 1 -- the value a block that is deep AND had a bucket past its slot
    // carries, which the tolerant pass does not take either: k_ss_retry_list lists flag == 2 only -- and the general
    // sorter (whose doubling rounds order a periodic stretch as chains, bwt_sa.hip) takes the block as it takes every flagged one.
    if (tid == 0 && emax >= n / 8 * 3 && (probe || flag[b] == 2u)) flag[b] = 3u;
}

// U of every taken block: T[0 .. Z L + 2 p + 1) | 0xFF | T[e - Z L .. n) | zeros up to nu   (below, L stands for Z L)
__global__ __launch_bounds__(256) void k_per_text(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                  const uint32_t *__restrict__ plist, const uint4 *__restrict__ info,
                                                  uint8_t *__restrict__ U, uint32_t nu)
{
    const uint32_t b = plist[blockIdx.y];
    const uint4 in = info[b];
    const uint32_t p = in.x, e = in.y, L = PER_Z * per_span(p, n - e), la = L + 2 * p + 1, lb = L + (n - e);
    const uint8_t *T = text + (size_t)b * stride;
    uint8_t *D = U + (size_t)blockIdx.y * PER_NU;
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nu; q += gridDim.x * 256) {
        uint8_t v = 0;
        if (q < la) v = T[q];
        else if (q == la) v = 0xFF;
        else if (q - la - 1 < lb) v = T[e - L + (q - la - 1)];
        D[q] = v;
    }
}

// rows of every representative: exclusive scan of the weights (a rotation: the members of its class; an explicit suffix: 1;
// anything else in U: 0) in the order of U's suffix array.  The weights must add up to n -- if they do not, the block is
// left to the general sorter (ok = 0).
__global__ __launch_bounds__(PER_NT) void k_per_bases(uint32_t n, const uint32_t *__restrict__ plist, uint4 *__restrict__ info,
                                                      const uint32_t *__restrict__ sa_u, uint32_t sa_stride, uint32_t nu,
                                                      uint32_t *__restrict__ base, uint32_t *__restrict__ ok)
{
    __shared__ uint32_t s_tmp[PER_NT / 64 + 1];
    const uint32_t slot = blockIdx.x, b = plist[slot], tid = threadIdx.x;
    const uint4 in = info[b];
    const uint32_t p = in.x, e = in.y, L = PER_Z * per_span(p, n - e), lb = L + (n - e), x0 = L + 2 * p + 2;
    const uint32_t *SA = sa_u + (size_t)slot * sa_stride;
    uint32_t *B = base + (size_t)slot * (PER_NU + 1);
    uint32_t carry = 0;
    for (uint32_t r0 = 0; r0 < nu; r0 += PER_NT) {
        const uint32_t r = r0 + tid;
        uint32_t w = 0;
        if (r < nu) {
            const uint32_t q = SA[r];
            if (q < p) w = (e - L - q + p - 1) / p;            // far members of class q: positions q, q + p, ... < e - L
            else if (q >= x0 && q - x0 < lb) w = 1;
        }
        uint32_t tot = 0;
        const uint32_t ex = block_excl_add<PER_NT>(w, s_tmp, &tot);
        if (r < nu) B[r] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) { B[nu] = carry; ok[slot] = carry == n ? 1u : 0u; }
}

// the rows: a thread takes 16 consecutive rows of the block's BWT (and suffix array)
__global__ __launch_bounds__(256) void k_per_rows(const uint8_t *__restrict__ text, size_t stride, uint32_t n,
                                                  const uint32_t *__restrict__ plist, const uint4 *__restrict__ info,
                                                  const uint32_t *__restrict__ sa_u, uint32_t sa_stride, uint32_t nu,
                                                  const uint32_t *__restrict__ base, const uint32_t *__restrict__ ok,
                                                  uint8_t *__restrict__ bwt_out, size_t bwt_stride, int *__restrict__ d_index,
                                                  uint32_t *__restrict__ ss_flag, uint32_t *__restrict__ lcnt,
                                                  uint32_t *__restrict__ ndone)
{
    const uint32_t slot = blockIdx.y, b = plist[slot];
    if (!ok[slot]) return;                                     // (uniform) left to the general sorter
    const uint4 in = info[b];
    const uint32_t p = in.x, e = in.y, X = in.z, L = PER_Z * per_span(p, n - e), x0 = L + 2 * p + 2;
    const uint8_t *T = text + (size_t)b * stride;
    const uint32_t *SA = sa_u + (size_t)slot * sa_stride;
    const uint32_t *B = base + (size_t)slot * (PER_NU + 1);
    uint8_t *O = bwt_out + (size_t)b * bwt_stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ss_flag[b] = 0; lcnt[b] = 0; atomicAdd(ndone, 1u); }   // this tier's block: finished
    const uint32_t row0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (row0 >= n) return;
    // the representative whose rows hold row0: the last r with B[r] <= row0 (representatives of weight 0 share their base
    // with the next one, so "the last" is one that has rows)
    uint32_t lo = 0, hi = nu;                                  // B[lo] <= row0 < B[hi]  (B[0] = 0, B[nu] = n)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (B[mid] <= row0) lo = mid; else hi = mid;
    }
    uint32_t r = lo, next = B[r + 1];
    uint32_t v[4] = {0, 0, 0, 0};
    const uint32_t rows = min(16u, n - row0);
    for (uint32_t k = 0; k < rows; k++) {
        const uint32_t row = row0 + k;
        while (row >= next) { r++; next = B[r + 1]; }
        const uint32_t q = SA[r];
        uint32_t byte;
        if (q < p) {
            const uint32_t m = (e - L - q + p - 1) / p, at = row - B[r];
            const uint32_t i = q + (X ? m - 1 - at : at) * p;  // the chain: nearest to the break first if X
            if (i == 0) { byte = T[n - 1]; d_index[b] = (int)row; }
            else byte = T[(q + p - 1) % p];
        } else byte = T[e - L + (q - x0) - 1];                 // an explicit suffix: position >= e - L >= 2 p + 1
        v[k >> 2] |= byte << (8 * (k & 3));
    }
    if (rows == 16 && (reinterpret_cast<uintptr_t>(O + row0) & 15) == 0) *reinterpret_cast<uint4 *>(O + row0) = make_uint4(v[0], v[1], v[2], v[3]);
    else for (uint32_t k = 0; k < rows; k++) O[row0 + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
}

#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t per_reserve(SaScratch &s)
{
    if (s.per_text) return hipSuccess;
    auto A = [&](void **p, size_t bytes) -> hipError_t { s.bytes += bytes; return hipMalloc(p, bytes); };
    GLC_TRY(A((void **)&s.per_info, (size_t)s.rows * sizeof(uint4)));
    GLC_TRY(A((void **)&s.per_list, (size_t)s.rows * 4));
    GLC_TRY(A((void **)&s.per_ok, (size_t)s.rows * 4));
    GLC_TRY(A((void **)&s.per_count, 16));
    const size_t take = s.rows < PER_TAKE ? s.rows : PER_TAKE;
    GLC_TRY(A((void **)&s.per_base, take * (PER_NU + 1) * 4));
    GLC_TRY(A((void **)&s.per_text, take * PER_NU));
    return hipSuccess;
}

hipError_t per_detect(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nlisted, SaScratch &s)
{
    GLC_TRY(hipMemsetAsync(s.per_count, 0, 16, st));
    hipLaunchKernelGGL(k_per_detect, dim3(nlisted), dim3(PER_NT), 0, st, text, text_stride, n, s.ss_list, s.ss_flag, s.per_info,
                       s.per_list, s.per_count, s.rows < PER_TAKE ? s.rows : PER_TAKE, false);
    return hipGetLastError();
}

hipError_t per_probe(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nlisted, SaScratch &s)
{
    if (n < 16 * PER_PMAX) return hipSuccess;
    hipLaunchKernelGGL(k_per_detect, dim3(nlisted), dim3(PER_NT), 0, st, text, text_stride, n, s.ss_list, s.ss_flag, (uint4 *)nullptr,
                       (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, true);
    return hipGetLastError();
}

hipError_t per_text(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nper, uint32_t nu, SaScratch &s)
{
    hipLaunchKernelGGL(k_per_text, dim3((nu + 255) / 256 < 16 ? (nu + 255) / 256 : 16, nper), dim3(256), 0, st, text, text_stride, n,
                       s.per_list, s.per_info, s.per_text, nu);
    return hipGetLastError();
}

hipError_t per_expand(hipStream_t st, const uint8_t *text, size_t text_stride, uint32_t n, uint32_t nper, uint32_t nu, SaScratch &s,
                      uint8_t *bwt_out, size_t bwt_stride, int *d_index)
{
    hipLaunchKernelGGL(k_per_bases, dim3(nper), dim3(PER_NT), 0, st, n, s.per_list, s.per_info, s.sa, s.nmax, nu, s.per_base, s.per_ok);
    hipLaunchKernelGGL(k_per_rows, dim3((n + 4095) / 4096, nper), dim3(256), 0, st, text, text_stride, n, s.per_list, s.per_info, s.sa,
                       s.nmax, nu, s.per_base, s.per_ok, bwt_out, bwt_stride, d_index, s.ss_flag, s.fs_lcnt, s.per_count + 2);
    return hipGetLastError();
}

} // namespace glc
