// culzss.hip -- CULZSS match search, token selection + flag packing, and decode
// for 4096-byte packets.  gfx950 / wave64.
//
// Byte-exact replacement of (cuda-lzss-cluster):
//   EncodeKernel + FindMatch      gpu_compress.cu:104-168,182-350
//   aftercomp (CPU, serial)       gpu_compress.cu:462-566   -> on the GPU here
//   trailer of aftercompression_wrapper                    gpu_compress.cu:620-657
//   DecodeKernel                  gpu_decompress.cu:120-244
//
// The reference runs one 128-thread CTA per packet through 32 barrier-separated
// steps over two 256-byte LDS rings.  Every lane's search only ever reads bytes
// of the packet itself (window = text[p-128 .. p-2], look-ahead = text[p .. p+127],
// spaces before the packet), so here the whole packet is staged ONCE into LDS
// behind a 128-byte run of spaces and all 4096 positions are searched with no
// barrier at all.  The two quirks of the last 128-byte chunk are kept exactly:
// the scan is shortened to max(1, 127-tx) window bytes and the look-ahead wraps
// into the previous chunk (stale ring half) past the end of the packet
// (gpu_compress.cu:120,149,303,313-317).
#include "glc_device.h"
#include "glc_internal.h"
#include "culzss_internal.h"

namespace glc {

// ---------------------------------------------------------------------------
// The 127 steps of one position's window scan, written out by hand.
//   st = window index where the current run of equal bytes started, R = A0 - st with A0 = LDS address of la[0]:
//   step k reads la[k - st] at R + k (k is the instruction's immediate offset), and only a MISMATCH changes R
//   (st = k + 1), which v_cmpx + one subtraction under the narrowed EXEC do: no select, no address arithmetic.
//   After step k the run length is j = k + 1 - st, so (R << 16) + ((k + 1) << 16 | 0xFFFF - k) orders records by
//   (A0 + j, earlier k first); the records of two steps go into one v_max3.  3.5 VALU operations + one LDS byte
//   read per step.  (Left to the compiler the loop became ~10 VALU per step: the 127 compare masks were computed
//   up front and parked in VGPR lanes with v_writelane / v_readlane.)
// ---------------------------------------------------------------------------
template <int K, int SH>
__device__ __forceinline__ void lz_scan(const uint32_t (&W)[33], uint32_t A0, uint32_t &R, uint32_t &best, uint32_t &c1,
                                        uint32_t &c2, uint64_t ex)
{
    if constexpr (K < 127) {
        uint32_t lb, C;
        if constexpr (K == 0) {
            asm volatile("ds_read_u8 %[lb], %[R]\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_cmpx_ne_u16_sdwa vcc, %[lb], %[w] src0_sel:DWORD src1_sel:BYTE_%[b]\n"
                         "v_subrev_u32_e32 %[R], 1, %[A0]\n"
                         "s_mov_b64 exec, %[ex]\n"
                         : [R] "+v"(R), [lb] "=&v"(lb) : [w] "v"(W[SH >> 2]), [A0] "v"(A0), [ex] "s"(ex), [b] "n"(SH & 3) : "vcc");
        } else if constexpr ((K & 1) == 1) {                  // record of step K - 1 (even) -> c1
            asm volatile("ds_read_u8 %[lb], %[R] offset:%[k]\n"
                         "s_mov_b32 %[C], %[ck]\n"
                         "v_lshl_add_u32 %[c1], %[R], 16, %[C]\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_cmpx_ne_u16_sdwa vcc, %[lb], %[w] src0_sel:DWORD src1_sel:BYTE_%[b]\n"
                         "v_subrev_u32_e32 %[R], %[k1], %[A0]\n"
                         "s_mov_b64 exec, %[ex]\n"
                         : [R] "+v"(R), [c1] "=&v"(c1), [lb] "=&v"(lb), [C] "=&s"(C)
                         : [w] "v"(W[(K + SH) >> 2]), [A0] "v"(A0), [ex] "s"(ex), [k] "n"(K), [k1] "n"(K + 1), [b] "n"((K + SH) & 3),
                           [ck] "n"((K << 16) | (0xFFFF - (K - 1)))
                         : "vcc");
        } else {                                               // record of step K - 1 (odd) -> c2, both into best
            asm volatile("ds_read_u8 %[lb], %[R] offset:%[k]\n"
                         "s_mov_b32 %[C], %[ck]\n"
                         "v_lshl_add_u32 %[c2], %[R], 16, %[C]\n"
                         "v_max3_u32 %[best], %[best], %[c1], %[c2]\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_cmpx_ne_u16_sdwa vcc, %[lb], %[w] src0_sel:DWORD src1_sel:BYTE_%[b]\n"
                         "v_subrev_u32_e32 %[R], %[k1], %[A0]\n"
                         "s_mov_b64 exec, %[ex]\n"
                         : [R] "+v"(R), [best] "+v"(best), [c2] "=&v"(c2), [lb] "=&v"(lb), [C] "=&s"(C)
                         : [w] "v"(W[(K + SH) >> 2]), [A0] "v"(A0), [ex] "s"(ex), [c1] "v"(c1), [k] "n"(K), [k1] "n"(K + 1),
                           [b] "n"((K + SH) & 3), [ck] "n"((K << 16) | (0xFFFF - (K - 1)))
                         : "vcc");
        }
        lz_scan<K + 1, SH>(W, A0, R, best, c1, c2, ex);
    } else {                                                   // record of step 126 (even)
        uint32_t C;
        asm volatile("s_mov_b32 %[C], %[ck]\n"
                     "v_lshl_add_u32 %[c1], %[R], 16, %[C]\n"
                     "v_max_u32_e32 %[best], %[best], %[c1]\n"
                     : [best] "+v"(best), [c1] "=&v"(c1), [C] "=&s"(C)
                     : [R] "v"(R), [ck] "n"((127 << 16) | (0xFFFF - 126)));
    }
}

// The same scan for a position of the LAST 128-byte chunk, where the reference stops after max(1, 127 - tx) window
// bytes (gpu_compress.cu:120,149): step K runs under EXEC = lanes with K < iters, everything else as above (one record
// per step, so the result is complete whenever the wave stops).
template <int K>
__device__ __forceinline__ void lz_scan_tail(const uint32_t (&W)[33], uint32_t A0, uint32_t iters, uint32_t &R, uint32_t &best,
                                             uint64_t ex)
{
    if constexpr (K < 127) {
        if constexpr (K % 16 == 0 && K > 0)
            if (__builtin_amdgcn_ballot_w64(iters > (uint32_t)K) == 0) return;               // wave-uniform
        uint32_t lb, C, c;
        uint64_t M;
        asm volatile("v_cmp_lt_u32_e32 vcc, %[k], %[it]\n"
                     "s_and_b64 exec, %[ex], vcc\n"
                     "ds_read_u8 %[lb], %[R] offset:%[k]\n"
                     "s_mov_b64 %[M], exec\n"
                     "s_mov_b32 %[C], %[ck]\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     "v_cmpx_ne_u16_sdwa vcc, %[lb], %[w] src0_sel:DWORD src1_sel:BYTE_%[b]\n"
                     "v_subrev_u32_e32 %[R], %[k1], %[A0]\n"
                     "s_mov_b64 exec, %[M]\n"
                     "v_lshl_add_u32 %[c], %[R], 16, %[C]\n"
                     "v_max_u32_e32 %[best], %[best], %[c]\n"
                     "s_mov_b64 exec, %[ex]\n"
                     : [R] "+v"(R), [best] "+v"(best), [lb] "=&v"(lb), [c] "=&v"(c), [C] "=&s"(C), [M] "=&s"(M)
                     : [w] "v"(W[K >> 2]), [A0] "v"(A0), [it] "v"(iters), [ex] "s"(ex), [k] "n"(K), [k1] "n"(K + 1),
                       [b] "n"(K & 3), [ck] "n"(((K + 1) << 16) | (0xFFFF - K))
                     : "vcc");
        lz_scan_tail<K + 1>(W, A0, iters, R, best, ex);
    }
}

// ---------------------------------------------------------------------------
// match search: one workgroup (256 threads) per packet, 16 positions / thread
// ---------------------------------------------------------------------------
// One position p = p4 + SH (p4 a multiple of 4) outside the last 128-byte chunk: returns its candidate
// (c0 | c1 << 8).  W = the 33 aligned dwords text[p4-128 .. p4+3], shared by the four positions of a lane: window
// byte k of position p is byte k + SH of W, a byte select of the compare, so the window costs no funnel shift and a
// quarter of the LDS reads.
// FindMatch (gpu_compress.cu:104-168) restated without its flag and its branch: a run of equal bytes starting at
// window byte `st` has length j; the reference records a run when it ends, keeps the first longest (strict >) and
// restarts at la[0] on the byte AFTER the mismatch.  Recording (j, st) at every step through a maximum is the same
// thing: a run's last record dominates its earlier ones, longer beats shorter, earlier start beats later on ties.
__device__ __forceinline__ uint32_t lz_candidate(int length, int offset, uint32_t la0)
{
    if (length >= LZ_MAXC) length = LZ_MAXC - 1;
    return length <= 2 ? (1u | (la0 << 8)) : ((uint32_t)length | ((uint32_t)offset << 8));
}

template <int SH>
__device__ __forceinline__ uint32_t lz_position(const uint32_t (&W)[33], const uint8_t *s_buf, int p4)
{
    const int p = p4 + SH;
    const uint8_t *la = s_buf + LZ_WIN + p;                   // text[p + j]
    const uint32_t A0 = (uint32_t)(uintptr_t)la;              // low half of the flat address = LDS offset
    uint32_t R = A0, best = 0, c1 = 0, c2 = 0;
    lz_scan<0, SH>(W, A0, R, best, c1, c2, __builtin_amdgcn_ballot_w64(true));
    const int j = (int)(best >> 16) - (int)A0;
    int length = 1, offset = 1;
    if (j >= 2) { length = j; offset = (p + (int)(0xFFFFu - (best & 0xFFFFu)) + 1 - j) & 255; }
    return lz_candidate(length, offset, la[0]);
}

// A position of the last 128-byte chunk: the reference's scan is shortened to max(1, 127 - tx) window bytes
// (gpu_compress.cu:120,149,303).  (Its look-ahead would wrap into the stale ring half past the end of the packet,
// :313-317, but a run that starts at window byte st has compared j <= iters - 1 - st < 128 - tx bytes when the scan
// stops, so the wrapped bytes are never read.)
__device__ __forceinline__ uint32_t lz_position_last(const uint8_t *s_buf, int p)
{
    const int tx = p & 127;
    const uint8_t *la = s_buf + LZ_WIN + p;
    uint32_t W[33];
    {
        const uint32_t *wa = reinterpret_cast<const uint32_t *>(s_buf + (p & ~3));
        const uint32_t sh = (uint32_t)(p & 3);
        uint32_t prev = wa[0];
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const uint32_t nxt = wa[q + 1];
            W[q] = __builtin_amdgcn_alignbyte(nxt, prev, sh);
            prev = nxt;
        }
        W[32] = 0;
    }
    const uint32_t A0 = (uint32_t)(uintptr_t)la;
    uint32_t R = A0, best = 0;
    lz_scan_tail<0>(W, A0, (uint32_t)max(1, 127 - tx), R, best, __builtin_amdgcn_ballot_w64(true));
    const int j = (int)(best >> 16) - (int)A0;
    int length = 1, offset = 1;
    if (j >= 2) { length = j; offset = (p + (int)(0xFFFFu - (best & 0xFFFFu)) + 1 - j) & 255; }
    if (length > 128 - tx) length = 128 - tx;
    return lz_candidate(length, offset, la[0]);
}

__global__ __launch_bounds__(256, 8) void k_lzss_match(const uint8_t *__restrict__ in, uint8_t *__restrict__ cand)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[LZ_WIN + LZ_PCKT];
    const uint32_t pk = blockIdx.x, tid = threadIdx.x;
    const uint8_t *src = in + (size_t)pk * LZ_PCKT;
    if (tid < LZ_WIN / 4) reinterpret_cast<uint32_t *>(s_buf)[tid] = 0x20202020u;           // ' ' (gpu_compress.cu:208)
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        reinterpret_cast<uint4 *>(s_buf + LZ_WIN)[tid] = s4[tid];                             // 256 x 16 B = 4096
    }
    __syncthreads();
    uint8_t *dst = cand + (size_t)pk * 2 * LZ_PCKT;
    constexpr int QUADS = (LZ_PCKT - 128) / 4;                // groups of four adjacent positions before the last chunk
#pragma unroll 1
    for (int g = (int)tid; g < QUADS; g += 256) {
        const int p4 = g * 4;
        uint32_t W[33];
        const uint32_t *wa = reinterpret_cast<const uint32_t *>(s_buf + p4);
#pragma unroll
        for (int q = 0; q < 33; q++) W[q] = wa[q];
        const uint32_t r0 = lz_position<0>(W, s_buf, p4), r1 = lz_position<1>(W, s_buf, p4);
        const uint32_t r2 = lz_position<2>(W, s_buf, p4), r3 = lz_position<3>(W, s_buf, p4);
        *reinterpret_cast<uint2 *>(dst + 2 * p4) = make_uint2(r0 | (r1 << 16), r2 | (r3 << 16));
    }
    if (tid >= 192) {                                          // the last chunk: wave 3, which had half a trip less above
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            const int p = LZ_PCKT - 128 + 64 * half + (int)tid - 192;
            reinterpret_cast<uint16_t *>(dst)[p] = (uint16_t)lz_position_last(s_buf, p);
        }
    }
}

// ---------------------------------------------------------------------------
// token selection + packing: one workgroup per packet (aftercomp's inner loop)
//   stage[pk][0..size) = flag/token bytes of the packet, meta[pk] = (size, last group bytes)
//   Second choice since k_lzss_pack_wave (below): it takes the packets that kernel gives up on -- candidate streams
//   whose walks from different starts never fall into step -- at a cost that does not depend on the data.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lzss_pack(const uint8_t *__restrict__ cand, uint8_t *__restrict__ stage,
                                                   uint2 *__restrict__ meta, const uint32_t *__restrict__ need)
{
    if (need[blockIdx.x] == 0) return;                         // k_lzss_pack_wave has done this packet
    __shared__ __attribute__((aligned(16))) uint8_t s_c[2 * LZ_PCKT];
    __shared__ __attribute__((aligned(16))) uint16_t s_tok[LZ_PCKT];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[LZ_STAGE];
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_ntok;
    const uint32_t pk = blockIdx.x, tid = threadIdx.x;
    {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(cand + (size_t)pk * 2 * LZ_PCKT);
        reinterpret_cast<uint4 *>(s_c)[tid] = c4[tid];
        reinterpret_cast<uint4 *>(s_c)[tid + 256] = c4[tid + 256];
    }
    __syncthreads();
    // The greedy walk p -> p + len(p) from 0 (gpu_compress.cu:498-515) without a serial chain:
    // positions reachable in fewer than 2^r steps are marked round by round while the jump table is
    // squared (J <- J o J), at most 12 rounds for 4096 positions; the marked positions, compacted in order,
    // are the tokens.  (One lane walking ~1500 dependent LDS reads held the other 255 idle.  Also measured: one WAVE
    // per packet walking on the scalar unit -- candidates in 64 VGPRs, a step = v_readlane + 5 SALU + the token's
    // store -- with no barrier anywhere: ~115 cycles per token, 7.8 ms per GiB against 6.1 ms for the rounds below,
    // because 129 VGPRs and 12.6 KB of LDS per packet leave only 12 such chains per CU.)
    {
        __shared__ __attribute__((aligned(16))) uint16_t s_j1[LZ_PCKT];
        __shared__ __attribute__((aligned(16))) uint8_t mark[LZ_PCKT];   // one byte per position: plain stores, no atomics
        uint16_t *J0 = s_tok, *J1 = s_j1;                     // s_tok is free until the tokens are compacted
#pragma unroll
        for (int r = 0; r < LZ_PCKT / 256; r++) {
            const uint32_t p = r * 256 + tid;
            const uint32_t c0 = s_c[2 * p];
            const uint32_t nx = p + ((c0 <= 1) ? 1u : c0);
            J0[p] = (uint16_t)(nx < LZ_PCKT ? nx : LZ_PCKT);
        }
        reinterpret_cast<uint4 *>(mark)[tid] = make_uint4(tid == 0 ? 1u : 0u, 0u, 0u, 0u);
        __syncthreads();
        uint16_t *J = J0, *Jn = J1;
        for (int round = 0; round < 12; round++) {
            // four consecutive positions per thread and trip: their table entries and flags are one 8-byte and one
            // 4-byte LDS read, the squared entries one 8-byte write; only the J[a] gathers are per position
            uint2 jn[LZ_PCKT / 1024];
#pragma unroll
            for (int r = 0; r < LZ_PCKT / 1024; r++) {
                const uint32_t p = r * 1024 + tid * 4;
                const uint2 a2 = *reinterpret_cast<const uint2 *>(J + p);
                const uint32_t m4 = *reinterpret_cast<const uint32_t *>(mark + p);
                const uint32_t a[4] = {a2.x & 0xFFFFu, a2.x >> 16, a2.y & 0xFFFFu, a2.y >> 16};
                uint32_t g[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (((m4 >> (8 * i)) & 1u) && a[i] < LZ_PCKT) mark[a[i]] = 1;
                    g[i] = a[i] < LZ_PCKT ? (uint32_t)J[a[i]] : (uint32_t)LZ_PCKT;
                }
                jn[r] = make_uint2(g[0] | (g[1] << 16), g[2] | (g[3] << 16));
            }
#pragma unroll
            for (int r = 0; r < LZ_PCKT / 1024; r++) *reinterpret_cast<uint2 *>(Jn + r * 1024 + tid * 4) = jn[r];
            __syncthreads();
            uint16_t *x = J; J = Jn; Jn = x;
            if (J[0] >= LZ_PCKT) break;                       // the walk from 0 ends within 2^(round+1) steps: all marked
        }
        // compact: thread t owns positions [16t, 16t+16)
        uint32_t mw;
        {
            const uint4 m4 = reinterpret_cast<const uint4 *>(mark)[tid];                    // 16 flag bytes -> 16 bits
            mw = ((m4.x * 0x01020408u) >> 24) | (((m4.y * 0x01020408u) >> 24) << 4) |
                 (((m4.z * 0x01020408u) >> 24) << 8) | (((m4.w * 0x01020408u) >> 24) << 12);
        }
        uint32_t tot = 0;
        const uint32_t pre = block_excl_add<256>((uint32_t)__popc(mw), s_tmp, &tot);
        __syncthreads();                                      // everyone has read J / marks: s_tok is reused
        uint32_t o = pre, m2 = mw;
        while (m2) {
            const uint32_t bit = (uint32_t)__builtin_ctz(m2);
            m2 &= m2 - 1;
            s_tok[o++] = (uint16_t)(tid * 16 + bit);
        }
        if (tid == 0) s_ntok = tot;
    }
    __syncthreads();
    const uint32_t T = s_ntok;
    // 16 consecutive tokens (= 2 flag groups) per thread
    const uint32_t k0 = tid * 16;
    uint32_t sz = 0, litmask = 0;
    uint16_t pos[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t k = k0 + j;
        pos[j] = 0;
        if (k < T) {
            pos[j] = s_tok[k];
            const bool lit = s_c[2 * pos[j]] == 1;
            if (lit) litmask |= 1u << j;
            sz += lit ? 1 : 2;
        }
    }
    uint32_t total = 0;
    uint32_t pre = block_excl_add<256>(sz, s_tmp, &total);
    const uint32_t ngroups = (T + 7) / 8;
    uint32_t o = pre + k0 / 8;                                // flag bytes of all earlier groups
    uint32_t last_group_bytes = 0;
#pragma unroll
    for (int g = 0; g < 2; g++) {
        if (k0 + 8 * g < T) {
            const uint32_t gstart = o;
            s_out[o++] = (uint8_t)(litmask >> (8 * g));       // flags, LSB first (gpu_compress.cu:505,529-531)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t k = k0 + 8 * g + j;
                if (k < T) {
                    const uint32_t p = pos[8 * g + j];
                    if ((litmask >> (8 * g + j)) & 1) s_out[o++] = s_c[2 * p + 1];
                    else { s_out[o++] = s_c[2 * p]; s_out[o++] = s_c[2 * p + 1]; }
                }
            }
            if (k0 / 8 + g == ngroups - 1) last_group_bytes = o - gstart;
        }
    }
    __syncthreads();
    const uint32_t size = total + ngroups;
    if (last_group_bytes) meta[pk] = make_uint2(size, last_group_bytes);
    uint8_t *dst = stage + (size_t)pk * LZ_STAGE;
    for (uint32_t i = tid; i < (size + 3) / 4; i += 256)
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(s_out)[i];
}

// ---------------------------------------------------------------------------
// token selection + packing, one WAVE per packet (four packets per workgroup, no barrier).
//   The greedy walk p -> p + len(p) from 0 (gpu_compress.cu:498-515) is a chain of ~1800 dependent steps, but two
//   walks that start at different positions fall into step within a few tokens.  So every lane first walks its own
//   64 positions from their START (speculative marks, 64 bits in registers); then it walks again from the position
//   where the real chain enters its segment -- first guess: where its left neighbour's speculative walk left -- only
//   until it lands on a speculative mark: from there on the speculative marks ARE the chain.  A lane whose real walk
//   leaves without meeting its marks hands its neighbour a new entry and that one walks again, until nothing
//   changes (lane 0's entry is exact, so at most 64 trips; one or two in practice, and after LP_MAX_TRIPS the packet
//   is left to k_lzss_pack above).  Tokens are then compacted in place (rank <= position) and the flag groups are
//   sized and written 64 at a time.
//   Candidates sit in LDS with one pad dword per 64 positions: the lanes' private walks are 132 bytes apart and
//   spread over the banks.
// ---------------------------------------------------------------------------
constexpr int LP_WAVES = 4;
constexpr int LP_MAX_TRIPS = 6;                                    // real walks per lane before the packet is handed over
constexpr int LP_CAND_BYTES = 2 * LZ_PCKT + 4 * (LZ_PCKT / 64);     // 8448

__device__ __forceinline__ uint32_t lp_off(uint32_t p) { return 2u * p + 4u * (p >> 6); }     // byte offset of position p

__global__ __launch_bounds__(LP_WAVES * 64) void k_lzss_pack_wave(const uint8_t *__restrict__ cand, uint8_t *__restrict__ stage,
                                                                  uint2 *__restrict__ meta, uint32_t total_pk,
                                                                  uint32_t *__restrict__ need)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_cand[LP_WAVES][LP_CAND_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[LP_WAVES][LZ_STAGE];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t pk = blockIdx.x * LP_WAVES + wave;
    if (pk >= total_pk) return;                                // wave-uniform; nothing below synchronises waves
    uint8_t *CD = s_cand[wave];
    uint8_t *OUT = s_out[wave];
    {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(cand + (size_t)pk * 2 * LZ_PCKT);
#pragma unroll
        for (int i = 0; i < 2 * LZ_PCKT / 16 / 64; i++) {
            const uint32_t q = i * 64 + lane;                  // 8 positions, inside one segment
            const uint4 v = c4[q];
            uint32_t *d = reinterpret_cast<uint32_t *>(CD + lp_off(q * 8));
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    auto next = [&](uint32_t p) { const uint32_t c0 = CD[lp_off(p)]; return p + (c0 <= 1u ? 1u : c0); };
    const uint32_t lo = lane * 64, hi = lo + 64;
    // 1. speculative walk from the segment's start
    uint64_t spec = 0;
    uint32_t spec_exit = lo;
    while (spec_exit < hi) { spec |= 1ull << (spec_exit - lo); spec_exit = next(spec_exit); }
    // 2. the real walk, from where the chain enters the segment, until it meets the speculative marks
    uint64_t mine = 0;
    uint32_t out = 0, cur = 0xFFFFFFFFu;
    uint32_t want = wave_prev(spec_exit);                      // lane 0: 0
    for (int trips = 0;; trips++) {
        const bool redo = want != cur;
        if (__builtin_amdgcn_ballot_w64(redo) == 0) break;
        if (trips == LP_MAX_TRIPS) {                           // walks that do not meet: leave the packet to k_lzss_pack
            if (lane == 0) need[pk] = 1;
            return;
        }
        if (redo) {
            cur = want;
            uint32_t p = cur;
            uint64_t m = 0;
            while (p < hi) {
                const uint64_t bit = 1ull << (p - lo);
                if (spec & bit) { m |= spec & (0ull - bit); p = spec_exit; break; }
                m |= bit;
                p = next(p);
            }
            mine = m; out = p;
        }
        want = wave_prev(out);
    }
    if (lane == 0) need[pk] = 0;
    // 3. compaction in place: token r of the packet <- candidate word of the r-th marked position
    uint16_t *TK = reinterpret_cast<uint16_t *>(CD);
    uint32_t T = 0;
    const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
    for (int c = 0; c < LZ_PCKT / 64; c++) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)mlo, c), b = (uint32_t)__builtin_amdgcn_readlane((int)mhi, c);
        if ((a | b) == 0) continue;
        const uint32_t v = *reinterpret_cast<const uint16_t *>(CD + lp_off(c * 64 + lane));
        const uint32_t r = T + __builtin_amdgcn_mbcnt_hi(b, __builtin_amdgcn_mbcnt_lo(a, 0));
        if ((lane < 32 ? a >> lane : b >> (lane - 32)) & 1u) TK[r] = (uint16_t)v;
        T += (uint32_t)__popc(a) + (uint32_t)__popc(b);
    }
    const uint32_t ngroups = (T + 7) / 8;
    // 4. groups g = g0 + lane: eight token words = one 16-byte LDS read
    uint32_t base = 0, last_group_bytes = 0;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 64) {
        const uint32_t g = g0 + lane;
        uint32_t w[4] = {0, 0, 0, 0};
        uint32_t nt = 0;
        if (g < ngroups) {
            const uint4 q = *reinterpret_cast<const uint4 *>(TK + g * 8);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            nt = min(8u, T - g * 8);
        }
        uint32_t flags = 0, sz = nt ? 1u : 0u;                // the flag byte
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t t = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
            const bool lit = (t & 0xFFu) == 1u;
            if ((uint32_t)j < nt) { flags |= (lit ? 1u : 0u) << j; sz += lit ? 1u : 2u; }
        }
        const uint32_t inc = wave_incl_add(sz);
        if (nt) {
            uint32_t o = base + inc - sz;
            const uint32_t gstart = o;
            OUT[o++] = (uint8_t)flags;                         // flags, LSB first (gpu_compress.cu:505,529-531)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t t = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                if ((uint32_t)j < nt) {
                    if ((flags >> j) & 1u) OUT[o++] = (uint8_t)(t >> 8);
                    else { OUT[o++] = (uint8_t)t; OUT[o++] = (uint8_t)(t >> 8); }
                }
            }
            if (g == ngroups - 1) last_group_bytes = o - gstart;
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    const uint32_t size = base;
    if (last_group_bytes) meta[pk] = make_uint2(size, last_group_bytes);
    uint8_t *dst = stage + (size_t)pk * LZ_STAGE;
    for (uint32_t i = lane; i < (size + 3) / 4; i += 64)
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(OUT)[i];
}

// ---------------------------------------------------------------------------
// per buffer: packet offsets, "took more" test, trailer.  One workgroup/buffer.
//   sizes[buf] = packed length incl. trailer, or 0 = store raw
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lzss_layout(const uint2 *__restrict__ meta, uint32_t npk, int buf_length,
                                                     uint32_t *__restrict__ pk_off, uint8_t *__restrict__ packed,
                                                     size_t pack_stride, int *__restrict__ sizes)
{
    __shared__ uint32_t s_tmp[8];
    __shared__ uint32_t s_run;
    const uint32_t bufi = blockIdx.x, tid = threadIdx.x;
    const uint2 *M = meta + (size_t)bufi * npk;
    uint32_t *PO = pk_off + (size_t)bufi * npk;
    uint8_t *P = packed + (size_t)bufi * pack_stride;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < npk; base += 256) {
        const uint32_t i = base + tid;
        const uint32_t sz = i < npk ? M[i].x : 0;
        uint32_t tot = 0;
        const uint32_t ex = block_excl_add<256>(sz, s_tmp, &tot);
        const uint32_t run = s_run;
        if (i < npk) PO[i] = run + ex;
        __syncthreads();
        if (tid == 0) s_run = run + tot;
        __syncthreads();
    }
    const uint32_t total = s_run;
    // aftercomp aborts when the bytes flushed BEFORE the last token iteration exceed
    // buf_length (gpu_compress.cu:492-497): everything but the last group of the last packet
    // ... and (not in the reference, which then writes its trailer past the end of the caller's buffer,
    // gpu_compress.cu:620-657, and whose container cannot tell a packed buffer of exactly BUFSIZE bytes from
    // a raw one, deculzss.c:94-95): a packed form that is not SMALLER than the buffer is stored raw too
    const bool ok = (total - M[npk - 1].y) <= (uint32_t)buf_length && total + 2 * npk + 6 < (uint32_t)buf_length;
    if (ok) {
        for (uint32_t i = tid; i < npk; i += 256) {           // packet sizes, big-endian u16 (:626-634)
            const uint32_t sz = M[i].x;
            P[total + 2 * i] = (uint8_t)(sz >> 8);
            P[total + 2 * i + 1] = (uint8_t)sz;
        }
        if (tid == 0) {
            uint8_t *t = P + total + 2 * npk;
            t[0] = (uint8_t)(buf_length >> 24); t[1] = (uint8_t)(buf_length >> 16);
            t[2] = (uint8_t)(buf_length >> 8);  t[3] = (uint8_t)buf_length;
            t[4] = 0; t[5] = 0;                                // pad size (:648-655)
            sizes[bufi] = (int)(total + 2 * npk + 6);
        }
    } else if (tid == 0) sizes[bufi] = 0;
}

__global__ __launch_bounds__(256) void k_lzss_gather(const uint8_t *__restrict__ stage, const uint2 *__restrict__ meta,
                                                     const uint32_t *__restrict__ pk_off, uint32_t npk,
                                                     uint8_t *__restrict__ packed, size_t pack_stride,
                                                     const int *__restrict__ sizes,
                                                     const uint8_t *__restrict__ raw_in)
{
    const uint32_t bufi = blockIdx.y, pk = blockIdx.x, tid = threadIdx.x;
    if (sizes[bufi] == 0) {                                    // "store raw" (culzss.c:177-181): keep the input
        if (raw_in) {
            const size_t o = (size_t)pk * LZ_PCKT;
            reinterpret_cast<uint4 *>(packed + (size_t)bufi * pack_stride + o)[tid] =
                reinterpret_cast<const uint4 *>(raw_in + ((size_t)bufi * npk) * LZ_PCKT + o)[tid];
        }
        return;
    }
    const size_t gp = (size_t)bufi * npk + pk;
    const uint32_t sz = meta[gp].x, off = pk_off[gp];
    const uint8_t *s = stage + gp * LZ_STAGE;
    uint8_t *d = packed + (size_t)bufi * pack_stride + off;
    // the destination starts at any byte: bytes up to its first 16-byte boundary, then aligned 16-byte stores fed by
    // unaligned 16-byte loads, then the tail (a byte per thread and trip was 0.85 ms per GiB, twice a copy's time)
    const uint32_t head = min(sz, (uint32_t)((0u - (uint32_t)reinterpret_cast<uintptr_t>(d)) & 15u));
    if (tid < head) d[tid] = s[tid];
    const uint32_t nvec = (sz - head) / 16;
    for (uint32_t i = tid; i < nvec; i += 256) {
        uint4 v;
        __builtin_memcpy(&v, s + head + 16 * i, 16);
        *reinterpret_cast<uint4 *>(d + head + 16 * i) = v;
    }
    const uint32_t done = head + 16 * nvec;
    if (done + tid < sz) d[done + tid] = s[done + tid];
}

// ---------------------------------------------------------------------------
// decode: one wave per packet; the token stream is parsed uniformly by the whole
// wave, match copies are lane-parallel (the window is a snapshot: bytes a match
// reads are always older than the bytes it writes, gpu_decompress.cu:217-236)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lzss_decode(const uint8_t *__restrict__ packed, size_t pack_stride,
                                                    const int *__restrict__ sizes, int buf_length,
                                                    uint8_t *__restrict__ out, int *__restrict__ d_err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_in[LZ_STAGE + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_o[LZ_WIN + LZ_PCKT + 256];
    const uint32_t bufi = blockIdx.y, pk = blockIdx.x, l = threadIdx.x;
    const uint32_t npk = (uint32_t)buf_length / LZ_PCKT;
    const int clen = sizes[bufi];
    const uint8_t *P = packed + (size_t)bufi * pack_stride;
    uint8_t *O = out + (size_t)bufi * buf_length + (size_t)pk * LZ_PCKT;
    if (clen == 0) {                                           // stored raw: slot holds the input bytes
        for (uint32_t i = l; i < LZ_PCKT / 16; i += 64)
            reinterpret_cast<uint4 *>(O)[i] = reinterpret_cast<const uint4 *>(P + (size_t)pk * LZ_PCKT)[i];
        return;
    }
    // a stream that cannot hold its own trailer, or is longer than its slot, is corrupt: the reference
    // trusts it (gpu_decompress.cu:257-294); here nothing outside the slot is read, the packet decodes to
    // zeros and the error word is raised
    if (clen < (int)(2 * npk + 6) || (size_t)clen > pack_stride) {
        if (d_err && l == 0) atomicOr(d_err, 1);
        for (uint32_t i = l; i < LZ_PCKT / 16; i += 64) reinterpret_cast<uint4 *>(O)[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    // trailer: npk big-endian u16 sizes, u32 length, u16 pad (gpu_decompress.cu:257-294)
    const uint32_t body = (uint32_t)clen - 6 - 2 * npk;
    const uint8_t *tr = P + body;
    uint32_t start = 0;
    {   // prefix of the packet sizes: all byte loads in flight together
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t i = r * 64 + l;
            hi[r] = tr[2 * (i < pk ? i : 0u)]; lo[r] = tr[2 * (i < pk ? i : 0u) + 1];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) if (r * 64 + l < pk) start += (hi[r] << 8) | lo[r];
        for (uint32_t i = 256 + l; i < pk; i += 64) start += ((uint32_t)tr[2 * i] << 8) | tr[2 * i + 1];
    }
    start = wave_sum(start);
    uint32_t size = ((uint32_t)tr[2 * pk] << 8) | tr[2 * pk + 1];
    if (size > LZ_STAGE) size = LZ_STAGE;
    if (start > body || start + size > body) {                 // packet sizes that do not add up: stay inside the body
        if (d_err && l == 0) atomicOr(d_err, 1);
        start = min(start, body); size = min(size, body - start);
    }
    // the packet's bytes: 16-byte loads from the aligned address below `start`, all in flight, then LDS;
    // s_in[sh + i] = stream byte i (a byte-per-trip copy loop costs one memory latency per 64 bytes)
    const uint8_t *src = P + start;
    uint32_t sh = (uint32_t)(reinterpret_cast<size_t>(src) & 15);
    if (src - sh < packed) {                                  // caller's buffer is not 16-byte aligned: byte copy
        sh = 0;
        for (uint32_t i = l; i < size; i += 64) s_in[i] = src[i];
    } else {
        const uint4 *a16 = reinterpret_cast<const uint4 *>(src - sh);
        const uint32_t n16 = (sh + size + 15) / 16;           // <= (15 + 4608 + 15) / 16 = 289
        uint4 q[5];
#pragma unroll
        for (int r = 0; r < 5; r++) { const uint32_t i = r * 64 + l; q[r] = a16[i < n16 ? i : 0u]; }
#pragma unroll
        for (int r = 0; r < 5; r++) { const uint32_t i = r * 64 + l; if (i < n16) reinterpret_cast<uint4 *>(s_in)[i] = q[r]; }
    }
    for (uint32_t i = l; i < LZ_WIN / 4; i += 64) reinterpret_cast<uint32_t *>(s_o)[i] = 0x20202020u;
    __syncthreads();
    // token parse, one flag group per trip: the group's flag byte and its <= 16 token bytes are read
    // with ONE LDS access (a byte per lane) and then picked out of the register with v_readlane, so
    // the only LDS round trip a token still waits for is the source read of a match copy
    uint32_t fp = 0, wp = 0;
    const uint8_t *sb = s_in + sh;
    bool done = false;
    while (!done) {
        if (fp >= size) break;
        const uint32_t v = (fp + l < size + 0u) ? (uint32_t)sb[fp + l] : 0u;
        uint32_t flags = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
        if (fp + 17 <= size) {
            // whole group present: lane t < 8 owns token t.  Stream position and output position of
            // every token come from the flag byte and an 8-lane scan of the token lengths; all
            // literals of the group go out in ONE store, only the matches are walked in order
            // (a match reads output produced before it -- earlier literals are already in place,
            // later ones do not overlap its source or destination).
            const uint32_t below = flags & ((1u << (l & 7)) - 1u);
            const uint32_t pos_t = 1u + 2u * (l & 7) - (uint32_t)__popc(below);
            const bool lit_t = (flags >> (l & 7)) & 1u;
            const uint32_t b0v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos_t << 2), (int)v);
            const uint32_t len_t = (l < 8) ? (lit_t ? 1u : b0v) : 0u;
            uint32_t inc = len_t;                                           // inclusive scan over lanes 0..7
            inc += GLC_DPP(inc, 0x111, 0xf);
            inc += GLC_DPP(inc, 0x112, 0xf);
            inc += GLC_DPP(inc, 0x114, 0xf);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 7);
            if (wp + total <= LZ_PCKT) {
                const uint32_t wp_t = wp + inc - len_t;
                if (l < 8 && lit_t) s_o[LZ_WIN + wp_t] = (uint8_t)b0v;
                uint32_t mm = ~flags & 0xFFu;
                while (mm) {
                    const uint32_t t = (uint32_t)__builtin_ctz(mm);
                    mm &= mm - 1;
                    const uint32_t pt = (uint32_t)__builtin_amdgcn_readlane((int)pos_t, t);
                    const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)v, pt);
                    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)v, pt + 1);
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wp_t, t);
                    uint32_t d = (w0 - off) & 127u;
                    if (d == 0) d = 128;
                    const uint32_t q = LZ_WIN + w0 - d;
                    uint8_t c0 = 0, c1 = 0;
                    const uint32_t i0 = l, i1 = l + 64;
                    if (i0 < len) c0 = s_o[(i0 < d) ? q + i0 : q + i0 - 128];
                    if (i1 < len) c1 = s_o[(i1 < d) ? q + i1 : q + i1 - 128];
                    __builtin_amdgcn_wave_barrier();
                    if (i0 < len) s_o[LZ_WIN + w0 + i0] = c0;
                    if (i1 < len) s_o[LZ_WIN + w0 + i1] = c1;
                    __builtin_amdgcn_wave_barrier();
                }
                wp += total;
                fp += 1u + 16u - (uint32_t)__popc(flags & 0xFFu);
                continue;
            }
        }
        uint32_t pos = 1;
#pragma unroll 1
        for (int t = 0; t < 8; t++) {
            if (fp + pos >= size || wp >= LZ_PCKT) { done = true; break; }
            if (flags & 1) {
                const uint32_t byte = (uint32_t)__builtin_amdgcn_readlane((int)v, pos);
                if (l == 0) s_o[LZ_WIN + wp] = (uint8_t)byte;
                pos++; wp++;
            } else {
                if (fp + pos + 1 >= size) { done = true; break; }
                const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)v, pos);
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)v, pos + 1);
                pos += 2;
                uint32_t d = (wp - off) & 127u;                // distance back to the ring slot `off`
                if (d == 0) d = 128;
                const uint32_t q = LZ_WIN + wp - d;            // s_o index of the first source byte
                uint8_t b0 = 0, b1 = 0;
                const uint32_t i0 = l, i1 = l + 64;
                if (i0 < len) b0 = s_o[(i0 < d) ? q + i0 : q + i0 - 128];
                if (i1 < len) b1 = s_o[(i1 < d) ? q + i1 : q + i1 - 128];
                __builtin_amdgcn_wave_barrier();
                if (i0 < len && wp + i0 < LZ_PCKT) s_o[LZ_WIN + wp + i0] = b0;
                if (i1 < len && wp + i1 < LZ_PCKT) s_o[LZ_WIN + wp + i1] = b1;
                wp += len;
            }
            flags >>= 1;
            __builtin_amdgcn_wave_barrier();
        }
        fp += pos;
    }
    __syncthreads();
    for (uint32_t i = l; i < LZ_PCKT / 16; i += 64)
        reinterpret_cast<uint4 *>(O)[i] = reinterpret_cast<const uint4 *>(s_o + LZ_WIN)[i];
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
KernelProf &lzss_prof()
{
    static KernelProf pr;
    static bool named = false;
    if (!named) {
        named = true;
        pr.name[LZP_MATCH] = "k_lzss_match"; pr.name[LZP_PACK] = "k_lzss_pack_wave+k_lzss_pack";
        pr.name[LZP_GATHER] = "k_lzss_layout+k_lzss_gather"; pr.name[LZP_DECODE] = "k_lzss_decode";
    }
    return pr;
}

size_t lzss_work_bytes(int buf_length, int nbuf)
{
    const size_t npk = (size_t)(buf_length / LZ_PCKT) * nbuf;
    return npk * (LZ_STAGE + sizeof(uint2) + sizeof(uint32_t)) + (size_t)2 * buf_length * nbuf + 256;
}

static void carve(void *work, int buf_length, int nbuf, uint8_t **stage, uint2 **meta, uint32_t **pk_off,
                  uint8_t **cand)
{
    const size_t npk = (size_t)(buf_length / LZ_PCKT) * nbuf;
    uint8_t *w = (uint8_t *)work;
    *stage = w; w += npk * LZ_STAGE;
    *meta = (uint2 *)w; w += npk * sizeof(uint2);
    *pk_off = (uint32_t *)w; w += npk * sizeof(uint32_t);
    w = (uint8_t *)(((uintptr_t)w + 255) & ~(uintptr_t)255);
    *cand = w;
}

hipError_t lzss_encode(hipStream_t st, const uint8_t *d_in, int buf_length, int nbuf, uint8_t *d_cand,
                       uint8_t *d_packed, int *d_sizes, void *d_work)
{
    if (buf_length <= 0 || buf_length % LZ_PCKT || nbuf <= 0) return hipErrorInvalidValue;
    uint8_t *stage, *cand; uint2 *meta; uint32_t *pk_off;
    carve(d_work, buf_length, nbuf, &stage, &meta, &pk_off, &cand);
    if (d_cand) cand = d_cand;
    const uint32_t npk = buf_length / LZ_PCKT;
    KernelProf &pr = lzss_prof();
    const int pi = pr.begin(LZP_MATCH, st);
    hipLaunchKernelGGL(k_lzss_match, dim3(npk * nbuf), dim3(256), 0, st, d_in, cand);
    pr.end(pi, (double)buf_length * nbuf, st);
    return lzss_pack(st, cand, buf_length, nbuf, d_packed, d_sizes, d_work, d_in);
}

hipError_t lzss_pack(hipStream_t st, const uint8_t *d_cand, int buf_length, int nbuf, uint8_t *d_packed,
                     int *d_sizes, void *d_work, const uint8_t *d_raw_in)
{
    if (buf_length <= 0 || buf_length % LZ_PCKT || nbuf <= 0) return hipErrorInvalidValue;
    uint8_t *stage, *cand; uint2 *meta; uint32_t *pk_off;
    carve(d_work, buf_length, nbuf, &stage, &meta, &pk_off, &cand);
    const uint32_t npk = buf_length / LZ_PCKT;
    const size_t stride = lzss_pack_stride(buf_length);
    // pk_off doubles as the "left to k_lzss_pack" flags until k_lzss_layout writes the offsets
    KernelProf &pr = lzss_prof();
    const double units = (double)buf_length * nbuf;
    int pi = pr.begin(LZP_PACK, st);
    hipLaunchKernelGGL(k_lzss_pack_wave, dim3((npk * nbuf + LP_WAVES - 1) / LP_WAVES), dim3(LP_WAVES * 64), 0, st, d_cand, stage,
                       meta, npk * nbuf, pk_off);
    hipLaunchKernelGGL(k_lzss_pack, dim3(npk * nbuf), dim3(256), 0, st, d_cand, stage, meta, pk_off);
    pr.end(pi, units, st);
    pi = pr.begin(LZP_GATHER, st);
    hipLaunchKernelGGL(k_lzss_layout, dim3(nbuf), dim3(256), 0, st, meta, npk, buf_length, pk_off, d_packed, stride,
                       d_sizes);
    hipLaunchKernelGGL(k_lzss_gather, dim3(npk, nbuf), dim3(256), 0, st, stage, meta, pk_off, npk, d_packed, stride,
                       d_sizes, d_raw_in);
    pr.end(pi, units, st);
    return hipGetLastError();
}

// device memory -> PINNED host memory by a kernel (posted writes over PCIe, 16 bytes per lane).  Why not hipMemcpyAsync: a
// copy-engine transfer queued BEHIND kernels of its stream is only handed to the engine when the host next looks at that
// stream (measured with the reference's four-slot ring, culzss_ring_bench.c: every slot's copy-out started at its
// onestream_finish_GPU, so nothing overlapped: 4.7 GB/s; 10.6 with AMD_DIRECT_DISPATCH=0, where a runtime thread keeps the
// queue moving).  A kernel is just the next packet of the stream.
__global__ __launch_bounds__(256) void k_lzss_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                      const uint8_t *__restrict__ tail_src, uint8_t *__restrict__ tail_dst,
                                                      uint32_t tail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}

hipError_t lzss_copy_to_host(hipStream_t st, const void *d_src, void *h_dst, size_t bytes)
{
    if (!bytes) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(h_dst)) & 15) return hipErrorInvalidValue;
    const size_t n16 = bytes / 16;
    const uint32_t tail = (uint32_t)(bytes % 16);
    const uint32_t grid = (uint32_t)(n16 / 256 < 1 ? 1 : (n16 / 256 > 128 ? 128 : n16 / 256));
    hipLaunchKernelGGL(k_lzss_to_host, dim3(grid), dim3(256), 0, st, (const uint4 *)d_src, (uint4 *)h_dst, n16,
                       (const uint8_t *)d_src + n16 * 16, (uint8_t *)h_dst + n16 * 16, tail);
    return hipGetLastError();
}

hipError_t lzss_decode(hipStream_t st, const uint8_t *d_packed, const int *d_sizes, int buf_length, int nbuf,
                       uint8_t *d_out, int *d_err)
{
    if (buf_length <= 0 || buf_length % LZ_PCKT || nbuf <= 0) return hipErrorInvalidValue;
    const uint32_t npk = buf_length / LZ_PCKT;
    KernelProf &pr = lzss_prof();
    const int pi = pr.begin(LZP_DECODE, st);
    hipLaunchKernelGGL(k_lzss_decode, dim3(npk, nbuf), dim3(64), 0, st, d_packed, lzss_pack_stride(buf_length),
                       d_sizes, buf_length, d_out, d_err);
    pr.end(pi, (double)buf_length * nbuf, st);
    return hipGetLastError();
}

} // namespace glc
