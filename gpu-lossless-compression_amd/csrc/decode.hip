// decode.hip -- GPU decoder for the cudppCompress stream.  gfx950 / wave64.
//
// The reference has NO GPU decoder: the only decoder is the CPU gold routine of
// its test (cudpp-inpar/apps/cudpp_testrig/test_compress.cpp:192-364), whose
// semantics this file implements on the device:
//   1. rebuild the Huffman tree from d_hist (+EOF)           test_compress.cpp:203-238
//   2. per 4096-symbol block, walk the codes MSB-first        test_compress.cpp:240-291
//      starting at word 1+encodeOffset[b]
//   3. inverse MTF                                            test_compress.cpp:293-310
//   4. inverse BWT.  The gold's plain LF walk (:351-354) is only valid when the
//      input ends in a unique minimal byte; here the sentinel-aware form of
//      SURVEY.md 8(f)1 is used so that every input round-trips.
// Parity criterion = round trip (decode(encode(x)) == x) plus equality with the
// oracle's decoder on the same stream.
//
// MI355X design (details at each kernel; measurements in DESIGN.md section 5):
//   Huffman : 12-bit first-level LUT in LDS per workgroup (built once per block by the tree
//             kernel); one WAVE per 4096-symbol block: 64 lanes decode 64 spans speculatively
//             and re-synchronise (the blocks are independently addressable via d_encodeOffset).
//   iMTF    : the MTF index stream of a chunk defines a permutation of list POSITIONS
//             independent of the list contents: one LANE per chunk produces position bytes +
//             the chunk permutation, a scan composes the permutations, a LUT pass maps
//             positions to symbols.
//   iBWT    : LF mapping by a stable 257-bucket counting sort (tile histograms + wave64 ballot
//             ranking, no data movement); the LF cycle is cut at every row that is a multiple
//             of 128, each piece is walked ONCE emitting into a bounded slot, pieces are
//             ordered by list ranking in LDS and copied to their text positions.
#include <stdlib.h>
#include <type_traits>
#include "glc_device.h"
#include "glc_internal.h"
#include "huff_tree.h"

namespace glc {

constexpr int      DEC_LUT_BITS = 12;
constexpr uint32_t DEC_FLAG     = 0x80000000u;
#ifndef GLC_LF_TILE
#define GLC_LF_TILE 4096
#endif
constexpr int      LF_TILE      = GLC_LF_TILE;                 // rows per tile of the LF construction (a 512-bin histogram per tile: 1 KiB per 2 KiB of rows at 2048)
constexpr uint32_t LF_MASK      = (1u << 21) - 1;
#ifndef GLC_SPLIT
#define GLC_SPLIT 128
#endif
constexpr uint32_t SPLIT        = GLC_SPLIT;                   // LF-cycle rows between splitters
constexpr uint32_t MAX_SPLITS   = (1u << 20) / SPLIT + 8;
#ifndef GLC_SLOT
#define GLC_SLOT 256
#endif
constexpr uint32_t SLOT         = GLC_SLOT;                   // bytes a walk emits before it continues in a new segment
constexpr uint32_t EMIT_SEGS    = 4;                     // segments per wave in k_ibwt_emit

// ---------------------------------------------------------------------------
// 1. tree -> 12-bit LUT + node table.  One workgroup per block.
//    lut entry : leaf within 12 bits -> symbol | len << 16 ; else DEC_FLAG | node
//    node entry: leaf -> DEC_FLAG | symbol ; composite -> left | right << 16
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_prepare(const uint32_t *__restrict__ d_hist, uint32_t *__restrict__ lut,
                                                     uint32_t *__restrict__ nodes, uint32_t n, uint32_t *__restrict__ d_status)
{
    __shared__ uint32_t s_hist[257];
    __shared__ uint32_t s_sum[4];
    __shared__ HuffTreeLds T;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    // The histogram is stream data (possibly from another process).  huff_tree_build keys its leaves as
    // count << 9 | slot in 32 bits, i.e. it relies on count < 2^23: a histogram of n symbols has every count <= n and
    // the counts add up to n.  Anything else is reported, and the counts are clamped to n (<= 2^20) so that the tree
    // that is built anyway is structurally valid and nothing downstream indexes with garbage.
    uint32_t c = d_hist[(size_t)b * 256 + tid];
    const bool over = c > n;
    if (over) c = n;
    s_hist[tid] = c;
    if (tid == 0) s_hist[256] = 1;
    uint32_t sum = c;
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((tid & 63) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (d_status && (over || (tid == 0 && s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3] != n))) atomicOr(d_status, ST_CORRUPT);
    huff_tree_build<256>(T, s_hist, tid);
    __syncthreads();
    const int head = T.head, used = 2 * T.nl - 1;
    for (int s = (int)tid; s < HUFF_NODES; s += 256) {
        uint32_t e = 0;
        if (s < used) e = (T.left[s] < 0) ? (DEC_FLAG | (uint32_t)(uint16_t)T.value[s])
                                          : ((uint32_t)T.left[s] | ((uint32_t)T.right[s] << 16));
        nodes[(size_t)b * HUFF_NODES + s] = e;
    }
    for (uint32_t idx = tid; idx < (1u << DEC_LUT_BITS); idx += 256) {
        int node = head, len = 0;
        while (T.left[node] >= 0 && len < DEC_LUT_BITS) {
            node = ((idx >> (DEC_LUT_BITS - 1 - len)) & 1u) ? T.right[node] : T.left[node];
            len++;
        }
        lut[((size_t)b << DEC_LUT_BITS) + idx] =
            (T.left[node] < 0) ? ((uint32_t)(uint16_t)T.value[node] | ((uint32_t)len << 16)) : (DEC_FLAG | (uint32_t)node);
    }
}

// ---------------------------------------------------------------------------
// 2. Huffman decode: one WAVE per 4096-symbol block.
//    The block's words are staged in LDS and cut into 64 equal spans, one per lane.  Every lane
//    decodes its span from bit 0 (speculatively), then lanes whose predecessor ended at a
//    different bit than they assumed re-decode from there until the chain that starts at lane 0
//    is consistent -- Huffman codes resynchronise within a few codewords, so this takes 2-3
//    rounds in practice and at most 64 (lane i is final after round i).  A scan of the symbol
//    counts gives every lane its output index; a last decode writes the symbols through LDS so
//    the global store is coalesced.  (A lane per block -- the obvious mapping, and the one the
//    CPU gold uses -- leaves a 256-block batch with 1024 waves of 4096 dependent steps each.
//    Also measured: stopping a re-decode as soon as it lands on a recorded boundary of the first
//    pass (the trick that pays in hd_decode.hip) -- 1.92 ms against 1.39 ms here: a round still
//    lasts as long as its slowest lane, and the per-unit checks are paid by every lane.)
// ---------------------------------------------------------------------------
constexpr uint32_t DH_WAVES     = 4;
constexpr uint32_t DH_MAX_SPAN  = (HUFF_MAX_WORDS + 63) / 64;          // 24 words
constexpr uint32_t DH_WORDS_LDS = 66 * (DH_MAX_SPAN | 1u);             // spans padded to an odd pitch

struct DhTables { const uint16_t *lut; const uint32_t *nodes; };

// decode lane span [0, span_bits) from bit `o`; returns (count << 8) | overshoot.
// WRITE: symbols go to s_out[base + k] while < limit.
template <bool WRITE>
__device__ __forceinline__ uint32_t dh_decode_span(const uint32_t *s_w, uint32_t addr, uint32_t S, uint32_t P,
                                                   uint32_t o, const DhTables T, uint8_t *s_out, uint32_t base,
                                                   uint32_t limit)
{
    const uint32_t span_bits = S * 32;
    uint32_t left = S;
    auto next_word = [&]() -> uint32_t {
        const uint32_t x = s_w[addr];
        addr++;
        if (--left == 0) { addr += P - S; left = S; }
        return x;
    };
    uint64_t buf = (uint64_t)next_word() << 32;
    buf |= next_word();
    buf <<= o;
    uint32_t nb = 64 - o, pos = o, cnt = 0;
    while (pos < span_bits) {
        if (nb <= 32) { buf |= (uint64_t)next_word() << (32 - nb); nb += 32; }
        const uint32_t e = T.lut[(uint32_t)(buf >> (64 - DEC_LUT_BITS))];
        uint32_t len, sym;
        if (!(e & 0x8000u)) { sym = e & 0x3FFu; len = e >> 10; }
        else {
            uint32_t ne = T.nodes[e & 0x3FFu];
            len = DEC_LUT_BITS;
            while (!(ne & DEC_FLAG) && len < 33) {
                const uint32_t bit = (uint32_t)(buf >> (63 - len)) & 1u;
                ne = T.nodes[bit ? (ne >> 16) : (ne & 0xFFFF)];
                len++;
            }
            sym = ne & 0xFFFF;
        }
        len = len ? len : 1u;
        if (WRITE) { if (base + cnt < limit) s_out[base + cnt] = (uint8_t)sym; }
        buf <<= len; nb -= len; pos += len; cnt++;
    }
    return (cnt << 8) | (pos - span_bits);
}

// where block b's words are and how many belong to it.  Strided layout: slot b of comp_stride words.  Compact layout
// (block_off, nblk + 1 entries; comp_stride = words in the whole array): [block_off[b], block_off[b + 1]) -- offsets
// come from the stream's producer, possibly another process: a range that does not ascend or leaves the array is empty.
__device__ __forceinline__ void dec_block_slot(const unsigned long long *block_off, uint32_t b, size_t comp_stride,
                                               uint64_t &base, uint64_t &slot)
{
    if (!block_off) { base = (uint64_t)b * comp_stride; slot = comp_stride; return; }
    const uint64_t b0 = block_off[b], b1 = block_off[b + 1];
    const bool ok = b1 >= b0 && b1 <= comp_stride;
    base = ok ? b0 : 0; slot = ok ? b1 - b0 : 0;
}

__global__ __launch_bounds__(DH_WAVES * 64) void k_dec_huff(const uint32_t *__restrict__ comp, size_t comp_stride,
                                                            const uint32_t *__restrict__ offsets, size_t offset_stride,
                                                            const uint32_t *__restrict__ lut,
                                                            const uint32_t *__restrict__ nodes, uint32_t n,
                                                            uint8_t *__restrict__ mtf, size_t mtf_stride,
                                                            uint32_t *__restrict__ d_status,
                                                            const unsigned long long *__restrict__ block_off)
{
    __shared__ uint16_t s_lut[1 << DEC_LUT_BITS];
    __shared__ uint32_t s_nodes[HUFF_NODES];
    __shared__ uint32_t s_words[DH_WAVES][DH_WORDS_LDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DH_WAVES][HUFF_BLOCK];
    const uint32_t b = blockIdx.y, tid = threadIdx.x, l = tid & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < (1u << DEC_LUT_BITS); i += DH_WAVES * 64) {
        const uint32_t e = lut[((size_t)b << DEC_LUT_BITS) + i];
        s_lut[i] = (e & DEC_FLAG) ? (uint16_t)(0x8000u | (e & 0x3FFu)) : (uint16_t)((e & 0x3FFu) | ((e >> 16) << 10));
    }
    for (uint32_t i = tid; i < HUFF_NODES; i += DH_WAVES * 64) s_nodes[i] = nodes[(size_t)b * HUFF_NODES + i];
    __syncthreads();
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    const uint32_t sub = blockIdx.x * DH_WAVES + wv;
    if (sub >= nsub) return;
    const uint32_t lo = sub * HUFF_BLOCK, cnt = min((uint32_t)HUFF_BLOCK, n - lo);
    // offsets and lengths come from the stream: stay inside the block's slot whatever they say
    uint32_t off = offsets[(size_t)b * offset_stride + sub];
    uint64_t wbase, slot;
    dec_block_slot(block_off, b, comp_stride, wbase, slot);
    bool bad = false;
    if ((uint64_t)off + 1 > slot) { off = 0; bad = true; }
    const uint32_t *w = comp + wbase + off;
    uint32_t nwords = 0;
    if (slot > 0) {
        nwords = w[0];
        if (nwords > HUFF_MAX_WORDS || (uint64_t)off + 1 + nwords > slot) {
            nwords = (uint32_t)min((uint64_t)min(nwords, (uint32_t)HUFF_MAX_WORDS), slot - off - 1);
            bad = true;
        }
    }
    if (bad && d_status && l == 0) atomicOr(d_status, ST_CORRUPT);
    w++;
    const uint32_t S = max(1u, (nwords + 63) / 64), P = S | 1u;
    uint32_t *sw = s_words[wv];
    for (uint32_t g = l; g < 66 * S; g += 64) {                   // 64 spans + two zero spans of look-ahead
        const uint32_t q = g / S;
        sw[q * P + (g - q * S)] = g < nwords ? w[g] : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    const DhTables T{s_lut, s_nodes};
    uint8_t *so = s_out[wv];
    uint32_t start = 0;
    uint32_t r = dh_decode_span<false>(sw, l * P, S, P, 0, T, so, 0, 0);
    for (uint32_t it = 0; it < 64; it++) {
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(r & 0xFFu), 0x138, 0xf, 0xf, false);  // wave_shr:1
        const bool need = prev != start;
        if (__ballot(need) == 0) break;
        if (need) { start = prev; r = dh_decode_span<false>(sw, l * P, S, P, start, T, so, 0, 0); }
    }
    const uint32_t c = r >> 8;
    const uint32_t base = wave_incl_add(c) - c;
    (void)dh_decode_span<true>(sw, l * P, S, P, start, T, so, base, cnt);
    __builtin_amdgcn_wave_barrier();
    uint8_t *dst = mtf + (size_t)b * mtf_stride + lo;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && cnt == HUFF_BLOCK) {
#pragma unroll
        for (uint32_t k = 0; k < HUFF_BLOCK / 1024; k++)
            reinterpret_cast<uint4 *>(dst)[k * 64 + l] = reinterpret_cast<const uint4 *>(so)[k * 64 + l];
    } else {
        for (uint32_t i = l; i < cnt; i += 64) dst[i] = so[i];
    }
}

// ---------------------------------------------------------------------------
// 2b. Huffman decode for large batches: one LANE per 4096-symbol block, the way the CPU gold walks a stream, 256 of
//     them per workgroup (= a 1 MiB block).  With >= 128 blocks in a call there are enough 4096-symbol blocks to fill
//     the machine with such lanes (1024 blocks: 4 waves per SIMD), and a lane decodes every symbol ONCE -- the
//     wave-per-block kernel above decodes every span three to four times to find where its codes start.
//     Each lane's words come through a 32-word ring in LDS, refilled 16 words (64 contiguous bytes) at a time, so a
//     cache line is fetched twice, not once per word; symbols leave as 16-byte stores.
// ---------------------------------------------------------------------------
constexpr uint32_t DL_NT = 256, DL_RING = 32, DL_PITCH = DL_RING + 1;
constexpr uint64_t DL_MIN_SUBS = 32768;                      // 4096-symbol blocks in a call from which the lane kernel is used

__global__ __launch_bounds__(DL_NT) void k_dec_huff_lanes(const uint32_t *__restrict__ comp, size_t comp_stride,
                                                          const uint32_t *__restrict__ offsets, size_t offset_stride,
                                                          const uint32_t *__restrict__ lut,
                                                          const uint32_t *__restrict__ nodes, uint32_t n,
                                                          uint8_t *__restrict__ mtf, size_t mtf_stride,
                                                          uint32_t *__restrict__ d_status,
                                                          const unsigned long long *__restrict__ block_off)
{
    __shared__ uint16_t s_lut[1 << DEC_LUT_BITS];
    __shared__ uint32_t s_nodes[HUFF_NODES];
    __shared__ uint32_t s_ring[DL_NT * DL_PITCH];
    const uint32_t b = blockIdx.y, tid = threadIdx.x;
    for (uint32_t i = tid; i < (1u << DEC_LUT_BITS); i += DL_NT) {
        const uint32_t e = lut[((size_t)b << DEC_LUT_BITS) + i];
        s_lut[i] = (e & DEC_FLAG) ? (uint16_t)(0x8000u | (e & 0x3FFu)) : (uint16_t)((e & 0x3FFu) | ((e >> 16) << 10));
    }
    for (uint32_t i = tid; i < HUFF_NODES; i += DL_NT) s_nodes[i] = nodes[(size_t)b * HUFF_NODES + i];
    __syncthreads();
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    const uint32_t sub = blockIdx.x * DL_NT + tid;
    if (sub >= nsub) return;
    const uint32_t lo = sub * HUFF_BLOCK, cnt = min((uint32_t)HUFF_BLOCK, n - lo);
    // offsets and lengths come from the stream: stay inside the block's slot whatever they say
    uint32_t off = offsets[(size_t)b * offset_stride + sub];
    uint64_t wbase, slot;
    dec_block_slot(block_off, b, comp_stride, wbase, slot);
    bool bad = false;
    if ((uint64_t)off + 1 > slot) { off = 0; bad = true; }
    const uint32_t *w = comp + wbase + off;
    uint32_t nwords = 0;
    if (slot > 0) {
        nwords = w[0];
        if (nwords > HUFF_MAX_WORDS || (uint64_t)off + 1 + nwords > slot) {
            nwords = (uint32_t)min((uint64_t)min(nwords, (uint32_t)HUFF_MAX_WORDS), slot - off - 1);
            bad = true;
        }
    }
    if (bad && d_status) atomicOr(d_status, ST_CORRUPT);
    w++;
    uint32_t *ring = s_ring + tid * DL_PITCH;
    uint32_t wi = 0, rd = 0, have = 0;                          // next word to fetch / ring read position / words in the ring
    // 16 words at a time, PREFETCHED: the loads issued at one refill are written to the ring at the next one, a batch of
    // 16 symbols later -- a lane that loaded and stored in the same breath sat out a global-memory round trip (~1 us)
    // per 16 symbols, a third of the kernel
    uint4 pre[4];
    auto fetch = [&]() {                                        // words wi .. wi + 16 -> pre (zeros past the end of the stream)
        if (wi + 16 <= nwords) {
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_memcpy(&pre[k], w + wi + 4 * k, 16);   // 16-byte loads at a 4-byte-aligned address
        } else {
            uint32_t t[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) t[k] = wi + k < nwords ? w[wi + k] : 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) pre[k] = make_uint4(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
        }
        wi += 16;
    };
    auto refill = [&]() {                                       // pre -> the 16 ring slots behind the ones in use; next fetch issued
        const uint32_t at = (rd + have) & (DL_RING - 1);        // (0 or 16: refills come in halves of the ring)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ring[(at + 4 * k + 0) & (DL_RING - 1)] = pre[k].x; ring[(at + 4 * k + 1) & (DL_RING - 1)] = pre[k].y;
            ring[(at + 4 * k + 2) & (DL_RING - 1)] = pre[k].z; ring[(at + 4 * k + 3) & (DL_RING - 1)] = pre[k].w;
        }
        have += 16;
        fetch();
    };
    fetch();
    refill();
    refill();
    auto next_word = [&]() -> uint32_t { const uint32_t x = ring[rd]; rd = (rd + 1) & (DL_RING - 1); have--; return x; };
    uint64_t buf = (uint64_t)next_word() << 32;
    buf |= next_word();
    uint32_t nb = 64;
    uint8_t *dst = mtf + (size_t)b * mtf_stride + lo;
    const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (uint32_t i0 = 0; i0 < cnt; i0 += 16) {
        if (have <= 16) refill();                               // 16 symbols take at most 14 words (codes <= 28 bits)
        uint32_t o4[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            if (nb <= 32) { buf |= (uint64_t)next_word() << (32 - nb); nb += 32; }
            const uint32_t e = s_lut[(uint32_t)(buf >> (64 - DEC_LUT_BITS))];
            uint32_t len, sym;
            if (!(e & 0x8000u)) { sym = e & 0x3FFu; len = e >> 10; }
            else {
                uint32_t ne = s_nodes[e & 0x3FFu];
                len = DEC_LUT_BITS;
                while (!(ne & DEC_FLAG) && len < 33) {
                    const uint32_t bit = (uint32_t)(buf >> (63 - len)) & 1u;
                    ne = s_nodes[bit ? (ne >> 16) : (ne & 0xFFFF)];
                    len++;
                }
                sym = ne & 0xFFFF;
            }
            len = len ? len : 1u;
            o4[j >> 2] |= (sym & 0xFFu) << (8 * (j & 3));
            buf <<= len; nb -= len;
        }
        if (vec && i0 + 16 <= cnt) *reinterpret_cast<uint4 *>(dst + i0) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        else for (uint32_t j = 0; j < 16 && i0 + j < cnt; j++) dst[i0 + j] = (uint8_t)(o4[j >> 2] >> (8 * (j & 3)));
    }
}

// ---------------------------------------------------------------------------
// 3. inverse MTF
// ---------------------------------------------------------------------------

// Pass 1 (k_imtf_pos): one LANE per chunk.  The lane keeps a 256-entry list of POSITIONS (identity at the chunk start)
// in LDS and, for every MTF index r, reads entry r and moves it to the front.  Output: the position byte of every
// symbol (symbol = start_list[pos]) and the chunk's final list = its position permutation.
// Pass 2 (k_imtf_scan) composes the permutations; pass 3 (k_imtf_apply) is a 256-byte LUT lookup per chunk -- the
// sequential work is done once, not twice.
// The list is 16 GROUPS of 16 entries, [group][lane] x 16 bytes.  Moving entry r = 16 g + o to the front shifts the
// r entries below it up by one; done word by word that is 2 g + 3 sixteen-byte LDS accesses per symbol FOR THE LANE
// WITH THE LARGEST r of the wave (PMC: LDS data path saturated, VALU at 38 %, 9.9 ms per GiB).  Here every group is a
// RING (its tail position is a nibble of tlo / thi): pushing an entry in at the front of group k < g and its last
// entry out is one byte read + one byte write at the ring's tail, which becomes the new head -- the 16 bytes stay
// where they are, and all the tails of a symbol step back in one nibble-wise addition.  Only group g itself is rewritten (rotated to its logical order, the o entries below the hit shifted,
// stored back with head 0).  Bytes moved per symbol: ~2 g + 48 instead of 32 g + 48.  Measured: 9.9 -> 9.0 ms per GiB
// -- the LDS data path is no longer the limit (active 105 -> 64 quad-cycles per 64 symbols, FIFO-full 40 -> 0.6), the
// kernel now runs at ~62 % of the VALU issue rate (~210 instructions per 64 symbols: a read, a write and a nested EXEC
// level per group) with 2.25 waves per SIMD (16.6 KB of LDS per wave) to cover three dependent LDS round trips per symbol.
#ifndef GLC_IMTF_CHUNK
#define GLC_IMTF_CHUNK 4096
#endif
constexpr uint32_t IMTF_CHUNK = GLC_IMTF_CHUNK;              // symbols per lane of pass 1 (4096 against 2048: permutation scan 251 -> 125, apply 535 -> 413 us per 1024 blocks)
static_assert(IMTF_CHUNK % 2048 == 0, "k_imtf_apply works in pieces of 2048 symbols");

// w rotated right by h bytes: byte i of the result = byte (i + h) & 15 of w
__device__ __forceinline__ uint4 imtf_rotr(uint4 w, uint32_t h)
{
    const bool q1 = (h & 4u) != 0, q2 = (h & 8u) != 0;
    uint32_t x = q1 ? w.y : w.x, y = q1 ? w.z : w.y, z = q1 ? w.w : w.z, v = q1 ? w.x : w.w;
    const uint32_t x2 = q2 ? z : x, y2 = q2 ? v : y, z2 = q2 ? x : z, v2 = q2 ? y : v;
    const uint32_t b = h & 3u;
    uint4 r;
    r.x = __builtin_amdgcn_alignbyte(y2, x2, b);
    r.y = __builtin_amdgcn_alignbyte(z2, y2, b);
    r.z = __builtin_amdgcn_alignbyte(v2, z2, b);
    r.w = __builtin_amdgcn_alignbyte(x2, v2, b);
    return r;
}

__global__ __launch_bounds__(64) void k_imtf_pos(const uint8_t *__restrict__ in, size_t in_stride, uint32_t n,
                                                 uint8_t *__restrict__ lists, uint32_t max_chunks,
                                                 uint8_t *__restrict__ pos_out, size_t out_stride)
{
    __shared__ uint4 s_list[16 * 64];
    __shared__ uint4 s_mask[17];                              // s_mask[c]: the low c bytes
    const uint32_t b = blockIdx.y, l = threadIdx.x;
    const uint32_t nchunks = (n + IMTF_CHUNK - 1) / IMTF_CHUNK;
    const uint32_t chunk = blockIdx.x * 64 + l;
    const bool live = chunk < nchunks;
    const uint32_t lo = chunk * IMTF_CHUNK;
    const uint32_t cnt = live ? min(IMTF_CHUNK, n - lo) : 0u;
    const uint8_t *src = in + (size_t)b * in_stride + lo;
    uint8_t *dst = pos_out + (size_t)b * out_stride + lo;
    uint8_t *s_bytes = reinterpret_cast<uint8_t *>(s_list);
    const bool vec_ok = ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        const uint32_t v = 0x03020100u + 0x10101010u * k;
        s_list[k * 64 + l] = make_uint4(v, v + 0x04040404u, v + 0x08080808u, v + 0x0C0C0C0Cu);
    }
    if (l < 17) {
        auto m = [&](int c) { return c >= 4 ? 0xFFFFFFFFu : (c <= 0 ? 0u : ((1u << (8 * c)) - 1u)); };
        s_mask[l] = make_uint4(m((int)l), m((int)l - 4), m((int)l - 8), m((int)l - 12));
    }
    __builtin_amdgcn_wave_barrier();
    auto load16 = [&](uint32_t j, uint32_t *rv) {
        rv[0] = rv[1] = rv[2] = rv[3] = 0;
        if (vec_ok && j + 16 <= cnt) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + j);
            rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
        } else if (j < cnt) {
            for (uint32_t t = 0; t < min(16u, cnt - j); t++) rv[t >> 2] |= (uint32_t)src[j + t] << (8 * (t & 3));
        }
    };
    uint32_t tlo = 0xFFFFFFFFu, thi = 0xFFFFFFFFu;           // ring TAILS (position of a group's last entry), groups 0..7 / 8..15, a nibble each
    const uint32_t lbase = l * 16;
    // nibble-wise x + y mod 16 (y = 15 in the nibbles that step back by one, 0 elsewhere)
    auto nib_add = [](uint32_t x, uint32_t y) { return ((x & 0x77777777u) + (y & 0x77777777u)) ^ ((x ^ y) & 0x88888888u); };
    uint32_t nx[4];
    load16(0, nx);
    for (uint32_t j = 0; j < IMTF_CHUNK; j += 16) {
        if (__ballot(j < cnt) == 0) break;
        uint32_t rv[4] = {nx[0], nx[1], nx[2], nx[3]}, ov[4] = {0, 0, 0, 0};
        load16(j + 16, nx);                                   // in flight while these 16 are processed
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) {
            // past the end of the chunk: index 0, which changes nothing
            const uint32_t r = (j + t < cnt) ? (rv[t >> 2] >> (8 * (t & 3))) & 0xFFu : 0u;
            const uint32_t g = r >> 4, o = r & 15u, sh = (g & 7u) * 4u;
            const uint32_t h = ((((g >= 8 ? thi : tlo) >> sh) & 15u) + 1u) & 15u;      // head of group g
            const uint32_t gbase = g * 1024u + lbase;
            const uint32_t sym = s_bytes[gbase + ((h + o) & 15u)];
            // groups below g: the entry pushed out at the tail makes room for the one pushed in; the slot becomes the
            // head.  A lane leaves the loop at its own g (the wave's EXEC shrinks); the next group's tail entry is
            // read one trip ahead.
            uint32_t carry = sym;
            {
                uint32_t acur = lbase + (tlo & 15u);
                uint32_t tcur = s_bytes[acur];
#pragma unroll
                for (uint32_t k = 0; k < 15; k++) {
                    if (k >= g) break;
                    const uint32_t k1 = k + 1;
                    const uint32_t anext = k1 * 1024u + lbase + (((k1 < 8 ? tlo : thi) >> (4 * (k1 & 7))) & 15u);
                    const uint32_t tnext = s_bytes[anext];
                    s_bytes[acur] = (uint8_t)carry;
                    carry = tcur; tcur = tnext; acur = anext;
                }
            }
            tlo = nib_add(tlo, g >= 8 ? 0xFFFFFFFFu : (1u << ((4 * g) & 31u)) - 1u);
            thi = nib_add(thi, g <= 8 ? 0u : (1u << ((4 * g) & 31u)) - 1u);
            {   // group g: logical order, entries 0..o-1 move up by one, the carry goes to the front, head = 0 (tail = 15)
                const uint4 w = imtf_rotr(s_list[g * 64 + l], h);
                const uint4 m = s_mask[o + 1];
                uint4 nw;
                nw.x = (w.x & ~m.x) | (((w.x << 8) | carry) & m.x);
                nw.y = (w.y & ~m.y) | (__builtin_amdgcn_alignbit(w.y, w.x, 24) & m.y);
                nw.z = (w.z & ~m.z) | (__builtin_amdgcn_alignbit(w.z, w.y, 24) & m.z);
                nw.w = (w.w & ~m.w) | (__builtin_amdgcn_alignbit(w.w, w.z, 24) & m.w);
                s_list[g * 64 + l] = nw;
                const uint32_t set = 15u << sh;
                if (g >= 8) thi |= set; else tlo |= set;
            }
            ov[t >> 2] |= sym << (8 * (t & 3));
        }
        if (vec_ok && j + 16 <= cnt) {
            *reinterpret_cast<uint4 *>(dst + j) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
        } else if (j < cnt) {
            for (uint32_t t = 0; t < min(16u, cnt - j); t++) dst[j + t] = (uint8_t)(ov[t >> 2] >> (8 * (t & 3)));
        }
    }
    if (live && chunk + 1 < nchunks) {                           // nobody needs the last permutation
        uint4 *LW = reinterpret_cast<uint4 *>(lists + ((size_t)b * max_chunks + chunk) * 256);
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
            LW[k] = imtf_rotr(s_list[k * 64 + l], ((((k < 8 ? tlo : thi) >> (4 * (k & 7))) & 15u) + 1u) & 15u);
    }
}

// ---------------------------------------------------------------------------
// Pass 1, second form (the default): the list as a DEQUE WITH HOLES.  Moving entry r to the front of an array shifts r
// entries; the ring form above cuts that to one byte per 16-entry group below the hit, and still spends ~135 of its 212
// wave instructions per step in the loop over those groups -- executed as often as the LARGEST index of the wave asks
// for, and MTF indices of near-incompressible data are large.  Here nothing is shifted at all.  A lane's list lives in
// 512 POSITIONS (lower = nearer the front): a 512-byte array A and a 512-bit bitmap V of the positions in use, exactly
// 256 of them.  A step with index r
//     finds the r-th position in use, p   -- SELECT on the bitmap: a 3-level tree of counts kept in registers names the
//                                            64-bit word, byte counts (SWAR) name the byte, a 2 KB table the bit;
//     reads the entry A[p], clears V[p], and writes the entry at the front: position f - 1, V[f - 1] set, f -= 1
// (f is the same for all lanes: it steps down by one per symbol whatever the data is; the counts are bytes packed in
// two registers, so a word's share of them is one multiply away).  After 256 steps the front has
// reached position 0 and the 256 live entries are packed back to positions 256..511, four bytes at a time (v_perm with
// a 16-entry selector table, no divergence: every lane holds exactly 256 entries).  ~75 VALU + 6 LDS instructions per
// step instead of 212 + 41; 36 KB of LDS per wave (one wave per SIMD), so a step is the latency of its chain
// (tree descent -> word -> byte -> bit: three LDS round trips), ~420 cycles.
// Same outputs as the ring form: the position byte of every symbol and the chunk's final list (its permutation).
// ---------------------------------------------------------------------------
#ifndef GLC_IMD_POS
#define GLC_IMD_POS 512
#endif
constexpr uint32_t IMD_POS = GLC_IMD_POS;                     // 512: 4 waves per CU, packed back every 256 steps: 1156 us per 256 blocks; 384: 5 waves, every 128 steps: 1404
constexpr uint32_t IMD_FRONT = IMD_POS - 256;                 // positions ahead of the 256 entries after packing
constexpr uint32_t IMD_WORDS = IMD_POS / 64;
static_assert(IMD_POS == 512 || IMD_POS == 384, "the count bytes hold words 0-3 and 4-7");

__global__ __launch_bounds__(64) void k_imtf_pos_deque(const uint8_t *__restrict__ in, size_t in_stride, uint32_t n,
                                                       uint8_t *__restrict__ lists, uint32_t max_chunks,
                                                       uint8_t *__restrict__ pos_out, size_t out_stride)
{
    __shared__ uint32_t s_a[(IMD_POS / 4) * 64];              // entry at position p of lane l: byte p & 3 of s_a[(p >> 2) * 64 + l] (lane l = bank l)
    __shared__ uint2 s_v[IMD_WORDS * 64];                     // bitmap word k of lane l: s_v[k * 64 + l]
    __shared__ uint8_t s_sel[256 * 8];                        // s_sel[b * 8 + j] = index of the j-th set bit of byte b
    __shared__ uint32_t s_pack[16];                           // v_perm selector that moves the bytes of mask m to the top of a dword, in order
    const uint32_t b = blockIdx.y, l = threadIdx.x;
    const uint32_t nchunks = (n + IMTF_CHUNK - 1) / IMTF_CHUNK;
    const uint32_t chunk = blockIdx.x * 64 + l;
    const bool live = chunk < nchunks;
    const uint32_t lo = chunk * IMTF_CHUNK;
    const uint32_t cnt = live ? min(IMTF_CHUNK, n - lo) : 0u;
    const uint8_t *src = in + (size_t)b * in_stride + lo;
    uint8_t *dst = pos_out + (size_t)b * out_stride + lo;
    const bool vec_ok = ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0;
    uint8_t *s_ab = reinterpret_cast<uint8_t *>(s_a);
    // tables
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t byte = l * 4 + t;
        uint32_t j = 0;
        for (uint32_t bit = 0; bit < 8; bit++)
            if ((byte >> bit) & 1u) s_sel[byte * 8 + j++] = (uint8_t)bit;
        for (; j < 8; j++) s_sel[byte * 8 + j] = 0;
    }
    if (l < 16) {
        // bytes of the dword whose mask bit is set, kept in order, moved to the top; 0x0C selects a zero byte
        uint32_t sel = 0x0C0C0C0Cu, c = (uint32_t)__popc(l), at = 4 - c;
        for (uint32_t i = 0; i < 4; i++)
            if ((l >> i) & 1u) { sel = (sel & ~(0xFFu << (8 * at))) | (i << (8 * at)); at++; }
        s_pack[l] = sel;
    }
    // identity list at the top 256 positions
    constexpr uint32_t D0 = IMD_FRONT / 4, K0 = IMD_FRONT / 64;  // first dword / bitmap word of the packed list
#pragma unroll
    for (uint32_t d = 0; d < 64; d++) s_a[(D0 + d) * 64 + l] = 0x03020100u + 0x04040404u * d;
#pragma unroll
    for (uint32_t k = 0; k < IMD_WORDS; k++) s_v[k * 64 + l] = k < K0 ? make_uint2(0u, 0u) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    __builtin_amdgcn_wave_barrier();
    const uint32_t lbase = l * 4;

    // the 256 entries in use back to positions 256 .. 511 (order kept); resets the counts
    auto compact = [&]() {
        uint32_t acc_hi = 0, acc_lo = 0, nacc = 0, dd = IMD_POS / 4 - 1;   // bytes waiting at the top of acc; next dword to fill
#pragma unroll 1
        for (int k = (int)IMD_WORDS - 1; k >= 0; k--) {
            const uint2 w = s_v[k * 64 + l];
#pragma unroll
            for (int q = 15; q >= 0; q--) {
                const uint32_t d = (uint32_t)k * 16u + (uint32_t)q;
                const uint32_t m = ((q >= 8 ? w.y : w.x) >> (4 * (q & 7))) & 15u;
                const uint32_t a = s_a[d * 64 + l];
                const uint32_t packed = __builtin_amdgcn_perm(0u, a, s_pack[m]);     // the bytes in use, top-aligned
                // below the nacc bytes already waiting
                const uint64_t sh = ((uint64_t)packed << 32) >> (8 * nacc);
                acc_hi |= (uint32_t)(sh >> 32); acc_lo |= (uint32_t)sh;
                nacc += (uint32_t)__popc(m);
                if (nacc >= 4) { s_a[dd * 64 + l] = acc_hi; dd--; acc_hi = acc_lo; acc_lo = 0; nacc -= 4; }
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < IMD_WORDS; k++) s_v[k * 64 + l] = k < K0 ? make_uint2(0u, 0u) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    };

    auto load16 = [&](uint32_t j, uint32_t *rv) {
        rv[0] = rv[1] = rv[2] = rv[3] = 0;
        if (vec_ok && j + 16 <= cnt) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + j);
            rv[0] = q.x; rv[1] = q.y; rv[2] = q.z; rv[3] = q.w;
        } else if (j < cnt) {
            for (uint32_t t = 0; t < min(16u, cnt - j); t++) rv[t >> 2] |= (uint32_t)src[j + t] << (8 * (t & 3));
        }
    };
    uint32_t nx[4];
    load16(0, nx);
    uint32_t f = IMD_FRONT;                                        // front: the next entry goes to position f - 1 (wave-uniform)
    // positions in use per bitmap word, a byte each: words 0-3 in cnt_lo, 4-7 in cnt_hi; tot_lo = all of words 0-3
    // (it reaches 256, one more than a byte holds)
    constexpr uint32_t CNT_LO0 = IMD_POS == 512 ? 0u : 0x40400000u, CNT_HI0 = IMD_POS == 512 ? 0x40404040u : 0x00004040u;
    constexpr uint32_t TOT_LO0 = IMD_POS == 512 ? 0u : 128u;
    uint32_t cnt_lo = CNT_LO0, cnt_hi = CNT_HI0, tot_lo = TOT_LO0;

    // which word holds the rr-th position in use: k = 4 g1 + kk, rr becomes the index inside the word
    struct Sel { uint32_t k, rr, kk; bool g1; };
    auto word_select = [&](uint32_t r) -> Sel {
        Sel o;
        o.g1 = r >= tot_lo;
        uint32_t rr = r - (o.g1 ? tot_lo : 0u);
        const uint32_t P = (o.g1 ? cnt_hi : cnt_lo) * 0x00010101u;   // bytes 0..2: positions in use in the half's words 0 / 0-1 / 0-2
        o.kk = (rr >= (P & 0xFFu) ? 1u : 0u) + (rr >= ((P >> 8) & 0xFFu) ? 1u : 0u) + (rr >= ((P >> 16) & 0xFFu) ? 1u : 0u);
        rr -= ((P << 8) >> (8 * o.kk)) & 0xFFu;
        o.rr = rr;
        o.k = (o.g1 ? 4u : 0u) + o.kk;
        return o;
    };
    // one batch of 16 steps.  FULL: every lane's chunk has all 16 symbols (no bounds).  Everything that depends on the
    // front alone is the same for all lanes and, inside a batch, a fixed offset from its value at the batch's start (f is
    // a multiple of 16 there): with one wave per SIMD every instruction is on the critical path, scalar ones included.
    auto batch = [&](uint32_t j, const uint32_t *rv, uint32_t *ov, auto FULL) {
        constexpr bool full = decltype(FULL)::value;
        const uint32_t F0 = f;                                     // positions F0 - 1 .. F0 - 16 are filled by this batch
        const uint32_t kf = (F0 - 1) >> 6;                         // one bitmap word for the whole batch
        const uint32_t inc = 1u << (8 * kf);
        const bool fy = ((F0 - 1) & 32u) != 0;
        uint32_t *vfront = fy ? &s_v[kf * 64 + l].y : &s_v[kf * 64 + l].x;
        uint32_t fb = (F0 & 16u) ? 0x8000u : 0x80000000u;          // bit of position F0 - 1 in its 32-bit half; one to the right per step
        uint8_t *afront = s_ab + ((((F0 >> 2) - 4u) << 8) + lbase);   // dwords F0/4 - 4 .. F0/4 - 1 of this lane
        auto index_at = [&](uint32_t t) {
            const uint32_t r = (rv[t >> 2] >> (8 * (t & 3))) & 0xFFu;
            return full ? r : ((j + t < cnt) ? r : 0u);            // past the end: index 0, the order stays
        };
        Sel cur = word_select(index_at(0));
        uint2 w = s_v[cur.k * 64 + l];
#pragma unroll
        for (uint32_t t = 0; t < 16; t++) {
            // ---- the counts after this step: one position less in word cur.k, one more in the front's word
            {
                const uint32_t dec = 1u << (8 * cur.kk);
                cnt_lo += inc - (cur.g1 ? 0u : dec);
                cnt_hi -= cur.g1 ? dec : 0u;
                tot_lo += cur.g1 ? 1u : 0u;                        // (+1 for the front, -1 if the entry came out of words 0-3)
            }
            // ---- next step: word, and its load
            Sel nxt = cur;
            uint2 wn = w;
            if (t < 15) {
                nxt = word_select(index_at(t + 1));
                wn = s_v[nxt.k * 64 + l];
            }
            // ---- this step: half, byte, bit
            uint32_t rr = cur.rr;
            const uint32_t ch = (uint32_t)__popc(w.x);
            const bool gh = rr >= ch;
            rr -= gh ? ch : 0u;
            const uint32_t x = gh ? w.y : w.x;
            const uint32_t x1 = x - ((x >> 1) & 0x55555555u);
            const uint32_t x2 = (x1 & 0x33333333u) + ((x1 >> 2) & 0x33333333u);
            const uint32_t x4 = (x2 + (x2 >> 4)) & 0x0F0F0F0Fu;
            const uint32_t cum = x4 * 0x00010101u;                 // bytes 0..2: positions in use in bytes 0 / 0-1 / 0-2 of x
            const uint32_t jb = (rr >= (cum & 0xFFu) ? 1u : 0u) + (rr >= ((cum >> 8) & 0xFFu) ? 1u : 0u) +
                                (rr >= ((cum >> 16) & 0xFFu) ? 1u : 0u);
            rr -= ((cum << 8) >> (8 * jb)) & 0xFFu;
            const uint32_t byte = (x >> (8 * jb)) & 0xFFu;
            const uint32_t bit = s_sel[byte * 8 + rr];
            const uint32_t pin = (gh ? 32u : 0u) + jb * 8u + bit;  // position inside the word
            const uint32_t p = cur.k * 64u + pin;
            const uint32_t sym = s_ab[((p >> 2) << 8) + lbase + (p & 3u)];
            // ---- out of its place, to the front (position F0 - 1 - t)
            const uint32_t clr = ~(1u << (pin & 31u));
            const uint32_t clrx = gh ? 0xFFFFFFFFu : clr, clry = gh ? clr : 0xFFFFFFFFu;
            w.x &= clrx; w.y &= clry;
            s_v[cur.k * 64 + l] = w;
            afront[((3u - (t >> 2)) << 8) + (3u - (t & 3u))] = (uint8_t)sym;
            atomicOr(vfront, fb);
            ov[t >> 2] |= sym << (8 * (t & 3));
            // ---- what this step did to the word the next step has already loaded
            if (t < 15) {
                const bool same = nxt.k == cur.k, front = nxt.k == kf;
                wn.x = (wn.x & (same ? clrx : 0xFFFFFFFFu)) | ((front && !fy) ? fb : 0u);
                wn.y = (wn.y & (same ? clry : 0xFFFFFFFFu)) | ((front && fy) ? fb : 0u);
            }
            fb >>= 1;
            cur = nxt; w = wn;
        }
        f = F0 - 16;
    };
    for (uint32_t j = 0; j < IMTF_CHUNK; j += 16) {
        if (__ballot(j < cnt) == 0) break;
        uint32_t rv[4] = {nx[0], nx[1], nx[2], nx[3]}, ov[4] = {0, 0, 0, 0};
        load16(j + 16, nx);                                        // in flight while these 16 are processed
        // Steps are software-pipelined: the word of step t + 1 is chosen and its load issued BEFORE step t has found its
        // bit -- the counts already know which words step t takes from and gives to -- and what step t then changes in
        // that word is applied to the loaded copy in registers.  (One wave per SIMD: nothing else hides the round trips.)
        if (__ballot(j + 16 > cnt) == 0) batch(j, rv, ov, std::true_type{});
        else batch(j, rv, ov, std::false_type{});
        if (vec_ok && j + 16 <= cnt) {
            *reinterpret_cast<uint4 *>(dst + j) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
        } else if (j < cnt) {
            for (uint32_t t = 0; t < min(16u, cnt - j); t++) dst[j + t] = (uint8_t)(ov[t >> 2] >> (8 * (t & 3)));
        }
        if (f == 0) {
            __builtin_amdgcn_wave_barrier();
            compact();
            f = IMD_FRONT; cnt_lo = CNT_LO0; cnt_hi = CNT_HI0; tot_lo = TOT_LO0;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (live && chunk + 1 < nchunks) {                           // nobody needs the last permutation
        if (f != IMD_FRONT) { __builtin_amdgcn_wave_barrier(); compact(); }
        __builtin_amdgcn_wave_barrier();
        uint4 *LW = reinterpret_cast<uint4 *>(lists + ((size_t)b * max_chunks + chunk) * 256);
#pragma unroll
        for (uint32_t k = 0; k < 16; k++)
            LW[k] = make_uint4(s_a[(D0 + 4 * k) * 64 + l], s_a[(D0 + 4 * k + 1) * 64 + l], s_a[(D0 + 4 * k + 2) * 64 + l],
                               s_a[(D0 + 4 * k + 3) * 64 + l]);
    }
}

// lists[c] <- list at the start of chunk c ; state' [k] = state[perm_c[k]]
__global__ __launch_bounds__(64) void k_imtf_scan(uint8_t *__restrict__ lists, uint32_t n, uint32_t max_chunks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_state[256];
    const uint32_t b = blockIdx.x, l = threadIdx.x;
    const uint32_t nchunks = (n + IMTF_CHUNK - 1) / IMTF_CHUNK;
    reinterpret_cast<uint32_t *>(s_state)[l] = 0x03020100u + 0x04040404u * l;
    __builtin_amdgcn_wave_barrier();
    uint32_t p4 = (1 < nchunks) ? reinterpret_cast<uint32_t *>(lists + (size_t)b * max_chunks * 256)[l] : 0u;
    for (uint32_t c = 0; c < nchunks; c++) {
        uint32_t *LW = reinterpret_cast<uint32_t *>(lists + ((size_t)b * max_chunks + c) * 256);
        const uint32_t pn = (c + 2 < nchunks) ? LW[64 + l] : 0u;  // prefetch the next chunk's permutation
        const uint32_t cur = reinterpret_cast<const uint32_t *>(s_state)[l];
        LW[l] = cur;
        if (c + 1 == nchunks) break;
        const uint32_t nv = (uint32_t)s_state[p4 & 0xFF] | ((uint32_t)s_state[(p4 >> 8) & 0xFF] << 8) |
                            ((uint32_t)s_state[(p4 >> 16) & 0xFF] << 16) | ((uint32_t)s_state[p4 >> 24] << 24);
        __builtin_amdgcn_wave_barrier();
        reinterpret_cast<uint32_t *>(s_state)[l] = nv;
        __builtin_amdgcn_wave_barrier();
        p4 = pn;
    }
}

// symbols = start_list[pos], in place.  One workgroup per chunk.
__global__ __launch_bounds__(256) void k_imtf_apply(uint8_t *__restrict__ buf, size_t stride, uint32_t n,
                                                    const uint8_t *__restrict__ lists, uint32_t max_chunks)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_lut[256];
    const uint32_t b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    if (tid < 64)
        reinterpret_cast<uint32_t *>(s_lut)[tid] =
            reinterpret_cast<const uint32_t *>(lists + ((size_t)b * max_chunks + chunk) * 256)[tid];
    __syncthreads();
#pragma unroll
    for (uint32_t g = 0; g < IMTF_CHUNK / 2048; g++) {
        const uint32_t lo = chunk * IMTF_CHUNK + g * 2048 + tid * 8;
        uint8_t *P = buf + (size_t)b * stride + lo;
        if (lo + 8 <= n && (reinterpret_cast<size_t>(P) & 7) == 0) {
            const uint2 q = *reinterpret_cast<const uint2 *>(P);
            uint2 o;
            o.x = (uint32_t)s_lut[q.x & 0xFF] | ((uint32_t)s_lut[(q.x >> 8) & 0xFF] << 8) |
                  ((uint32_t)s_lut[(q.x >> 16) & 0xFF] << 16) | ((uint32_t)s_lut[q.x >> 24] << 24);
            o.y = (uint32_t)s_lut[q.y & 0xFF] | ((uint32_t)s_lut[(q.y >> 8) & 0xFF] << 8) |
                  ((uint32_t)s_lut[(q.y >> 16) & 0xFF] << 16) | ((uint32_t)s_lut[q.y >> 24] << 24);
            *reinterpret_cast<uint2 *>(P) = o;
        } else {
            for (uint32_t i = lo; i < n && i < lo + 8; i++) P[i - lo] = s_lut[P[i - lo]];
        }
    }
}

// ---------------------------------------------------------------------------
// 4. inverse BWT
//    rows r = 0..n of the full (n+1)-row matrix of T$: L'[0] = T[n-1] = bwt[index],
//    L'[index+1] = '$', L'[r] = bwt[r-1] otherwise.  Symbols: '$' = 0, byte c = c+1.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t row_symbol(const uint8_t *__restrict__ B, uint32_t r, uint32_t index)
{
    const uint32_t x = B[r == 0 ? index : r - 1];          // one branch-free load, so unrolled callers overlap them
    return (r == index + 1) ? 0u : x + 1;
}

// (8 copies of the 512-bin histogram per workgroup, copy = lane & 7 at a stride of 513 words: equal symbols inside a wave
//  serialise on an LDS atomic, and Zipf data puts ten lanes of a wave on the most frequent symbol -- the form k_fs_hist
//  uses on the encode side; with one copy per wave this kernel ran at 1.1 TB/s, a third of that one)
__global__ __launch_bounds__(256) void k_ibwt_hist(const uint8_t *__restrict__ bwt, size_t bwt_stride,
                                                   const int *__restrict__ d_index, uint32_t n,
                                                   uint32_t *__restrict__ tile_hist, uint32_t max_tiles,
                                                   uint32_t *__restrict__ d_status)
{
    __shared__ uint32_t s_h[8 * 513];
    const uint32_t b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const uint32_t rows = n + 1, base = t * LF_TILE;
    if (base >= rows) return;
    for (uint32_t i = tid; i < 8 * 513; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint8_t *B = bwt + (size_t)b * bwt_stride;
    uint32_t index = (uint32_t)d_index[b];
    if (index >= n) { index = n - 1; if (d_status && t == 0 && tid == 0) atomicOr(d_status, ST_CORRUPT); }   // row index from the stream
    uint32_t *H8 = s_h + (tid & 7) * 513;
    // rows base .. base + LF_TILE are the bytes bwt[base - 1 ..]: a thread takes groups of 8 consecutive rows, each read
    // as one unaligned 8-byte load where that stays inside the block and clear of the two special rows
#pragma unroll
    for (uint32_t g = 0; g < LF_TILE / 2048; g++) {
        const uint32_t r0 = base + g * 2048 + tid * 8;
        if (r0 >= 1 && r0 + 8 <= rows && !(index + 1 >= r0 && index + 1 < r0 + 8)) {
            uint64_t q;
            __builtin_memcpy(&q, B + r0 - 1, 8);
#pragma unroll
            for (int k = 0; k < 8; k++) atomicAdd(&H8[(uint32_t)((q >> (8 * k)) & 0xFFu) + 1u], 1u);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t r = r0 + k;
                if (r < rows) atomicAdd(&H8[row_symbol(B, r, index)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *H = tile_hist + ((size_t)b * max_tiles + t) * 512;
    for (uint32_t d = tid; d < 512; d += 256) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) c += s_h[k * 513 + d];
        H[d] = c;
    }
}

// LF(r) = start of the symbol's bucket + number of equal symbols in earlier rows
__global__ __launch_bounds__(256) void k_ibwt_lf(const uint8_t *__restrict__ bwt, size_t bwt_stride,
                                                 const int *__restrict__ d_index, uint32_t n,
                                                 const uint32_t *__restrict__ tile_hist,
                                                 const uint32_t *__restrict__ digit_base, uint32_t max_tiles,
                                                 uint32_t *__restrict__ lf, size_t lf_stride)
{
    __shared__ uint32_t s_wc[4][512];
    // (peers found by a ballot per symbol bit, wave_match<9>.  Measured against it: every lane ORs its bit into a per-symbol
    //  64-bit LDS entry and reads the entry back -- the form k_mtf_encode uses -- 548 us per 256 blocks against 523: the
    //  same-address atomics of the frequent symbols cost what the ballots do.)
    const uint32_t b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, l = tid & 63;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rows = n + 1, base = t * LF_TILE;
    if (base >= rows) return;
    for (uint32_t i = tid; i < 4 * 512; i += 256) (&s_wc[0][0])[i] = 0;
    __syncthreads();
    const uint8_t *B = bwt + (size_t)b * bwt_stride;
    const uint32_t index = min((uint32_t)d_index[b], n - 1);
    uint32_t sy[LF_TILE / 256], rk[LF_TILE / 256];
#pragma unroll
    for (int k = 0; k < LF_TILE / 256; k++) {                 // loads first: the wave barriers below pin them
        const uint32_t r = base + w * (LF_TILE / 4) + k * 64 + l;
        sy[k] = row_symbol(B, r < rows ? r : 0u, index);
    }
#pragma unroll
    for (int k = 0; k < LF_TILE / 256; k++) {
        const uint32_t r = base + w * (LF_TILE / 4) + k * 64 + l;
        const bool valid = r < rows;
        const uint32_t d = valid ? sy[k] : 0u;
        sy[k] = d;
        const uint64_t peers = wave_match<9>(d, __ballot(valid));
        const uint32_t pre = mbcnt(peers), tot = (uint32_t)__popcll(peers);
        const uint32_t old = s_wc[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && pre == 0) s_wc[w][d] = old + tot;
        __builtin_amdgcn_wave_barrier();
        rk[k] = old + pre;
    }
    __syncthreads();
    // per-wave bases: tile prefix + earlier waves of this tile
    const uint32_t *TH = tile_hist + ((size_t)b * max_tiles + t) * 512;
    const uint32_t *DB = digit_base + (size_t)b * 512;
    for (uint32_t d = tid; d < 512; d += 256) {
        const uint32_t c0 = s_wc[0][d], c1 = s_wc[1][d], c2 = s_wc[2][d];
        const uint32_t g = DB[d] + TH[d];
        s_wc[0][d] = g; s_wc[1][d] = g + c0; s_wc[2][d] = g + c0 + c1; s_wc[3][d] = g + c0 + c1 + c2;
    }
    __syncthreads();
    uint32_t *LF = lf + (size_t)b * lf_stride;
#pragma unroll
    for (int k = 0; k < LF_TILE / 256; k++) {
        const uint32_t r = base + w * (LF_TILE / 4) + k * 64 + l;
        if (r < rows) LF[r] = (sy[k] << 21) | (s_wc[w][sy[k]] + rk[k]);
    }
}

// Single LF walk.  Measured on MI355X (tools/probes/gather_probe.hip): dependent random 4-byte
// reads over a >256 MiB working set top out at ~55 G accesses/s (one 128-byte line from HBM per
// access), and the walk below runs at that ceiling -- so the text is emitted DURING the one walk
// instead of walking twice.  A lane starts at splitter row s*SPLIT and emits into a private
// SLOT-byte slot; a walk that fills its slot before reaching the next splitter row continues in
// a freshly allocated ("dynamic") segment, so slots are bounded whatever the cycle looks like.
//   seg_info[id] = len | next << 9 ; ids < nsplit are the static splitters.
#ifndef GLC_WALK_PAD
#define GLC_WALK_PAD 100
#endif
template <int PAD>
__global__ __launch_bounds__(256) void k_ibwt_walk(const uint32_t *__restrict__ lf, size_t lf_stride, uint32_t n,
                                                   uint32_t *__restrict__ seg_info, uint32_t max_seg,
                                                   uint32_t *__restrict__ seg_count, uint8_t *__restrict__ tmp)
{
    // PAD extra VGPRs are held live across the walk to cap its residency at 4 waves per SIMD (1024 lanes
    // per CU): a pointer chase gains nothing beyond that, and under stage pipelining every slot it does
    // not hold goes to the LDS/VALU-bound stage A of the next batch.  Registers, not LDS, are the resource
    // to spend on the cap, because the kernels it shares the CU with live on LDS.  Measured, 4 GiB
    // pipelined decode: no cap 25.3 GB/s, 40 KB LDS cap 25.9, 118-VGPR cap 26.6; back to back 22.8 -> 23.5.
    uint32_t pad[PAD > 0 ? PAD : 1];
    if (PAD > 0) {
#pragma unroll
        for (int i = 0; i < PAD; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(pad[i]) : "v"(threadIdx.x + i));
    }
    // XCD-aware order: all the walks of a block start on ONE XCD, so its 4 MiB LF table competes for one L2 with the
    // three other blocks in flight there instead of for all eight with thirty-one (decode 27.6 -> 31.5 GB/s one plan)
    uint32_t gx, gy;
    xcd_order(gx, gy);
    const uint32_t b = gy, s = gx * 256 + threadIdx.x;
    const uint32_t rows = n + 1, nsplit = (rows + SPLIT - 1) / SPLIT;
    if (s >= nsplit) return;
    const uint32_t *LF = lf + (size_t)b * lf_stride;
    uint32_t *SI = seg_info + (size_t)b * max_seg;
    uint4 *T = reinterpret_cast<uint4 *>(tmp + (size_t)b * max_seg * SLOT);
    uint32_t id = s, r = s * SPLIT, len = 0, steps = 0;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (;;) {
        const uint32_t wv = LF[r];
        const uint32_t byte = ((wv >> 21) - 1u) & 0xFFu;
        const uint32_t sh = byte << (8 * (len & 3)), q = (len >> 2) & 3;
        w0 |= (q == 0) ? sh : 0u; w1 |= (q == 1) ? sh : 0u; w2 |= (q == 2) ? sh : 0u; w3 |= (q == 3) ? sh : 0u;
        r = wv & LF_MASK;
        len++; steps++;
        const bool at_split = (r & (SPLIT - 1)) == 0 || steps > rows;
        if ((len & 15) == 0 || at_split) {
            T[(size_t)id * (SLOT / 16) + ((len - 1) >> 4)] = make_uint4(w0, w1, w2, w3);
            w0 = w1 = w2 = w3 = 0;
        }
        if (at_split) { SI[id] = len | ((r / SPLIT) << 9); break; }
        if (len == SLOT) {
            const uint32_t nid = atomicAdd(&seg_count[b], 1u);
            if (nid >= max_seg) { SI[id] = len; break; }            // corrupt stream: give up on this path
            SI[id] = len | (nid << 9);
            id = nid; len = 0;
        }
    }
    if (PAD > 0) {
#pragma unroll
        for (int i = 0; i < PAD; i++) asm volatile("; keep %0" :: "v"(pad[i]));
    }
}

// (Measured and rejected, three times: persistent forms -- 32 K to 512 K lanes taking (block, splitter)
//  pieces in block-major order so that only a few blocks' LF tables are live.  A bare chase over such
//  a window reaches 125 G steps/s with 64 K lanes (tools/probes/window_probe.hip), but with the real
//  step (symbol packing, slot stores, piece bookkeeping) and one wave per SIMD the issue time of the
//  step is no longer hidden: wave-pooled tickets 5.2-5.6 ms, strided static pieces 6.3-6.9 ms
//  (imbalance), against 5.2 ms for the plain launch below.  Round 3, a fourth time, with what round 2 added -- the
//  XCD-aware order -- built in: persistent lanes that take the next (block, splitter) piece from a pool their wave
//  refills from ONE counter per XCD, so that every lane stays busy (in the plain launch a wave lasts as long as its
//  longest walk, ~4.7 x the mean: four lanes of five idle) and all the waves of an XCD work through the same block's
//  table.  1 / 2 / 4 / 8 waves per SIMD: walk 4.07 / 4.6 / 5.2 / 5.3 ms per 256 blocks against 3.6 for the plain
//  launch -- MORE lanes on one table is slower, fewer blocks in flight is not faster: the walk is not short of
//  parallelism or of cache, it runs at the rate the memory system takes dependent random 4-byte reads.
//  The way forward is more pieces per block (16-row splitters, two-level ordering) only if that rate can be raised.)
__global__ void k_ibwt_seg_init(uint32_t *__restrict__ seg_count, uint32_t n, uint32_t nblk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nblk) seg_count[i] = (n + 1 + SPLIT - 1) / SPLIT;
}

// Text position of every segment: list ranking (pointer jumping) in LDS over the <= MAX_SEG
// segments of the cycle that starts at segment 0 (row 0 = the "$" suffix).  d[s] = symbols
// from the start of s to the end of the cycle, so segment s emits positions d[s]-2, d[s]-3, ...
constexpr uint32_t MAX_SEG  = MAX_SPLITS + ((1u << 20) + 1) / SLOT + 8;
constexpr uint32_t RANK_NT  = 1024;
constexpr uint32_t RANK_E   = (MAX_SEG + RANK_NT - 1) / RANK_NT;
constexpr uint32_t SEG_NIL  = 0xFFFFu;
static_assert(MAX_SEG < SEG_NIL, "segment ids are kept as 16 bits in k_ibwt_rank");

__global__ __launch_bounds__(RANK_NT) void k_ibwt_rank(const uint32_t *__restrict__ seg_info, uint32_t max_seg,
                                                       const uint32_t *__restrict__ seg_count,
                                                       int *__restrict__ seg_pos)
{
    __shared__ uint32_t s_d[MAX_SEG];
    __shared__ uint16_t s_nx[MAX_SEG];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t nseg = min(seg_count[b], min(max_seg, MAX_SEG));
    const uint32_t *SI = seg_info + (size_t)b * max_seg;
    for (uint32_t i = tid; i < nseg; i += RANK_NT) {
        const uint32_t v = SI[i], nx = v >> 9;
        s_d[i] = v & 511u;
        s_nx[i] = (uint16_t)((nx == 0 || nx >= nseg) ? SEG_NIL : nx);
    }
    __syncthreads();
    uint32_t nd[RANK_E], nn[RANK_E];
    for (uint32_t span = 1; span < nseg; span <<= 1) {
#pragma unroll
        for (uint32_t e = 0; e < RANK_E; e++) {
            const uint32_t i = tid + e * RANK_NT;
            if (i < nseg) {
                const uint32_t j = s_nx[i];
                nd[e] = s_d[i] + (j != SEG_NIL ? s_d[j] : 0u);
                nn[e] = (j != SEG_NIL) ? s_nx[j] : SEG_NIL;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < RANK_E; e++) {
            const uint32_t i = tid + e * RANK_NT;
            if (i < nseg) { s_d[i] = nd[e]; s_nx[i] = (uint16_t)nn[e]; }
        }
        __syncthreads();
    }
    int *P = seg_pos + (size_t)b * max_seg;
    for (uint32_t i = tid; i < nseg; i += RANK_NT) P[i] = (int)s_d[i] - 2;
}

// slots -> text (reversed within a segment).  One wave per segment.
__global__ __launch_bounds__(256) void k_ibwt_emit(const uint8_t *__restrict__ tmp, const uint32_t *__restrict__ seg_info,
                                                   const int *__restrict__ seg_pos, uint32_t max_seg,
                                                   const uint32_t *__restrict__ seg_count, uint32_t n,
                                                   uint8_t *__restrict__ out, size_t out_stride)
{
    const uint32_t b = blockIdx.y, l = threadIdx.x & 63;
    const uint32_t nseg = min(seg_count[b], max_seg);
    uint8_t *O = out + (size_t)b * out_stride;
    const uint32_t id0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * EMIT_SEGS;
    // the records of all EMIT_SEGS segments first, then all of their bytes, then the stores: three
    // memory round trips per wave instead of three per segment
    uint32_t len[EMIT_SEGS];
    int pos[EMIT_SEGS];
#pragma unroll
    for (uint32_t k = 0; k < EMIT_SEGS; k++) {
        const uint32_t id = id0 + k;
        const bool ok = id < nseg;
        const uint32_t si = seg_info[(size_t)b * max_seg + (ok ? id : 0u)];
        pos[k] = seg_pos[(size_t)b * max_seg + (ok ? id : 0u)];
        len[k] = ok ? (si & 511u) : 0u;
    }
    uint8_t v[EMIT_SEGS][SLOT / 64];
#pragma unroll
    for (uint32_t k = 0; k < EMIT_SEGS; k++) {
        const uint8_t *S = tmp + ((size_t)b * max_seg + min(id0 + k, max_seg - 1)) * SLOT;
#pragma unroll
        for (uint32_t j = 0; j < SLOT / 64; j++) {
            const uint32_t i = l + 64 * j;
            v[k][j] = S[i < len[k] ? i : 0u];
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < EMIT_SEGS; k++) {
#pragma unroll
        for (uint32_t j = 0; j < SLOT / 64; j++) {
            const uint32_t i = l + 64 * j;
            const uint32_t t = (uint32_t)(pos[k] - (int)i);
            if (i < len[k] && t < n) O[t] = v[k][j];
        }
    }
}

// ---------------------------------------------------------------------------
#define GLC_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t decode_scratch_alloc(DecodeScratch &s, uint32_t nmax, uint32_t rows)
{
    s.nmax = nmax; s.rows = rows;
    s.max_tiles = (nmax + 1 + LF_TILE - 1) / LF_TILE;
    s.max_split = (nmax + 1 + SPLIT - 1) / SPLIT;
    s.max_chunks = (nmax + IMTF_CHUNK - 1) / IMTF_CHUNK;
    size_t total = 0;
    auto A = [&](void **p, size_t bytes) -> hipError_t { total += bytes; return hipMalloc(p, bytes); };
    GLC_TRY(A((void **)&s.mtf, (size_t)nmax * rows));
    GLC_TRY(A((void **)&s.bwt, (size_t)nmax * rows));
    GLC_TRY(A((void **)&s.bwt2, (size_t)nmax * rows));
    GLC_TRY(A((void **)&s.lf, ((size_t)nmax + 4) * rows * 4));
    GLC_TRY(A((void **)&s.lut, ((size_t)rows << DEC_LUT_BITS) * 4));
    GLC_TRY(A((void **)&s.nodes, (size_t)rows * HUFF_NODES * 4));
    GLC_TRY(A((void **)&s.tile_hist, (size_t)rows * s.max_tiles * 512 * 4));
    GLC_TRY(A((void **)&s.digit_base, (size_t)rows * 512 * 4));
    s.max_seg = s.max_split + (nmax + 1) / SLOT + 8;
    GLC_TRY(A((void **)&s.seg, (size_t)rows * s.max_seg * 4));
    GLC_TRY(A((void **)&s.seg_pos, (size_t)rows * s.max_seg * 4));
    GLC_TRY(A((void **)&s.seg_count, (size_t)rows * 4));
    GLC_TRY(A((void **)&s.slots, (size_t)rows * s.max_seg * SLOT));
    GLC_TRY(A((void **)&s.ilists, (size_t)rows * s.max_chunks * 256));
    s.bytes = total;
    return hipSuccess;
}

void decode_scratch_free(DecodeScratch &s)
{
    void *ps[] = {s.mtf, s.bwt, s.bwt2, s.lf, s.lut, s.nodes, s.tile_hist, s.digit_base, s.seg, s.ilists, s.seg_pos, s.seg_count, s.slots};
    for (void *p : ps) if (p) (void)hipFree(p);
    s = DecodeScratch();
}

// Stage A: Huffman decode + inverse MTF -> BWT bytes in `bwt` (s.bwt or s.bwt2).
// Stage B: inverse BWT of `bwt` -> d_out.  The two stages use disjoint scratch apart from `bwt`, so
// stage A of the next batch can run on another stream while stage B of this one walks its LF cycles
// (glcPlanSetPipelining): A is LDS/VALU work, B is a memory-latency-bound pointer chase.
hipError_t decode_stage_a(hipStream_t st, const uint32_t *d_hist, const uint32_t *d_offsets, size_t offset_stride,
                          const uint32_t *d_comp, size_t comp_stride_words, uint32_t n, uint32_t nblk, DecodeScratch &s,
                          uint8_t *bwt, uint32_t *d_status, const unsigned long long *d_block_off)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    const uint32_t nchunks = (n + IMTF_CHUNK - 1) / IMTF_CHUNK;
    const double units = (double)n * nblk;
    int pi = s.prof ? s.prof->begin(PROF_DEC_HUFF, st) : -1;
    hipLaunchKernelGGL(k_dec_prepare, dim3(nblk), dim3(256), 0, st, d_hist, s.lut, s.nodes, n, d_status);
    if ((uint64_t)nsub * nblk >= DL_MIN_SUBS)
        hipLaunchKernelGGL(k_dec_huff_lanes, dim3((nsub + DL_NT - 1) / DL_NT, nblk), dim3(DL_NT), 0, st, d_comp, comp_stride_words,
                           d_offsets, offset_stride, s.lut, s.nodes, n, s.mtf, (size_t)s.nmax, d_status, d_block_off);
    else
        hipLaunchKernelGGL(k_dec_huff, dim3((nsub + DH_WAVES - 1) / DH_WAVES, nblk), dim3(DH_WAVES * 64), 0, st, d_comp, comp_stride_words,
                           d_offsets, offset_stride, s.lut, s.nodes, n, s.mtf, (size_t)s.nmax, d_status, d_block_off);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_IMTF_POS, st) : -1;
    static const bool rings = getenv("GLC_IMTF_RINGS") != nullptr;   // A/B: the ring form of pass 1
    if (rings)
        hipLaunchKernelGGL(k_imtf_pos, dim3((nchunks + 63) / 64, nblk), dim3(64), 0, st, s.mtf, (size_t)s.nmax, n, s.ilists,
                           s.max_chunks, bwt, (size_t)s.nmax);
    else
        hipLaunchKernelGGL(k_imtf_pos_deque, dim3((nchunks + 63) / 64, nblk), dim3(64), 0, st, s.mtf, (size_t)s.nmax, n, s.ilists,
                           s.max_chunks, bwt, (size_t)s.nmax);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_IMTF_REST, st) : -1;
    hipLaunchKernelGGL(k_imtf_scan, dim3(nblk), dim3(64), 0, st, s.ilists, n, s.max_chunks);
    hipLaunchKernelGGL(k_imtf_apply, dim3(nchunks, nblk), dim3(256), 0, st, bwt, (size_t)s.nmax, n, s.ilists,
                       s.max_chunks);
    if (pi >= 0) s.prof->end(pi, units, st);
    return hipGetLastError();
}

hipError_t decode_stage_b(hipStream_t st, const int *d_bwt_index, const uint8_t *bwt, uint8_t *d_out, uint32_t n,
                          uint32_t nblk, DecodeScratch &s, uint32_t *d_status)
{
    if (n == 0 || n > s.nmax || nblk == 0 || nblk > s.rows) return hipErrorInvalidValue;
    const uint32_t rows = n + 1, tiles = (rows + LF_TILE - 1) / LF_TILE, nsplit = (rows + SPLIT - 1) / SPLIT;
    const size_t lf_stride = (size_t)s.nmax + 4;
    const double units = (double)n * nblk;
    int pi = s.prof ? s.prof->begin(PROF_IBWT_LF, st) : -1;
    hipLaunchKernelGGL(k_ibwt_hist, dim3(tiles, nblk), dim3(256), 0, st, bwt, (size_t)s.nmax, d_bwt_index, n,
                       s.tile_hist, s.max_tiles, d_status);
    GLC_TRY(tile_hist_scan9(st, s.tile_hist, rows, s.digit_base, s.max_tiles, nblk, LF_TILE));
    hipLaunchKernelGGL(k_ibwt_lf, dim3(tiles, nblk), dim3(256), 0, st, bwt, (size_t)s.nmax, d_bwt_index, n,
                       s.tile_hist, s.digit_base, s.max_tiles, s.lf, lf_stride);
    hipLaunchKernelGGL(k_ibwt_seg_init, dim3((nblk + 255) / 256), dim3(256), 0, st, s.seg_count, n, nblk);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_IBWT_WALK, st) : -1;
    hipLaunchKernelGGL(k_ibwt_walk<GLC_WALK_PAD>, dim3((nsplit + 255) / 256, nblk), dim3(256), 0, st, s.lf, lf_stride, n, s.seg,
                       s.max_seg, s.seg_count, s.slots);
    if (pi >= 0) s.prof->end(pi, units, st);
    pi = s.prof ? s.prof->begin(PROF_IBWT_EMIT, st) : -1;
    hipLaunchKernelGGL(k_ibwt_rank, dim3(nblk), dim3(RANK_NT), 0, st, s.seg, s.max_seg, s.seg_count, s.seg_pos);
    const uint32_t seg_bound = nsplit + rows / SLOT + 1;
    hipLaunchKernelGGL(k_ibwt_emit, dim3((seg_bound + 4 * EMIT_SEGS - 1) / (4 * EMIT_SEGS), nblk), dim3(256), 0, st,
                       s.slots, s.seg, s.seg_pos, s.max_seg, s.seg_count, n, d_out, (size_t)n);
    if (pi >= 0) s.prof->end(pi, units, st);
    return hipGetLastError();
}

hipError_t decode_blocks(hipStream_t st, const int *d_bwt_index, const uint32_t *d_hist, const uint32_t *d_offsets,
                         size_t offset_stride, const uint32_t *d_comp, size_t comp_stride_words, uint8_t *d_out,
                         uint32_t n, uint32_t nblk, DecodeScratch &s, MtfScratch & /*ms*/, uint32_t *d_status,
                         const unsigned long long *d_block_off)
{
    GLC_TRY(decode_stage_a(st, d_hist, d_offsets, offset_stride, d_comp, comp_stride_words, n, nblk, s, s.bwt, d_status, d_block_off));
    return decode_stage_b(st, d_bwt_index, s.bwt, d_out, n, nblk, s, d_status);
}

} // namespace glc
