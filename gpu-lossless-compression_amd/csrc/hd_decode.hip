// hd_decode.hip -- CUHD-style Huffman-only decoder (include/glc_hd.h).  gfx950 / wave64.
//
// Same job as the reference's 4-phase self-synchronising decoder
// (cuhd-icpp/src/cuhd_gpu_decoder.cu:16-523): <= 11-bit canonical codes, 32-bit units,
// MSB first, a 2048-entry {length, symbol} table (cuhd_codetable.h:20-23), but a
// different way to find where each thread must start:
//
//   A codeword straddling a span boundary leaves 0..10 bits in the next span, so a span
//   of 32 units is a FUNCTION  f: start offset (11 values) -> (offset into the next
//   span, symbols decoded).  Functions compose associatively:
//     k_hd_span_functions  each lane decodes its span once per start offset (LUT in LDS,
//                          units staged in LDS with a 33-unit pitch = conflict-free),
//                          then a workgroup-wide inclusive scan of the 11-entry tables;
//                          writes each span's exclusive prefix and the workgroup total.
//     k_hd_walk            two-level walk over the workgroup totals (512 per chunk):
//                          compose chunks, walk the <= 512 chunks, expand.
//     k_hd_emit            every lane now knows its exact start bit and output index and
//                          decodes its span once more, writing symbols.
//   No data-dependent iteration count and no device->host convergence flag (the
//   reference re-launches phase 2 until a flag copied back to the host says "synced",
//   cuhd_gpu_decoder.cu:459-495).
#include "glc_device.h"
#include "glc_internal.h"
#include "../../include/glc_hd.h"

#include <algorithm>
#include <string.h>
#include <vector>

namespace glc {

constexpr int HD_SPAN   = 32;                    // units per lane
constexpr int HD_PITCH  = HD_SPAN + 1;           // LDS pitch (bank-conflict-free) + look-ahead unit
constexpr int HD_LANES  = 256;
constexpr int HD_WG_UNITS = HD_SPAN * HD_LANES;  // 8192 units = 32 KiB per workgroup
constexpr int HD_NOFF   = 11;                    // start offsets 0..10
constexpr int HD_TSTRIDE = 12;                   // words per stored table
constexpr int HD_CHUNK  = 512;                   // workgroup functions per walk chunk
constexpr int HD_SPAN_BITS = HD_SPAN * 32;

__device__ __forceinline__ void hd_stage(const uint32_t *__restrict__ units, size_t nunits, size_t base_unit,
                                         uint32_t *s_u, const uint16_t *__restrict__ lut, uint16_t *s_lut)
{
    const uint32_t tid = threadIdx.x;
    {   // batches of 8 independent loads, then the LDS stores (a load->store loop is one latency per trip)
        uint32_t q[8];
#pragma unroll
        for (int r = 0; r < 8; r++) q[r] = lut[r * HD_LANES + tid];        // two 16-bit entries per word are not
#pragma unroll                                                            // worth it: the table is read once
        for (int r = 0; r < 8; r++) s_lut[r * HD_LANES + tid] = (uint16_t)q[r];
        const bool full = base_unit + HD_WG_UNITS <= nunits;
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < HD_WG_UNITS; i0 += 8 * HD_LANES) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = i0 + r * HD_LANES + tid;
                const size_t gu = base_unit + i;
                q[r] = units[full || gu < nunits ? gu : 0];
                if (!full && gu >= nunits) q[r] = 0u;
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = i0 + r * HD_LANES + tid;
                s_u[(i >> 5) * HD_PITCH + (i & 31)] = q[r];
            }
        }
    }
    __syncthreads();
    {   // look-ahead unit of every lane = first unit of the next lane / next workgroup
        const size_t gu = base_unit + HD_WG_UNITS;
        const uint32_t nxt = (tid + 1 < HD_LANES) ? s_u[(tid + 1) * HD_PITCH] : (gu < nunits ? units[gu] : 0u);
        s_u[tid * HD_PITCH + HD_SPAN] = nxt;
    }
    __syncthreads();
}

// decode one span from bit `o`; EMIT: write symbols to out[base + k] (k-th symbol) while < nsym
template <bool EMIT>
__device__ __forceinline__ uint32_t hd_decode_span(const uint32_t *U, const uint16_t *s_lut, uint32_t o,
                                                   uint8_t *__restrict__ out, size_t base, size_t nsym)
{
    uint32_t pos = o, cnt = 0;
    uint64_t w = (((uint64_t)U[0] << 32) | U[1]) << o;
    uint32_t valid = 64 - o, next = 2;
    while (pos < HD_SPAN_BITS) {
        const uint32_t e = s_lut[(uint32_t)(w >> (64 - GLC_HD_MAX_LEN))];
        const uint32_t len = e >> 8;
        if (EMIT) { if (base + cnt < nsym) out[base + cnt] = (uint8_t)e; }
        w <<= len; pos += len; valid -= len; cnt++;
        if (valid <= 32 && next <= HD_SPAN) { w |= (uint64_t)U[next] << (32 - valid); valid += 32; next++; }
    }
    return (cnt << 4) | (pos - HD_SPAN_BITS);
}

// Path from offset 0, recording for every unit the first codeword boundary inside it (every 32-bit
// unit holds at least two boundaries: codewords are <= 11 bits): chk[u] = symbols before it << 5 | bit.
__device__ __forceinline__ uint32_t hd_decode_ref(const uint32_t *U, const uint16_t *s_lut, uint16_t *chk)
{
    uint32_t pos = 0, cnt = 0, last_u = 0xFFFFFFFFu;
    uint64_t w = ((uint64_t)U[0] << 32) | U[1];
    uint32_t valid = 64, next = 2;
    while (pos < HD_SPAN_BITS) {
        const uint32_t u = pos >> 5;
        if (u != last_u) { chk[u] = (uint16_t)((cnt << 5) | (pos & 31)); last_u = u; }
        const uint32_t len = s_lut[(uint32_t)(w >> (64 - GLC_HD_MAX_LEN))] >> 8;
        w <<= len; pos += len; valid -= len; cnt++;
        if (valid <= 32 && next <= HD_SPAN) { w |= (uint64_t)U[next] << (32 - valid); valid += 32; next++; }
    }
    return (cnt << 4) | (pos - HD_SPAN_BITS);
}

// Path from offset o > 0: Huffman codes self-synchronise, so it usually falls onto the reference
// path within a few codewords; from there on the two are identical, and the result is the
// reference's (end offset, count) corrected by the symbols decoded so far.  Checked once per unit.
__device__ __forceinline__ uint32_t hd_decode_merge(const uint32_t *U, const uint16_t *s_lut, const uint16_t *chk,
                                                    uint32_t ref, uint32_t o)
{
    uint32_t pos = o, cnt = 0, last_u = 0;                     // unit 0 holds the start itself: no check there
    uint64_t w = (((uint64_t)U[0] << 32) | U[1]) << o;
    uint32_t valid = 64 - o, next = 2;
    while (pos < HD_SPAN_BITS) {
        const uint32_t u = pos >> 5;
        if (u != last_u) {
            const uint32_t c = chk[u];
            if ((c & 31u) == (pos & 31u)) return (((ref >> 4) - (c >> 5) + cnt) << 4) | (ref & 15u);
            last_u = u;
        }
        const uint32_t len = s_lut[(uint32_t)(w >> (64 - GLC_HD_MAX_LEN))] >> 8;
        w <<= len; pos += len; valid -= len; cnt++;
        if (valid <= 32 && next <= HD_SPAN) { w |= (uint64_t)U[next] << (32 - valid); valid += 32; next++; }
    }
    return (cnt << 4) | (pos - HD_SPAN_BITS);
}

constexpr int HD_CHK_PITCH = 34;                                // u16 per lane (17 words: odd, conflict-free)

__global__ __launch_bounds__(HD_LANES) void k_hd_span_functions(const uint32_t *__restrict__ units, size_t nunits,
                                                                const uint16_t *__restrict__ lut,
                                                                uint32_t *__restrict__ pexcl,
                                                                uint32_t *__restrict__ fwg)
{
    __shared__ uint32_t s_u[HD_LANES * HD_PITCH];
    __shared__ uint16_t s_lut[2048];
    __shared__ uint32_t s_tab[2][HD_LANES][HD_NOFF];
    __shared__ uint16_t s_chk[HD_LANES * HD_CHK_PITCH];
    const uint32_t tid = threadIdx.x;
    const size_t wg = blockIdx.x;
    hd_stage(units, nunits, wg * (size_t)HD_WG_UNITS, s_u, lut, s_lut);
    const uint32_t *U = s_u + tid * HD_PITCH;
    uint16_t *chk = s_chk + tid * HD_CHK_PITCH;
    const uint32_t ref = hd_decode_ref(U, s_lut, chk);
    s_tab[0][tid][0] = ref;
#pragma unroll 1
    for (uint32_t o = 1; o < HD_NOFF; o++) s_tab[0][tid][o] = hd_decode_merge(U, s_lut, chk, ref, o);
    // inclusive scan of the span functions across the 256 lanes (Hillis-Steele, composition B(A(.)))
    int src = 0;
    for (uint32_t d = 1; d < HD_LANES; d <<= 1) {
        __syncthreads();
        for (uint32_t o = 0; o < HD_NOFF; o++) {
            uint32_t r = s_tab[src][tid][o];
            if (tid >= d) {
                const uint32_t a = s_tab[src][tid - d][o];            // earlier part, applied first
                const uint32_t bb = s_tab[src][tid][a & 15];
                r = (((a >> 4) + (bb >> 4)) << 4) | (bb & 15);
            }
            s_tab[src ^ 1][tid][o] = r;
        }
        src ^= 1;
    }
    __syncthreads();
    uint32_t *P = pexcl + (wg * HD_LANES + tid) * HD_TSTRIDE;
    for (uint32_t o = 0; o < HD_NOFF; o++) P[o] = tid ? s_tab[src][tid - 1][o] : o;   // exclusive; identity for lane 0
    if (tid < HD_NOFF) fwg[wg * HD_TSTRIDE + tid] = s_tab[src][HD_LANES - 1][tid];
}

// mode 0: compose chunk c of up to 512 functions F -> G[c]
// mode 1: (single workgroup) walk the chunk functions G from (offset 0, base 0) -> start_chunk
// mode 2: expand chunk c: per-function start (offset, base) from start_chunk[c]
__global__ __launch_bounds__(64) void k_hd_walk(int mode, const uint32_t *__restrict__ F, size_t nF,
                                                uint32_t *__restrict__ G, uint32_t *__restrict__ chunk_off,
                                                unsigned long long *__restrict__ chunk_base,
                                                uint32_t *__restrict__ start_off,
                                                unsigned long long *__restrict__ start_base)
{
    __shared__ uint32_t s_row[16];
    const uint32_t l = threadIdx.x;
    if (mode == 1) {
        if (l == 0) {
            uint32_t o = 0; unsigned long long base = 0;
            for (size_t c = 0; c < nF; c++) {
                chunk_off[c] = o; chunk_base[c] = base;
                const uint32_t e = G[c * HD_TSTRIDE + o];
                base += e >> 4; o = e & 15;
            }
        }
        return;
    }
    const size_t c = blockIdx.x, lo = c * HD_CHUNK, hi = min(nF, lo + (size_t)HD_CHUNK);
    uint32_t cur = (mode == 0) ? l : chunk_off[c];
    unsigned long long acc = (mode == 0) ? 0ull : chunk_base[c];
    for (size_t i = lo; i < hi; i++) {
        if (l < HD_TSTRIDE) s_row[l] = F[i * HD_TSTRIDE + l];
        __syncthreads();
        if (mode == 2 && l == 0) { start_off[i] = cur; start_base[i] = acc; }
        const uint32_t e = s_row[cur < HD_NOFF ? cur : 0];
        acc += e >> 4; cur = e & 15;
        __syncthreads();
    }
    if (mode == 0 && l < HD_NOFF) G[c * HD_TSTRIDE + l] = (uint32_t)((acc << 4) | cur);
}

__global__ __launch_bounds__(HD_LANES) void k_hd_emit(const uint32_t *__restrict__ units, size_t nunits,
                                                      const uint16_t *__restrict__ lut,
                                                      const uint32_t *__restrict__ pexcl,
                                                      const uint32_t *__restrict__ start_off,
                                                      const unsigned long long *__restrict__ start_base,
                                                      uint8_t *__restrict__ out, size_t nsym)
{
    __shared__ uint32_t s_u[HD_LANES * HD_PITCH];
    __shared__ uint16_t s_lut[2048];
    const uint32_t tid = threadIdx.x;
    const size_t wg = blockIdx.x;
    hd_stage(units, nunits, wg * (size_t)HD_WG_UNITS, s_u, lut, s_lut);
    const uint32_t ow = start_off[wg];
    const uint32_t p = pexcl[(wg * HD_LANES + tid) * HD_TSTRIDE + ow];
    const size_t base = (size_t)start_base[wg] + (p >> 4);
    if (base >= nsym) return;
    // decode and write: four symbols per dword store once the output index is 4-aligned (the bytes
    // before that belong to the previous lane's dword and go out one by one, as does the tail)
    const uint32_t *U = s_u + tid * HD_PITCH;
    const uint32_t o = p & 15;
    uint32_t pos = o;
    uint64_t w = (((uint64_t)U[0] << 32) | U[1]) << o;
    uint32_t valid = 64 - o, next = 2, acc = 0;
    size_t idx = base;
    const bool al = (reinterpret_cast<size_t>(out) & 3) == 0;
    while (pos < HD_SPAN_BITS && idx < nsym) {
        const uint32_t e = s_lut[(uint32_t)(w >> (64 - GLC_HD_MAX_LEN))];
        const uint32_t len = e >> 8, k = (uint32_t)idx & 3u;
        acc |= (e & 0xFFu) << (8 * k);
        if (k == 3) {
            if (al && idx - base >= 3) *reinterpret_cast<uint32_t *>(out + idx - 3) = acc;
            else for (uint32_t q = (idx - base >= 3) ? 0u : 3u - (uint32_t)(idx - base); q < 4; q++) out[idx - 3 + q] = (uint8_t)(acc >> (8 * q));
            acc = 0;
        }
        idx++;
        w <<= len; pos += len; valid -= len;
        if (valid <= 32 && next <= HD_SPAN) { w |= (uint64_t)U[next] << (32 - valid); valid += 32; next++; }
    }
    {   // tail: the bytes of an unfinished dword
        const uint32_t k = (uint32_t)idx & 3u;                 // bytes [idx - k, idx) pending, but not before `base`
        const uint32_t have = (uint32_t)((idx - base) < k ? (idx - base) : k);
        for (uint32_t q = k - have; q < k; q++) out[idx - k + q] = (uint8_t)(acc >> (8 * q));
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct HdLayout { size_t nspans, nwg, nchunks, o_pexcl, o_fwg, o_g, o_coff, o_cbase, o_soff, o_sbase, o_lut, total; };

static HdLayout hd_layout(size_t nunits)
{
    HdLayout L;
    L.nspans = (nunits + HD_SPAN - 1) / HD_SPAN;
    L.nwg = (L.nspans + HD_LANES - 1) / HD_LANES;
    L.nchunks = (L.nwg + HD_CHUNK - 1) / HD_CHUNK;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = (o + bytes + 255) & ~(size_t)255; return r; };
    L.o_pexcl = take(L.nwg * HD_LANES * HD_TSTRIDE * 4);
    L.o_fwg = take(L.nwg * HD_TSTRIDE * 4);
    L.o_g = take(L.nchunks * HD_TSTRIDE * 4);
    L.o_coff = take(L.nchunks * 4);
    L.o_cbase = take(L.nchunks * 8);
    L.o_soff = take(L.nwg * 4);
    L.o_sbase = take(L.nwg * 8);
    L.o_lut = take(2048 * 2);
    L.total = o;
    return L;
}

} // namespace glc

using namespace glc;

extern "C" {

// Length-limited Huffman by package-merge (max 11 bits), then canonical codes by (length, symbol).
// Same job as LLHuffmanEncoder::get_encoder_table (llhuffman_encoder.cc:160-262).
int glcHdBuildTable(const unsigned long long hist[256], unsigned char lens[256], unsigned short codes[256])
{
    if (!hist || !lens || !codes) return 0;
    struct Item { unsigned long long w; std::vector<uint16_t> leaf; };
    std::vector<int> syms;
    for (int s = 0; s < 256; s++) { lens[s] = 0; codes[s] = 0; if (hist[s]) syms.push_back(s); }
    const int m = (int)syms.size();
    if (m == 0) return 0;
    if (m == 1) { lens[syms[0]] = 1; codes[syms[0]] = 0; return 1; }
    std::stable_sort(syms.begin(), syms.end(), [&](int a, int b) { return hist[a] < hist[b]; });
    std::vector<Item> leaves(m);
    for (int i = 0; i < m; i++) { leaves[i].w = hist[syms[i]]; leaves[i].leaf.assign(m, 0); leaves[i].leaf[i] = 1; }
    std::vector<Item> prev = leaves;
    for (int level = 1; level < GLC_HD_MAX_LEN; level++) {
        std::vector<Item> pk;
        for (size_t i = 0; i + 1 < prev.size(); i += 2) {
            Item it; it.w = prev[i].w + prev[i + 1].w; it.leaf = prev[i].leaf;
            for (int k = 0; k < m; k++) it.leaf[k] = (uint16_t)(it.leaf[k] + prev[i + 1].leaf[k]);
            pk.push_back(std::move(it));
        }
        std::vector<Item> cur;
        size_t a = 0, b = 0;
        while (a < leaves.size() || b < pk.size()) {
            if (b >= pk.size() || (a < leaves.size() && leaves[a].w <= pk[b].w)) cur.push_back(leaves[a++]);
            else cur.push_back(std::move(pk[b++]));
        }
        prev.swap(cur);
    }
    std::vector<int> len(m, 0);
    for (int i = 0; i < 2 * m - 2 && i < (int)prev.size(); i++)
        for (int k = 0; k < m; k++) len[k] += prev[i].leaf[k];
    for (int k = 0; k < m; k++) { if (len[k] < 1 || len[k] > GLC_HD_MAX_LEN) return 0; lens[syms[k]] = (unsigned char)len[k]; }
    // canonical assignment
    std::vector<int> order;
    for (int s = 0; s < 256; s++) if (lens[s]) order.push_back(s);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] < lens[b]; });
    unsigned code = 0; int pl = lens[order[0]];
    for (int s : order) { code <<= (lens[s] - pl); pl = lens[s]; codes[s] = (unsigned short)code; code++; }
    return m;
}

size_t glcHdEncodeHost(const unsigned char *in, size_t nsym, const unsigned char lens[256],
                       const unsigned short codes[256], unsigned int *out_units, size_t cap_units)
{
    if (!in || !lens || !codes || !out_units) return 0;
    unsigned long long acc = 0; int na = 0; size_t nu = 0;
    for (size_t i = 0; i < nsym; i++) {
        const int l = lens[in[i]];
        if (l == 0) return 0;
        acc = (acc << l) | codes[in[i]]; na += l;
        if (na >= 32) {
            if (nu >= cap_units) return 0;
            na -= 32;
            out_units[nu++] = (unsigned int)(acc >> na);
            acc &= (1ull << na) - 1ull;
        }
    }
    if (na > 0) { if (nu >= cap_units) return 0; out_units[nu++] = (unsigned int)(acc << (32 - na)); }
    if (nu >= cap_units) return 0;
    out_units[nu++] = 0;                                     // pad unit (cuhd_input_buffer.cc:20-27)
    return nu;
}

size_t glcHdWorkBytes(size_t nunits) { return hd_layout(nunits ? nunits : 1).total; }

enum { HDP_SPANS = 0, HDP_WALK, HDP_EMIT, HDP_NSLOT };
static glc::KernelProf &hd_prof()
{
    static glc::KernelProf pr;
    static bool named = false;
    if (!named) { named = true; pr.name[HDP_SPANS] = "k_hd_span_functions"; pr.name[HDP_WALK] = "k_hd_walk x3"; pr.name[HDP_EMIT] = "k_hd_emit"; }
    return pr;
}

int glcHdEnableProfile(int on)
{
    glc::KernelProf &pr = hd_prof();
    (void)hipDeviceSynchronize();
    pr.collect();
    pr.reset();
    return pr.enable(on != 0) ? 1 : 0;
}

int glcHdKernelProfile(int index, char *name, size_t nameCap, double *out3)
{
    return glc::global_prof_get(hd_prof(), HDP_NSLOT, index, name, nameCap, out3);
}

// the reference's device table {num_bits, symbol}[2048] -> the u16 form (bits << 8 | symbol) the kernels read
__global__ void k_hd_table_from_device(const unsigned char *__restrict__ table2048, uint16_t *__restrict__ lut)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2048) {
        uint32_t bits = table2048[2 * i];
        const uint32_t sym = table2048[2 * i + 1];
        if (bits == 0 || bits > GLC_HD_MAX_LEN) bits = 1;     // prefixes no codeword reaches / out of range: never met in a valid stream
        lut[i] = (uint16_t)((bits << 8) | sym);
    }
}

// lut: host table (copied in), or nullptr with d_table = the reference's table already in device memory
static int hd_decode_lut(const unsigned int *d_units, size_t nunits, const uint16_t *lut, unsigned char *d_out,
                         size_t nsym, void *d_work, void *stream, const unsigned char *d_table = nullptr)
{
    const HdLayout L = hd_layout(nunits);
    if (L.nchunks > HD_CHUNK) return 0;
    uint8_t *W = (uint8_t *)d_work;
    hipStream_t st = (hipStream_t)stream;
    uint16_t *d_lut = (uint16_t *)(W + L.o_lut);
    if (lut) {
        if (hipMemcpyAsync(d_lut, lut, 2048 * sizeof(uint16_t), hipMemcpyHostToDevice, st) != hipSuccess) return 0;
        if (hipStreamSynchronize(st) != hipSuccess) return 0; // lut is a stack array of the caller
    } else hipLaunchKernelGGL(k_hd_table_from_device, dim3(8), dim3(256), 0, st, d_table, d_lut);
    uint32_t *pexcl = (uint32_t *)(W + L.o_pexcl), *fwg = (uint32_t *)(W + L.o_fwg), *G = (uint32_t *)(W + L.o_g);
    uint32_t *coff = (uint32_t *)(W + L.o_coff), *soff = (uint32_t *)(W + L.o_soff);
    unsigned long long *cbase = (unsigned long long *)(W + L.o_cbase), *sbase = (unsigned long long *)(W + L.o_sbase);
    glc::KernelProf &pr = hd_prof();
    int pi = pr.begin(HDP_SPANS, st);
    hipLaunchKernelGGL(k_hd_span_functions, dim3((unsigned)L.nwg), dim3(HD_LANES), 0, st, d_units, nunits, d_lut, pexcl, fwg);
    pr.end(pi, (double)nsym, st);
    pi = pr.begin(HDP_WALK, st);
    hipLaunchKernelGGL(k_hd_walk, dim3((unsigned)L.nchunks), dim3(64), 0, st, 0, fwg, L.nwg, G, coff, cbase, soff, sbase);
    hipLaunchKernelGGL(k_hd_walk, dim3(1), dim3(64), 0, st, 1, fwg, L.nchunks, G, coff, cbase, soff, sbase);
    hipLaunchKernelGGL(k_hd_walk, dim3((unsigned)L.nchunks), dim3(64), 0, st, 2, fwg, L.nwg, G, coff, cbase, soff, sbase);
    pr.end(pi, (double)nsym, st);
    pi = pr.begin(HDP_EMIT, st);
    hipLaunchKernelGGL(k_hd_emit, dim3((unsigned)L.nwg), dim3(HD_LANES), 0, st, d_units, nunits, d_lut, pexcl, soff, sbase,
                       d_out, nsym);
    pr.end(pi, (double)nsym, st);
    return hipGetLastError() == hipSuccess ? 1 : 0;
}

int glcHdDecodeDevice(const unsigned int *d_units, size_t nunits, const unsigned char lens[256],
                      const unsigned short codes[256], unsigned char *d_out, size_t nsym, void *d_work, void *stream)
{
    if (!d_units || !lens || !codes || !d_out || !d_work || nunits == 0 || nunits > (1ull << 31)) return 0;
    // 2048-entry table: top 11 bits -> (length << 8) | symbol ; unused prefixes decode as length 1
    uint16_t lut[2048];
    int first = -1;
    for (int s = 0; s < 256; s++) if (lens[s]) { if (lens[s] > GLC_HD_MAX_LEN) return 0; if (first < 0) first = s; }
    if (first < 0) return 0;
    for (int i = 0; i < 2048; i++) lut[i] = (uint16_t)((1 << 8) | first);
    for (int s = 0; s < 256; s++) {
        if (!lens[s]) continue;
        const int l = lens[s], span = 1 << (GLC_HD_MAX_LEN - l), lo = (int)codes[s] << (GLC_HD_MAX_LEN - l);
        if (lo + span > 2048) return 0;
        for (int i = 0; i < span; i++) lut[lo + i] = (uint16_t)((l << 8) | s);
    }
    return hd_decode_lut(d_units, nunits, lut, d_out, nsym, d_work, stream);
}

// the table exactly as the reference holds it: cuhd::CUHDCodetableItemSingle[2048] = {num_bits, symbol} byte
// pairs indexed by the next 11 bits of the stream (cuhd_codetable.h:20-23, llhuffman_encoder.cc:240-262)
int glcHdDecodeDeviceTable(const unsigned int *d_units, size_t nunits, const unsigned char *table2048,
                           unsigned char *d_out, size_t nsym, void *d_work, void *stream)
{
    if (!d_units || !table2048 || !d_out || !d_work || nunits == 0 || nunits > (1ull << 31)) return 0;
    uint16_t lut[2048];
    for (int i = 0; i < 2048; i++) {
        const unsigned bits = table2048[2 * i], sym = table2048[2 * i + 1];
        if (bits > GLC_HD_MAX_LEN) return 0;
        lut[i] = (uint16_t)(((bits ? bits : 1u) << 8) | sym);   // prefixes no codeword reaches: never met in a valid stream
    }
    return hd_decode_lut(d_units, nunits, lut, d_out, nsym, d_work, stream);
}

// As glcHdDecodeDeviceTable, with the table where cuhd::CUHDGPUCodetable keeps it: in DEVICE memory.  Nothing is copied
// and the host is not held: the whole decode is enqueued on `stream`.
int glcHdDecodeDeviceTableOnDevice(const unsigned int *d_units, size_t nunits, const unsigned char *d_table2048,
                                   unsigned char *d_out, size_t nsym, void *d_work, void *stream)
{
    if (!d_units || !d_table2048 || !d_out || !d_work || nunits == 0 || nunits > (1ull << 31)) return 0;
    return hd_decode_lut(d_units, nunits, nullptr, d_out, nsym, d_work, stream, d_table2048);
}

} // extern "C"
