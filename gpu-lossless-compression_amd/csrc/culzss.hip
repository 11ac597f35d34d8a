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
// match search: one workgroup (256 threads) per packet, 16 positions / thread
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lzss_match(const uint8_t *__restrict__ in, uint8_t *__restrict__ cand)
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
#pragma unroll 1
    for (int it = 0; it < LZ_PCKT / 256; it++) {
        const int p = it * 256 + tid;
        const int tx = p & 127;
        const bool last = (p >> 7) == (LZ_PCKT / 128 - 1);
        const int iters = last ? max(1, 127 - tx) : 127;
        const int la_wrap = last ? (128 - tx) : 1 << 20;      // look-ahead index where the stale half begins
        int length = 1, offset = 1, j = 0;
        bool matching = false;
        const uint8_t *win = s_buf + p;                       // text[p-128 + k]
        const uint8_t *la = s_buf + LZ_WIN + p;               // text[p + j]
        for (int k = 0; k < iters; k++) {
            const uint8_t lb = (j < la_wrap) ? la[j] : la[j - 256];
            if (win[k] == lb) { j++; matching = true; }
            else {
                if (matching && j > length) { length = j; offset = (p + k - j) & 255; }
                j = 0; matching = false;
            }
        }
        if (j > length && matching) { length = j; offset = (p + iters - j) & 255; }
        if (last && length > 128 - tx) length = 128 - tx;
        if (length >= LZ_MAXC) length = LZ_MAXC - 1;
        uint8_t c0, c1;
        if (length <= 2) { c0 = 1; c1 = la[0]; }
        else { c0 = (uint8_t)length; c1 = (uint8_t)offset; }
        reinterpret_cast<uint16_t *>(dst)[p] = (uint16_t)c0 | ((uint16_t)c1 << 8);
    }
}

// ---------------------------------------------------------------------------
// token selection + packing: one workgroup per packet (aftercomp's inner loop)
//   stage[pk][0..size) = flag/token bytes of the packet, meta[pk] = (size, last group bytes)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lzss_pack(const uint8_t *__restrict__ cand, uint8_t *__restrict__ stage,
                                                   uint2 *__restrict__ meta)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_c[2 * LZ_PCKT];
    __shared__ uint16_t s_tok[LZ_PCKT];
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
    if (tid == 0) {                                           // greedy walk (gpu_compress.cu:498-515)
        uint32_t t = 0, p = 0;
        while (p < LZ_PCKT) {
            s_tok[t++] = (uint16_t)p;
            const uint32_t c0 = s_c[2 * p];
            p += (c0 <= 1) ? 1u : c0;
        }
        s_ntok = t;
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
    const bool ok = (total - M[npk - 1].y) <= (uint32_t)buf_length;
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
    for (uint32_t i = tid; i < sz; i += 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------
// decode: one wave per packet; the token stream is parsed uniformly by the whole
// wave, match copies are lane-parallel (the window is a snapshot: bytes a match
// reads are always older than the bytes it writes, gpu_decompress.cu:217-236)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lzss_decode(const uint8_t *__restrict__ packed, size_t pack_stride,
                                                    const int *__restrict__ sizes, int buf_length,
                                                    uint8_t *__restrict__ out)
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
    // trailer: npk big-endian u16 sizes, u32 length, u16 pad (gpu_decompress.cu:257-294)
    const uint8_t *tr = P + clen - 6 - 2 * npk;
    uint32_t start = 0;
    for (uint32_t i = l; i < pk; i += 64) start += ((uint32_t)tr[2 * i] << 8) | tr[2 * i + 1];
    start = wave_sum(start);
    uint32_t size = ((uint32_t)tr[2 * pk] << 8) | tr[2 * pk + 1];
    if (size > LZ_STAGE) size = LZ_STAGE;
    for (uint32_t i = l; i < size; i += 64) s_in[i] = P[start + i];
    for (uint32_t i = l; i < LZ_WIN / 4; i += 64) reinterpret_cast<uint32_t *>(s_o)[i] = 0x20202020u;
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    uint32_t fp = 0, wp = 0, flags = 0, used = 8;
    for (;;) {
        if (used == 8) {
            if (fp >= size) break;
            flags = __builtin_amdgcn_readfirstlane((uint32_t)s_in[fp]); fp++; used = 0;
        }
        if (fp >= size || wp >= LZ_PCKT) break;
        if (flags & 1) {
            if (l == 0) s_o[LZ_WIN + wp] = s_in[fp];
            fp++; wp++;
        } else {
            if (fp + 1 >= size) break;
            const uint32_t len = __builtin_amdgcn_readfirstlane((uint32_t)s_in[fp]);
            const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)s_in[fp + 1]);
            fp += 2;
            uint32_t d = (wp - off) & 127u;                    // distance back to the ring slot `off`
            if (d == 0) d = 128;
            const uint32_t q = LZ_WIN + wp - d;                // s_o index of the first source byte
            uint8_t b0 = 0, b1 = 0;
            const uint32_t i0 = l, i1 = l + 64;
            if (i0 < len) b0 = s_o[(i0 < d) ? q + i0 : q + i0 - 128];
            if (i1 < len) b1 = s_o[(i1 < d) ? q + i1 : q + i1 - 128];
            __builtin_amdgcn_wave_barrier();
            if (i0 < len && wp + i0 < LZ_PCKT) s_o[LZ_WIN + wp + i0] = b0;
            if (i1 < len && wp + i1 < LZ_PCKT) s_o[LZ_WIN + wp + i1] = b1;
            wp += len;
        }
        flags >>= 1; used++;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (uint32_t i = l; i < LZ_PCKT / 16; i += 64)
        reinterpret_cast<uint4 *>(O)[i] = reinterpret_cast<const uint4 *>(s_o + LZ_WIN)[i];
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
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
    hipLaunchKernelGGL(k_lzss_match, dim3(npk * nbuf), dim3(256), 0, st, d_in, cand);
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
    hipLaunchKernelGGL(k_lzss_pack, dim3(npk * nbuf), dim3(256), 0, st, d_cand, stage, meta);
    hipLaunchKernelGGL(k_lzss_layout, dim3(nbuf), dim3(256), 0, st, meta, npk, buf_length, pk_off, d_packed, stride,
                       d_sizes);
    hipLaunchKernelGGL(k_lzss_gather, dim3(npk, nbuf), dim3(256), 0, st, stage, meta, pk_off, npk, d_packed, stride,
                       d_sizes, d_raw_in);
    return hipGetLastError();
}

hipError_t lzss_decode(hipStream_t st, const uint8_t *d_packed, const int *d_sizes, int buf_length, int nbuf,
                       uint8_t *d_out)
{
    if (buf_length <= 0 || buf_length % LZ_PCKT || nbuf <= 0) return hipErrorInvalidValue;
    const uint32_t npk = buf_length / LZ_PCKT;
    hipLaunchKernelGGL(k_lzss_decode, dim3(npk, nbuf), dim3(64), 0, st, d_packed, lzss_pack_stride(buf_length),
                       d_sizes, buf_length, d_out);
    return hipGetLastError();
}

} // namespace glc
