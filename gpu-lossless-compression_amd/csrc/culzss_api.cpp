// culzss_api.cpp -- the C ABI of include/culzss.h.
//
// Mirrors the wrapper layer of the reference (cuda-lzss-cluster/gpu_compress.cu:352-460,
// 569-670; gpu_decompress.cu:98-118,247-358): same symbols, argument meaning and
// return values, called from the reference's pthread pipeline (culzss.c:85-86,108,
// 133-134,170,176; deculzss.c:78-79,98,119-120) on different threads ("launch on
// thread A, wait on thread B"), so all shared state sits behind one mutex.
//
// Differences, all deliberate: token selection/packing runs on the GPU inside the
// compression stream instead of on a CPU thread (aftercomp, gpu_compress.cu:462-566);
// device scratch is cached per ring slot instead of cudaMalloc/cudaFree per call
// (gpu_decompress.cu:306-349); HIP errors are reported by return value 0 rather
// than exit() (gpu_compress.cu:170-179).
#include "../../include/culzss.h"
#include "culzss_internal.h"

#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

using namespace glc;

namespace {

constexpr int NSLOTS = 4;                       // ring slots of the reference queue (culzss.c:273-346)

struct Slot {
    hipStream_t stream = nullptr;
    int cap = 0;                                // buf_length the scratch below was sized for
    uint8_t *d_packed = nullptr;
    int *d_size = nullptr;
    void *d_work = nullptr;
    uint8_t *h_packed = nullptr;                // pinned
    int *h_size = nullptr;                      // pinned
    const unsigned char *key = nullptr;         // host candidate buffer of the in-flight call
    int len = 0;
    bool valid = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

struct State {
    std::mutex mu;
    bool inited = false;
    Slot slot[NSLOTS + 1];                      // +1: scratch slot for stand-alone packing / conveniences
    // decode scratch
    uint8_t *dd_in = nullptr, *dd_out = nullptr;
    int *dd_size = nullptr;
    int dd_cap = 0;
    float last_ms = 0.f;
} g;

bool ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    fprintf(stderr, "culzss (hip): %s: %s\n", what, hipGetErrorString(e));
    return false;
}

void init_locked()
{
    if (g.inited) return;
    (void)hipSetDevice(0);                       // gpu_compress.cu:395
    for (auto &s : g.slot) {
        (void)hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        (void)hipEventCreate(&s.e0);
        (void)hipEventCreate(&s.e1);
    }
    g.inited = true;
}

void free_slot(Slot &s)
{
    if (s.d_packed) (void)hipFree(s.d_packed);
    if (s.d_size) (void)hipFree(s.d_size);
    if (s.d_work) (void)hipFree(s.d_work);
    if (s.h_packed) (void)hipHostFree(s.h_packed);
    if (s.h_size) (void)hipHostFree(s.h_size);
    s.d_packed = nullptr; s.d_size = nullptr; s.d_work = nullptr; s.h_packed = nullptr; s.h_size = nullptr;
    s.cap = 0; s.valid = false;
}

bool ensure_slot(Slot &s, int buf_length)
{
    if (s.cap >= buf_length) return true;
    free_slot(s);
    const size_t stride = lzss_pack_stride(buf_length);
    if (!ok(hipMalloc((void **)&s.d_packed, stride), "slot packed")) return false;
    if (!ok(hipMalloc((void **)&s.d_size, sizeof(int)), "slot size")) return false;
    if (!ok(hipMalloc(&s.d_work, lzss_work_bytes(buf_length, 1)), "slot work")) return false;
    if (!ok(hipHostMalloc((void **)&s.h_packed, stride, hipHostMallocDefault), "slot pinned")) return false;
    if (!ok(hipHostMalloc((void **)&s.h_size, sizeof(int), hipHostMallocDefault), "slot pinned size")) return false;
    s.cap = buf_length;
    return true;
}

bool valid_len(int n) { return n > 0 && n % GLC_LZSS_PACKET == 0 && n <= GLC_LZSS_MAX_BUF; }

} // namespace

extern "C" {

void initGPU(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
}

void resetGPU(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto &s : g.slot) {
        free_slot(s);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.e0) (void)hipEventDestroy(s.e0);
        if (s.e1) (void)hipEventDestroy(s.e1);
        s = Slot();
    }
    if (g.dd_in) (void)hipFree(g.dd_in);
    if (g.dd_out) (void)hipFree(g.dd_out);
    if (g.dd_size) (void)hipFree(g.dd_size);
    g.dd_in = g.dd_out = nullptr; g.dd_size = nullptr; g.dd_cap = 0;
    g.inited = false;
    (void)hipDeviceReset();
}

int streams_in_GPU(void) { return 1; }

void deleteGPUStreams(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto &s : g.slot) {
        if (s.stream) { (void)hipStreamSynchronize(s.stream); }
        free_slot(s);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.e0) (void)hipEventDestroy(s.e0);
        if (s.e1) (void)hipEventDestroy(s.e1);
        s = Slot();
    }
    g.inited = false;
}

void signalExitThreads(void) {}

unsigned char *initGPUmem(int buf_length)
{
    void *p = nullptr;
    if (buf_length <= 0 || !ok(hipMalloc(&p, (size_t)buf_length), "initGPUmem")) return nullptr;
    return (unsigned char *)p;
}

unsigned char *initCPUmem(int buf_length)
{
    void *p = nullptr;
    if (buf_length <= 0 || !ok(hipHostMalloc(&p, (size_t)buf_length, hipHostMallocDefault), "initCPUmem")) return nullptr;
    return (unsigned char *)p;
}

void deleteGPUmem(unsigned char *mem_d) { if (mem_d) (void)hipFree(mem_d); }
void deleteCPUmem(unsigned char *mem_d) { if (mem_d) (void)hipHostFree(mem_d); }
unsigned char *deinitGPUmem(int buf_length) { return initGPUmem(buf_length); }
void dedeleteGPUmem(unsigned char *mem_d) { deleteGPUmem(mem_d); }
void deinitGPU(void) { (void)hipSetDevice(0); }

int compression_kernel_wrapper(unsigned char *buffer, int buf_length, unsigned char *compressed_buffer,
                               int /*compression_type*/, int /*wsize*/, int /*numthre*/, int /*nstreams*/,
                               int index, unsigned char *in_d, unsigned char *out_d)
{
    if (!buffer || !compressed_buffer || !in_d || !out_d || !valid_len(buf_length)) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    Slot &s = g.slot[((index % NSLOTS) + NSLOTS) % NSLOTS];
    (void)hipStreamSynchronize(s.stream);          // slot reuse: previous call on this slot must be done
    if (!ensure_slot(s, buf_length)) return 0;
    hipStream_t st = s.stream;
    const size_t stride = lzss_pack_stride(buf_length);
    if (!ok(hipMemcpyAsync(in_d, buffer, (size_t)buf_length, hipMemcpyHostToDevice, st), "H2D")) return 0;
    (void)hipEventRecord(s.e0, st);
    if (!ok(lzss_encode(st, in_d, buf_length, 1, out_d, s.d_packed, s.d_size, s.d_work), "encode")) return 0;
    (void)hipEventRecord(s.e1, st);
    if (!ok(hipMemcpyAsync(compressed_buffer, out_d, (size_t)2 * buf_length, hipMemcpyDeviceToHost, st), "D2H cand")) return 0;
    if (!ok(hipMemcpyAsync(s.h_packed, s.d_packed, stride, hipMemcpyDeviceToHost, st), "D2H packed")) return 0;
    if (!ok(hipMemcpyAsync(s.h_size, s.d_size, sizeof(int), hipMemcpyDeviceToHost, st), "D2H size")) return 0;
    s.key = compressed_buffer; s.len = buf_length; s.valid = true;
    return 1;
}

int onestream_finish_GPU(int index)
{
    hipStream_t st;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        init_locked();
        st = g.slot[((index % NSLOTS) + NSLOTS) % NSLOTS].stream;
    }
    return ok(hipStreamSynchronize(st), "stream sync") ? 1 : 0;
}

int aftercompression_wrapper(unsigned char *buffer, int buf_length, unsigned char *bufferout, int *comp_length)
{
    if (!buffer || !bufferout || !comp_length || !valid_len(buf_length)) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    Slot *hit = nullptr;
    for (int i = 0; i < NSLOTS; i++)
        if (g.slot[i].valid && g.slot[i].key == bufferout && g.slot[i].len == buf_length) hit = &g.slot[i];
    if (!hit) {
        // candidates that did not come from a tracked call: pack them on the GPU now
        Slot &s = g.slot[NSLOTS];
        if (!ensure_slot(s, buf_length)) return 0;
        uint8_t *d_cand = nullptr;
        if (!ok(hipMalloc((void **)&d_cand, (size_t)2 * buf_length), "cand upload")) return 0;
        bool good = ok(hipMemcpyAsync(d_cand, bufferout, (size_t)2 * buf_length, hipMemcpyHostToDevice, s.stream), "H2D cand")
                 && ok(lzss_pack(s.stream, d_cand, buf_length, 1, s.d_packed, s.d_size, s.d_work), "pack")
                 && ok(hipMemcpyAsync(s.h_packed, s.d_packed, lzss_pack_stride(buf_length), hipMemcpyDeviceToHost, s.stream), "D2H")
                 && ok(hipMemcpyAsync(s.h_size, s.d_size, sizeof(int), hipMemcpyDeviceToHost, s.stream), "D2H size")
                 && ok(hipStreamSynchronize(s.stream), "sync");
        (void)hipFree(d_cand);
        if (!good) return 0;
        hit = &s;
    } else if (!ok(hipStreamSynchronize(hit->stream), "sync")) return 0;
    hit->valid = false;
    const int size = *hit->h_size;
    if (size <= 0) return 0;                        // "compression took more": caller stores the buffer raw
    memcpy(buffer, hit->h_packed, (size_t)size);
    *comp_length = size;
    return 1;
}

int decompression_kernel_wrapper(unsigned char *buffer, int buf_length, int *decomp_length,
                                 int /*compression_type*/, int /*wsize*/, int /*numthre*/)
{
    if (!buffer || !decomp_length || buf_length < 8) return 0;
    // trailer (gpu_decompress.cu:257-270)
    const int orig = (int)(((unsigned)buffer[buf_length - 6] << 24) ^ ((unsigned)buffer[buf_length - 5] << 16) ^
                           ((unsigned)buffer[buf_length - 4] << 8) ^ (unsigned)buffer[buf_length - 3]);
    const int pad = (int)(((unsigned)buffer[buf_length - 2] << 8) ^ (unsigned)buffer[buf_length - 1]);
    if (!valid_len(orig) || pad < 0 || pad > orig || buf_length < 2 * (orig / GLC_LZSS_PACKET) + 6) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    if (g.dd_cap < orig) {
        if (g.dd_in) (void)hipFree(g.dd_in);
        if (g.dd_out) (void)hipFree(g.dd_out);
        if (g.dd_size) (void)hipFree(g.dd_size);
        g.dd_in = g.dd_out = nullptr; g.dd_size = nullptr; g.dd_cap = 0;
        if (!ok(hipMalloc((void **)&g.dd_in, lzss_pack_stride(orig)), "decode in")) return 0;
        if (!ok(hipMalloc((void **)&g.dd_out, (size_t)orig), "decode out")) return 0;
        if (!ok(hipMalloc((void **)&g.dd_size, sizeof(int)), "decode size")) return 0;
        g.dd_cap = orig;
    }
    if ((size_t)buf_length > lzss_pack_stride(orig)) return 0;
    hipStream_t st = g.slot[NSLOTS].stream;
    bool good = ok(hipMemcpyAsync(g.dd_in, buffer, (size_t)buf_length, hipMemcpyHostToDevice, st), "H2D")
             && ok(hipMemcpyAsync(g.dd_size, &buf_length, sizeof(int), hipMemcpyHostToDevice, st), "H2D size")
             && ok(hipStreamSynchronize(st), "sync")       // &buf_length is a stack variable
             && ok(lzss_decode(st, g.dd_in, g.dd_size, orig, 1, g.dd_out), "decode")
             && ok(hipMemcpyAsync(buffer, g.dd_out, (size_t)(orig - pad), hipMemcpyDeviceToHost, st), "D2H")
             && ok(hipStreamSynchronize(st), "sync");
    if (!good) return 0;
    *decomp_length = orig - pad;
    return 1;
}

// --------------------------------------------------------------------------
// conveniences
// --------------------------------------------------------------------------
int culzss_compress(const unsigned char *in, int len, unsigned char *out, int *out_len)
{
    if (!in || !out || !out_len || !valid_len(len)) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    Slot &s = g.slot[NSLOTS];
    if (!ensure_slot(s, len)) return 0;
    uint8_t *d_in = nullptr;
    if (!ok(hipMalloc((void **)&d_in, (size_t)len), "culzss_compress in")) return 0;
    bool good = ok(hipMemcpyAsync(d_in, in, (size_t)len, hipMemcpyHostToDevice, s.stream), "H2D")
             && ok(lzss_encode(s.stream, d_in, len, 1, nullptr, s.d_packed, s.d_size, s.d_work), "encode")
             && ok(hipMemcpyAsync(s.h_packed, s.d_packed, lzss_pack_stride(len), hipMemcpyDeviceToHost, s.stream), "D2H")
             && ok(hipMemcpyAsync(s.h_size, s.d_size, sizeof(int), hipMemcpyDeviceToHost, s.stream), "D2H size")
             && ok(hipStreamSynchronize(s.stream), "sync");
    (void)hipFree(d_in);
    if (!good) return 0;
    const int size = *s.h_size;
    if (size <= 0) { memcpy(out, in, (size_t)len); *out_len = len; return 2; }
    memcpy(out, s.h_packed, (size_t)size);
    *out_len = size;
    return 1;
}

int culzss_decompress(const unsigned char *in, int len, unsigned char *out, int *out_len)
{
    if (!in || !out || !out_len || len < 8) return 0;
    const int orig = (int)(((unsigned)in[len - 6] << 24) ^ ((unsigned)in[len - 5] << 16) ^
                           ((unsigned)in[len - 4] << 8) ^ (unsigned)in[len - 3]);
    if (!valid_len(orig) || (size_t)len > lzss_pack_stride(orig)) return 0;
    unsigned char *tmp = (unsigned char *)malloc(lzss_pack_stride(orig) > (size_t)orig ? lzss_pack_stride(orig) : (size_t)orig);
    if (!tmp) return 0;
    memcpy(tmp, in, (size_t)len);
    int n = 0;
    const int rc = decompression_kernel_wrapper(tmp, len, &n, 0, 1, 1);
    if (rc == 1) { memcpy(out, tmp, (size_t)n); *out_len = n; }
    free(tmp);
    return rc;
}

// --------------------------------------------------------------------------
// device-resident batch API
// --------------------------------------------------------------------------
unsigned long long glcLzssPackStride(int buf_length)
{
    return valid_len(buf_length) ? (unsigned long long)lzss_pack_stride(buf_length) : 0ull;
}

unsigned long long glcLzssWorkBytes(int buf_length, int nbuf)
{
    if (!valid_len(buf_length) || nbuf <= 0) return 0;
    return (unsigned long long)lzss_work_bytes(buf_length, nbuf);
}

int glcLzssEncodeDevice(const unsigned char *d_in, int buf_length, int nbuf, unsigned char *d_cand,
                        unsigned char *d_packed, int *d_sizes, void *d_work, void *stream)
{
    if (!d_in || !d_packed || !d_sizes || !d_work || !valid_len(buf_length) || nbuf <= 0) return 0;
    return ok(lzss_encode((hipStream_t)stream, d_in, buf_length, nbuf, d_cand, d_packed, d_sizes, d_work), "encode") ? 1 : 0;
}

int glcLzssDecodeDevice(const unsigned char *d_packed, const int *d_sizes, int buf_length, int nbuf,
                        unsigned char *d_out, void *stream)
{
    if (!d_packed || !d_sizes || !d_out || !valid_len(buf_length) || nbuf <= 0) return 0;
    return ok(lzss_decode((hipStream_t)stream, d_packed, d_sizes, buf_length, nbuf, d_out), "decode") ? 1 : 0;
}

float glcLzssLastKernelMs(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    float best = 0.f;
    for (int i = 0; i < NSLOTS; i++) {
        float ms = 0.f;
        if (g.slot[i].e0 && hipEventElapsedTime(&ms, g.slot[i].e0, g.slot[i].e1) == hipSuccess && ms > best) best = ms;
    }
    return best;
}

} // extern "C"
