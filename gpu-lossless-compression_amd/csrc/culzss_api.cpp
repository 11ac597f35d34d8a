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
#include "glc_internal.h"

#include <algorithm>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

using namespace glc;

namespace {

constexpr int NSLOTS = 4;                       // ring slots of the reference queue (culzss.c:273-346)

struct Slot {
    hipStream_t stream = nullptr;
    int cap = 0;                                // buf_length the scratch below was sized for
    uint8_t *d_packed = nullptr;
    uint8_t *d_in = nullptr, *d_cand = nullptr; // the slot's own input / candidate buffers (see compression_kernel_wrapper)
    int *d_size = nullptr;
    void *d_work = nullptr;
    uint8_t *h_packed = nullptr;                // pinned
    int *h_size = nullptr;                      // pinned
    const unsigned char *key = nullptr;         // host candidate buffer of the in-flight call
    int len = 0;
    bool valid = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // One lock per ring slot: the reference's callers run a producer, a GPU thread and a CPU thread on DIFFERENT slots
    // at the same time (culzss.c:85-176); under one global lock the launch of slot k + 1 (a dozen HIP calls) waited for
    // the copy-out of slot k.  Order: g.mu (slot lookup, init) is never held while a slot lock is taken for long work.
    std::mutex mu;
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
    // the reference pins device 0 (gpu_compress.cu:395); here the streams belong to whatever device is current
    // in the calling thread, so one process per GPU (rank r on device r) works without HIP_VISIBLE_DEVICES
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
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_cand) (void)hipFree(s.d_cand);
    if (s.d_size) (void)hipFree(s.d_size);
    if (s.d_work) (void)hipFree(s.d_work);
    if (s.h_packed) (void)hipHostFree(s.h_packed);
    if (s.h_size) (void)hipHostFree(s.h_size);
    s.d_packed = nullptr; s.d_in = nullptr; s.d_cand = nullptr; s.d_size = nullptr; s.d_work = nullptr; s.h_packed = nullptr; s.h_size = nullptr;
    s.cap = 0;                                                 // (the tracking entry {valid, key, len} is read and written under g.mu only:
}                                                              //  the callers that hold it clear it; the wrapper drops it before reusing a slot)

bool ensure_slot(Slot &s, int buf_length)
{
    if (s.cap >= buf_length) return true;
    free_slot(s);
    const size_t stride = lzss_pack_stride(buf_length);
    if (!ok(hipMalloc((void **)&s.d_packed, stride), "slot packed")) return false;
    if (!ok(hipMalloc((void **)&s.d_in, (size_t)buf_length), "slot in")) return false;
    if (!ok(hipMalloc((void **)&s.d_cand, (size_t)2 * buf_length), "slot candidates")) return false;
    if (!ok(hipMalloc((void **)&s.d_size, 16), "slot size")) return false;          // (leaves as one 16-byte piece: lzss_copy_to_host)
    if (!ok(hipMalloc(&s.d_work, lzss_work_bytes(buf_length, 1)), "slot work")) return false;
    if (!ok(hipHostMalloc((void **)&s.h_packed, stride, hipHostMallocDefault), "slot pinned")) return false;
    if (!ok(hipHostMalloc((void **)&s.h_size, 16, hipHostMallocDefault), "slot pinned size")) return false;
    s.cap = buf_length;
    return true;
}

// is [p, p + bytes) pinned host memory the device can write (hipHostMalloc / initCPUmem) and 16-byte aligned?
bool host_mapped(const void *p, size_t bytes)
{
    if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    // the kernel is handed p itself: only memory whose device address IS its host address (hipHostMalloc on this platform) -- a
    // registered or otherwise mapped range with another device address takes the copy-engine path
    if (a.type != hipMemoryTypeHost || a.devicePointer != p) return false;
    hipPointerAttribute_t b;                                   // ... to its last byte
    const void *last = (const char *)p + bytes - 1;
    if (hipPointerGetAttributes(&b, last) != hipSuccess) { (void)hipGetLastError(); return false; }
    return b.type == hipMemoryTypeHost && b.devicePointer == last;
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
        s.valid = false;
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.e0) (void)hipEventDestroy(s.e0);
        if (s.e1) (void)hipEventDestroy(s.e1);
        s.stream = nullptr; s.e0 = s.e1 = nullptr; s.key = nullptr; s.len = 0;
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
        s.valid = false;
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.e0) (void)hipEventDestroy(s.e0);
        if (s.e1) (void)hipEventDestroy(s.e1);
        s.stream = nullptr; s.e0 = s.e1 = nullptr; s.key = nullptr; s.len = 0;
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
    Slot *sp;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        init_locked();
        sp = &g.slot[((index % NSLOTS) + NSLOTS) % NSLOTS];
        sp->valid = false;                         // the slot is taken again: what it tracked is gone (written under g.mu, as it is read)
    }
    Slot &s = *sp;
    std::lock_guard<std::mutex> lk(s.mu);
    (void)hipStreamSynchronize(s.stream);          // slot reuse: previous call on this slot must be done
    if (!ensure_slot(s, buf_length)) return 0;
    hipStream_t st = s.stream;
    // The reference's pipeline hands EVERY ring slot the same in_d / out_d (culzss.c:85-86,108) and queues the slots
    // on different streams without waiting (gpu_compress.cu:426-460): slot s+1's copy-in can overwrite what slot
    // s's kernel is still reading.  The caller's device buffers are therefore accepted but not used: each slot
    // stages through buffers of its own.
    (void)in_d; (void)out_d;
#ifdef GLC_LZ_HOSTTRACE
    auto tnow = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    const double t_0 = tnow();
#endif
    if (!ok(hipMemcpyAsync(s.d_in, buffer, (size_t)buf_length, hipMemcpyHostToDevice, st), "H2D")) return 0;
#ifdef GLC_LZ_HOSTTRACE
    const double t_1 = tnow();
#endif
    (void)hipEventRecord(s.e0, st);
    if (!ok(lzss_encode(st, s.d_in, buf_length, 1, s.d_cand, s.d_packed, s.d_size, s.d_work), "encode")) return 0;
    (void)hipEventRecord(s.e1, st);
#ifdef GLC_LZ_HOSTTRACE
    const double t_2 = tnow();
#endif
    // what leaves the device here: the candidate stream (the interface's compressed_buffer: 2 B per input byte) and the
    // packed size.  The packed bytes stay in the slot until aftercompression_wrapper knows how many there are and copies
    // exactly those, straight into the caller's buffer (a whole-slot copy into pinned staging + a host memcpy were
    // 1 MiB more over PCIe and ~60 us of the CPU thread per buffer).
    if (host_mapped(compressed_buffer, (size_t)2 * buf_length)) {
        // pinned (initCPUmem, as the reference's callers allocate it): written by a kernel -- see k_lzss_to_host
        if (!ok(lzss_copy_to_host(st, s.d_cand, compressed_buffer, (size_t)2 * buf_length), "candidates to host")) return 0;
    } else if (!ok(hipMemcpyAsync(compressed_buffer, s.d_cand, (size_t)2 * buf_length, hipMemcpyDeviceToHost, st), "D2H cand")) return 0;
#ifdef GLC_LZ_HOSTTRACE
    const double t_3 = tnow();
#endif
    if (!ok(lzss_copy_to_host(st, s.d_size, s.h_size, 16), "size to host")) return 0;
#ifdef GLC_LZ_HOSTTRACE
    fprintf(stderr, "wrapper slot %d: H2D %.0f us, kernels %.0f us, D2H cand %.0f us, D2H size %.0f us\n", index, t_1 - t_0, t_2 - t_1, t_3 - t_2, tnow() - t_3);
#endif
    {
        std::lock_guard<std::mutex> lg(g.mu);
        s.key = compressed_buffer; s.len = buf_length; s.valid = true;
    }
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
    Slot *hit = nullptr;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        init_locked();
        for (int i = 0; i < NSLOTS; i++)
            if (g.slot[i].valid && g.slot[i].key == bufferout && g.slot[i].len == buf_length) { hit = &g.slot[i]; hit->valid = false; }
    }
    if (!hit) {
        // candidates that did not come from a tracked call: pack them on the GPU now
        Slot &s = g.slot[NSLOTS];
        std::lock_guard<std::mutex> lk(s.mu);
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
        const int size = *s.h_size;
        if (size <= 0) return 0;                    // "compression took more": caller stores the buffer raw
        memcpy(buffer, s.h_packed, (size_t)size);
        *comp_length = size;
        return 1;
    }
    std::lock_guard<std::mutex> lk(hit->mu);
    if (!ok(hipStreamSynchronize(hit->stream), "sync")) return 0;
    const int size = *hit->h_size;
    if (size <= 0) return 0;                        // "compression took more": caller stores the buffer raw (untouched)
    if (!ok(hipMemcpyAsync(buffer, hit->d_packed, (size_t)size, hipMemcpyDeviceToHost, hit->stream), "D2H packed") ||
        !ok(hipStreamSynchronize(hit->stream), "sync")) return 0;
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
    std::lock_guard<std::mutex> ls(g.slot[NSLOTS].mu);        // (the scratch slot's stream: shared with the untracked packing path)
    if (g.dd_cap < orig) {
        if (g.dd_in) (void)hipFree(g.dd_in);
        if (g.dd_out) (void)hipFree(g.dd_out);
        if (g.dd_size) (void)hipFree(g.dd_size);
        g.dd_in = g.dd_out = nullptr; g.dd_size = nullptr; g.dd_cap = 0;
        if (!ok(hipMalloc((void **)&g.dd_in, lzss_pack_stride(orig)), "decode in")) return 0;
        if (!ok(hipMalloc((void **)&g.dd_out, (size_t)orig), "decode out")) return 0;
        if (!ok(hipMalloc((void **)&g.dd_size, 2 * sizeof(int)), "decode size")) return 0;   // {size, error word}
        g.dd_cap = orig;
    }
    if ((size_t)buf_length > lzss_pack_stride(orig)) return 0;
    hipStream_t st = g.slot[NSLOTS].stream;
    int hdr[2] = {buf_length, 0}, err = 0;
    bool good = ok(hipMemcpyAsync(g.dd_in, buffer, (size_t)buf_length, hipMemcpyHostToDevice, st), "H2D")
             && ok(hipMemcpyAsync(g.dd_size, hdr, sizeof hdr, hipMemcpyHostToDevice, st), "H2D size")
             && ok(hipStreamSynchronize(st), "sync")       // hdr is a stack variable
             && ok(lzss_decode(st, g.dd_in, g.dd_size, orig, 1, g.dd_out, g.dd_size + 1), "decode")
             && ok(hipMemcpyAsync(&err, g.dd_size + 1, sizeof(int), hipMemcpyDeviceToHost, st), "D2H err")
             && ok(hipStreamSynchronize(st), "sync");
    if (!good || err) return 0;                             // malformed stream: nothing is written back
    good = ok(hipMemcpyAsync(buffer, g.dd_out, (size_t)(orig - pad), hipMemcpyDeviceToHost, st), "D2H")
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
    std::lock_guard<std::mutex> ls(s.mu);
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

int glcLzssEnableProfile(int on)
{
    KernelProf &pr = lzss_prof();
    (void)hipDeviceSynchronize();
    pr.collect();
    pr.reset();
    return pr.enable(on != 0) ? 1 : 0;
}

int glcLzssKernelProfile(int index, char *name, size_t nameCap, double *out3)
{
    return global_prof_get(lzss_prof(), LZP_NSLOT, index, name, nameCap, out3);
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

// ==========================================================================
// Container + pipeline: the file format written by the
// reference's pthread pipeline (cuda-lzss-cluster/main.c:236-245, culzss.c:204-269,
// decompression.c:66-173, deculzss.c:125-180):
//     u32 nbufs | u32 padding | u32 cumulative_size[nbufs] | payload_0 | payload_1 ...
// (host-endian u32, as fwrite'd by culzss.c:220,263-264).  payload_i is the packed
// form of 1 MiB buffer i, or the raw buffer when packing "took more"
// (size == BUFSIZE, culzss.c:241-242; recognised by deculzss.c:94-95).  `padding`
// = bytes missing from the last buffer.  One deliberate fix: the reference
// compresses the last partial buffer together with stale bytes of the ring slot
// (main.c:122-130), so its bytes are run-dependent; here the tail is zero-filled.
//
// The reference overlaps file I/O, PCIe and the GPU with four pthreads and a
// 4-slot ledger.  Here PCIe and the GPU overlap through two HIP streams working on
// alternating groups of buffers (copy-in / kernels / copy-out of group g+1 run
// while the host assembles group g) -- no thread ring, no busy-wait on
// cudaStreamQuery (gpu_compress.cu:415-424).  File I/O is NOT overlapped: the
// *_file functions read the whole input and hold the whole output in memory
// (inputs are bounded by the 4 GiB - 1 payload limit of the u32 offsets anyway).
// ==========================================================================
namespace {

constexpr int CBUF = 1 << 20;                   // BUFSIZE (main.c:62)
constexpr int GROUP = 16;                       // buffers per in-flight group

struct Group {
    hipStream_t st = nullptr;
    uint8_t *d_in = nullptr, *d_packed = nullptr, *h_in = nullptr, *h_packed = nullptr;
    int *d_sizes = nullptr, *h_sizes = nullptr; // GROUP sizes + one error word
    void *d_work = nullptr;
    uint8_t *d_out = nullptr, *h_out = nullptr;  // decode side
    int nbuf = 0;
    size_t first = 0;
};

bool group_alloc(Group &g, bool decode)
{
    const size_t stride = lzss_pack_stride(CBUF);
    if (!ok(hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking), "group stream")) return false;
    if (!ok(hipMalloc((void **)&g.d_packed, stride * GROUP), "group packed")) return false;
    if (!ok(hipMalloc((void **)&g.d_sizes, sizeof(int) * (GROUP + 1)), "group sizes")) return false;
    if (!ok(hipHostMalloc((void **)&g.h_packed, stride * GROUP, hipHostMallocDefault), "group h_packed")) return false;
    if (!ok(hipHostMalloc((void **)&g.h_sizes, sizeof(int) * (GROUP + 1), hipHostMallocDefault), "group h_sizes")) return false;
    if (!decode) {
        if (!ok(hipMalloc((void **)&g.d_in, (size_t)CBUF * GROUP), "group in")) return false;
        if (!ok(hipHostMalloc((void **)&g.h_in, (size_t)CBUF * GROUP, hipHostMallocDefault), "group h_in")) return false;
        if (!ok(hipMalloc(&g.d_work, lzss_work_bytes(CBUF, GROUP)), "group work")) return false;
    } else {
        if (!ok(hipMalloc((void **)&g.d_out, (size_t)CBUF * GROUP), "group out")) return false;
        if (!ok(hipHostMalloc((void **)&g.h_out, (size_t)CBUF * GROUP, hipHostMallocDefault), "group h_out")) return false;
    }
    return true;
}

void group_free(Group &g)
{
    if (g.st) { (void)hipStreamSynchronize(g.st); (void)hipStreamDestroy(g.st); }
    void *dp[] = {g.d_in, g.d_packed, g.d_sizes, g.d_work, g.d_out};
    for (void *p : dp) if (p) (void)hipFree(p);
    void *hp[] = {g.h_in, g.h_packed, g.h_sizes, g.h_out};
    for (void *p : hp) if (p) (void)hipHostFree(p);
    g = Group();
}

} // namespace

extern "C" {

unsigned long long culzss_container_bound(unsigned long long len)
{
    const unsigned long long nb = (len + CBUF - 1) / CBUF;
    return 8 + 4 * nb + nb * (unsigned long long)CBUF;       // every buffer stored raw is the worst case
}

int culzss_container_compress(const unsigned char *in, unsigned long long len, unsigned char *out,
                              unsigned long long out_cap, unsigned long long *out_len)
{
    if (!in || !out || !out_len) return 0;
    if (len < (unsigned long long)CBUF) return 0;            // "too small to benefit from GPU" (main.c:228-232)
    const size_t nb = (size_t)((len + CBUF - 1) / CBUF);
    if (nb > 0x3FFFFFFFu || out_cap < culzss_container_bound(len)) return 0;
    const uint32_t padding = (uint32_t)(nb * (size_t)CBUF - len);
    uint32_t *hdr = reinterpret_cast<uint32_t *>(out);       // out comes from malloc/numpy: aligned
    uint32_t w0 = (uint32_t)nb; memcpy(out, &w0, 4); memcpy(out + 4, &padding, 4);
    size_t wpos = 8 + 4 * nb, cum = 0;
    (void)hdr;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    Group grp[2];
    if (!group_alloc(grp[0], false) || !group_alloc(grp[1], false)) { group_free(grp[0]); group_free(grp[1]); return 0; }
    const size_t stride = lzss_pack_stride(CBUF);
    bool good = true;
    auto submit = [&](Group &G, size_t first) {
        G.first = first; G.nbuf = (int)std::min((size_t)GROUP, nb - first);
        for (int i = 0; i < G.nbuf; i++) {
            const size_t off = (first + i) * (size_t)CBUF;
            const size_t take = std::min((size_t)CBUF, (size_t)len - off);
            memcpy(G.h_in + (size_t)i * CBUF, in + off, take);
            if (take < (size_t)CBUF) memset(G.h_in + (size_t)i * CBUF + take, 0, CBUF - take);   // zero-filled tail
        }
        good = good && ok(hipMemcpyAsync(G.d_in, G.h_in, (size_t)G.nbuf * CBUF, hipMemcpyHostToDevice, G.st), "H2D")
            && ok(lzss_encode(G.st, G.d_in, CBUF, G.nbuf, nullptr, G.d_packed, G.d_sizes, G.d_work), "encode")
            && ok(hipMemcpyAsync(G.h_sizes, G.d_sizes, sizeof(int) * G.nbuf, hipMemcpyDeviceToHost, G.st), "D2H sizes")
            && ok(hipMemcpyAsync(G.h_packed, G.d_packed, stride * G.nbuf, hipMemcpyDeviceToHost, G.st), "D2H packed");
    };
    auto collect = [&](Group &G) {
        good = good && ok(hipStreamSynchronize(G.st), "sync");
        for (int i = 0; good && i < G.nbuf; i++) {
            int sz = G.h_sizes[i];
            if (sz >= CBUF) sz = 0;                                           // never packed to >= BUFSIZE: a payload of exactly
                                                                              // BUFSIZE bytes MEANS raw (deculzss.c:94-95)
            const size_t bytes = sz > 0 ? (size_t)sz : (size_t)CBUF;          // 0 => stored raw (culzss.c:241-242)
            if (wpos + bytes > out_cap || cum + bytes > 0xFFFFFFFFull) { good = false; break; }   // u32 offsets: format limit
            memcpy(out + wpos, sz > 0 ? G.h_packed + (size_t)i * stride : G.h_in + (size_t)i * CBUF, bytes);
            wpos += bytes; cum += bytes;
            const uint32_t c32 = (uint32_t)cum;
            memcpy(out + 8 + 4 * (G.first + i), &c32, 4);
        }
    };
    size_t next = 0;
    int cur = 0;
    submit(grp[cur], next); next += GROUP;
    while (good) {
        const bool more = next < nb;
        if (more) { submit(grp[cur ^ 1], next); next += GROUP; }
        collect(grp[cur]);
        if (!more) break;
        cur ^= 1;
    }
    group_free(grp[0]); group_free(grp[1]);
    if (!good) return 0;
    *out_len = wpos;
    return 1;
}

int culzss_container_decompress(const unsigned char *in, unsigned long long len, unsigned char *out,
                                unsigned long long out_cap, unsigned long long *out_len)
{
    if (!in || !out || !out_len || len < 8) return 0;
    uint32_t nb32, padding;
    memcpy(&nb32, in, 4); memcpy(&padding, in + 4, 4);
    const size_t nb = nb32;
    if (nb == 0 || len < 8 + 4 * nb || padding >= (uint32_t)CBUF) return 0;
    const unsigned long long total = (unsigned long long)nb * CBUF - padding;
    if (out_cap < total) return 0;
    std::lock_guard<std::mutex> lk(g.mu);
    init_locked();
    Group grp[2];
    if (!group_alloc(grp[0], true) || !group_alloc(grp[1], true)) { group_free(grp[0]); group_free(grp[1]); return 0; }
    const size_t stride = lzss_pack_stride(CBUF);
    const size_t payload = 8 + 4 * nb;
    bool good = true;
    auto cumat = [&](size_t i) -> size_t { if (i == 0) return 0; uint32_t c; memcpy(&c, in + 8 + 4 * (i - 1), 4); return c; };
    auto submit = [&](Group &G, size_t first) {
        G.first = first; G.nbuf = (int)std::min((size_t)GROUP, nb - first);
        for (int i = 0; good && i < G.nbuf; i++) {
            const size_t a = cumat(first + i), b = cumat(first + i + 1);
            const size_t sz = b - a;
            // (a packed chunk may be LONGER than the buffer: the reference's packer only gives up when the bytes flushed
            //  before the last group outgrow it, so up to BUFSIZE + 535 bytes reach the file -- include/culzss.h -- and
            //  its decoder takes everything that is not exactly BUFSIZE as packed, deculzss.c:92-98)
            if (b < a || sz > stride || payload + b > len) { good = false; break; }
            const uint8_t *src = in + payload + a;
            if (sz != (size_t)CBUF) {                                          // packed: must hold its trailer, and the trailer
                constexpr size_t TR = 2 * (CBUF / GLC_LZSS_PACKET) + 6;        // must describe a 1 MiB buffer without padding
                if (sz < TR) { good = false; break; }
                const uint32_t orig = ((uint32_t)src[sz - 6] << 24) | ((uint32_t)src[sz - 5] << 16) |
                                      ((uint32_t)src[sz - 4] << 8) | (uint32_t)src[sz - 3];
                if (orig != (uint32_t)CBUF || src[sz - 2] || src[sz - 1]) { good = false; break; }
            }
            memcpy(G.h_packed + (size_t)i * stride, src, sz);
            G.h_sizes[i] = (sz == (size_t)CBUF) ? 0 : (int)sz;             // raw buffers: deculzss.c:94-95
        }
        G.h_sizes[GROUP] = 0;                                              // error word
        good = good && ok(hipMemcpyAsync(G.d_packed, G.h_packed, stride * G.nbuf, hipMemcpyHostToDevice, G.st), "H2D")
            && ok(hipMemcpyAsync(G.d_sizes, G.h_sizes, sizeof(int) * (GROUP + 1), hipMemcpyHostToDevice, G.st), "H2D sizes")
            && ok(lzss_decode(G.st, G.d_packed, G.d_sizes, CBUF, G.nbuf, G.d_out, G.d_sizes + GROUP), "decode")
            && ok(hipMemcpyAsync(G.h_sizes + GROUP, G.d_sizes + GROUP, sizeof(int), hipMemcpyDeviceToHost, G.st), "D2H err")
            && ok(hipMemcpyAsync(G.h_out, G.d_out, (size_t)G.nbuf * CBUF, hipMemcpyDeviceToHost, G.st), "D2H");
    };
    auto collect = [&](Group &G) {
        good = good && ok(hipStreamSynchronize(G.st), "sync");
        if (good && G.h_sizes[GROUP]) good = false;                        // a packet table that does not add up
        for (int i = 0; good && i < G.nbuf; i++) {
            const size_t off = (G.first + i) * (size_t)CBUF;
            const size_t take = std::min((size_t)CBUF, (size_t)total - off);  // last buffer loses the padding (deculzss.c:156-159)
            memcpy(out + off, G.h_out + (size_t)i * CBUF, take);
        }
    };
    size_t next = 0;
    int cur = 0;
    submit(grp[cur], next); next += GROUP;
    while (good) {
        const bool more = next < nb;
        if (more) { submit(grp[cur ^ 1], next); next += GROUP; }
        collect(grp[cur]);
        if (!more) break;
        cur ^= 1;
    }
    group_free(grp[0]); group_free(grp[1]);
    if (!good) return 0;
    *out_len = total;
    return 1;
}

static int slurp(const char *path, unsigned char **data, unsigned long long *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return 0; }          // pipes, FIFOs, directories: not seekable
    const long sz = ftell(f);
    if (sz < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return 0; }
    unsigned char *p = (unsigned char *)malloc(sz > 0 ? (size_t)sz : 1);
    if (!p) { fclose(f); return 0; }
    const size_t got = fread(p, 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) { free(p); return 0; }
    *data = p; *len = (unsigned long long)sz;
    return 1;
}

/* ./main -i in -o out   (main.c:160-186, compress branch) */
int culzss_compress_file(const char *in_path, const char *out_path)
{
    unsigned char *in = nullptr; unsigned long long len = 0, olen = 0;
    if (!in_path || !out_path || !slurp(in_path, &in, &len)) return 0;
    unsigned char *out = (unsigned char *)malloc((size_t)culzss_container_bound(len) + 16);
    int rc = out ? culzss_container_compress(in, len, out, culzss_container_bound(len), &olen) : 0;
    if (rc) { FILE *f = fopen(out_path, "wb"); rc = f && fwrite(out, 1, (size_t)olen, f) == (size_t)olen; if (f) fclose(f); }
    free(in); free(out);
    return rc;
}

/* ./main -d 1 -i in -o out   (main.c:207-222) */
int culzss_decompress_file(const char *in_path, const char *out_path)
{
    unsigned char *in = nullptr; unsigned long long len = 0, olen = 0;
    if (!in_path || !out_path || !slurp(in_path, &in, &len) || len < 8) { free(in); return 0; }
    uint32_t nb; memcpy(&nb, in, 4);
    const unsigned long long cap = (unsigned long long)nb * CBUF;
    unsigned char *out = (unsigned char *)malloc(cap ? (size_t)cap : 1);
    int rc = out ? culzss_container_decompress(in, len, out, cap, &olen) : 0;
    if (rc) { FILE *f = fopen(out_path, "wb"); rc = f && fwrite(out, 1, (size_t)olen, f) == (size_t)olen; if (f) fclose(f); }
    free(in); free(out);
    return rc;
}

} // extern "C"
