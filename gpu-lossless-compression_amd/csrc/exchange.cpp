// exchange.cpp -- the C ABI of include/glc_exchange.h: result collection of the multi-GPU path over RCCL.
//
// The reference is single-GPU (gpu_compress.cu:395, sa_app.cu:370); this is the one exchange step SURVEY.md 8(e)
// defines for the sharded path: counts by ncclAllGather, then exact-length gather-v / scatter-v of records and
// compacted streams as ONE group of point-to-point operations per rank (xGMI is point-to-point: the root
// receives over its 7 links at once, no ring).
#include "../../include/glc_exchange.h"
#include "glc_device.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <new>
#include <string.h>

struct glcComm_st {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int nranks = 0, rank = 0;
    // GLC_COUNT_SLOTS count exchanges may be in flight (one per batch of a pipelined encode): a slot = device
    // [2 + 2 * nranks] (mine, then everybody's), pinned host [2 * nranks], an event behind the copy to the host
    unsigned long long *d_counts = nullptr;
    unsigned long long *h_counts = nullptr;
    hipEvent_t done[GLC_COUNT_SLOTS] = {};
    bool busy[GLC_COUNT_SLOTS] = {};
    unsigned next = 0;
    size_t slot_words() const { return 2 + 2 * (size_t)nranks; }
};

namespace {

#define GLC_NCCL(x) do { if ((x) != ncclSuccess) return CUDPP_ERROR_UNKNOWN; } while (0)
#define GLC_HIP(x) do { if ((x) != hipSuccess) return CUDPP_ERROR_UNKNOWN; } while (0)

// A group of point-to-point operations that is ALWAYS closed: the first error is remembered, later operations are
// skipped, ncclGroupEnd runs in any case (a group left open on this thread would swallow every later RCCL call and
// hang the peers), and end() returns what happened.
struct P2PGroup {
    ncclResult_t err;
    bool open;
    P2PGroup() : err(ncclGroupStart()), open(err == ncclSuccess) {}
    void send(const void *p, size_t n, int peer, ncclComm_t comm, hipStream_t st)
    { if (open && err == ncclSuccess) err = ncclSend(p, n, ncclUint32, peer, comm, st); }
    void recv(void *p, size_t n, int peer, ncclComm_t comm, hipStream_t st)
    { if (open && err == ncclSuccess) err = ncclRecv(p, n, ncclUint32, peer, comm, st); }
    CUDPPResult end()
    {
        if (open) { const ncclResult_t e = ncclGroupEnd(); open = false; if (err == ncclSuccess) err = e; }
        return err == ncclSuccess ? CUDPP_SUCCESS : CUDPP_ERROR_UNKNOWN;
    }
    ~P2PGroup() { if (open) (void)ncclGroupEnd(); }
};

__global__ void k_pack_records(const int *__restrict__ idx, const unsigned int *__restrict__ hist,
                               const unsigned int *__restrict__ off, size_t off_stride,
                               const unsigned int *__restrict__ size, uint32_t nsub, unsigned int *__restrict__ rec)
{
    const uint32_t b = blockIdx.x, R = GLC_RECORD_FIXED_WORDS + nsub;
    unsigned int *r = rec + (size_t)b * R;
    for (uint32_t i = threadIdx.x; i < R; i += blockDim.x)
        r[i] = i == 0 ? size[b] : i == 1 ? (unsigned int)idx[b] : i < GLC_RECORD_FIXED_WORDS ? hist[(size_t)b * 256 + i - 2]
                                                                                              : off[(size_t)b * off_stride + i - GLC_RECORD_FIXED_WORDS];
}

__global__ void k_unpack_records(const unsigned int *__restrict__ rec, uint32_t nsub, int *__restrict__ idx,
                                 unsigned int *__restrict__ hist, unsigned int *__restrict__ off, size_t off_stride,
                                 unsigned int *__restrict__ size)
{
    const uint32_t b = blockIdx.x, R = GLC_RECORD_FIXED_WORDS + nsub;
    const unsigned int *r = rec + (size_t)b * R;
    for (uint32_t i = threadIdx.x; i < R; i += blockDim.x) {
        const unsigned int v = r[i];
        if (i == 0) { if (size) size[b] = v; }
        else if (i == 1) idx[b] = (int)v;
        else if (i < GLC_RECORD_FIXED_WORDS) hist[(size_t)b * 256 + i - 2] = v;
        else off[(size_t)b * off_stride + i - GLC_RECORD_FIXED_WORDS] = v;
    }
}

__global__ void k_set_counts(unsigned long long *dst, unsigned long long nblocks, unsigned long long nwords,
                             const unsigned long long *d_nwords)
{
    dst[0] = nblocks;
    dst[1] = d_nwords ? *d_nwords : nwords;
}

CUDPPResult finish_init(glcComm_st *c)
{
    GLC_NCCL(ncclCommCount(c->comm, &c->nranks));
    GLC_NCCL(ncclCommUserRank(c->comm, &c->rank));
    GLC_HIP(hipMalloc((void **)&c->d_counts, sizeof(unsigned long long) * c->slot_words() * GLC_COUNT_SLOTS));
    GLC_HIP(hipHostMalloc((void **)&c->h_counts, sizeof(unsigned long long) * 2 * (size_t)c->nranks * GLC_COUNT_SLOTS, hipHostMallocDefault));
    for (int i = 0; i < GLC_COUNT_SLOTS; i++) GLC_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
    return CUDPP_SUCCESS;
}

} // namespace

extern "C" {

CUDPPResult glcCommGetUniqueId(void *id128)
{
    if (!id128) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    ncclUniqueId id;
    GLC_NCCL(ncclGetUniqueId(&id));
    static_assert(sizeof(id) == GLC_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof id);
    return CUDPP_SUCCESS;
}

CUDPPResult glcCommInitRank(glcComm_t *comm, int nranks, const void *id128, int rank)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    glcComm_st *c = new (std::nothrow) glcComm_st();
    if (!c) return CUDPP_ERROR_INSUFFICIENT_RESOURCES;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (ncclCommInitRank(&c->comm, nranks, id, rank) != ncclSuccess) { delete c; return CUDPP_ERROR_UNKNOWN; }
    c->owned = true;
    const CUDPPResult r = finish_init(c);
    if (r != CUDPP_SUCCESS) { (void)glcCommDestroy(c); return r; }
    *comm = c;
    return CUDPP_SUCCESS;
}

CUDPPResult glcCommAdopt(glcComm_t *comm, void *ncclComm)
{
    if (!comm || !ncclComm) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    glcComm_st *c = new (std::nothrow) glcComm_st();
    if (!c) return CUDPP_ERROR_INSUFFICIENT_RESOURCES;
    c->comm = (ncclComm_t)ncclComm;
    const CUDPPResult r = finish_init(c);
    if (r != CUDPP_SUCCESS) { (void)glcCommDestroy(c); return r; }
    *comm = c;
    return CUDPP_SUCCESS;
}

CUDPPResult glcCommDestroy(glcComm_t c)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    for (int i = 0; i < GLC_COUNT_SLOTS; i++) if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
    return CUDPP_SUCCESS;
}

CUDPPResult glcCommInfo(glcComm_t c, int *nranks, int *rank)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    return CUDPP_SUCCESS;
}

CUDPPResult glcPackRecords(const int *d_bwtIndex, const unsigned int *d_hist, const unsigned int *d_encodeOffset,
                           size_t offsetStride, const unsigned int *d_compressedSize, size_t nsub, size_t numBlocks,
                           unsigned int *d_records, void *hipStream)
{
    if (!d_bwtIndex || !d_hist || !d_encodeOffset || !d_compressedSize || !d_records || offsetStride < nsub || nsub > 0xFFFFu)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numBlocks == 0) return CUDPP_SUCCESS;
    hipLaunchKernelGGL(k_pack_records, dim3((unsigned)numBlocks), dim3(256), 0, (hipStream_t)hipStream, d_bwtIndex, d_hist,
                       d_encodeOffset, offsetStride, d_compressedSize, (uint32_t)nsub, d_records);
    return hipGetLastError() == hipSuccess ? CUDPP_SUCCESS : CUDPP_ERROR_UNKNOWN;
}

CUDPPResult glcUnpackRecords(const unsigned int *d_records, size_t nsub, size_t numBlocks, int *d_bwtIndex,
                             unsigned int *d_hist, unsigned int *d_encodeOffset, size_t offsetStride,
                             unsigned int *d_compressedSize, void *hipStream)
{
    if (!d_records || !d_bwtIndex || !d_hist || !d_encodeOffset || offsetStride < nsub || nsub > 0xFFFFu)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numBlocks == 0) return CUDPP_SUCCESS;
    hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)numBlocks), dim3(256), 0, (hipStream_t)hipStream, d_records,
                       (uint32_t)nsub, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compressedSize);
    return hipGetLastError() == hipSuccess ? CUDPP_SUCCESS : CUDPP_ERROR_UNKNOWN;
}

CUDPPResult glcGatherCountsBegin(glcComm_t c, unsigned long long numBlocks, unsigned long long numWords,
                                 const unsigned long long *d_numWords, int *ticket, void *hipStream)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (!ticket) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    // any free slot (tickets may be ended out of order: the ticket names the slot).  All ranks begin and end their
    // exchanges in the same order, so they fail here together or not at all.
    unsigned slot = GLC_COUNT_SLOTS;
    for (unsigned k = 0; k < GLC_COUNT_SLOTS && slot == GLC_COUNT_SLOTS; k++)
        if (!c->busy[(c->next + k) % GLC_COUNT_SLOTS]) slot = (c->next + k) % GLC_COUNT_SLOTS;
    if (slot == GLC_COUNT_SLOTS) return CUDPP_ERROR_INSUFFICIENT_RESOURCES;   // GLC_COUNT_SLOTS exchanges begun and not ended
    hipStream_t st = (hipStream_t)hipStream;
    unsigned long long *d = c->d_counts + slot * c->slot_words();
    hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(1), 0, st, d, numBlocks, numWords, d_numWords);
    GLC_HIP(hipGetLastError());
    GLC_NCCL(ncclAllGather(d, d + 2, 2, ncclUint64, c->comm, st));
    GLC_HIP(hipMemcpyAsync(c->h_counts + slot * 2 * (size_t)c->nranks, d + 2, sizeof(unsigned long long) * 2 * (size_t)c->nranks,
                           hipMemcpyDeviceToHost, st));
    GLC_HIP(hipEventRecord(c->done[slot], st));
    c->busy[slot] = true;
    c->next++;
    *ticket = (int)slot;
    return CUDPP_SUCCESS;
}

CUDPPResult glcGatherCountsReady(glcComm_t c, int ticket, int *ready)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (ticket < 0 || ticket >= GLC_COUNT_SLOTS || !c->busy[ticket] || !ready) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const hipError_t e = hipEventQuery(c->done[ticket]);
    if (e != hipSuccess && e != hipErrorNotReady) return CUDPP_ERROR_UNKNOWN;
    *ready = e == hipSuccess;
    return CUDPP_SUCCESS;
}

CUDPPResult glcGatherCountsEnd(glcComm_t c, int ticket, unsigned long long *h_counts)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (ticket < 0 || ticket >= GLC_COUNT_SLOTS || !c->busy[ticket] || !h_counts) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    GLC_HIP(hipEventSynchronize(c->done[ticket]));                   // waits for THIS exchange only, not for the stream
    memcpy(h_counts, c->h_counts + (size_t)ticket * 2 * (size_t)c->nranks, sizeof(unsigned long long) * 2 * (size_t)c->nranks);
    c->busy[ticket] = false;                                         // (only now: the slot's buffers are read out)
    return CUDPP_SUCCESS;
}

CUDPPResult glcGatherCounts(glcComm_t c, unsigned long long numBlocks, unsigned long long numWords,
                            const unsigned long long *d_numWords, unsigned long long *h_counts, void *hipStream)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (!h_counts) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    int ticket = -1;
    const CUDPPResult r = glcGatherCountsBegin(c, numBlocks, numWords, d_numWords, &ticket, hipStream);
    return r != CUDPP_SUCCESS ? r : glcGatherCountsEnd(c, ticket, h_counts);
}

CUDPPResult glcGatherStreams(glcComm_t c, int root, const unsigned int *d_words, const unsigned int *d_records,
                             size_t recordWords, const unsigned long long *h_counts, unsigned int *d_allWords,
                             unsigned int *d_allRecords, void *hipStream)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (!h_counts || root < 0 || root >= c->nranks || recordWords < GLC_RECORD_FIXED_WORDS) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    hipStream_t st = (hipStream_t)hipStream;
    const unsigned long long myb = h_counts[2 * c->rank], myw = h_counts[2 * c->rank + 1];
    if ((myb && !d_records) || (myw && !d_words)) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (c->rank != root) {
        P2PGroup g;
        if (myb) g.send(d_records, (size_t)myb * recordWords, root, c->comm, st);
        if (myw) g.send(d_words, (size_t)myw, root, c->comm, st);
        return g.end();
    }
    if (!d_allWords || !d_allRecords) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    unsigned long long wo = 0, bo = 0;
    {
        P2PGroup g;
        for (int r = 0; r < c->nranks; r++) {
            const unsigned long long nb = h_counts[2 * r], nw = h_counts[2 * r + 1];
            if (r != root) {
                if (nb) g.recv(d_allRecords + bo * recordWords, (size_t)nb * recordWords, r, c->comm, st);
                if (nw) g.recv(d_allWords + wo, (size_t)nw, r, c->comm, st);
            }
            wo += nw; bo += nb;
        }
        const CUDPPResult gr = g.end();
        if (gr != CUDPP_SUCCESS) return gr;
    }
    wo = 0; bo = 0;
    for (int r = 0; r < root; r++) { bo += h_counts[2 * r]; wo += h_counts[2 * r + 1]; }
    if (myb && d_allRecords + bo * recordWords != d_records)
        GLC_HIP(hipMemcpyAsync(d_allRecords + bo * recordWords, d_records, (size_t)myb * recordWords * 4, hipMemcpyDeviceToDevice, st));
    if (myw && d_allWords + wo != d_words)
        GLC_HIP(hipMemcpyAsync(d_allWords + wo, d_words, (size_t)myw * 4, hipMemcpyDeviceToDevice, st));
    return CUDPP_SUCCESS;
}

CUDPPResult glcScatterStreams(glcComm_t c, int root, const unsigned int *d_allWords, const unsigned int *d_allRecords,
                              size_t recordWords, const unsigned long long *h_counts, unsigned int *d_words,
                              unsigned int *d_records, void *hipStream)
{
    if (!c) return CUDPP_ERROR_INVALID_HANDLE;
    if (!h_counts || root < 0 || root >= c->nranks || recordWords < GLC_RECORD_FIXED_WORDS) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    hipStream_t st = (hipStream_t)hipStream;
    const unsigned long long myb = h_counts[2 * c->rank], myw = h_counts[2 * c->rank + 1];
    if ((myb && !d_records) || (myw && !d_words)) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (c->rank != root) {
        P2PGroup g;
        if (myb) g.recv(d_records, (size_t)myb * recordWords, root, c->comm, st);
        if (myw) g.recv(d_words, (size_t)myw, root, c->comm, st);
        return g.end();
    }
    if (!d_allWords || !d_allRecords) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    unsigned long long wo = 0, bo = 0, mywo = 0, mybo = 0;
    {
        P2PGroup g;
        for (int r = 0; r < c->nranks; r++) {
            const unsigned long long nb = h_counts[2 * r], nw = h_counts[2 * r + 1];
            if (r != root) {
                if (nb) g.send(d_allRecords + bo * recordWords, (size_t)nb * recordWords, r, c->comm, st);
                if (nw) g.send(d_allWords + wo, (size_t)nw, r, c->comm, st);
            } else { mywo = wo; mybo = bo; }
            wo += nw; bo += nb;
        }
        const CUDPPResult gr = g.end();
        if (gr != CUDPP_SUCCESS) return gr;
    }
    if (myb && d_records != d_allRecords + mybo * recordWords)
        GLC_HIP(hipMemcpyAsync(d_records, d_allRecords + mybo * recordWords, (size_t)myb * recordWords * 4, hipMemcpyDeviceToDevice, st));
    if (myw && d_words != d_allWords + mywo)
        GLC_HIP(hipMemcpyAsync(d_words, d_allWords + mywo, (size_t)myw * 4, hipMemcpyDeviceToDevice, st));
    return CUDPP_SUCCESS;
}

} // extern "C"
