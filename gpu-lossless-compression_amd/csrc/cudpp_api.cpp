// cudpp_api.cpp -- the C ABI of include/cudpp.h: library/plan handles, argument
// validation and the per-call orchestration of the HIP stages.
//
// Mirrors the reference's public layer (cudpp-inpar/src/cudpp/cudpp.cpp:764-919,
// 1000-1034; cudpp_plan.cpp:29-46,81-292,712-799; cudpp_manager.cpp:40-63):
// same entry points, same handle representation (a pointer cast to size_t,
// cudpp_plan.h:51-54), same validation order and result codes.  Differences are
// deliberate and listed in DESIGN.md: plans are reusable (the reference leaks /
// drifts, sa_app.cu:201-202,340-351), nothing is allocated per call
// (compress_app.cu:257,263; sa_app.cu:73-100), HIP failures are returned as
// CUDPP_ERROR_UNKNOWN instead of exit() (cuda_util.h:13-21).
#include "../../include/cudpp.h"
#include "glc_internal.h"

#include <new>
#include <stdlib.h>
#include <string.h>

using namespace glc;

namespace {

struct Manager {
    int device = 0;
};

struct PlanBase {
    CUDPPConfiguration config{};
    uint32_t n = 0, rows = 1;
    hipStream_t stream = nullptr;
    uint32_t *d_status = nullptr;
    uint32_t *h_status = nullptr;
    bool timing = false;
    KernelProf prof;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
    float last_ms[4] = {0, 0, 0, 0};

    hipError_t init_common()
    {
        hipError_t e = hipMalloc((void **)&d_status, 4);
        if (e != hipSuccess) return e;
        e = hipMemset(d_status, 0, 4);
        if (e != hipSuccess) return e;
        return hipHostMalloc((void **)&h_status, 4, hipHostMallocDefault);
    }
    virtual ~PlanBase()
    {
        if (d_status) (void)hipFree(d_status);
        if (h_status) (void)hipHostFree(h_status);
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
    }
};

struct SaPlan : PlanBase {                       // CUDPPSaPlan (cudpp_plan.h:289-305)
    SaScratch sa;
    ~SaPlan() override { sa_scratch_free(sa); }
};
struct BwtPlan : PlanBase {                      // CUDPPBwtPlan (cudpp_plan.h:343-356)
    SaScratch sa;
    ~BwtPlan() override { sa_scratch_free(sa); }
};
struct MtfPlan : PlanBase {                      // CUDPPMtfPlan (cudpp_plan.h:358-369)
    MtfScratch mtf;
    ~MtfPlan() override { mtf_scratch_free(mtf); }
};
struct CompressPlan : PlanBase {                 // CUDPPCompressPlan (cudpp_plan.h:307-341)
    SaScratch sa;
    MtfScratch mtf;
    HuffScratch huff;
    DecodeScratch dec;
    uint8_t *d_bwt = nullptr, *d_mtf = nullptr;  // [rows][n]
    // stage pipelining (glcPlanSetPipelining): the suffix sort of batch i+1 runs on `side` while the
    // MTF + Huffman stages of batch i run on the plan's stream; d_bwt is double-buffered between them
    bool pipelined = false;
    uint8_t *d_bwt2 = nullptr;
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_sorted[2] = {nullptr, nullptr}, ev_released[2] = {nullptr, nullptr}, ev_s2 = nullptr;
    bool released_valid[2] = {false, false};
    uint32_t calls = 0;
    hipEvent_t ev_dec_a[2] = {nullptr, nullptr}, ev_dec_released[2] = {nullptr, nullptr};
    bool dec_released_valid[2] = {false, false};
    uint32_t dec_calls = 0;
    bool side_busy = false;                      // side-stream work issued since the last join
    void join_side()                             // make the plan's stream wait for everything on the side stream
    {
        if (!side || !side_busy) return;
        (void)hipEventRecord(ev_in, side);
        (void)hipStreamWaitEvent(stream, ev_in, 0);
        side_busy = false;
    }
    hipError_t pipeline_init()
    {
        if (side) return hipSuccess;
        hipError_t e = hipMalloc((void **)&d_bwt2, (size_t)n * rows);
        if (e == hipSuccess) {
            // lowest priority: the second halves fill the slots the first halves leave free.  Measured on the
            // 4 GiB bench: decode 26.6 (default priority) / 27.3 (lowest) / 25.9 GB/s (highest); encode unchanged.
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            // (GLC_SIDE_PRIO = least | same | greatest: A/B knob for the probes)
            const char *pr = getenv("GLC_SIDE_PRIO");
            const int prio = !pr ? least : (pr[0] == 's' ? 0 : (pr[0] == 'g' ? greatest : least));
            e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio);
        }
        hipEvent_t *evs[] = {&ev_in, &ev_sorted[0], &ev_sorted[1], &ev_released[0], &ev_released[1],
                             &ev_dec_a[0], &ev_dec_a[1], &ev_dec_released[0], &ev_dec_released[1]};
        for (auto pe : evs) if (e == hipSuccess) e = hipEventCreateWithFlags(pe, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&ev_s2);
        return e;
    }
    ~CompressPlan() override
    {
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
        hipEvent_t evs[] = {ev_in, ev_sorted[0], ev_sorted[1], ev_released[0], ev_released[1], ev_s2,
                            ev_dec_a[0], ev_dec_a[1], ev_dec_released[0], ev_dec_released[1]};
        for (auto e : evs) if (e) (void)hipEventDestroy(e);
        sa_scratch_free(sa); mtf_scratch_free(mtf); huff_scratch_free(huff); decode_scratch_free(dec);
        if (d_bwt) (void)hipFree(d_bwt);
        if (d_bwt2) (void)hipFree(d_bwt2);
        if (d_mtf) (void)hipFree(d_mtf);
    }
};

template <class T> T *plan_from(CUDPPHandle h) { return reinterpret_cast<T *>(h); }

CUDPPResult validate_options(const CUDPPConfiguration &c)
{   // cudpp_plan.cpp:29-46
    if ((c.options & CUDPP_OPTION_BACKWARD) && (c.options & CUDPP_OPTION_FORWARD))
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if ((c.options & CUDPP_OPTION_EXCLUSIVE) && (c.options & CUDPP_OPTION_INCLUSIVE))
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    return CUDPP_SUCCESS;
}

CUDPPResult hip_result(hipError_t e)
{
    if (e == hipSuccess) return CUDPP_SUCCESS;
    if (e == hipErrorOutOfMemory) return CUDPP_ERROR_INSUFFICIENT_RESOURCES;
    return CUDPP_ERROR_UNKNOWN;
}

struct StageTimer {
    PlanBase *p;
    explicit StageTimer(PlanBase *pl) : p(pl)
    {
        if (p->timing && !p->ev[0])
            for (auto &e : p->ev) (void)hipEventCreate(&e);
    }
    void mark(int i) { if (p->timing) (void)hipEventRecord(p->ev[i], p->stream); }
    void done() { if (p->timing) p->ev_valid = true; }
};

} // namespace

extern "C" {

CUDPPResult cudppCreate(CUDPPHandle *theCudpp)
{
    if (!theCudpp) return CUDPP_ERROR_INVALID_HANDLE;
    Manager *m = new (std::nothrow) Manager();
    if (!m) return CUDPP_ERROR_UNKNOWN;
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; *theCudpp = 0; return CUDPP_ERROR_UNKNOWN; }
    *theCudpp = reinterpret_cast<CUDPPHandle>(m);
    return CUDPP_SUCCESS;
}

CUDPPResult cudppDestroy(CUDPPHandle theCudpp)
{
    if (theCudpp == 0 || theCudpp == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    delete reinterpret_cast<Manager *>(theCudpp);
    return CUDPP_SUCCESS;
}

CUDPPResult cudppPlan(const CUDPPHandle cudppHandle, CUDPPHandle *planHandle, CUDPPConfiguration config,
                      size_t n, size_t rows, size_t /*rowPitch*/)
{
    if (!planHandle) return CUDPP_ERROR_INVALID_HANDLE;
    *planHandle = CUDPP_INVALID_HANDLE;
    if (cudppHandle == 0 || cudppHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    CUDPPResult r = validate_options(config);
    if (r != CUDPP_SUCCESS) return r;
    if (rows == 0) rows = 1;
    if (n == 0 || rows > 65535) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;

    PlanBase *plan = nullptr;
    hipError_t e = hipSuccess;
    switch (config.algorithm) {
    case CUDPP_COMPRESS: {
        if (n > MAX_BLOCK_ELEMS) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
        CompressPlan *p = new (std::nothrow) CompressPlan();
        if (!p) return CUDPP_ERROR_UNKNOWN;
        plan = p;
        e = sa_scratch_alloc(p->sa, (uint32_t)n, (uint32_t)rows);
        if (e == hipSuccess) e = mtf_scratch_alloc(p->mtf, (uint32_t)n, (uint32_t)rows);
        if (e == hipSuccess) e = huff_scratch_alloc(p->huff, (uint32_t)n, (uint32_t)rows);
        if (e == hipSuccess) e = hipMalloc((void **)&p->d_bwt, n * rows);
        if (e == hipSuccess) e = hipMalloc((void **)&p->d_mtf, n * rows);
        break;
    }
    case CUDPP_BWT: {
        if (n > MAX_BLOCK_ELEMS) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
        BwtPlan *p = new (std::nothrow) BwtPlan();
        if (!p) return CUDPP_ERROR_UNKNOWN;
        plan = p;
        e = sa_scratch_alloc(p->sa, (uint32_t)n, (uint32_t)rows);
        break;
    }
    case CUDPP_SA: {
        if (n > MAX_BLOCK_ELEMS) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
        SaPlan *p = new (std::nothrow) SaPlan();
        if (!p) return CUDPP_ERROR_UNKNOWN;
        plan = p;
        e = sa_scratch_alloc(p->sa, (uint32_t)n, (uint32_t)rows);
        break;
    }
    case CUDPP_MTF: {
        if (n > (1u << 30)) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
        MtfPlan *p = new (std::nothrow) MtfPlan();
        if (!p) return CUDPP_ERROR_UNKNOWN;
        plan = p;
        e = mtf_scratch_alloc(p->mtf, (uint32_t)n, (uint32_t)rows);
        break;
    }
    default:
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;   // not on the compression path
    }
    if (e == hipSuccess) e = plan->init_common();
    if (e != hipSuccess) { delete plan; (void)hipGetLastError(); return hip_result(e); }
    plan->config = config;
    plan->n = (uint32_t)n;
    plan->rows = (uint32_t)rows;
    *planHandle = reinterpret_cast<CUDPPHandle>(plan);
    return CUDPP_SUCCESS;
}

CUDPPResult cudppDestroyPlan(CUDPPHandle planHandle)
{
    if (planHandle == CUDPP_INVALID_HANDLE || planHandle == 0) return CUDPP_ERROR_INVALID_HANDLE;
    PlanBase *p = plan_from<PlanBase>(planHandle);
    switch (p->config.algorithm) {
    case CUDPP_COMPRESS: case CUDPP_BWT: case CUDPP_MTF: case CUDPP_SA:
        (void)hipStreamSynchronize(p->stream);
        delete p;
        return CUDPP_SUCCESS;
    default:
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    }
}

// --------------------------------------------------------------------------
// batched entry points
// --------------------------------------------------------------------------
// One body for the two output layouts.  Strided (d_blockOffsets == nullptr): block b's words at d_compressed +
// b * compressedStrideWords, the reference's layout per block.  Compact: block b's words at d_compressed +
// d_blockOffsets[b], the blocks back to back from *d_startOffset on -- the sizes are known before anything is packed
// (k_huff_build), so the packer writes every block where it ends up and no copy pass follows.  In that mode the packer
// runs once, after the host knows that no block's size can still change (the sorter's tiers are through).
static CUDPPResult compress_batch(CUDPPHandle planHandle, const unsigned char *d_uncompressed, int *d_bwtIndex,
                                  unsigned int *d_hist, unsigned int *d_encodeOffset, size_t offsetStride,
                                  unsigned int *d_compressedSize, unsigned int *d_compressed,
                                  size_t compressedStrideWords, size_t numElements, size_t numBlocks,
                                  unsigned long long *d_blockOffsets, const unsigned long long *d_startOffset,
                                  size_t capacityWords)
{
    const bool compact = d_blockOffsets != nullptr;
    CompressPlan *p = plan_from<CompressPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_COMPRESS) return CUDPP_ERROR_INVALID_PLAN;
    if (p->config.datatype != CUDPP_UCHAR) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numElements == 0 || numElements > p->n || numBlocks == 0 || numBlocks > p->rows)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const uint32_t n = (uint32_t)numElements, nb = (uint32_t)numBlocks;
    const uint32_t nsub = (n + HUFF_BLOCK - 1) / HUFF_BLOCK;
    if (offsetStride < nsub) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    hipStream_t st = p->stream;
    StageTimer tm(p);
    hipError_t e = hipSuccess;
    uint8_t *bwt = p->d_bwt;
    const uint32_t k = p->calls++ & 1u;
    hipStream_t s2 = st;                                       // stream of the MTF + Huffman stages
    if (p->pipelined) {
        // The sort stays on the plan's stream (inputs keep their stream order).  MTF + Huffman move to the side
        // stream, where they overlap the sort of the NEXT call; the outputs they write are complete after
        // glcPlanSynchronize / glcCompactStreams / a device synchronize (include/cudpp.h).
        e = p->pipeline_init();
        if (e != hipSuccess) return hip_result(e);
        bwt = k ? p->d_bwt2 : p->d_bwt;
        if (p->released_valid[k]) (void)hipStreamWaitEvent(st, p->ev_released[k], 0);   // this BWT half is free again
        s2 = p->side;
    }
    // The stages after the sort are queued BEFORE the host waits for the sorter's one readback (how many blocks the
    // bucket sorter handed to the general sorter): the GPU has MTF + Huffman to do while the host wakes up and queues
    // the next call.  In the rare batch with flagged blocks they run again on the corrected BWT.
    tm.mark(0);
    p->sa.parity = k;
    e = sa_build_begin(st, d_uncompressed, n, n, nb, p->sa, bwt, p->n, d_bwtIndex);
    tm.mark(1);
    auto after_sort = [&](const uint32_t *redo_flag, const uint32_t *only, bool skewed) {
        if (p->pipelined) {
            (void)hipEventRecord(p->ev_sorted[k], st);
            (void)hipStreamWaitEvent(s2, p->ev_sorted[k], 0);
            if (p->timing) (void)hipEventRecord(p->ev_s2, s2);
        }
        if (e == hipSuccess) e = mtf_forward(s2, bwt, p->n, n, nb, p->d_mtf, p->n, p->mtf, p->huff.sub_hist, only, skewed);
        if (p->timing) (void)hipEventRecord(p->ev[2], s2);
        // (compact layout: a block has no slot of its own to overflow -- the array's capacity is checked with the offsets)
        if (e == hipSuccess) e = huff_build(s2, n, nb, p->huff, d_hist, d_encodeOffset, offsetStride, d_compressedSize,
                                            compact ? (size_t)(HUFF_MAX_WORDS + 1) * nsub : compressedStrideWords, p->d_status,
                                            redo_flag, only);
        if (e == hipSuccess && !compact) e = huff_pack(s2, p->d_mtf, p->n, n, nb, p->huff, d_encodeOffset, offsetStride,
                                                       d_compressed, compressedStrideWords, only);
        if (p->timing) (void)hipEventRecord(p->ev[3], s2);
    };
    // Blocks the bucket sorter has given up on (flagged up front as text-like, or after its attempt) are skipped by this
    // speculative pass (`only` = the sorter's keep mask of this call) and encoded below, once their sort is final.
    const bool tiers = p->sa.sorter == 0 || p->sa.sorter == 3 || p->sa.sorter == 4;
    const bool speculate = !tiers || !sa_skips_tier1(p->sa, nb);   // (no attempt of the bucket sorter: nothing to speculate on)
    if (speculate) after_sort(nullptr, tiers ? p->sa.fs_keep[k] : nullptr, false);
    uint32_t nflag = 0;
    // (not in the pipelined mode, whose stages have a stream of their own already, nor under the stage timer, whose events sit on
    //  the plan's stream: the blocks the sample sorter's first attempt finished get their MTF + Huffman beside its second attempt)
    if (speculate && tiers && !p->pipelined && !p->timing)
        p->sa.stage_partial = [&](hipStream_t aux, const uint32_t *only) -> hipError_t {
            hipError_t e2 = mtf_forward(aux, bwt, p->n, n, nb, p->d_mtf, p->n, p->mtf, p->huff.sub_hist, only, true);
            if (e2 == hipSuccess) e2 = huff_build(aux, n, nb, p->huff, d_hist, d_encodeOffset, offsetStride, d_compressedSize,
                                                  compact ? (size_t)(HUFF_MAX_WORDS + 1) * nsub : compressedStrideWords, p->d_status, nullptr, only);
            if (e2 == hipSuccess && !compact) e2 = huff_pack(aux, p->d_mtf, p->n, n, nb, p->huff, d_encodeOffset, offsetStride, d_compressed,
                                                             compressedStrideWords, only);
            return e2;
        };
    if (e == hipSuccess) e = sa_build_finish(st, d_uncompressed, n, n, nb, p->sa, bwt, p->n, d_bwtIndex, &nflag);
    p->sa.stage_partial = nullptr;                             // (it refers to this call's arguments)
    if (e == hipSuccess && (nflag || !speculate)) {
        // sa_build_finish has queued the other sorters for the flagged blocks on st; this pass is ordered after the
        // last of them (ev_sorted) and touches only those blocks -- those still open after the sample sorter's first attempt if
        // the others' stages have been queued beside its second one
        tm.mark(1);                                            // the sort stage ends here: the other tiers' time is the sort's
        after_sort(nullptr, p->sa.partial_used ? p->sa.ss_mask[1] : (speculate && tiers ? p->sa.fs_redo[k] : nullptr), true);   // (the blocks of the other tiers: text-like)
    }
    if (e == hipSuccess && compact) {
        e = huff_block_offsets(s2, d_compressedSize, nb, d_blockOffsets, d_startOffset, capacityWords, p->d_status);
        if (e == hipSuccess) e = huff_pack(s2, p->d_mtf, p->n, n, nb, p->huff, d_encodeOffset, offsetStride, d_compressed, 0,
                                           nullptr, d_blockOffsets, capacityWords);
        if (p->timing) (void)hipEventRecord(p->ev[3], s2);
    }
    tm.done();
    if (p->pipelined) { (void)hipEventRecord(p->ev_released[k], s2); p->released_valid[k] = true; p->side_busy = true; }
    return hip_result(e);
}

CUDPPResult glcCompressBatch(CUDPPHandle planHandle, const unsigned char *d_uncompressed, int *d_bwtIndex,
                             unsigned int *d_hist, unsigned int *d_encodeOffset, size_t offsetStride,
                             unsigned int *d_compressedSize, unsigned int *d_compressed,
                             size_t compressedStrideWords, size_t numElements, size_t numBlocks)
{
    return compress_batch(planHandle, d_uncompressed, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compressedSize,
                          d_compressed, compressedStrideWords, numElements, numBlocks, nullptr, nullptr, 0);
}

CUDPPResult glcCompressBatchCompact(CUDPPHandle planHandle, const unsigned char *d_uncompressed, int *d_bwtIndex,
                                    unsigned int *d_hist, unsigned int *d_encodeOffset, size_t offsetStride,
                                    unsigned int *d_compressedSize, unsigned int *d_compact, size_t capacityWords,
                                    unsigned long long *d_blockOffsets, const unsigned long long *d_startOffset,
                                    size_t numElements, size_t numBlocks)
{
    if (!d_blockOffsets || !d_compact) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    return compress_batch(planHandle, d_uncompressed, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compressedSize,
                          d_compact, 0, numElements, numBlocks, d_blockOffsets, d_startOffset, capacityWords);
}

// the Huffman half of cudppCompress on its own (histogram, tree + codes, bit packer, offsets: rows a5-a8)
CUDPPResult glcHuffmanEncodeBatch(CUDPPHandle planHandle, const unsigned char *d_symbols, unsigned int *d_hist,
                                  unsigned int *d_encodeOffset, size_t offsetStride, unsigned int *d_compressedSize,
                                  unsigned int *d_compressed, size_t compressedStrideWords, size_t numElements,
                                  size_t numBlocks)
{
    CompressPlan *p = plan_from<CompressPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_COMPRESS) return CUDPP_ERROR_INVALID_PLAN;
    if (numElements == 0 || numElements > p->n || numBlocks == 0 || numBlocks > p->rows)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const uint32_t n = (uint32_t)numElements, nb = (uint32_t)numBlocks;
    if (offsetStride < (n + HUFF_BLOCK - 1) / HUFF_BLOCK) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    p->join_side();
    hipStream_t st = p->stream;
    hipError_t e = huff_histogram(st, d_symbols, n, n, nb, p->huff);
    if (e == hipSuccess) e = huff_build(st, n, nb, p->huff, d_hist, d_encodeOffset, offsetStride, d_compressedSize,
                                        compressedStrideWords, p->d_status);
    if (e == hipSuccess) e = huff_pack(st, d_symbols, n, n, nb, p->huff, d_encodeOffset, offsetStride, d_compressed,
                                       compressedStrideWords);
    return hip_result(e);
}

CUDPPResult glcPlanSetPipelining(CUDPPHandle planHandle, int on)
{
    CompressPlan *p = plan_from<CompressPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_COMPRESS) return CUDPP_ERROR_INVALID_PLAN;
    p->join_side();
    if (p->side) (void)hipStreamSynchronize(p->side);
    (void)hipStreamSynchronize(p->stream);
    p->pipelined = on != 0;
    p->released_valid[0] = p->released_valid[1] = false;
    p->dec_released_valid[0] = p->dec_released_valid[1] = false;
    return CUDPP_SUCCESS;
}

CUDPPResult glcBwtBatch(CUDPPHandle planHandle, const unsigned char *d_in, unsigned char *d_out, int *d_index,
                        size_t numElements, size_t numBlocks)
{
    BwtPlan *p = plan_from<BwtPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_BWT) return CUDPP_ERROR_INVALID_PLAN;
    if (p->config.datatype != CUDPP_UCHAR) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numElements == 0 || numElements > p->n || numBlocks == 0 || numBlocks > p->rows)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const uint32_t n = (uint32_t)numElements, nb = (uint32_t)numBlocks;
    hipError_t e = sa_build(p->stream, d_in, n, n, nb, p->sa, d_out, n, d_index);
    return hip_result(e);
}

CUDPPResult glcMtfBatch(CUDPPHandle planHandle, const unsigned char *d_in, unsigned char *d_out,
                        size_t numElements, size_t numBlocks)
{
    MtfPlan *p = plan_from<MtfPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_MTF) return CUDPP_ERROR_INVALID_PLAN;
    if (p->config.datatype != CUDPP_UCHAR) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numElements == 0 || numElements > p->n || numBlocks == 0 || numBlocks > p->rows)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const uint32_t n = (uint32_t)numElements, nb = (uint32_t)numBlocks;
    return hip_result(mtf_forward(p->stream, d_in, n, n, nb, d_out, n, p->mtf, nullptr));
}

static CUDPPResult decompress_batch(CUDPPHandle planHandle, const int *d_bwtIndex, const unsigned int *d_hist,
                                    const unsigned int *d_encodeOffset, size_t offsetStride,
                                    const unsigned int *d_compressed, size_t compressedStrideWords,
                                    unsigned char *d_out, size_t numElements, size_t numBlocks,
                                    const unsigned long long *d_blockOffsets)
{
    CompressPlan *p = plan_from<CompressPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_COMPRESS) return CUDPP_ERROR_INVALID_PLAN;
    if (numElements == 0 || numElements > p->n || numBlocks == 0 || numBlocks > p->rows)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (!p->dec.lf) {
        hipError_t e = decode_scratch_alloc(p->dec, p->n, p->rows);
        if (e != hipSuccess) return hip_result(e);
    }
    if (!p->pipelined)
        return hip_result(decode_blocks(p->stream, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compressed,
                                        compressedStrideWords, d_out, (uint32_t)numElements, (uint32_t)numBlocks,
                                        p->dec, p->mtf, p->d_status, d_blockOffsets));
    // pipelined: Huffman + inverse MTF on the plan's stream (inputs keep their stream order), the inverse
    // BWT -- a memory-latency-bound walk -- on the side stream, where it overlaps stage A of the next call.
    // d_out is complete after glcPlanSynchronize / a device synchronize.
    hipError_t e = p->pipeline_init();
    if (e != hipSuccess) return hip_result(e);
    hipStream_t st = p->stream;
    const uint32_t k = p->dec_calls++ & 1u;
    uint8_t *bwt = k ? p->dec.bwt2 : p->dec.bwt;
    if (p->dec_released_valid[k]) (void)hipStreamWaitEvent(st, p->ev_dec_released[k], 0);
    e = decode_stage_a(st, d_hist, d_encodeOffset, offsetStride, d_compressed, compressedStrideWords,
                       (uint32_t)numElements, (uint32_t)numBlocks, p->dec, bwt, p->d_status, d_blockOffsets);
    (void)hipEventRecord(p->ev_dec_a[k], st);
    (void)hipStreamWaitEvent(p->side, p->ev_dec_a[k], 0);
    if (e == hipSuccess) e = decode_stage_b(p->side, d_bwtIndex, bwt, d_out, (uint32_t)numElements, (uint32_t)numBlocks, p->dec, p->d_status);
    (void)hipEventRecord(p->ev_dec_released[k], p->side);
    p->dec_released_valid[k] = true;
    p->side_busy = true;
    return hip_result(e);
}

CUDPPResult glcDecompressBatch(CUDPPHandle planHandle, const int *d_bwtIndex, const unsigned int *d_hist,
                               const unsigned int *d_encodeOffset, size_t offsetStride,
                               const unsigned int *d_compressed, size_t compressedStrideWords,
                               unsigned char *d_out, size_t numElements, size_t numBlocks)
{
    return decompress_batch(planHandle, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compressed,
                            compressedStrideWords, d_out, numElements, numBlocks, nullptr);
}

CUDPPResult glcDecompressBatchCompact(CUDPPHandle planHandle, const int *d_bwtIndex, const unsigned int *d_hist,
                                      const unsigned int *d_encodeOffset, size_t offsetStride,
                                      const unsigned int *d_compact, size_t compactWords,
                                      const unsigned long long *d_blockOffsets, unsigned char *d_out,
                                      size_t numElements, size_t numBlocks)
{
    if (!d_blockOffsets) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    return decompress_batch(planHandle, d_bwtIndex, d_hist, d_encodeOffset, offsetStride, d_compact, compactWords, d_out,
                            numElements, numBlocks, d_blockOffsets);
}

// --------------------------------------------------------------------------
// the reference's single-block entry points
// --------------------------------------------------------------------------
CUDPPResult cudppCompress(CUDPPHandle planHandle, unsigned char *d_uncompressed, int *d_bwtIndex,
                          unsigned int * /*d_histSize: ignored, as compress_app.cu:507-526*/,
                          unsigned int *d_hist, unsigned int *d_encodeOffset, unsigned int *d_compressedSize,
                          unsigned int *d_compressed, size_t numElements)
{
    if (planHandle == 0) return CUDPP_ERROR_INVALID_HANDLE;
    const size_t nsub = (numElements + HUFF_BLOCK - 1) / HUFF_BLOCK;
    return glcCompressBatch(planHandle, d_uncompressed, d_bwtIndex, d_hist, d_encodeOffset, nsub ? nsub : 1,
                            d_compressedSize, d_compressed, (size_t)(HUFF_MAX_WORDS + 1) * (nsub ? nsub : 1),
                            numElements, 1);
}

CUDPPResult cudppBurrowsWheelerTransform(CUDPPHandle planHandle, unsigned char *d_in, unsigned char *d_out,
                                         int *d_index, size_t numElements)
{
    if (planHandle == 0) return CUDPP_ERROR_INVALID_HANDLE;
    return glcBwtBatch(planHandle, d_in, d_out, d_index, numElements, 1);
}

CUDPPResult cudppMoveToFrontTransform(CUDPPHandle planHandle, unsigned char *d_in, unsigned char *d_out,
                                      size_t numElements)
{
    if (planHandle == 0) return CUDPP_ERROR_INVALID_HANDLE;
    return glcMtfBatch(planHandle, d_in, d_out, numElements, 1);
}

CUDPPResult cudppSuffixArray(CUDPPHandle planHandle, unsigned char *d_str, unsigned int *d_keys_sa,
                             size_t numElements)
{
    SaPlan *p = plan_from<SaPlan>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm != CUDPP_SA) return CUDPP_ERROR_INVALID_PLAN;
    if (p->config.datatype != CUDPP_UCHAR) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (numElements == 0 || numElements > p->n) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    const uint32_t n = (uint32_t)numElements;
    hipError_t e = sa_build(p->stream, d_str, n, n, 1, p->sa);
    if (e == hipSuccess) e = sa_export(p->stream, p->sa.sa, n, d_keys_sa);
    return hip_result(e);
}

// --------------------------------------------------------------------------
// plan utilities
// --------------------------------------------------------------------------
CUDPPResult glcPlanSetStream(CUDPPHandle planHandle, void *hipStream)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    p->stream = reinterpret_cast<hipStream_t>(hipStream);
    return CUDPP_SUCCESS;
}

__global__ void k_status_fetch(uint32_t *__restrict__ d_status, uint32_t *__restrict__ h_status)
{
    *reinterpret_cast<volatile uint32_t *>(h_status) = *d_status;
    *d_status = 0;
}

CUDPPResult glcPlanSynchronize(CUDPPHandle planHandle)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (p->config.algorithm == CUDPP_COMPRESS) static_cast<CompressPlan *>(p)->join_side();
    // status word: fetched into pinned memory and cleared by ONE small kernel (a copy command + a fill command were two more
    // ~5 us links in the chain a cudppCompress caller waits for)
    hipLaunchKernelGGL(k_status_fetch, dim3(1), dim3(1), 0, p->stream, p->d_status, p->h_status);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) return CUDPP_ERROR_UNKNOWN;
    if (p->timing && p->ev_valid) {
        (void)hipEventElapsedTime(&p->last_ms[0], p->ev[0], p->ev[1]);
        CompressPlan *cp = p->config.algorithm == CUDPP_COMPRESS ? static_cast<CompressPlan *>(p) : nullptr;
        if (cp && cp->pipelined && cp->ev_s2) {                 // stages ran on two streams: report their own spans
            (void)hipEventElapsedTime(&p->last_ms[1], cp->ev_s2, p->ev[2]);
            (void)hipEventElapsedTime(&p->last_ms[2], p->ev[2], p->ev[3]);
            p->last_ms[3] = p->last_ms[0] + p->last_ms[1] + p->last_ms[2];
        } else {
            (void)hipEventElapsedTime(&p->last_ms[1], p->ev[1], p->ev[2]);
            (void)hipEventElapsedTime(&p->last_ms[2], p->ev[2], p->ev[3]);
            (void)hipEventElapsedTime(&p->last_ms[3], p->ev[0], p->ev[3]);
        }
    }
    return *p->h_status ? CUDPP_ERROR_UNKNOWN : CUDPP_SUCCESS;
}

static SaScratch *sa_of(PlanBase *p)
{
    switch (p->config.algorithm) {
    case CUDPP_COMPRESS: return &static_cast<CompressPlan *>(p)->sa;
    case CUDPP_BWT: return &static_cast<BwtPlan *>(p)->sa;
    case CUDPP_SA: return &static_cast<SaPlan *>(p)->sa;
    default: return nullptr;
    }
}

CUDPPResult glcPlanEnableTiming(CUDPPHandle planHandle, int enable)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    p->timing = enable != 0;
    p->ev_valid = false;
    if (!p->prof.enable((enable & 2) != 0)) return CUDPP_ERROR_INSUFFICIENT_RESOURCES;
    p->prof.reset();
    if (SaScratch *s = sa_of(p)) s->prof = &p->prof;
    if (p->config.algorithm == CUDPP_COMPRESS) {
        CompressPlan *cp = static_cast<CompressPlan *>(p);
        cp->mtf.prof = &p->prof; cp->huff.prof = &p->prof; cp->dec.prof = &p->prof;
    } else if (p->config.algorithm == CUDPP_MTF) static_cast<MtfPlan *>(p)->mtf.prof = &p->prof;
    return CUDPP_SUCCESS;
}

CUDPPResult glcPlanSetSorter(CUDPPHandle planHandle, int mode)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    if (mode < 0 || mode > 7) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    s->sorter = mode >= 5 ? 0 : mode;
    s->resume_min = mode == 5 ? 0u : (mode == 6 ? 1u : 2u);
    s->periodic = mode != 7 && mode != 5;                      // (5: "from scratch" for everything the sample sorter gives up on)
    return CUDPP_SUCCESS;
}

CUDPPResult glcPlanLastSortStats(CUDPPHandle planHandle, unsigned int *flaggedBlocks)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !flaggedBlocks) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    *flaggedBlocks = s->last_flagged;
    return CUDPP_SUCCESS;
}

CUDPPResult glcPlanLastSortStatsEx(CUDPPHandle planHandle, unsigned int *out2)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out2) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    out2[0] = s->last_flagged;
    out2[1] = s->last_general;
    return CUDPP_SUCCESS;
}

// out[0] = blocks of the plan's last call that the sample sorter finished only in its second attempt (other samples)
CUDPPResult glcPlanLastSortRetries(CUDPPHandle planHandle, unsigned int *out)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    out[0] = s->last_retried;
    return CUDPP_SUCCESS;
}

// out[0] = 1 if the plan's last call skipped the bucket sorter's attempt (sorter mode 4, or a small call behind a streak of calls
// whose every block the text-likeness probe flagged), out[1] = the streak
CUDPPResult glcPlanLastSortSkipped(CUDPPHandle planHandle, unsigned int *out2)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out2) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    out2[0] = s->last_skipped ? 1u : 0u;
    out2[1] = s->textlike_streak;
    return CUDPP_SUCCESS;
}

// out[0] = blocks of the plan's last call whose doubling rounds resumed from the sample sorter's tolerant form
CUDPPResult glcPlanLastSortPeriodic(CUDPPHandle planHandle, unsigned int *out)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    out[0] = s->last_periodic;
    return CUDPP_SUCCESS;
}

CUDPPResult glcPlanLastSortResumed(CUDPPHandle planHandle, unsigned int *out)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s) return CUDPP_ERROR_INVALID_PLAN;
    out[0] = s->last_resumed;
    return CUDPP_SUCCESS;
}

// diagnostic: the give-up flags of the plan's last sort, per block (waits for the plan's stream).  out_fs[b]: bucket sorter
// (1 = a bucket overflowed / flagged up front as text-like, 2 = equal codes deeper than the cap, 4 = work list full);
// out_ss[b]: sample sorter (1 = a bucket overflowed, 2 = deeper than its cap / a run no window holds)
CUDPPResult glcPlanDebugSortFlags(CUDPPHandle planHandle, unsigned int *out_fs, unsigned int *out_ss, size_t numBlocks)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s || numBlocks > s->rows) return CUDPP_ERROR_INVALID_PLAN;
    if (hipStreamSynchronize(p->stream) != hipSuccess) return CUDPP_ERROR_UNKNOWN;
    if (out_fs && hipMemcpy(out_fs, s->fs_flag, numBlocks * 4, hipMemcpyDeviceToHost) != hipSuccess) return CUDPP_ERROR_UNKNOWN;
    if (out_ss && hipMemcpy(out_ss, s->ss_flag, numBlocks * 4, hipMemcpyDeviceToHost) != hipSuccess) return CUDPP_ERROR_UNKNOWN;
    return CUDPP_SUCCESS;
}

// diagnostic: bucket fills of block `block` as the last bucketing pass left them (FS_MAXNB = 512 entries)
CUDPPResult glcPlanDebugBucketFill(CUDPPHandle planHandle, size_t block, unsigned int *out512)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out512) return CUDPP_ERROR_INVALID_HANDLE;
    SaScratch *s = sa_of(p);
    if (!s || block >= s->rows) return CUDPP_ERROR_INVALID_PLAN;
    if (hipStreamSynchronize(p->stream) != hipSuccess) return CUDPP_ERROR_UNKNOWN;
    return hipMemcpy(out512, s->fs_fill + block * FS_MAXNB, FS_MAXNB * 4, hipMemcpyDeviceToHost) == hipSuccess ? CUDPP_SUCCESS : CUDPP_ERROR_UNKNOWN;
}

// the event pairs are read when the plan's streams are idle: both getters wait for them first
static void prof_collect(PlanBase *p)
{
    if (p->prof.npend == 0) return;
    if (p->config.algorithm == CUDPP_COMPRESS) static_cast<CompressPlan *>(p)->join_side();
    (void)hipStreamSynchronize(p->stream);
    p->prof.collect();
}

CUDPPResult glcPlanKernelProfileEx(CUDPPHandle planHandle, int index, char *name, size_t nameCap, double *out3)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out3) return CUDPP_ERROR_INVALID_HANDLE;
    if (index < 0 || index >= PROF_NSLOT) return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    prof_collect(p);
    out3[0] = p->prof.ms[index]; out3[1] = (double)p->prof.launches[index]; out3[2] = p->prof.units[index];
    if (name && nameCap) { strncpy(name, p->prof.name[index], nameCap - 1); name[nameCap - 1] = 0; }
    return CUDPP_SUCCESS;
}

// the kernel with the largest accumulated launch time: {ms, launches, input bytes processed}; resets the profile
CUDPPResult glcPlanKernelProfile(CUDPPHandle planHandle, double *out3)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out3) return CUDPP_ERROR_INVALID_HANDLE;
    prof_collect(p);
    int best = 0;
    for (int k = 1; k < PROF_NSLOT; k++) if (p->prof.ms[k] > p->prof.ms[best]) best = k;
    out3[0] = p->prof.ms[best]; out3[1] = (double)p->prof.launches[best]; out3[2] = p->prof.units[best];
    p->prof.reset();
    return CUDPP_SUCCESS;
}

// launches the live profile could not account for: out2[0] = not bracketed (more than 4096 launches pending between two
// reads), out2[1] = bracketed but unreadable
CUDPPResult glcPlanKernelProfileLost(CUDPPHandle planHandle, unsigned long long *out2)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !out2) return CUDPP_ERROR_INVALID_HANDLE;
    out2[0] = (unsigned long long)p->prof.dropped; out2[1] = (unsigned long long)p->prof.unread;
    return CUDPP_SUCCESS;
}

CUDPPResult glcCompactStreams(CUDPPHandle planHandle, const unsigned int *d_compressed,
                              size_t compressedStrideWords, const unsigned int *d_compressedSize,
                              size_t numBlocks, unsigned int *d_out, unsigned long long *d_outOffsets)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (!d_compressed || !d_compressedSize || !d_out || !d_outOffsets || numBlocks == 0)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (p->config.algorithm == CUDPP_COMPRESS) static_cast<CompressPlan *>(p)->join_side();
    return hip_result(compact_streams(p->stream, d_compressed, compressedStrideWords, d_compressedSize,
                                      (uint32_t)numBlocks, d_out, d_outOffsets));
}

CUDPPResult glcExpandStreams(CUDPPHandle planHandle, const unsigned int *d_in, const unsigned long long *d_inOffsets,
                             size_t numBlocks, unsigned int *d_compressed, size_t compressedStrideWords,
                             unsigned int *d_compressedSize)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE) return CUDPP_ERROR_INVALID_HANDLE;
    if (!d_in || !d_inOffsets || !d_compressed || numBlocks == 0 || compressedStrideWords == 0)
        return CUDPP_ERROR_ILLEGAL_CONFIGURATION;
    if (p->config.algorithm == CUDPP_COMPRESS) static_cast<CompressPlan *>(p)->join_side();
    return hip_result(expand_streams(p->stream, d_in, d_inOffsets, (uint32_t)numBlocks, d_compressed,
                                     compressedStrideWords, d_compressedSize, p->d_status));
}

CUDPPResult glcPlanLastTiming(CUDPPHandle planHandle, float *ms4)
{
    PlanBase *p = plan_from<PlanBase>(planHandle);
    if (!p || planHandle == CUDPP_INVALID_HANDLE || !ms4) return CUDPP_ERROR_INVALID_HANDLE;
    for (int i = 0; i < 4; i++) ms4[i] = p->last_ms[i];
    return CUDPP_SUCCESS;
}

} // extern "C"
