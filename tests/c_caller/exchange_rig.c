/* A plain-C rank of the multi-GPU path (include/glc_exchange.h over include/cudpp.h), world size 1: the call
 * sequence INTEGRATION.md 3b shows, end to end -- encode a batch, compact, pack records, exchange counts, gather to
 * the root, scatter back, unpack, expand, decode, compare with the input.  Built with gcc (no hipcc, no RCCL
 * headers): the exchange's boundary is a C ABI too.  One GPU can only hold one rank, so the root's own share is the
 * whole exchange here; the layouts and every entry point are the ones N ranks use. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "glc_exchange.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define OK(x) do { CUDPPResult r_ = (x); if (r_ != CUDPP_SUCCESS) { fprintf(stderr, "%s -> %d (line %d)\n", #x, (int)r_, __LINE__); return 1; } } while (0)

int main(void)
{
    const size_t N = 1 << 16, NBLK = 6, NSUB = N / 4096, STRIDE = (1536 + 1) * NSUB, RW = GLC_RECORD_FIXED_WORDS + NSUB;
    unsigned char *h_in = (unsigned char *)malloc(N * NBLK), *h_back = (unsigned char *)malloc(N * NBLK);
    srand(20240917);
    for (size_t b = 0; b < NBLK; b++)                          /* blocks of different compressibility */
        for (size_t j = 0; j < N; j++) {
            const int r = rand();
            h_in[b * N + j] = (unsigned char)(b == 0 ? 7 : (b & 1) ? (r % (3 + 40 * (int)b)) : ((r & 0xFF) & ((r >> 8) & 0xFF)));
        }

    CUDPPHandle lib = 0, plan = 0;
    CUDPPConfiguration cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.algorithm = CUDPP_COMPRESS; cfg.op = CUDPP_ADD; cfg.datatype = CUDPP_UCHAR; cfg.bucket_mapper = CUDPP_DEFAULT_BUCKET_MAPPER;
    OK(cudppCreate(&lib));
    OK(cudppPlan(lib, &plan, cfg, N, NBLK, 0));

    hipStream_t side;
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    unsigned char *d_in, *d_back;
    int *d_idx, *d_idx2;
    unsigned int *d_hist, *d_off, *d_size, *d_words, *d_compact, *d_rec;
    unsigned int *d_allw, *d_allr, *d_w2, *d_r2, *d_hist2, *d_off2, *d_size2, *d_strided;
    unsigned long long *d_coff;
    CK(hipMalloc((void **)&d_in, N * NBLK));                 CK(hipMalloc((void **)&d_back, N * NBLK));
    CK(hipMalloc((void **)&d_idx, NBLK * 4));                CK(hipMalloc((void **)&d_idx2, NBLK * 4));
    CK(hipMalloc((void **)&d_hist, NBLK * 256 * 4));         CK(hipMalloc((void **)&d_hist2, NBLK * 256 * 4));
    CK(hipMalloc((void **)&d_off, NBLK * NSUB * 4));         CK(hipMalloc((void **)&d_off2, NBLK * NSUB * 4));
    CK(hipMalloc((void **)&d_size, NBLK * 4));               CK(hipMalloc((void **)&d_size2, NBLK * 4));
    CK(hipMalloc((void **)&d_words, NBLK * STRIDE * 4));     CK(hipMalloc((void **)&d_strided, NBLK * STRIDE * 4));
    CK(hipMalloc((void **)&d_compact, NBLK * STRIDE * 4));   CK(hipMalloc((void **)&d_coff, (NBLK + 1) * 8));
    CK(hipMalloc((void **)&d_rec, NBLK * RW * 4));
    CK(hipMalloc((void **)&d_allw, NBLK * STRIDE * 4));      CK(hipMalloc((void **)&d_allr, NBLK * RW * 4));
    CK(hipMalloc((void **)&d_w2, NBLK * STRIDE * 4));        CK(hipMalloc((void **)&d_r2, NBLK * RW * 4));
    CK(hipMemcpy(d_in, h_in, N * NBLK, hipMemcpyHostToDevice));
    CK(hipMemset(d_strided, 0, NBLK * STRIDE * 4));

    /* this rank's share: encode, compact, records */
    OK(glcCompressBatch(plan, d_in, d_idx, d_hist, d_off, NSUB, d_size, d_words, STRIDE, N, NBLK));
    OK(glcCompactStreams(plan, d_words, STRIDE, d_size, NBLK, d_compact, d_coff));
    OK(glcPlanSynchronize(plan));
    OK(glcPackRecords(d_idx, d_hist, d_off, NSUB, d_size, NSUB, NBLK, d_rec, side));

    /* communicator of one rank; an application hands `id` from rank 0 to the others */
    unsigned char id[GLC_UNIQUE_ID_BYTES];
    glcComm_t comm = 0;
    int nranks = -1, rank = -1;
    OK(glcCommGetUniqueId(id));
    OK(glcCommInitRank(&comm, 1, id, 0));
    OK(glcCommInfo(comm, &nranks, &rank));

    unsigned long long counts[2] = {0, 0};
    OK(glcGatherCounts(comm, NBLK, 0, d_coff + NBLK, counts, side));     /* word count read from the device */
    OK(glcGatherStreams(comm, 0, d_compact, d_rec, RW, counts, d_allw, d_allr, side));
    /* ... the root would write d_allw / d_allr out here; the decode side starts from them */
    OK(glcScatterStreams(comm, 0, d_allw, d_allr, RW, counts, d_w2, d_r2, side));
    OK(glcUnpackRecords(d_r2, NSUB, NBLK, d_idx2, d_hist2, d_off2, NSUB, d_size2, side));
    CK(hipStreamSynchronize(side));

    /* offsets of the compacted layout = prefix sums of the sizes that travelled in the records */
    unsigned int h_size[6], h_size2[6];
    unsigned long long h_coff[7];
    CK(hipMemcpy(h_size, d_size, NBLK * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_size2, d_size2, NBLK * 4, hipMemcpyDeviceToHost));
    h_coff[0] = 0;
    for (size_t b = 0; b < NBLK; b++) h_coff[b + 1] = h_coff[b] + h_size2[b];
    CK(hipMemcpy(d_coff, h_coff, (NBLK + 1) * 8, hipMemcpyHostToDevice));
    OK(glcExpandStreams(plan, d_w2, d_coff, NBLK, d_strided, STRIDE, NULL));
    OK(glcDecompressBatch(plan, d_idx2, d_hist2, d_off2, NSUB, d_strided, STRIDE, d_back, N, NBLK));
    OK(glcPlanSynchronize(plan));
    CK(hipMemcpy(h_back, d_back, N * NBLK, hipMemcpyDeviceToHost));

    const int sizes_equal = memcmp(h_size, h_size2, NBLK * 4) == 0;
    const int round_trip = memcmp(h_in, h_back, N * NBLK) == 0;
    printf("nranks=%d rank=%d blocks=%llu words=%llu sum_sizes=%llu sizes_equal=%d round_trip=%d\n", nranks, rank, counts[0], counts[1],
           h_coff[NBLK], sizes_equal, round_trip);
    OK(glcCommDestroy(comm));
    OK(cudppDestroyPlan(plan));
    OK(cudppDestroy(lib));
    if (nranks != 1 || rank != 0 || counts[0] != NBLK || counts[1] != h_coff[NBLK] || !sizes_equal || !round_trip) return 1;
    printf("ALL OK\n");
    return 0;
}
