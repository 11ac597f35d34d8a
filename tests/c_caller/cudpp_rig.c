/* A plain-C caller of include/cudpp.h shaped like the reference's own compress test
 * (cudpp-inpar/apps/cudpp_testrig/test_compress.cpp:672-797): same input generator (glibc
 * srand(95835), rand()%255+1, last byte 0), same plan / buffer sizes / call, outputs copied back with
 * blocking copies.  Prints the known answers of BASELINE.md section 4 so the test can compare them.
 * Built with gcc (no hipcc): the library's boundary is a C ABI. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "cudpp.h"

static uint32_t crc32(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) { c ^= b[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
    return c ^ 0xFFFFFFFFu;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)

int main(void)
{
    const size_t N = 1048576;
    unsigned char *h_in = (unsigned char *)malloc(N);
    srand(95835);                                             /* test_compress.cpp:439 */
    for (size_t j = 0; j < N; j++) h_in[j] = (unsigned char)(rand() % 255 + 1);
    h_in[N - 1] = 0;                                          /* test_compress.cpp:687-692 */

    CUDPPHandle lib = 0, plan = 0;
    CUDPPConfiguration cfg;
    cfg.algorithm = CUDPP_COMPRESS; cfg.op = CUDPP_ADD; cfg.datatype = CUDPP_UCHAR; cfg.options = 0;
    cfg.bucket_mapper = CUDPP_DEFAULT_BUCKET_MAPPER;
    if (cudppCreate(&lib) != CUDPP_SUCCESS) { fprintf(stderr, "cudppCreate failed\n"); return 1; }
    if (cudppPlan(lib, &plan, cfg, N, 1, 0) != CUDPP_SUCCESS) { fprintf(stderr, "cudppPlan failed\n"); return 1; }

    unsigned char *d_in; int *d_idx; unsigned int *d_hist, *d_off, *d_size, *d_comp, *d_histsize;
    const size_t comp_words = (1536 + 1) * 256;               /* test_compress.cpp:717-718 */
    CK(hipMalloc((void **)&d_in, N));            CK(hipMalloc((void **)&d_idx, sizeof(int)));
    CK(hipMalloc((void **)&d_hist, 256 * 4));    CK(hipMalloc((void **)&d_off, 256 * 4));
    CK(hipMalloc((void **)&d_size, 4));          CK(hipMalloc((void **)&d_comp, comp_words * 4));
    CK(hipMalloc((void **)&d_histsize, 4));
    CK(hipMemcpy(d_in, h_in, N, hipMemcpyHostToDevice));

    const CUDPPResult r = cudppCompress(plan, d_in, d_idx, d_histsize, d_hist, d_off, d_size, d_comp, N);
    if (r != CUDPP_SUCCESS) { fprintf(stderr, "cudppCompress -> %d\n", (int)r); return 1; }

    int idx = -1; unsigned int size = 0;
    unsigned int *h_hist = (unsigned int *)malloc(256 * 4), *h_off = (unsigned int *)malloc(256 * 4);
    CK(hipMemcpy(&idx, d_idx, 4, hipMemcpyDeviceToHost));     /* blocking copies = the reference's sync point */
    CK(hipMemcpy(&size, d_size, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_hist, d_hist, 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_off, d_off, 256 * 4, hipMemcpyDeviceToHost));
    if (size == 0 || size > comp_words) { fprintf(stderr, "bad size %u\n", size); return 1; }
    unsigned int *h_comp = (unsigned int *)malloc((size_t)size * 4);
    CK(hipMemcpy(h_comp, d_comp, (size_t)size * 4, hipMemcpyDeviceToHost));

    printf("in_crc=%08x bwt_index=%d size_words=%u crc_words=%08x crc_offsets=%08x crc_hist=%08x\n", crc32(h_in, N), idx, size,
           crc32(h_comp, (size_t)size * 4), crc32(h_off, 256 * 4), crc32(h_hist, 256 * 4));

    if (cudppDestroyPlan(plan) != CUDPP_SUCCESS || cudppDestroy(lib) != CUDPP_SUCCESS) return 1;
    hipFree(d_in); hipFree(d_idx); hipFree(d_hist); hipFree(d_off); hipFree(d_size); hipFree(d_comp); hipFree(d_histsize);
    free(h_in); free(h_hist); free(h_off); free(h_comp);
    return 0;
}
