/*
 * cudpp.h -- C ABI of the MI355X-native cudppCompress path.
 *
 * Drop-in for the subset of the reference's public header
 * (cudpp-inpar/include/cudpp.h) that is on the compression hot path:
 *
 *   cudppCreate / cudppDestroy                 cudpp.h:199-205, cudpp_manager.cpp:40-63
 *   cudppPlan / cudppDestroyPlan               cudpp.h:209-217, cudpp_plan.cpp:81-292
 *   cudppCompress                              cudpp.h:327-335, cudpp.cpp:764-806
 *   cudppBurrowsWheelerTransform               cudpp.h:339-343, cudpp.cpp:829-864
 *   cudppMoveToFrontTransform                  cudpp.h:347-350, cudpp.cpp:886-919
 *   cudppSuffixArray                           cudpp.h:363-366, cudpp.cpp:1000-1034
 *
 * Enumerator VALUES, the CUDPPConfiguration layout, CUDPPHandle = size_t and
 * CUDPP_INVALID_HANDLE are identical to the reference, so a caller compiled
 * against the reference header links against this library unchanged.  All data
 * pointers are DEVICE pointers (HIP) owned by the caller; a plan owns its
 * scratch.  Algorithms the reference header lists but that are not on this
 * path (scan, sort, rand, ...) keep their enumerators (values matter) but
 * cudppPlan() answers CUDPP_ERROR_ILLEGAL_CONFIGURATION for them.
 *
 * Extensions (new, prefixed glc): batched entry points that run `rows`
 * independent blocks per call -- see the end of this file.
 */
#ifndef GLC_CUDPP_H
#define GLC_CUDPP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* result codes -- values as reference cudpp.h:32-49 */
enum CUDPPResult
{
    CUDPP_SUCCESS = 0,
    CUDPP_ERROR_INVALID_HANDLE,
    CUDPP_ERROR_ILLEGAL_CONFIGURATION,
    CUDPP_ERROR_INVALID_PLAN,
    CUDPP_ERROR_INSUFFICIENT_RESOURCES,
    CUDPP_ERROR_UNKNOWN = 9999
};

/* option bits -- reference cudpp.h:62-84 */
enum CUDPPOption
{
    CUDPP_OPTION_FORWARD         = 0x1,
    CUDPP_OPTION_BACKWARD        = 0x2,
    CUDPP_OPTION_EXCLUSIVE       = 0x4,
    CUDPP_OPTION_INCLUSIVE       = 0x8,
    CUDPP_OPTION_CTA_LOCAL       = 0x10,
    CUDPP_OPTION_KEYS_ONLY       = 0x20,
    CUDPP_OPTION_KEY_VALUE_PAIRS = 0x40
};

/* datatypes -- reference cudpp.h:90-103 (CUDPP_UCHAR == 1) */
enum CUDPPDatatype
{
    CUDPP_CHAR, CUDPP_UCHAR, CUDPP_SHORT, CUDPP_USHORT, CUDPP_INT, CUDPP_UINT,
    CUDPP_FLOAT, CUDPP_DOUBLE, CUDPP_LONGLONG, CUDPP_ULONGLONG, CUDPP_DATATYPE_INVALID
};

/* operators -- reference cudpp.h:109-116 */
enum CUDPPOperator
{
    CUDPP_ADD, CUDPP_MULTIPLY, CUDPP_MIN, CUDPP_MAX, CUDPP_OPERATOR_INVALID
};

/* algorithms -- reference cudpp.h:128-147 (COMPRESS=10, BWT=12, MTF=13, SA=14) */
enum CUDPPAlgorithm
{
    CUDPP_SCAN, CUDPP_SEGMENTED_SCAN, CUDPP_COMPACT, CUDPP_REDUCE, CUDPP_SORT_RADIX,
    CUDPP_SORT_MERGE, CUDPP_SORT_STRING, CUDPP_SPMVMULT, CUDPP_RAND_MD5, CUDPP_TRIDIAGONAL,
    CUDPP_COMPRESS, CUDPP_LISTRANK, CUDPP_BWT, CUDPP_MTF, CUDPP_SA, CUDPP_MULTISPLIT,
    CUDPP_ALGORITHM_INVALID
};

/* bucket mapper (multisplit only; kept for struct layout) -- cudpp.h:153-159 */
enum CUDPPBucketMapper
{
    CUDPP_LSB_BUCKET_MAPPER, CUDPP_MSB_BUCKET_MAPPER, CUDPP_DEFAULT_BUCKET_MAPPER,
    CUDPP_CUSTOM_BUCKET_MAPPER
};

/* plan configuration -- same five fields, same order: cudpp.h:171-178 */
struct CUDPPConfiguration
{
    enum CUDPPAlgorithm    algorithm;
    enum CUDPPOperator     op;
    enum CUDPPDatatype     datatype;
    unsigned int           options;
    enum CUDPPBucketMapper bucket_mapper;
};

#ifndef __cplusplus
typedef enum CUDPPResult CUDPPResult;
typedef enum CUDPPOption CUDPPOption;
typedef enum CUDPPDatatype CUDPPDatatype;
typedef enum CUDPPOperator CUDPPOperator;
typedef enum CUDPPAlgorithm CUDPPAlgorithm;
typedef enum CUDPPBucketMapper CUDPPBucketMapper;
typedef struct CUDPPConfiguration CUDPPConfiguration;
#endif

#define CUDPP_INVALID_HANDLE 0xC0DABAD1      /* cudpp.h:182 */
typedef size_t CUDPPHandle;                   /* cudpp.h:183 */

/* Library object.  One per host thread / device context by convention
 * (cudpp_manager.cpp:30-34).  The device is whatever is current at the call. */
CUDPPResult cudppCreate(CUDPPHandle *theCudpp);
CUDPPResult cudppDestroy(CUDPPHandle theCudpp);

/* Plan for `n` elements at most.  `rows` = how many independent blocks of up
 * to n elements the plan can run per batched call (reference callers pass 1,
 * test_compress.cpp:405); `rowPitch` is ignored as in the reference.
 * Errors: ILLEGAL_CONFIGURATION (bad option combination, algorithm not on this
 * path, n == 0 or n > 1048576 for COMPRESS/BWT), INSUFFICIENT_RESOURCES
 * (device allocation failed), INVALID_HANDLE (library handle is 0). */
CUDPPResult cudppPlan(const CUDPPHandle cudppHandle, CUDPPHandle *planHandle,
                      CUDPPConfiguration config, size_t n, size_t rows, size_t rowPitch);
CUDPPResult cudppDestroyPlan(CUDPPHandle plan);

/* BWT -> MTF -> Huffman of one block (numElements <= plan n, <= 1 MiB).
 *   d_bwtIndex[1], d_hist[256], d_encodeOffset[ceil(n/4096)] (256 at 1 MiB),
 *   d_compressedSize[1] (words), d_compressed[>= (1536+1)*ceil(n/4096)] words,
 *   d_histSize is ignored (may be NULL) exactly as in the reference.
 * Stream layout: SURVEY.md App. A.  Errors: INVALID_HANDLE (plan 0),
 * INVALID_PLAN (plan is not a COMPRESS plan), ILLEGAL_CONFIGURATION (datatype
 * != UCHAR, numElements out of range), UNKNOWN (a HIP call failed; the
 * reference exit()s instead, cuda_util.h:13-21). */
CUDPPResult cudppCompress(CUDPPHandle planHandle, unsigned char *d_uncompressed,
                          int *d_bwtIndex, unsigned int *d_histSize, unsigned int *d_hist,
                          unsigned int *d_encodeOffset, unsigned int *d_compressedSize,
                          unsigned int *d_compressed, size_t numElements);

CUDPPResult cudppBurrowsWheelerTransform(CUDPPHandle planHandle, unsigned char *d_in,
                                         unsigned char *d_out, int *d_index, size_t numElements);

CUDPPResult cudppMoveToFrontTransform(CUDPPHandle planHandle, unsigned char *d_in,
                                      unsigned char *d_out, size_t numElements);

/* d_keys_sa must hold numElements+1 words; the suffix array (0-based, sentinel
 * row dropped) is written at d_keys_sa[1..numElements] (test_sa.cpp:112-113,161);
 * d_keys_sa[0] receives numElements (the position of the virtual sentinel). */
CUDPPResult cudppSuffixArray(CUDPPHandle planHandle, unsigned char *d_str,
                             unsigned int *d_keys_sa, size_t numElements);

/* ------------------------------------------------------------------------ */
/* Extensions (not in the reference): batched calls and stream control.     */
/* A batched call runs `numBlocks` (<= plan rows) independent blocks of      */
/* `numElements` each; block b reads d_uncompressed + b*numElements and      */
/* writes d_bwtIndex[b], d_hist[b*256..], d_encodeOffset[b*offsetStride..],  */
/* d_compressedSize[b], d_compressed + b*compressedStrideWords.  Each        */
/* block's outputs are bit-identical to a single cudppCompress() call.       */
/* ------------------------------------------------------------------------ */
CUDPPResult glcCompressBatch(CUDPPHandle planHandle, const unsigned char *d_uncompressed,
                             int *d_bwtIndex, unsigned int *d_hist, unsigned int *d_encodeOffset,
                             size_t offsetStride, unsigned int *d_compressedSize,
                             unsigned int *d_compressed, size_t compressedStrideWords,
                             size_t numElements, size_t numBlocks);

/* The same encode with the COMPACT output layout: block b's words are written at d_compact + d_blockOffsets[b], the
 * blocks back to back (d_blockOffsets: numBlocks + 1 entries, the last one = the end; d_compressedSize[b] =
 * d_blockOffsets[b + 1] - d_blockOffsets[b]).  The first block starts at *d_startOffset (a DEVICE value, read in
 * stream order; NULL = 0): pass the previous batch's d_blockOffsets + numBlocks and a sequence of batches fills one
 * contiguous array with no host involvement -- the layout glcCompactStreams produces from the strided one, without
 * the copy pass.  ORDERING: the start offset is read by work queued on THIS plan (its stream, or its internal stream
 * when stage pipelining is on), so chaining is ordered only between batches that go through the SAME plan; a batch
 * whose start offset was written by ANOTHER plan (or host thread) must be queued after glcPlanSynchronize of that
 * plan (or after an event the caller records behind it once it has been synchronised).  With glcPlanSetPipelining on,
 * d_blockOffsets, d_compressedSize and d_compact are complete only after glcPlanSynchronize, as for every output of a
 * pipelined plan.  capacityWords = size of the d_compact array; streams that would pass it are not written and
 * glcPlanSynchronize reports CUDPP_ERROR_UNKNOWN.  Everything else as glcCompressBatch. */
CUDPPResult glcCompressBatchCompact(CUDPPHandle planHandle, const unsigned char *d_uncompressed,
                                    int *d_bwtIndex, unsigned int *d_hist, unsigned int *d_encodeOffset,
                                    size_t offsetStride, unsigned int *d_compressedSize,
                                    unsigned int *d_compact, size_t capacityWords,
                                    unsigned long long *d_blockOffsets, const unsigned long long *d_startOffset,
                                    size_t numElements, size_t numBlocks);

CUDPPResult glcBwtBatch(CUDPPHandle planHandle, const unsigned char *d_in, unsigned char *d_out,
                        int *d_index, size_t numElements, size_t numBlocks);

CUDPPResult glcMtfBatch(CUDPPHandle planHandle, const unsigned char *d_in, unsigned char *d_out,
                        size_t numElements, size_t numBlocks);

/* Inverse of glcCompressBatch (the reference has no GPU decoder; semantics =
 * the gold decoder of test_compress.cpp:192-311 with the sentinel-aware
 * inverse BWT of SURVEY.md 8(f)1).  Plan must be a COMPRESS plan. */
CUDPPResult glcDecompressBatch(CUDPPHandle planHandle, const int *d_bwtIndex,
                               const unsigned int *d_hist, const unsigned int *d_encodeOffset,
                               size_t offsetStride, const unsigned int *d_compressed,
                               size_t compressedStrideWords, unsigned char *d_out,
                               size_t numElements, size_t numBlocks);

/* glcDecompressBatch reading the compact layout: block b's words are d_compact[d_blockOffsets[b] ..
 * d_blockOffsets[b + 1]) (numBlocks + 1 entries; absolute word offsets into d_compact, which holds compactWords
 * words).  A range that does not ascend or leaves the array is treated as empty and reported. */
CUDPPResult glcDecompressBatchCompact(CUDPPHandle planHandle, const int *d_bwtIndex,
                                      const unsigned int *d_hist, const unsigned int *d_encodeOffset,
                                      size_t offsetStride, const unsigned int *d_compact, size_t compactWords,
                                      const unsigned long long *d_blockOffsets, unsigned char *d_out,
                                      size_t numElements, size_t numBlocks);

/* Run a plan's work on `hipStream` (a hipStream_t cast to void*; NULL = the
 * default stream, which is what the reference uses). */
CUDPPResult glcPlanSetStream(CUDPPHandle planHandle, void *hipStream);

/* Blocks until everything queued by the plan has finished; returns
 * CUDPP_ERROR_UNKNOWN if a kernel faulted or a block overflowed the 1536-word
 * per-4096-symbol capacity of the reference format. */
CUDPPResult glcPlanSynchronize(CUDPPHandle planHandle);

/* Stage pipelining across batched calls (off by default).  When on, the second half of a call -- the
 * MTF + Huffman stages of glcCompressBatch / cudppCompress, the inverse BWT of glcDecompressBatch -- runs
 * on an internal stream, where it overlaps the first half (suffix sort; Huffman decode + inverse MTF) of
 * the NEXT call on the plan's stream.  Inputs keep their meaning: they are read in stream order on the
 * plan's stream.  OUTPUT buffers written by the second half (d_hist, d_encodeOffset, d_compressedSize,
 * d_compressed; the decoder's d_out) are complete after glcPlanSynchronize, glcCompactStreams, turning
 * pipelining off, or a device synchronize -- not merely "later on the plan's stream". */
CUDPPResult glcPlanSetPipelining(CUDPPHandle planHandle, int on);

/* Per-stage device time of the plan's last batched compress, in milliseconds,
 * measured with hipEvents on the plan's stream: [0]=BWT(suffix sort+gather),
 * [1]=MTF, [2]=Huffman, [3]=total.  Enables timing when `enable` != 0. */
CUDPPResult glcPlanEnableTiming(CUDPPHandle planHandle, int enable);
CUDPPResult glcPlanLastTiming(CUDPPHandle planHandle, float *ms4);

/* With glcPlanEnableTiming(plan, 3) the library also brackets every launch of its main kernels with hipEvents
 * on the stream they are launched on (encoder and decoder).  glcPlanKernelProfileEx(plan, i, name, cap, out3)
 * returns for slot i = 0, 1, ... (CUDPP_ERROR_ILLEGAL_CONFIGURATION past the last) the kernel's name and
 * out3 = {sum of launch durations in ms, number of launches, input bytes those launches processed};
 * glcPlanKernelProfile returns the slot with the largest total and resets the accumulators. */
CUDPPResult glcPlanKernelProfileEx(CUDPPHandle planHandle, int index, char *name, size_t nameCap, double *out3);
CUDPPResult glcPlanKernelProfile(CUDPPHandle planHandle, double *out3);
/* launches the live profile could not account for since it was switched on: out2[0] = not bracketed (more than 4096
 * launches between two reads), out2[1] = bracketed but unreadable.  Both getters above wait for the plan's streams. */
CUDPPResult glcPlanKernelProfileLost(CUDPPHandle planHandle, unsigned long long *out2);

/* The Huffman half of cudppCompress on caller-supplied symbols (what the pipeline feeds with the MTF output):
 * histogram, tree + codes, bit packer and offsets -- huffman_build_histogram_kernel / huffman_build_tree_kernel /
 * huffman_kernel_en / huffman_datapack_kernel, compress_kernel.cuh:2037-2750.  Plan: CUDPP_COMPRESS.  Outputs as
 * glcCompressBatch.  Lets tests pin the tree's tie-break rule on chosen histograms. */
CUDPPResult glcHuffmanEncodeBatch(CUDPPHandle planHandle, const unsigned char *d_symbols, unsigned int *d_hist,
                                  unsigned int *d_encodeOffset, size_t offsetStride, unsigned int *d_compressedSize,
                                  unsigned int *d_compressed, size_t compressedStrideWords, size_t numElements,
                                  size_t numBlocks);

/* Suffix sorter selection (plans of CUDPP_COMPRESS / CUDPP_BWT / CUDPP_SA).  0 (default): three tiers -- the
 * bucket sorter (one bucketing pass + in-LDS sort; i.i.d.-like data), for the blocks it flags the sample sorter
 * (buckets cut at sampled splitter suffixes, runs of equal codes refined from the text; text, logs), and for
 * what that flags (repeats deeper than ~500 symbols) the general sorter; 1: general sorter only; 2: general
 * sorter, prefix doubling from the first refinement round; 3: bucket sorter, then general sorter; 4: sample
 * sorter first, for a caller that knows its data is text-like (saves the bucket sorter's wasted attempt); 5: as 0, but
 * a block the sample sorter gives up on for depth always goes to the general sorter from scratch (0: when a call has four
 * or more such blocks, the sample sorter finishes what it can of them and the doubling rounds resume from there); 6: as 0,
 * resuming for a single such block too; 7: as 0 without the periodic tier (a block the sample sorter gives up on that is ONE
 * periodic stretch -- a page repeated to the end, a short pattern, one byte up to a different last one -- gets its suffix
 * array in closed form from the sorted rotations of its period: bwt_periodic.hip; 5 switches that off too).  All
 * produce the same bytes (the suffix array of a block is unique); the knob exists for tests and A/B timing. */
CUDPPResult glcPlanSetSorter(CUDPPHandle planHandle, int mode);
/* number of blocks of the plan's last call the bucket sorter gave up on (0 for i.i.d.-like data) */
CUDPPResult glcPlanLastSortStats(CUDPPHandle planHandle, unsigned int *flaggedBlocks);
/* out2[0] = the same count, out2[1] = how many of those the sample sorter gave up on too (general sorter) */
CUDPPResult glcPlanLastSortStatsEx(CUDPPHandle planHandle, unsigned int *out2);
/* out[0] = how many blocks of the last call the sample sorter finished in its SECOND attempt (a bucket past its slot with the
 * first samples; other samples are drawn once before the block would go to the general sorter) */
CUDPPResult glcPlanLastSortRetries(CUDPPHandle planHandle, unsigned int *out);
/* out[0] = how many of the blocks the sample sorter gave up on FOR DEPTH ALONE (out2[1] above counts them) were finished by
 * prefix doubling RESUMED from its order -- the sample sorter run once more in a form that leaves suffixes agreeing in more
 * than 64 symbols (SS_TOL_CAP; 128 until round 6) as they come, doubling from that depth over the rows that still tie -- instead of from scratch */
CUDPPResult glcPlanLastSortResumed(CUDPPHandle planHandle, unsigned int *out);
/* out[0] = how many of the blocks the sample sorter gave up on were finished by the periodic tier (see glcPlanSetSorter) */
CUDPPResult glcPlanLastSortPeriodic(CUDPPHandle planHandle, unsigned int *out);
/* out2[0] = 1 if the plan's last call went straight to the sample sorter (sorter mode 4, or adaptively: a call of up to 4 blocks
   behind eight calls in a row whose every block the text-likeness probe flagged -- the reference's callers hand over one block per
   call, test_compress.cpp:744, and a text block's call spent a fifth of its time on launches that found the block flagged);
   out2[1] = the length of that streak.  A wrong guess costs time, never correctness. */
CUDPPResult glcPlanLastSortSkipped(CUDPPHandle planHandle, unsigned int *out2);
/* diagnostics (tests, tools/exp): per-block give-up flags of the last sort (bucket sorter: 1 bucket overflow / text-like, 2 deep,
 * 4 work list full; sample sorter: 1 bucket overflow, 2 deep; 3 also marks a deep block whose beginning is periodic for an eighth of the block or more: not worth the tolerant pass), numBlocks entries each, either pointer may be NULL; and the
 * 512 bucket fills of one block as the last bucketing pass left them.  Both wait for the plan's stream. */
CUDPPResult glcPlanDebugSortFlags(CUDPPHandle planHandle, unsigned int *out_fs, unsigned int *out_ss, size_t numBlocks);
CUDPPResult glcPlanDebugBucketFill(CUDPPHandle planHandle, size_t block, unsigned int *out512);

/* Result collection: packs the strided per-block streams of a batched compress
 * back to back.  d_outOffsets has numBlocks+1 entries (word offsets; the last is
 * the total).  d_out must hold sum(d_compressedSize) words. */
CUDPPResult glcCompactStreams(CUDPPHandle planHandle, const unsigned int *d_compressed,
                              size_t compressedStrideWords, const unsigned int *d_compressedSize,
                              size_t numBlocks, unsigned int *d_out, unsigned long long *d_outOffsets);

/* The mirror of glcCompactStreams (decode side of the multi-GPU exchange): block b's words move from
 * d_in[d_inOffsets[b] .. d_inOffsets[b+1]) back to the strided layout glcDecompressBatch reads;
 * d_compressedSize (optional) receives the word counts.  d_in holds d_inOffsets[numBlocks] words; offsets that do
 * not ascend or pass that total are reported (glcPlanSynchronize -> CUDPP_ERROR_UNKNOWN) and the block is left empty. */
CUDPPResult glcExpandStreams(CUDPPHandle planHandle, const unsigned int *d_in, const unsigned long long *d_inOffsets,
                             size_t numBlocks, unsigned int *d_compressed, size_t compressedStrideWords,
                             unsigned int *d_compressedSize);

/* Measurement aid (not part of the reference): launches a trivial 16-byte-per-lane streaming-read kernel over
 * d_buf `iters` times on `stream` and returns the average launch duration in *ms -- the read ceiling of this
 * box, measured in the same run as the numbers it is quoted beside.  Returns 1 on success. */
int glcProbeStreamRead(const void *d_buf, size_t bytes, int iters, float *ms, void *stream);
/* Measurement aid: the config-2 workload of SURVEY.md 8(d).  Bytes [first_byte, first_byte + bytes) (multiples of 16) of
 * the Zipf(1.0) byte stream defined by a counter-based Philox4x32-10 generator: byte i = number of thresholds
 * d_thr255[s] <= word (i & 3) of Philox(counter = i / 4, key = seed).  tests/datagen.py zipf_philox_bytes is the host
 * twin (same bytes).  Enqueues on the stream; returns 1 on success. */
int glcGenZipfPhilox(void *d_out, size_t bytes, unsigned long long first_byte, unsigned int seed,
                     const unsigned int *d_thr255, void *stream);
/* The config-4 workload: bytes [first_byte, first_byte + bytes) (multiples of 16) of a float32 ~ N(0, 1) stream as raw
 * little-endian bytes -- value i = ((sum of the four Philox4x32-10 words of counter i, each >> 10) * 2^-22 - 2) * sqrt(3), the
 * same float32 bits as tests/datagen.py float_philox_bytes on the host. */
int glcGenFloatPhilox(void *d_out, size_t bytes, unsigned long long first_byte, unsigned int seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GLC_CUDPP_H */
