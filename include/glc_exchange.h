/* glc_exchange.h -- the ONE exchange step of the multi-GPU path, as a C ABI over RCCL.
 *
 * The reference is single-GPU (device 0 only: cuda-lzss-cluster/gpu_compress.cu:395, cudpp-inpar sa_app.cu:370), so
 * nothing here replaces a reference function; it is what BASELINE.json's north_star adds on top of the
 * cudppCompress path: "independent 1 MB blocks shard embarrassingly across the 8 GPUs of one node with a single
 * RCCL gather over xGMI for the output bitstream", host code in C/C++ through a thin C-ABI layer.
 *
 * Model: one process per GPU.  Global block g is block g / N of rank g % N; encoding and decoding never cross GPUs.
 * Result collection on a root rank, and its mirror for decoding:
 *
 *   glcGatherCounts    all-gather of {blocks, words} of every rank (16 bytes each; one ncclAllGather), read back
 *                      to the host -- the only host wait of the exchange, and glcGatherCountsBegin / ...End move
 *                      it behind the queueing of the next batch;
 *   glcGatherStreams   gather-v of the per-block RECORDS and of the compacted word streams with grouped
 *                      point-to-point operations: ONE ncclGroup of ncclRecv on the root, one of ncclSend on every
 *                      other rank -- exact lengths, no padding to the largest rank, and on xGMI (point-to-point
 *                      links, no switch) 7 direct links into the root instead of a ring;
 *   glcScatterStreams  the mirror (root -> ranks).
 *
 * A RECORD is everything a decoder needs for a block besides its words, fixed size: recordWords = 258 + nsub
 * 32-bit words {compressedSize, bwtIndex, hist[256], encodeOffset[nsub]} (nsub = blocks of 4096 symbols,
 * 256 at 1 MiB); glcPackRecords / glcUnpackRecords move between that and the output arrays of glcCompressBatch.
 * The words travel compacted (glcCompactStreams / glcExpandStreams of include/cudpp.h).
 *
 * Everything is enqueued on the caller's stream (a hipStream_t cast to void*, NULL = default stream): a caller
 * overlaps the gather of batch i with the encode of batch i + 1 by giving it a side stream (bench.py does).
 * Returns CUDPP_SUCCESS, or CUDPP_ERROR_UNKNOWN when RCCL / HIP report an error, CUDPP_ERROR_INVALID_HANDLE for a
 * NULL communicator, CUDPP_ERROR_ILLEGAL_CONFIGURATION for inconsistent arguments.
 */
#ifndef GLC_EXCHANGE_H
#define GLC_EXCHANGE_H

#include "cudpp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct glcComm_st *glcComm_t;
#define GLC_UNIQUE_ID_BYTES 128        /* = NCCL_UNIQUE_ID_BYTES */
#define GLC_RECORD_FIXED_WORDS 258     /* compressedSize, bwtIndex, hist[256]; encodeOffset[nsub] follows */
#define GLC_COUNT_SLOTS 4              /* count exchanges that may be begun and not yet ended, per communicator */

/* Communicator.  Rank 0 calls glcCommGetUniqueId and hands the 128 bytes to the other ranks by whatever means the
 * application has (MPI, a socket, torch.distributed); every rank then calls glcCommInitRank with the GPU it
 * encodes on current (ncclGetUniqueId / ncclCommInitRank).  glcCommAdopt wraps a communicator the application
 * already owns (an ncclComm_t cast to void*); glcCommDestroy does not destroy an adopted one. */
CUDPPResult glcCommGetUniqueId(void *id128);
CUDPPResult glcCommInitRank(glcComm_t *comm, int nranks, const void *id128, int rank);
CUDPPResult glcCommAdopt(glcComm_t *comm, void *ncclComm);
CUDPPResult glcCommDestroy(glcComm_t comm);
CUDPPResult glcCommInfo(glcComm_t comm, int *nranks, int *rank);

/* records <-> the output arrays of glcCompressBatch (device pointers; d_records holds numBlocks * (258 + nsub) words) */
CUDPPResult glcPackRecords(const int *d_bwtIndex, const unsigned int *d_hist, const unsigned int *d_encodeOffset,
                           size_t offsetStride, const unsigned int *d_compressedSize, size_t nsub, size_t numBlocks,
                           unsigned int *d_records, void *hipStream);
CUDPPResult glcUnpackRecords(const unsigned int *d_records, size_t nsub, size_t numBlocks, int *d_bwtIndex,
                             unsigned int *d_hist, unsigned int *d_encodeOffset, size_t offsetStride,
                             unsigned int *d_compressedSize, void *hipStream);

/* h_counts[2 r] = blocks, h_counts[2 r + 1] = words of rank r (host array of 2 * nranks entries).  d_numWords: a
 * device word count (e.g. the last entry of glcCompactStreams' offsets) or NULL with the count in numWords.
 * Waits for the exchange (not for the rest of the stream): the counts size the point-to-point operations that
 * follow.  = glcGatherCountsBegin + glcGatherCountsEnd. */
CUDPPResult glcGatherCounts(glcComm_t comm, unsigned long long numBlocks, unsigned long long numWords,
                            const unsigned long long *d_numWords, unsigned long long *h_counts, void *hipStream);

/* The same in two halves, so that the encoding thread never waits: Begin ENQUEUES the count exchange of a batch on
 * the stream (set counts, ncclAllGather, copy to pinned host memory, an event) and returns a ticket; the caller goes on
 * queueing the next batch; End waits for that ticket's event only and hands out the counts (Ready polls instead).
 * Up to GLC_COUNT_SLOTS tickets may be outstanding per communicator (Begin returns
 * CUDPP_ERROR_INSUFFICIENT_RESOURCES beyond that); every rank must Begin its exchanges in the same order. */
CUDPPResult glcGatherCountsBegin(glcComm_t comm, unsigned long long numBlocks, unsigned long long numWords,
                                 const unsigned long long *d_numWords, int *ticket, void *hipStream);
CUDPPResult glcGatherCountsReady(glcComm_t comm, int ticket, int *ready);
CUDPPResult glcGatherCountsEnd(glcComm_t comm, int ticket, unsigned long long *h_counts);

/* Every rank: d_words (its compacted streams, h_counts[2 rank + 1] words) and d_records (h_counts[2 rank] records).
 * Root only: d_allWords / d_allRecords receive the ranks' data back to back in rank order (rank r's words start at
 * sum of words of the ranks below it); ignored elsewhere (may be NULL).  Enqueues only. */
CUDPPResult glcGatherStreams(glcComm_t comm, int root, const unsigned int *d_words, const unsigned int *d_records,
                             size_t recordWords, const unsigned long long *h_counts, unsigned int *d_allWords,
                             unsigned int *d_allRecords, void *hipStream);

/* The mirror: the root sends rank r its words and records out of the rank-ordered arrays.  h_counts must be valid on
 * every rank (glcGatherCounts of the gather, or broadcast by the application).  Enqueues only. */
CUDPPResult glcScatterStreams(glcComm_t comm, int root, const unsigned int *d_allWords, const unsigned int *d_allRecords,
                              size_t recordWords, const unsigned long long *h_counts, unsigned int *d_words,
                              unsigned int *d_records, void *hipStream);

#ifdef __cplusplus
}
#endif
#endif /* GLC_EXCHANGE_H */
