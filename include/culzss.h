/*
 * culzss.h -- C ABI of the MI355X-native CULZSS path.
 *
 * Drop-in for the GPU wrapper symbols the reference's pthread pipeline binds
 * (cuda-lzss-cluster/culzss.h:84-96, gpu_compress.h:121-133,
 * gpu_decompress.h:109-112; definitions gpu_compress.cu:352-670,
 * gpu_decompress.cu:98-358).  Signatures follow the DEFINITIONS (the three
 * reference headers disagree with each other about
 * decompression_kernel_wrapper; C linkage hides that -- the definition
 * gpu_decompress.cu:247 takes 6 arguments and returns int, and that is what
 * deculzss.c:98 calls).
 *
 * Format constants (gpu_compress.h:62-69): 128-byte window, matches of 3..127
 * bytes, 4096-byte packets, buffers of 1 MiB (main.c:62); any buffer length
 * that is a multiple of 4096 and <= GLC_LZSS_MAX_BUF is accepted here.
 */
#ifndef GLC_CULZSS_H
#define GLC_CULZSS_H

#ifdef __cplusplus
extern "C" {
#endif

#define GLC_LZSS_PACKET   4096
#define GLC_LZSS_MAX_BUF  (16 * 1024 * 1024)

/* ---- compression side (gpu_compress.cu) --------------------------------- */

/* Device 0 + stream pool (reference: 64 streams = 4 ring slots x 16;
 * here one stream per ring slot, 4 slots).  gpu_compress.cu:392-403. */
void initGPU(void);
void resetGPU(void);                 /* hipDeviceReset, gpu_compress.cu:405-408 */
int  streams_in_GPU(void);           /* always 1, gpu_compress.cu:410-413 */
void deleteGPUStreams(void);         /* gpu_compress.cu:383-390 */
void signalExitThreads(void);        /* declared by the reference headers, defined in culzss.c; no-op here */

unsigned char *initGPUmem(int buf_length);     /* hipMalloc,      gpu_compress.cu:352-360 */
unsigned char *initCPUmem(int buf_length);     /* pinned host,    gpu_compress.cu:362-370 */
void deleteGPUmem(unsigned char *mem_d);
void deleteCPUmem(unsigned char *mem_d);

/* Asynchronous: H2D of buffer -> match kernel (2 candidate bytes per input
 * byte: (len,offset) or (1,literal)) -> token selection + flag packing ON THE
 * GPU (the reference does that part on the CPU in aftercomp) -> D2H of the
 * candidate stream into compressed_buffer (kept for ABI fidelity) and of the
 * packed bytes into a pinned staging slot.  `index` selects the ring slot /
 * stream (0..3).  in_d/out_d are device scratch of buf_length / 2*buf_length
 * bytes from initGPUmem.  Returns 1.  gpu_compress.cu:426-460. */
int compression_kernel_wrapper(unsigned char *buffer, int buf_length, unsigned char *compressed_buffer,
                               int compression_type, int wsize, int numthre, int nstreams, int index,
                               unsigned char *in_d, unsigned char *out_d);

/* Blocks until slot `index` has finished.  Returns 1.  gpu_compress.cu:415-424. */
int onestream_finish_GPU(int index);

/* Packed form of the buffer whose candidates are in `bufferout`, written over
 * `buffer`: per packet flag bytes + tokens, then the trailer (big-endian u16
 * packet sizes, u32 buf_length, u16 pad=0).  Returns 1 and *comp_length, or 0
 * ("store raw": packed form outgrew buf_length; buffer left untouched).
 * If `bufferout` is the compressed_buffer of a finished
 * compression_kernel_wrapper call the GPU-packed bytes are used directly;
 * otherwise the candidates are uploaded and packed on the GPU now.
 * gpu_compress.cu:462-670. */
int aftercompression_wrapper(unsigned char *buffer, int buf_length, unsigned char *bufferout, int *comp_length);

/* ---- decompression side (gpu_decompress.cu) ----------------------------- */
unsigned char *deinitGPUmem(int buf_length);
void dedeleteGPUmem(unsigned char *mem_d);
void deinitGPU(void);

/* Synchronous, in place: `buffer` holds buf_length packed bytes (with
 * trailer) on entry and the decoded bytes on return; *decomp_length = original
 * size - pad.  Returns 1 (0 on a malformed trailer).  gpu_decompress.cu:247-358. */
int decompression_kernel_wrapper(unsigned char *buffer, int buf_length, int *decomp_length,
                                 int compression_type, int wsize, int numthre);

/* ---- conveniences named by the north star (new, thin) -------------------- */
/* host in -> host out; out must hold glcLzssPackStride(len) bytes.  Returns 1 packed,
 * 2 stored raw (*out_len == len), 0 error. */
int culzss_compress(const unsigned char *in, int len, unsigned char *out, int *out_len);
int culzss_decompress(const unsigned char *in, int len, unsigned char *out, int *out_len);

/* ---- container + pipeline (the reference's main.c / culzss.c / deculzss.c) -- */
/* File format: u32 nbufs | u32 padding | u32 cumulative_size[nbufs] | payloads
 * (host-endian; main.c:236-245, culzss.c:220,241-243,263-264).  payload i = packed
 * form of 1 MiB buffer i, or the raw 1 MiB when packing took more (size == 1 MiB,
 * deculzss.c:94-95).  Inputs shorter than 1 MiB are refused like main.c:228-232.
 * The last partial buffer is zero-filled (the reference leaves stale ring-slot
 * bytes there, main.c:122-130).  All return 1 on success, 0 on failure.
 * Two more deliberate differences, both format-compatible: (1) a buffer whose packed form incl. trailer
 * is not SMALLER than 1 MiB is stored raw (aftercomp only checks the bytes flushed before the last token
 * group, gpu_compress.cu:492-497, so the reference can produce up to 1 MiB + 535 bytes, written past the end
 * of its slot, and exactly 1 MiB, which deculzss.c:94-95 then reads back as raw); the same rule applies to
 * culzss_compress / aftercompression_wrapper / glcLzssEncodeDevice (size 0 = store raw).  (2) Offsets are
 * u32: a container whose payloads exceed 4 GiB - 1 is refused (return 0) instead of written with wrapped
 * offsets.  The decoder validates what it reads (payload sizes, trailers, packet size tables) and fails
 * rather than reading outside the stream. */
unsigned long long culzss_container_bound(unsigned long long len);
int culzss_container_compress(const unsigned char *in, unsigned long long len, unsigned char *out,
                              unsigned long long out_cap, unsigned long long *out_len);
int culzss_container_decompress(const unsigned char *in, unsigned long long len, unsigned char *out,
                                unsigned long long out_cap, unsigned long long *out_len);
int culzss_compress_file(const char *in_path, const char *out_path);     /* ./main -i in -o out      */
int culzss_decompress_file(const char *in_path, const char *out_path);   /* ./main -d 1 -i in -o out */

/* ---- device-resident batch API (no PCIe in the timed region) ------------- */
/* nbuf buffers of buf_length bytes at d_in (contiguous).  d_cand: 2x input
 * bytes (candidate stream, may be NULL to skip exporting it); d_packed: nbuf
 * slots of glcLzssPackStride(buf_length) bytes; d_sizes[nbuf]: packed length
 * incl. trailer, or 0 = "store raw" (the slot then holds the input bytes).
 * stream = hipStream_t or NULL. */
unsigned long long glcLzssPackStride(int buf_length);
int glcLzssEncodeDevice(const unsigned char *d_in, int buf_length, int nbuf, unsigned char *d_cand,
                        unsigned char *d_packed, int *d_sizes, void *d_work, void *stream);
/* d_work for the call above: glcLzssWorkBytes(buf_length, nbuf) bytes of device memory */
unsigned long long glcLzssWorkBytes(int buf_length, int nbuf);
/* inverse: d_packed slots of glcLzssPackStride(buf_length) bytes -> d_out */
int glcLzssDecodeDevice(const unsigned char *d_packed, const int *d_sizes, int buf_length, int nbuf,
                        unsigned char *d_out, void *stream);
float glcLzssLastKernelMs(void);
/* Measurement aid: live per-kernel profile of the device entry points above (hipEvent pairs on the call's stream around
 * k_lzss_match / the token walk + packer / layout + gather / k_lzss_decode).  glcLzssEnableProfile(1) switches it on and
 * resets it; glcLzssKernelProfile(i, name, cap, out3) waits for the device and returns 1 with the slot's name and
 * out3 = {sum of launch durations in ms, launches, input bytes processed}, 0 past the last slot. */
int glcLzssEnableProfile(int on);
int glcLzssKernelProfile(int index, char *name, size_t nameCap, double *out3);

#ifdef __cplusplus
}
#endif
#endif
