/*
 * glc_hd.h -- C ABI of the CUHD-style Huffman-only decoder (BASELINE.json configs[4],
 * SURVEY.md 8(f)3).
 *
 * Replaces, for the same stream shape, the C++ interface of the reference
 *   cuhd::CUHDGPUDecoder::decode(...)            cuhd-icpp/include/cuhd_gpu_decoder.h,
 *                                                 src/cuhd_gpu_decoder.cu:422-523
 *   cuhd::CUHDCodetable / LLHuffmanEncoder        cuhd-icpp/src/cuhd_codetable.cc,
 *                                                 encoder/src/llhuffman_encoder.cc:160-262
 * Stream shape (cuhd_constants.h:15-24, cuhd_input_buffer.cc:20-27): symbols are
 * bytes, codewords are at most 11 bits (length-limited by package-merge), packed
 * MSB-first into 32-bit units, followed by one zero pad unit.  The reference's own
 * code assignment depends on libstdc++ sort/hash order (llhuffman_encoder.cc:48-51,
 * 143-155), so parity for this path is "decoded bytes == original bytes"
 * (demo.cc:176-178); this library fixes the assignment to plain canonical order
 * (by length, then symbol).
 *
 * The decoder does not iterate to self-synchronise (phase 2 of the reference loops with
 * a device->host flag copy per iteration, cuhd_gpu_decoder.cu:459-495).  Every 32-unit
 * span is summarised as a function {start offset 0..10} -> {offset into the next span,
 * symbols decoded}; those 11-entry functions compose associatively, so a scan gives the
 * exact start offset and output index of every span in a fixed number of passes.
 */
#ifndef GLC_HD_H
#define GLC_HD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLC_HD_MAX_LEN 11

/* Length-limited canonical Huffman table from a 256-bin histogram (host).
 * lens[s] = 0 for absent symbols.  Returns the number of coded symbols (0 on error). */
int glcHdBuildTable(const unsigned long long hist[256], unsigned char lens[256], unsigned short codes[256]);

/* Host encoder (the reference's encoder is CPU code too): returns units written incl. the
 * zero pad unit, or 0 if out_units (capacity cap_units) is too small. */
size_t glcHdEncodeHost(const unsigned char *in, size_t nsym, const unsigned char lens[256],
                       const unsigned short codes[256], unsigned int *out_units, size_t cap_units);

/* Device scratch needed by glcHdDecodeDevice for a stream of `nunits` units. */
size_t glcHdWorkBytes(size_t nunits);

/* d_units: nunits 32-bit units in device memory (incl. the pad unit); d_out: nsym bytes.
 * Returns 1 on success.  stream = hipStream_t or NULL.  nunits <= 2^31. */
int glcHdDecodeDevice(const unsigned int *d_units, size_t nunits, const unsigned char lens[256],
                      const unsigned short codes[256], unsigned char *d_out, size_t nsym,
                      void *d_work, void *stream);

/* Same, with the decoder table exactly as the reference holds it: cuhd::CUHDCodetableItemSingle[2048], i.e.
 * {num_bits, symbol} byte pairs indexed by the next 11 stream bits (cuhd_codetable.h:20-23, built by
 * LLHuffmanEncoder::get_decoder_table, llhuffman_encoder.cc:240-262).  This is the call
 * cuhd::CUHDGPUDecoder::decode (cuhd_gpu_decoder.cu:422-431) maps to: units, table, output. */
int glcHdDecodeDeviceTable(const unsigned int *d_units, size_t nunits, const unsigned char *table2048,
                           unsigned char *d_out, size_t nsym, void *d_work, void *stream);

/* Same again, with the table where cuhd::CUHDGPUCodetable keeps it -- in DEVICE memory (its get() pointer).  Nothing is
 * copied and the host is not held: the decode is only enqueued on `stream`.  include/glc_cuhd_adapter.hpp wraps this in
 * the reference's own call signature. */
int glcHdDecodeDeviceTableOnDevice(const unsigned int *d_units, size_t nunits, const unsigned char *d_table2048,
                                   unsigned char *d_out, size_t nsym, void *d_work, void *stream);

/* Measurement aid: live per-kernel profile of the two decode entry points (hipEvent pairs on the call's stream around
 * k_hd_span_functions / the three k_hd_walk launches / k_hd_emit).  glcHdEnableProfile(1) switches it on and resets it;
 * glcHdKernelProfile(i, name, cap, out3) waits for the device and returns 1 with the slot's name and
 * out3 = {sum of launch durations in ms, launches, decoded bytes}, 0 past the last slot. */
int glcHdEnableProfile(int on);
int glcHdKernelProfile(int index, char *name, size_t nameCap, double *out3);

#ifdef __cplusplus
}
#endif
#endif
