// culzss_internal.h -- constants + launchers shared by culzss.hip and culzss_api.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace glc {

constexpr int LZ_WIN   = 128;     // WINDOW_SIZE  (cuda-lzss-cluster/gpu_compress.h:62)
constexpr int LZ_MAXC  = 128;     // MAX_CODED    (gpu_compress.h:66)
constexpr int LZ_PCKT  = 4096;    // PCKTSIZE     (gpu_compress.h:69)
constexpr int LZ_STAGE = 4608;    // worst packed packet: 4096 literals + 512 flag bytes

// bytes reserved per buffer for the packed form: data + last group slack + trailer
inline size_t lzss_pack_stride(int buf_length)
{
    return ((size_t)buf_length + (size_t)buf_length / 2048 + 32 + 255) & ~(size_t)255;
}

struct KernelProf;
enum { LZP_MATCH = 0, LZP_PACK, LZP_GATHER, LZP_DECODE, LZP_NSLOT };
KernelProf &lzss_prof();             // process-wide live profile of the launchers below (glcLzssEnableProfile)
size_t     lzss_work_bytes(int buf_length, int nbuf);
hipError_t lzss_encode(hipStream_t st, const uint8_t *d_in, int buf_length, int nbuf, uint8_t *d_cand,
                       uint8_t *d_packed, int *d_sizes, void *d_work);
hipError_t lzss_pack(hipStream_t st, const uint8_t *d_cand, int buf_length, int nbuf, uint8_t *d_packed,
                     int *d_sizes, void *d_work, const uint8_t *d_raw_in = nullptr);
// device -> pinned (device-mapped) host memory by a kernel on the stream; both pointers 16-byte aligned
hipError_t lzss_copy_to_host(hipStream_t st, const void *d_src, void *h_dst, size_t bytes);
// d_err (optional): device int, bit 0 is raised when a stream is malformed (trailer / packet sizes out of range)
hipError_t lzss_decode(hipStream_t st, const uint8_t *d_packed, const int *d_sizes, int buf_length, int nbuf,
                       uint8_t *d_out, int *d_err = nullptr);

} // namespace glc
