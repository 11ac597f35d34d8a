// glc_cuhd_adapter.hpp -- the reference's C++ call for config 5, over the C ABI of glc_hd.h.  Header only.
//
// Replaces  cuhd::CUHDGPUDecoder::decode  (cuhd-icpp/include/cuhd_gpu_decoder.h:22-31, implementation
// cuhd-icpp/src/cuhd_gpu_decoder.cu:422-523): same argument list, same meaning -- `input` holds `input_size` 32-bit
// units in device memory (incl. the zero pad unit, cuhd_input_buffer.cc:20-27), `output` receives `output_size`
// symbols, `table` is the 2048-entry {num_bits, symbol} device table of cuhd::CUHDGPUCodetable
// (cuhd_codetable.h:20-23).  The reference's buffer classes are used AS THEY ARE: anything whose get() returns the
// device pointer fits, so a caller keeps cuhd::CUHDGPUInputBuffer / CUHDGPUOutputBuffer / CUHDGPUCodetable (their
// allocate / cpy_host_to_device members become hipMalloc / hipMemcpy in the caller's port) and only this call changes:
//
//     cuhd::CUHDGPUDecoder::decode(in, n_units, out, n_sym, table, aux, 11, 4, 128);      // reference
//     glc::cuhd::CUHDGPUDecoder::decode(in, n_units, out, n_sym, table, aux, 11, 4, 128); // here
//
// Differences, deliberate: `aux` (the reference's synchronisation scratch, cuhd_gpu_decoder_memory.h) may be any type.
// A `std::shared_ptr<glc::cuhd::DecoderMemory>` IS the work buffer of the decode (one per decoder object, as the
// reference holds its CUHDGPUDecoderMemory; the caller then owns the rule "one decode in flight per DecoderMemory").
// Any other type (or a null pointer) makes the adapter keep its own work buffers, ONE PER (calling thread, stream):
// two decodes enqueued from one thread on different streams never share scratch, and decodes on the same stream are
// ordered by the stream.  A buffer that has to grow waits for its stream before the old allocation is freed.  The subsequence
// size and threads-per-block hints are accepted and ignored (the span functions of hd_decode.hip need no
// self-synchronisation rounds, so there is nothing to tune and no device->host flag copy per round,
// cuhd_gpu_decoder.cu:459-495); codewords longer than 11 bits are refused, as the reference's table format does.
#pragma once
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <unordered_map>
#include <hip/hip_runtime.h>
#include "glc_hd.h"

namespace glc { namespace cuhd {

// work buffer of a decode (what cuhd::CUHDGPUDecoderMemory is to the reference): sized for the largest stream seen
class DecoderMemory {
  public:
    DecoderMemory() = default;
    DecoderMemory(const DecoderMemory &) = delete;
    DecoderMemory &operator=(const DecoderMemory &) = delete;
    ~DecoderMemory() { if (ptr_) (void)hipFree(ptr_); }
    // `stream`: where the decodes that used this buffer were enqueued; waited for before a smaller buffer is released
    void *reserve(std::size_t units, hipStream_t stream = nullptr)
    {
        const std::size_t need = glcHdWorkBytes(units);
        if (need > bytes_) {
            if (ptr_) {
                // a stream handle that has been destroyed since (the per-stream cache below keeps its buffer): nothing of it can
                // still be running, so the buffer is free to go
                const hipError_t e = hipStreamSynchronize(stream);
                if (e == hipErrorInvalidHandle || e == hipErrorContextIsDestroyed || e == hipErrorInvalidResourceHandle) (void)hipGetLastError();
                else if (e != hipSuccess) throw std::runtime_error("glc::cuhd::DecoderMemory: hipStreamSynchronize failed");
                (void)hipFree(ptr_);
            }
            ptr_ = nullptr; bytes_ = 0;
            if (hipMalloc(&ptr_, need) != hipSuccess) throw std::runtime_error("glc::cuhd::DecoderMemory: hipMalloc failed");
            bytes_ = need;
        }
        return ptr_;
    }
  private:
    void *ptr_ = nullptr;
    std::size_t bytes_ = 0;
};

class CUHDGPUDecoder {
  public:
    template <class InputBuffer, class OutputBuffer, class Codetable, class Aux>
    static void decode(std::shared_ptr<InputBuffer> input, std::size_t input_size, std::shared_ptr<OutputBuffer> output,
                       std::size_t output_size, std::shared_ptr<Codetable> table, std::shared_ptr<Aux> aux,
                       std::size_t max_codeword_length, std::size_t /*preferred_subsequence_size*/,
                       std::size_t /*threads_per_block*/, hipStream_t stream = nullptr)
    {
        if (!input || !output || !table) throw std::invalid_argument("glc::cuhd::CUHDGPUDecoder::decode: null buffer");
        if (max_codeword_length > GLC_HD_MAX_LEN) throw std::invalid_argument("glc::cuhd::CUHDGPUDecoder::decode: codewords longer than 11 bits");
        void *w = nullptr;
        if constexpr (std::is_same<Aux, DecoderMemory>::value) {
            if (aux) w = aux->reserve(input_size, stream);     // the caller's decoder memory, as in the reference
        }
        if (!w) {
            // one per (thread, stream); bounded: a caller that keeps creating streams gets the cache emptied at 16 entries
            // (every buffer is released behind a wait for its stream) instead of one work buffer per handle ever seen.
            // release_work_buffers() empties it on request.
            auto &work = cache();
            if (work.size() >= 16 && work.find(stream) == work.end()) release_work_buffers();
            w = work[stream].reserve(input_size, stream);
        }
        const int ok = glcHdDecodeDeviceTableOnDevice(reinterpret_cast<const unsigned int *>(input->get()), input_size,
                                                      reinterpret_cast<const unsigned char *>(table->get()),
                                                      reinterpret_cast<unsigned char *>(output->get()), output_size, w, stream);
        if (!ok) throw std::runtime_error("glc::cuhd::CUHDGPUDecoder::decode: glcHdDecodeDeviceTableOnDevice failed");
    }
    // frees the calling thread's cached work buffers (each behind a wait for the stream it was used on)
    static void release_work_buffers()
    {
        auto &work = cache();
        for (auto &kv : work) {
            const hipError_t e = hipStreamSynchronize(kv.first);
            if (e != hipSuccess) (void)hipGetLastError();      // (a destroyed stream: nothing of it is running)
        }
        work.clear();
    }
  private:
    static std::unordered_map<hipStream_t, DecoderMemory> &cache()
    {
        static thread_local std::unordered_map<hipStream_t, DecoderMemory> work;
        return work;
    }
};

}} // namespace glc::cuhd
