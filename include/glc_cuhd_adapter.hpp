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
// ordered by the stream.  A buffer that has to grow (or go) waits for the EVENT its last decode recorded, never for a stream handle.  The subsequence
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

// work buffer of a decode (what cuhd::CUHDGPUDecoderMemory is to the reference): sized for the largest stream seen.  It never
// touches a stream handle it does not own (the caller may have destroyed the stream since): every decode that used the buffer
// RECORDS an event behind itself (used()), and the buffer is released -- to grow, or for good -- only behind that event.
class DecoderMemory {
  public:
    DecoderMemory() = default;
    DecoderMemory(const DecoderMemory &) = delete;
    DecoderMemory &operator=(const DecoderMemory &) = delete;
    ~DecoderMemory()
    {
        wait_idle();
        if (ptr_) (void)hipFree(ptr_);
        if (done_) (void)hipEventDestroy(done_);
    }
    void *reserve(std::size_t units)
    {
        const std::size_t need = glcHdWorkBytes(units);
        if (need > bytes_) {
            wait_idle();                                       // the last decode that used the smaller buffer is through
            if (ptr_) (void)hipFree(ptr_);
            ptr_ = nullptr; bytes_ = 0;
            if (hipMalloc(&ptr_, need) != hipSuccess) throw std::runtime_error("glc::cuhd::DecoderMemory: hipMalloc failed");
            bytes_ = need;
        }
        return ptr_;
    }
    // called right behind a decode enqueued on `stream` (a live stream: the caller has just used it)
    void used(hipStream_t stream)
    {
        if (!done_ && hipEventCreateWithFlags(&done_, hipEventDisableTiming) != hipSuccess) {
            done_ = nullptr;
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);                // no event to be had: the decode is waited for here instead
            pending_ = false;
            return;
        }
        pending_ = hipEventRecord(done_, stream) == hipSuccess;
        if (!pending_) { (void)hipGetLastError(); (void)hipStreamSynchronize(stream); }
    }
  private:
    void wait_idle()
    {
        if (pending_ && done_ && hipEventSynchronize(done_) != hipSuccess) (void)hipGetLastError();
        pending_ = false;
    }
    void *ptr_ = nullptr;
    std::size_t bytes_ = 0;
    hipEvent_t done_ = nullptr;
    bool pending_ = false;
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
        DecoderMemory *mem = nullptr;
        if constexpr (std::is_same<Aux, DecoderMemory>::value) {
            if (aux) mem = aux.get();                          // the caller's decoder memory, as in the reference
        }
        if (!mem) {
            // one per (thread, stream); bounded: a caller that keeps creating streams gets the cache emptied at 16 entries
            // (every buffer is released behind the event of its last decode) instead of one work buffer per handle ever seen.
            // release_work_buffers() empties it on request.
            auto &work = cache();
            if (work.size() >= 16 && work.find(stream) == work.end()) release_work_buffers();
            mem = &work[stream];
        }
        void *w = mem->reserve(input_size);
        const int ok = glcHdDecodeDeviceTableOnDevice(reinterpret_cast<const unsigned int *>(input->get()), input_size,
                                                      reinterpret_cast<const unsigned char *>(table->get()),
                                                      reinterpret_cast<unsigned char *>(output->get()), output_size, w, stream);
        mem->used(stream);
        if (!ok) throw std::runtime_error("glc::cuhd::CUHDGPUDecoder::decode: glcHdDecodeDeviceTableOnDevice failed");
    }
    // frees the calling thread's cached work buffers (each behind the event its last decode recorded; no stream handle is touched)
    static void release_work_buffers() { cache().clear(); }
  private:
    static std::unordered_map<hipStream_t, DecoderMemory> &cache()
    {
        static thread_local std::unordered_map<hipStream_t, DecoderMemory> work;
        return work;
    }
};

}} // namespace glc::cuhd
