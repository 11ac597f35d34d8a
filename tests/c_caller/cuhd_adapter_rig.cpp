// A C++ caller shaped like the reference's demo (cuhd-icpp/demo.cc:150-178): buffers with get(), one decode call with
// the reference's argument list -- here through include/glc_cuhd_adapter.hpp.  Reads a stream written by the
// reference's own encoder (units, 2048-entry table, expected symbols as raw files), decodes it on the GPU and compares.
//   usage: cuhd_adapter_rig units.bin table.bin symbols.bin
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#include <hip/hip_runtime.h>
#include "glc_cuhd_adapter.hpp"

template <class T> struct DeviceBuffer {                       // stands for cuhd::CUHDGPUMemoryBuffer<T>
    explicit DeviceBuffer(size_t n) : n_(n) { if (hipMalloc((void **)&p_, n * sizeof(T) + 16) != hipSuccess) p_ = nullptr; }
    ~DeviceBuffer() { if (p_) (void)hipFree(p_); }
    T *get() { return p_; }
    size_t n_;
    T *p_ = nullptr;
};
struct TableItem { uint8_t num_bits, symbol; };               // cuhd::CUHDCodetableItemSingle

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n);
    if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const auto units = slurp(argv[1]), table = slurp(argv[2]), want = slurp(argv[3]);
    if (units.empty() || table.size() != 4096 || want.empty()) { printf("bad input files\n"); return 2; }
    const size_t nunits = units.size() / 4, nsym = want.size();
    auto in = std::make_shared<DeviceBuffer<uint32_t>>(nunits);
    auto out = std::make_shared<DeviceBuffer<uint8_t>>(nsym);
    auto tab = std::make_shared<DeviceBuffer<TableItem>>(2048);
    std::shared_ptr<glc::cuhd::DecoderMemory> aux;            // null: the adapter keeps its own work buffer
    if (!in->get() || !out->get() || !tab->get()) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemcpy(in->get(), units.data(), units.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(tab->get(), table.data(), table.size(), hipMemcpyHostToDevice);
    (void)hipMemset(out->get(), 0, nsym);
    for (int rep = 0; rep < 2; rep++)                          // twice: the cached work buffer is reused
        glc::cuhd::CUHDGPUDecoder::decode(in, nunits, out, nsym, tab, aux, 11, 4, 128);
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    std::vector<uint8_t> got(nsym);
    (void)hipMemcpy(got.data(), out->get(), nsym, hipMemcpyDeviceToHost);
    bool ok = memcmp(got.data(), want.data(), nsym) == 0;
    // two decodes in flight from ONE thread on two streams (each gets its own scratch), and one with the caller's own
    // DecoderMemory as `aux`, the way the reference passes its CUHDGPUDecoderMemory
    hipStream_t st[2];
    std::shared_ptr<DeviceBuffer<uint8_t>> outs[3];
    for (int i = 0; i < 3; i++) { outs[i] = std::make_shared<DeviceBuffer<uint8_t>>(nsym); (void)hipMemset(outs[i]->get(), 0, nsym); }
    for (int i = 0; i < 2; i++) if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) { printf("stream\n"); return 1; }
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 4; rep++)
        for (int i = 0; i < 2; i++)
            glc::cuhd::CUHDGPUDecoder::decode(in, nunits, outs[i], nsym, tab, aux, 11, 4, 128, st[i]);
    auto mine = std::make_shared<glc::cuhd::DecoderMemory>();
    glc::cuhd::CUHDGPUDecoder::decode(in, nunits, outs[2], nsym, tab, mine, 11, 4, 128);
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    for (int i = 0; i < 3; i++) {
        (void)hipMemcpy(got.data(), outs[i]->get(), nsym, hipMemcpyDeviceToHost);
        ok = ok && memcmp(got.data(), want.data(), nsym) == 0;
    }
    for (int i = 0; i < 2; i++) (void)hipStreamDestroy(st[i]);
    printf("units=%zu symbols=%zu decoded_equals_original=%d\n", nunits, nsym, ok ? 1 : 0);
    return ok ? 0 : 1;
}
