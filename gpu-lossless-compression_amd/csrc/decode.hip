// decode.hip -- GPU decoder for the cudppCompress stream (placeholder until the
// encode path is parity-green; fails loudly).
#include "glc_device.h"
#include "glc_internal.h"
namespace glc {
hipError_t decode_scratch_alloc(DecodeScratch &, uint32_t, uint32_t) { return hipErrorNotSupported; }
void decode_scratch_free(DecodeScratch &s) { s = DecodeScratch(); }
hipError_t decode_blocks(hipStream_t, const int *, const uint32_t *, const uint32_t *, size_t, const uint32_t *,
                         size_t, uint8_t *, uint32_t, uint32_t, DecodeScratch &, MtfScratch &, uint32_t *)
{ return hipErrorNotSupported; }
}
