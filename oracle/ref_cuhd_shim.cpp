// ref_cuhd_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" shim (this repo's code) over the PUBLIC interface of the
// reference's CUHD encoder, so that tests/golden/make_cuhd_gold.py can produce config-5 streams with the
// reference's own code.  Compiled by oracle/mk_ref_cuhd_encoder.sh together with the reference's
//   cuhd-icpp/encoder/src/llhuffman_encoder.cc   (LLHuffmanEncoder: package-merge lengths, codes, encoder)
//   cuhd-icpp/src/cuhd_codetable.cc              (CUHDCodetable: the 2048-entry decoder table)
// from where they lie.  The call sequence is the demo's (cuhd-icpp/src/demo.cc.ori:64-91) and the pad unit is
// CUHDInputBuffer's (cuhd_input_buffer.cc:20-27).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "llhuffman_encoder.h"

extern "C" {

// in: nsym symbols.  out_units (capacity cap): the encoded stream incl. the zero pad unit.  lut2048: the
// reference's decoder table as {num_bits, symbol} byte pairs.  lens/codes[256]: the encoder dictionary.
// Returns the number of units (incl. pad), 0 if cap is too small.
size_t ref_cuhd_encode(const uint8_t *in, size_t nsym, uint32_t *out_units, size_t cap, uint8_t *lut2048,
                       uint8_t *lens, uint32_t *codes)
{
    std::vector<SYMBOL_TYPE> buf(in, in + nsym);
    auto lengths = llhuff::LLHuffmanEncoder::get_symbol_lengths(buf.data(), nsym);
    auto enc = llhuff::LLHuffmanEncoder::get_encoder_table(lengths);
    auto dec = llhuff::LLHuffmanEncoder::get_decoder_table(enc);
    const size_t units = enc->compressed_size;
    if (units + 1 > cap) return 0;
    memset(out_units, 0, (units + 1) * sizeof(uint32_t));
    llhuff::LLHuffmanEncoder::encode_memory(out_units, units, buf.data(), nsym, enc);
    memset(lens, 0, 256); memset(codes, 0, 256 * sizeof(uint32_t));
    for (auto &kv : enc->dict) { lens[kv.first] = (uint8_t)kv.second.length; codes[kv.first] = kv.second.codeword; }
    // entries no codeword reaches stay {0, 0}: value-initialised by make_unique (cuhd_codetable.cc)
    const cuhd::CUHDCodetableItemSingle *t = dec->get();
    for (size_t i = 0; i < dec->get_size(); i++) { lut2048[2 * i] = t[i].num_bits; lut2048[2 * i + 1] = t[i].symbol; }
    return units + 1;
}

}
