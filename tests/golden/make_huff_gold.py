#!/usr/bin/env python3
"""Generates tests/golden/ref_huff_gold.npz -- run in the build container where /root/reference exists
(after `make -C oracle`).  Pins rows a5-a8 (histogram, tree, codes, packer, offsets) to the REFERENCE:

  * code lengths / codes come from the reference's own huffman_build_tree_cpu + FindMinimumCountTest
    (test_compress.cpp:55-78,127-190, compiled from the reference's lines by oracle/mk_ref_compress_gold.sh),
    run on histograms chosen to stress its tie-break (count, then level, then slot) and its relocation rule;
  * the expected stream is those codes packed per SURVEY.md App. A steps 6-7 (4096-symbol blocks, MSB first,
    word count in front), and it is ACCEPTED only if the reference's own gold decoder
    (computeCompressGold, test_compress.cpp:192-311: tree walk + inverse MTF) decodes it back to the symbols;
  * the end-to-end cases go input -> computeBwtGold -> computeMtfGold (the reference's lines) -> tree -> stream.

The file holds data only: histograms (the symbols follow from datagen.symbols_from_hist), code lengths,
sizes, CRC32s and the first words of every expected stream."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

N = 1 << 20


def pack(symbols, codes, lens):
    """App. A steps 6-7: per 4096 symbols, codes MSB-first into u32 words, preceded by the word count."""
    words, offsets = [], []
    for b in range(0, symbols.size, 4096):
        acc, nbits = 0, 0
        for s in symbols[b:b + 4096].tolist():
            acc = (acc << int(lens[s])) | codes[s]
            nbits += int(lens[s])
        nw = (nbits + 31) // 32
        acc <<= nw * 32 - nbits
        offsets.append(len(words))
        words.append(nw)
        words.extend((acc >> (32 * (nw - 1 - k))) & 0xFFFFFFFF for k in range(nw))
    return np.array(words, dtype=np.uint32), np.array(offsets, dtype=np.uint32)


def hist_cases():
    rng = np.random.default_rng(20260928)
    c = {}
    c["all_equal_256"] = np.full(256, N // 256)
    h = np.zeros(256, dtype=np.int64); h[0] = h[1] = N // 2; c["two_equal"] = h
    h = np.zeros(256, dtype=np.int64); h[3] = 1; h[200] = N - 1; c["two_unequal"] = h
    h = np.zeros(256, dtype=np.int64); h[7] = N; c["single_symbol"] = h
    h = np.zeros(256, dtype=np.int64); h[:20] = 2 ** np.arange(19, -1, -1); h[20] = 1; c["powers_of_two"] = h
    fib = [1, 1]
    while sum(fib) + fib[-1] + fib[-2] <= N:
        fib.append(fib[-1] + fib[-2])
    h = np.zeros(256, dtype=np.int64); h[10:10 + len(fib)] = fib; h[10 + len(fib) - 1] += N - sum(fib); c["fibonacci"] = h
    h = np.zeros(256, dtype=np.int64)
    base = rng.integers(100, 8000, 128); base = (base * (N // 2) // base.sum())
    h[0::2] = base; h[1::2] = base; h[0] += N - h.sum(); c["equal_pairs"] = h
    h = np.zeros(256, dtype=np.int64); h[1:] = N // 255; h[1] += N - h.sum(); c["kat_like_255"] = h
    z = 1.0 / np.arange(1, 257); h = np.floor(z / z.sum() * N).astype(np.int64); h[0] += N - h.sum(); c["zipf_ranks"] = h
    h = np.zeros(256, dtype=np.int64); h[:16] = N // 16; c["all_equal_16"] = h
    h = rng.integers(0, 5000, 256).astype(np.int64); h[h < 600] = 0; h[5] += N - h.sum(); c["random_sparse"] = h
    h = np.zeros(256, dtype=np.int64); h[20:220] = 1; h[0] = N - 200; c["many_ones"] = h          # ties with EOF (count 1)
    h = np.zeros(256, dtype=np.int64); h[:64] = np.repeat([1, 2, 4, 8], 16); h[255] = N - h.sum(); c["small_ties"] = h
    h = np.zeros(256, dtype=np.int64); h[:3] = [N // 4, N // 4, N // 2]; c["three_levels"] = h
    for k, v in c.items():
        assert v.sum() == N and (v >= 0).all(), k
    return c


def one(symbols):
    hist = np.bincount(symbols, minlength=256).astype(np.uint32)
    tree = O.ref_huffman_tree(hist)
    codes, lens = O.ref_codes_from_tree(tree)
    words, offsets = pack(symbols, codes, lens)
    sym_back, bytes_back = O.ref_compress_gold_decode(hist, offsets, words, symbols.size)
    assert np.array_equal(sym_back, symbols), "the reference's gold decoder does not read this stream"
    return dict(hist=hist, lens=lens.astype(np.uint8), size=np.array([words.size], dtype=np.uint32),
                crc_words=np.array([O.crc32(words.view(np.uint8))], dtype=np.uint32),
                crc_offsets=np.array([O.crc32(offsets.view(np.uint8))], dtype=np.uint32),
                head_words=words[:64].copy(), crc_imtf=np.array([O.crc32(bytes_back)], dtype=np.uint32)), words, offsets


def main():
    assert O.have_ref_compress_gold(), "run `make -C oracle` with /root/reference present"
    out = {}
    names = []
    for name, h in hist_cases().items():
        sym = datagen.symbols_from_hist(h)
        r, _, _ = one(sym)
        for k, v in r.items():
            out["h_%s_%s" % (name, k)] = v
        names.append(name)
        print(name, "nsym", int((h > 0).sum()), "maxlen", int(r["lens"].max()), "words", int(r["size"][0]))
    out["hist_cases"] = np.array(names)
    # end to end with the reference's own BWT and MTF gold: the two 1 MiB vectors of its tests + synthetic blocks
    e2e = {"ref_compressTest": None, "zipf": datagen.zipf_bytes(N), "float": datagen.float_bytes(N),
           "text": datagen.text_bytes(N)}
    x = O.glibc_rand_bytes(N, 255); x[-1] = 0                  # test_compress.cpp:687-692
    e2e["ref_compressTest"] = x
    en = []
    for name, x in e2e.items():
        b, idx = O.ref_bwt_gold(x)
        m = O.ref_mtf_gold(b)
        r, words, offsets = one(m)
        out["e_%s_bwt_index" % name] = np.array([idx], dtype=np.int32)
        out["e_%s_crc_in" % name] = np.array([O.crc32(x)], dtype=np.uint32)
        out["e_%s_crc_bwt" % name] = np.array([O.crc32(b)], dtype=np.uint32)
        out["e_%s_crc_mtf" % name] = np.array([O.crc32(m)], dtype=np.uint32)
        for k, v in r.items():
            out["e_%s_%s" % (name, k)] = v
        en.append(name)
        print("e2e", name, "idx", idx, "words", int(r["size"][0]), "crc %08x" % int(r["crc_words"][0]))
    out["e2e_cases"] = np.array(en)
    np.savez_compressed(os.path.join(HERE, "ref_huff_gold.npz"), **out)


if __name__ == "__main__":
    main()
