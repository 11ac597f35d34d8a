#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run in the build container where
/root/reference exists.  Vectors are DATA (inputs + expected outputs):

  ref_gold_small.npz  the reference's own test inputs (glibc srand(95835);
                      test_compress.cpp:439-441,552-556; test_sa.cpp:124-126) at
                      the small sizes of its test matrix, with the suffix arrays
                      produced by the reference's gold routine computeSaGold
                      (apps/cudpp_testrig/sa_gold.cpp, compiled unmodified into
                      oracle/_ref/libsagold.so) and BWT/MTF derived from them with
                      the semantics of computeBwtGold/computeMtfGold
                      (test_compress.cpp:79-125) restated in numpy below.
  ref_gold_1m.json    CRC32s / index of the 1 MiB reference vectors.
  stream_kats.npz     restatement-derived known answers for the Huffman stream and
                      CULZSS ("parity unpinned" by reference tests): produced by
                      the oracle, validated by round trip; they pin the oracle
                      against silent drift, not against the reference.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402


def bwt_from_sa(x, sa):          # computeBwtGold, test_compress.cpp:79-91
    out = np.where(sa == 0, x[-1], x[(sa.astype(np.int64) - 1) % x.size])
    return out.astype(np.uint8), int(np.nonzero(sa == 0)[0][0])


def mtf_gold(x):                 # computeMtfGold, test_compress.cpp:93-125
    lst = list(range(256))
    out = np.empty(x.size, dtype=np.uint8)
    for i, c in enumerate(x.tolist()):
        j = lst.index(c)
        out[i] = j
        lst.pop(j)
        lst.insert(0, c)
    return out


def main():
    assert O.have_ref_gold(), "oracle/_ref/libsagold.so missing: run make -C oracle with /root/reference present"
    small = {}
    sa_in = O.glibc_rand_bytes(65536, 128)          # test_sa.cpp:124-126
    bw_in = O.glibc_rand_bytes(65536, 255)          # test_compress.cpp:552-556
    for n in (39, 128, 256, 512, 513, 1000, 1024, 1025, 32768, 45537, 65536):
        x = sa_in[:n]
        small["sa_in_%d" % n] = x
        small["sa_out_%d" % n] = O.ref_sa_gold(x)
    for n in (39, 128, 1000, 1025, 45537, 65536):
        x = bw_in[:n]
        sa = O.ref_sa_gold(x)
        b, idx = bwt_from_sa(x, sa)
        small["bwt_in_%d" % n] = x
        small["bwt_out_%d" % n] = b
        small["bwt_idx_%d" % n] = np.array([idx], dtype=np.int32)
        small["mtf_of_in_%d" % n] = mtf_gold(x)       # MTF test applies MTF to the raw input
    for name, s in (("mississippi", b"mississippi"), ("banana", b"banana")):
        x = np.frombuffer(s, dtype=np.uint8)
        sa = O.ref_sa_gold(x)
        b, idx = bwt_from_sa(x, sa)
        small["str_%s_sa" % name] = sa
        small["str_%s_bwt" % name] = b
        small["str_%s_idx" % name] = np.array([idx], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_gold_small.npz"), **small)

    N = 1 << 20
    x = O.glibc_rand_bytes(N, 255)
    sa = O.ref_sa_gold(x)
    b, idx = bwt_from_sa(x, sa)
    x2 = x.copy(); x2[-1] = 0
    sa2 = O.ref_sa_gold(x2)
    b2, idx2 = bwt_from_sa(x2, sa2)
    j = {"bwtTest": {"crc_in": "%08x" % O.crc32(x), "bwt_index": idx, "crc_bwt": "%08x" % O.crc32(b),
                     "crc_mtf": "%08x" % O.crc32(O.mtf(b)), "crc_sa": "%08x" % O.crc32(sa.view(np.uint8))},
         "compressTest": {"crc_in": "%08x" % O.crc32(x2), "bwt_index": idx2, "crc_bwt": "%08x" % O.crc32(b2),
                          "crc_mtf": "%08x" % O.crc32(O.mtf(b2)), "crc_sa": "%08x" % O.crc32(sa2.view(np.uint8))}}
    r = O.compress(x2)
    j["compressTest_stream_restatement"] = {
        "size_words": r["size"], "crc_words": "%08x" % O.crc32(r["words"].view(np.uint8)),
        "crc_offsets": "%08x" % O.crc32(r["offsets"].view(np.uint8)), "crc_hist": "%08x" % O.crc32(r["hist"].view(np.uint8))}
    json.dump(j, open(os.path.join(HERE, "ref_gold_1m.json"), "w"), indent=1)

    kat = {}
    t = datagen.text_bytes(8192, seed=42)
    r = O.compress(t)
    kat["huff_in"] = t; kat["huff_words"] = r["words"]; kat["huff_offsets"] = r["offsets"]; kat["huff_hist"] = r["hist"]
    kat["huff_idx"] = np.array([r["bwt_index"]], dtype=np.int32)
    lz = datagen.log_bytes(8192, seed=43)
    c = O.lzss_candidates(lz)
    kat["lz_in"] = lz; kat["lz_cand"] = c; kat["lz_packed"] = O.lzss_pack(c, lz.size)
    np.savez_compressed(os.path.join(HERE, "stream_kats.npz"), **kat)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
