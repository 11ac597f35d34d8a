#!/usr/bin/env python3
"""Generates tests/golden/ref_huff_ties_gold.npz -- run in the build container where /root/reference exists (after
`make -C oracle`).  For the tie-heavy histograms of tests/test_gpu_huffman_ties.py: the codes and code lengths of the
REFERENCE's own huffman_build_tree_cpu + FindMinimumCountTest (test_compress.cpp:55-78,127-190, compiled from the
reference's lines by oracle/mk_ref_compress_gold.sh).  Data only: histograms, codes, lengths."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import test_gpu_huffman_ties as T  # noqa: E402

out = {"cases": np.array([c[0] for c in T.CASES])}
for name, hist in T.CASES:
    codes, lens = O.ref_codes_from_tree(O.ref_huffman_tree(hist))
    out[name + "_hist"] = hist.astype(np.uint32)
    out[name + "_codes"] = np.array(codes, dtype=np.uint64)
    out[name + "_lens"] = lens.astype(np.uint8)
np.savez_compressed(os.path.join(HERE, "ref_huff_ties_gold.npz"), **out)
print("wrote", len(T.CASES), "cases")
