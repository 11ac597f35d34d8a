#!/usr/bin/env python3
"""Generates tests/golden/ref_lzss_gold.npz -- run in the build container where /root/reference exists
(after `make -C oracle`).  Pins row a13 (token selection + packing + trailer) and, through it, the format that
row a14 (DecodeKernel) reads, to the REFERENCE:

  * every candidate stream (2 B per input position: `length | 1`, `offset | literal`, as EncodeKernel leaves them,
    gpu_compress.cu:329-345) is handed to the reference's own aftercompression_wrapper / aftercomp
    (cuda-lzss-cluster/gpu_compress.cu:462-672, compiled from the reference's lines by oracle/mk_ref_aftercomp.sh
    into oracle/_ref/libaftercomp.so); the fixture keeps its return code, packed size, CRC-32, first and last
    bytes, and -- for buffers up to 64 KiB -- the packed bytes themselves, so that the HIP decoder can be run on
    bytes the reference wrote;
  * candidate streams are (i) the oracle's lock-step EncodeKernel restatement on the seeded inputs of
    datagen.lzss_gold_inputs() -- log lines, text, zeros, Zipf/float bytes (store raw), the two boundary inputs
    whose packed form is not smaller than the buffer, buffers of 1..4 packets -- and (ii) the synthetic streams of
    datagen.lzss_synthetic_candidates() (one length everywhere, all literals, jumps over a lane's segment, mixtures).
    The CRC-32 of every candidate stream is kept, so a test can tell that it packs the SAME candidates.

EncodeKernel itself (a11) is CUDA only and has no CPU twin in the reference: the candidates of (i) are the
oracle's, cross-checked in tests/test_cpu_oracle.py against the survey's independent restatement.

The file holds data only."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

KEEP_BYTES_UP_TO = 65536


def ref_aftercomp(cand, buf_length):
    """the reference's aftercompression_wrapper on a candidate stream -> (rc, packed bytes or None).
    `buffer` gets generous slack: the reference writes up to 535 bytes past buf_length (see include/culzss.h)."""
    L = O.ref_aftercomp_lib()
    buf = np.zeros(buf_length + 8192, dtype=np.uint8)
    c = np.ascontiguousarray(cand, dtype=np.uint8).copy()
    n = C.c_int(-1)
    rc = L.aftercompression_wrapper(buf.ctypes.data, buf_length, c.ctypes.data, C.byref(n))
    return rc, (buf[: n.value].copy() if rc == 1 else None)


def main():
    out = {}
    names = []

    def add(name, kind, cand, n):
        rc, packed = ref_aftercomp(cand, n)
        names.append(name)
        out[name + "/kind"] = np.array(kind)
        out[name + "/n"] = np.int64(n)
        out[name + "/cand_crc"] = np.uint32(zlib.crc32(cand.tobytes()))
        out[name + "/rc"] = np.int32(rc)
        if rc == 1:
            out[name + "/size"] = np.int64(packed.size)
            out[name + "/crc"] = np.uint32(zlib.crc32(packed.tobytes()))
            out[name + "/head"] = packed[:64].copy()
            out[name + "/tail"] = packed[-32:].copy()
            if n <= KEEP_BYTES_UP_TO:
                out[name + "/packed"] = packed
            # the oracle's packer must say the same thing
            mine = O.lzss_pack(cand, n)
            assert mine is not None and np.array_equal(mine, packed), name
        else:
            assert O.lzss_pack(cand, n) is None, name
        print("%-28s n=%8d rc=%d size=%s" % (name, n, rc, packed.size if rc == 1 else "-"))

    for name, x in datagen.lzss_gold_inputs().items():
        add(name, "input", O.lzss_candidates(x), x.size)
    n, syn = datagen.lzss_synthetic_candidates()
    for name, c in syn.items():
        add("syn_" + name, "synthetic", c, n)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_lzss_gold.npz"), **out)
    print("wrote ref_lzss_gold.npz: %d cases" % len(names))


if __name__ == "__main__":
    main()
