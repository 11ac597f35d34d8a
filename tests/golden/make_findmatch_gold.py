#!/usr/bin/env python3
"""Generates tests/golden/ref_findmatch_gold.npz -- run in the build container where /root/reference exists
(after `make -C oracle`).  Pins row a11 (the CULZSS match search) to the REFERENCE:

  * the reference's own FindMatch (cuda-lzss-cluster/gpu_compress.cu:104-168: the body is plain C, compiled from
    the reference's lines by oracle/mk_ref_findmatch.sh into oracle/_ref/libfindmatch.so) is called once per
    byte position -- 128 lanes x 32 chunks per 4096-byte packet -- on the two rings exactly as EncodeKernel
    (gpu_compress.cu:182-350) has filled them at that call;
  * the ring choreography around it (which input byte sits in which ring slot at which call, the emit rule of
    :251-274, the last-chunk clamp of :313-317) is the part that stays restated, here in Python, phase by phase
    where the kernel has __syncthreads: EncodeKernel itself needs __shared__ / threadIdx / __syncthreads and is
    not compilable in this container.  It moves bytes; every comparison is made by the reference's code.

The fixture keeps, per input of datagen.findmatch_gold_inputs(): size, CRC-32 of the input and of the candidate
stream (2 B per input position: `length | 1`, `offset | literal`), and -- for inputs up to 64 KiB -- the candidate
bytes themselves.  The oracle's lock-step restatement (oracle/glc_oracle.c orc_lzss_candidates) must produce the
same bytes (asserted here, and again by tests/test_cpu_oracle.py from the fixture); the HIP match kernel is
compared with the fixture directly (tests/test_gpu_lzss_refgold.py).

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
WIN, MAXC, RING, PCKT = 128, 128, 256, 4096      # gpu_compress.h:62-69


def encode_packet_with_ref_findmatch(L, pkt, out):
    """EncodeKernel's data movement for one packet (gpu_compress.cu:182-350), FindMatch = the reference's."""
    win = (C.c_ubyte * RING)()
    la = (C.c_ubyte * RING)()
    find = L.ref_FindMatch
    length = [1] * MAXC
    offset = [1] * MAXC
    for tx in range(MAXC):                               # :208
        win[tx] = 0x20
    whead0 = uhead0 = 0                                  # lane tx: windowHead = (tx + whead0) % 256, likewise uncoded
    filepoint = wfile = lastcheck = loadcounter = 0
    for tx in range(MAXC):                               # :224
        la[tx] = pkt[tx]
    filepoint += MAXC
    for tx in range(MAXC):                               # :227
        win[(tx + WIN) % RING] = la[tx]
    for tx in range(MAXC):                               # :233
        la[MAXC + tx] = pkt[filepoint + tx]
    filepoint += MAXC
    loadcounter += 1

    def search():
        for tx in range(MAXC):
            r = find((tx + whead0) % RING, (tx + uhead0) % RING, win, la, tx, 0, wfile, lastcheck, loadcounter)
            length[tx], offset[tx] = r.length, r.offset

    def emit():                                          # :251-274 / :319-345
        for tx in range(MAXC):
            n = length[tx]
            if n >= MAXC:
                n = MAXC - 1
            if n <= 2:
                out[wfile + 2 * tx] = 1
                out[wfile + 2 * tx + 1] = la[(tx + uhead0) % RING]
            else:
                out[wfile + 2 * tx] = n & 255
                out[wfile + 2 * tx + 1] = offset[tx] & 255

    search()                                             # :242
    while filepoint <= PCKT and not lastcheck:           # :246
        emit()
        wfile += 2 * MAXC
        whead0 = (whead0 + MAXC) % RING
        uhead0 = (uhead0 + MAXC) % RING
        if filepoint < PCKT:                             # :291-298
            for tx in range(MAXC):
                la[(tx + uhead0 + MAXC) % RING] = pkt[filepoint + tx]
            filepoint += MAXC
            for tx in range(MAXC):
                win[(tx + whead0 + WIN) % RING] = la[(tx + uhead0) % RING]
        else:                                            # :301-303
            lastcheck += 1
            for tx in range(MAXC):
                win[(tx + whead0 + MAXC) % RING] = ord("^")
        loadcounter += 1
        search()                                         # :308
    if lastcheck == 1:                                   # :313-317
        for tx in range(MAXC):
            if length[tx] > MAXC - tx:
                length[tx] = MAXC - tx
    emit()


def ref_candidates(x):
    L = O.ref_findmatch_lib()
    x = np.ascontiguousarray(x, dtype=np.uint8)
    assert x.size % PCKT == 0
    out = np.zeros(2 * x.size, dtype=np.uint8)
    for g in range(x.size // PCKT):
        encode_packet_with_ref_findmatch(L, x[g * PCKT:(g + 1) * PCKT].tolist(), out[g * 2 * PCKT:(g + 1) * 2 * PCKT])
    return out


def main():
    out = {}
    names = []
    for name, x in datagen.findmatch_gold_inputs().items():
        cand = ref_candidates(x)
        mine = O.lzss_candidates(x)
        diff = int(np.count_nonzero(mine != cand))
        assert diff == 0, "%s: the oracle's candidates differ from the reference FindMatch in %d bytes" % (name, diff)
        names.append(name)
        out[name + "/n"] = np.int64(x.size)
        out[name + "/in_crc"] = np.uint32(zlib.crc32(x.tobytes()))
        out[name + "/cand_crc"] = np.uint32(zlib.crc32(cand.tobytes()))
        out[name + "/matches"] = np.int64(np.count_nonzero(cand[0::2] > 1))
        if x.size <= KEEP_BYTES_UP_TO:
            out[name + "/cand"] = cand
        print("%-28s n=%8d cand_crc=%08x matches=%d" % (name, x.size, zlib.crc32(cand.tobytes()), out[name + "/matches"]))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "ref_findmatch_gold.npz"), **out)
    print("wrote ref_findmatch_gold.npz: %d cases" % len(names))


if __name__ == "__main__":
    main()
