"""CULZSS on the GPU against REFERENCE-PRODUCED vectors, with nothing of this repo's oracle in between:

  tests/golden/ref_lzss_gold.npz   what the reference's own aftercompression_wrapper / aftercomp
                                   (cuda-lzss-cluster/gpu_compress.cu:462-672, compiled from the reference's lines by
                                   oracle/mk_ref_aftercomp.sh) returned for 31 candidate streams
                                   (tests/golden/make_lzss_gold.py): return code, packed size, CRC-32, first and last
                                   bytes, the bytes themselves for buffers up to 64 KiB.

a13: the HIP token walk + packer (k_lzss_pack_wave / k_lzss_pack / layout / gather) on the same candidates must
     give exactly those bytes;
a14: the HIP decoder (k_lzss_decode) must read bytes THE REFERENCE packed back to the input;
a11: the HIP match kernel's candidate stream (k_lzss_match) against
  tests/golden/ref_findmatch_gold.npz   the candidate streams the reference's own FindMatch
                                   (gpu_compress.cu:104-168, compiled from the reference's lines by
                                   oracle/mk_ref_findmatch.sh) produced inside EncodeKernel's ring choreography
                                   (tests/golden/make_findmatch_gold.py): 24 inputs, CRC-32 each, every byte up to 64 KiB.
Checksums are zlib's CRC-32."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_lzss_gold.npz"))
NAMES = [str(s) for s in GOLD["names"]]
INPUT_CASES = [s for s in NAMES if str(GOLD[s + "/kind"]) == "input"]
SYN_CASES = [s for s in NAMES if str(GOLD[s + "/kind"]) == "synthetic"]


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def inputs():
    return datagen.lzss_gold_inputs()


def _check_packed(name, got):
    assert got.size == int(GOLD[name + "/size"]), "%s: %d bytes, the reference packed %d" % (name, got.size, int(GOLD[name + "/size"]))
    assert np.array_equal(got[:64], GOLD[name + "/head"]), name + ": first bytes"
    assert np.array_equal(got[-32:], GOLD[name + "/tail"]), name + ": trailer bytes"
    assert _crc(got) == int(GOLD[name + "/crc"]), name + ": CRC of the packed bytes"
    if name + "/packed" in GOLD:
        assert np.array_equal(got, GOLD[name + "/packed"]), name


@pytest.mark.parametrize("name", INPUT_CASES)
def test_encode_matches_reference_packed_bytes(glc, cuda, inputs, name):
    """input -> HIP match kernel -> HIP packer == what the reference's packer made of the same candidates"""
    import torch
    L = glc.lib()
    x = inputs[name]
    n = x.size
    d_in = torch.from_numpy(x.copy()).cuda()
    d_cand = torch.zeros(2 * n, dtype=torch.uint8, device=cuda)
    d_packed = torch.zeros(L.glcLzssPackStride(n), dtype=torch.uint8, device=cuda)
    d_size = torch.full((1,), -7, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, 1), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, 1, d_cand.data_ptr(), d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert _crc(d_cand.cpu().numpy()) == int(GOLD[name + "/cand_crc"]), name + ": candidate stream"
    size = int(d_size.item())
    ref_rc = int(GOLD[name + "/rc"])
    if ref_rc == 0 or int(GOLD[name + "/size"]) >= n:
        # the reference gives up (rc 0), or its packed form is not smaller than the buffer (run_118685: 1 049 103 bytes
        # written into a 1 MiB slot; run_119175: exactly BUFSIZE, read back as raw by deculzss.c:94-95): stored raw here,
        # see include/culzss.h
        assert size == 0 and np.array_equal(d_packed.cpu().numpy()[:n], x)
        return
    _check_packed(name, d_packed.cpu().numpy()[:size])


@pytest.mark.parametrize("name", SYN_CASES + ["log_4pkt", "spaces_then_text", "zeros_1m"])
def test_packer_on_given_candidates_matches_reference(glc, cuda, inputs, name):
    """aftercompression_wrapper of include/culzss.h (candidates handed in by the caller, as culzss.c:133-134 does)
    == the reference's aftercompression_wrapper on the same bytes"""
    import oracle_lib as O
    L = glc.lib()
    if name in inputs:
        cand = O.lzss_candidates(inputs[name])             # (checked against the fixture's CRC below)
        n = inputs[name].size
    else:
        n, syn = datagen.lzss_synthetic_candidates()
        cand = syn[name[4:]]
    assert _crc(cand) == int(GOLD[name + "/cand_crc"])
    L.initGPU()
    buf = L.initCPUmem(n)
    hc = L.initCPUmem(2 * n)
    C.memmove(hc, cand.ctypes.data, 2 * n)
    m = C.c_int(-1)
    rc = L.aftercompression_wrapper(buf, n, hc, C.byref(m))
    got = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(n,))[: max(m.value, 0)].copy()
    L.deleteCPUmem(buf); L.deleteCPUmem(hc)
    L.deleteGPUStreams()
    assert rc == int(GOLD[name + "/rc"]), name
    if rc == 1:
        _check_packed(name, got)


@pytest.mark.parametrize("name", [s for s in INPUT_CASES if s + "/packed" in GOLD])
def test_decoder_reads_reference_packed_bytes(glc, cuda, inputs, name):
    """bytes written by the reference's packer -> HIP k_lzss_decode == the input (cross-pins a14's reading of the
    format: flag bits LSB first, literal = 1, (length, offset) pairs, 256-entry ring, big-endian trailer)"""
    import torch
    L = glc.lib()
    x = inputs[name]
    n = x.size
    packed = GOLD[name + "/packed"]
    slot = np.zeros(L.glcLzssPackStride(n), dtype=np.uint8)
    slot[: packed.size] = packed
    d_packed = torch.from_numpy(slot).cuda()
    d_size = torch.tensor([packed.size], dtype=torch.int32, device=cuda)
    d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
    assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_size.data_ptr(), n, 1, d_out.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), x), name
    # and through the reference's in-place wrapper ABI (deculzss.c:98)
    L.initGPU()
    buf = L.initCPUmem(max(n, packed.size))
    C.memmove(buf, packed.ctypes.data, packed.size)
    k = C.c_int(0)
    assert L.decompression_kernel_wrapper(buf, packed.size, C.byref(k), 0, 1, 1) == 1 and k.value == n
    back = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    L.deleteCPUmem(buf)
    L.deleteGPUStreams()
    assert np.array_equal(back, x), name


# ------------------------------------------------ a11: the match kernel against the reference's FindMatch ----
FM = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_findmatch_gold.npz"))
FM_NAMES = [str(s) for s in FM["names"]]


@pytest.fixture(scope="module")
def fm_inputs():
    return datagen.findmatch_gold_inputs()


@pytest.mark.parametrize("name", FM_NAMES)
def test_match_kernel_equals_reference_findmatch(glc, cuda, fm_inputs, name):
    """input -> k_lzss_match (through glcLzssEncodeDevice) == the bytes the reference's own FindMatch produced"""
    import torch
    L = glc.lib()
    x = fm_inputs[name]
    n = x.size
    assert n == int(FM[name + "/n"]) and _crc(x) == int(FM[name + "/in_crc"]), name + ": input drifted"
    d_in = torch.from_numpy(x.copy()).cuda()
    d_cand = torch.full((2 * n,), 0xEE, dtype=torch.uint8, device=cuda)
    d_packed = torch.zeros(L.glcLzssPackStride(n), dtype=torch.uint8, device=cuda)
    d_size = torch.full((1,), -7, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, 1), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, 1, d_cand.data_ptr(), d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    cand = d_cand.cpu().numpy()
    if name + "/cand" in FM:
        want = FM[name + "/cand"]
        bad = np.flatnonzero(cand != want)
        assert bad.size == 0, "%s: %d candidate bytes differ from the reference FindMatch, first at position %d" % (name, bad.size, int(bad[0]) // 2)
    assert _crc(cand) == int(FM[name + "/cand_crc"]), name + ": candidate stream"


def test_match_kernel_batch_of_reference_buffers(glc, cuda, fm_inputs):
    """the 1 MiB fixture inputs as ONE batched launch (buffers side by side): every buffer's candidates == the reference's"""
    import torch
    L = glc.lib()
    names = [s for s in FM_NAMES if int(FM[s + "/n"]) == 1 << 20]
    n, nb = 1 << 20, len(names)
    d_in = torch.from_numpy(np.concatenate([fm_inputs[s] for s in names])).cuda()
    d_cand = torch.zeros(2 * n * nb, dtype=torch.uint8, device=cuda)
    stride = L.glcLzssPackStride(n)
    d_packed = torch.zeros(stride * nb, dtype=torch.uint8, device=cuda)
    d_size = torch.zeros(nb, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, nb), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, nb, d_cand.data_ptr(), d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    cand = d_cand.cpu().numpy()
    for i, s in enumerate(names):
        assert _crc(cand[2 * n * i: 2 * n * (i + 1)]) == int(FM[s + "/cand_crc"]), s

