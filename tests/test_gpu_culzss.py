"""GPU parity tests for the CULZSS path: candidate stream, packed stream and
decode of the HIP kernels (through the C ABI of include/culzss.h) vs the
lock-step CPU oracle, byte-exact.  The reference has no tests for this codec
(SURVEY.md section 4); the cases follow its README transcript (compress,
decompress, diff) plus the raw-store fallback of gpu_compress.cu:494-498."""
import ctypes as C
import os

import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
MiB = 1 << 20


def _inputs():
    rng = np.random.default_rng(11)
    return {
        "log_1m": datagen.log_bytes(MiB),
        "text_1m": datagen.text_bytes(MiB),
        "zeros_64k": np.zeros(65536, dtype=np.uint8),
        "spaces_then_text": np.concatenate([np.full(8192, 0x20, dtype=np.uint8), datagen.text_bytes(57344, seed=5)]),
        "period3_8k": np.tile(np.array([1, 2, 3], dtype=np.uint8), 2731)[:8192].copy(),
        "caret_tail_4k": np.concatenate([datagen.log_bytes(3968, seed=9), np.full(128, ord("^"), dtype=np.uint8)]),
        "repeat_across_last_chunk": np.tile(datagen.text_bytes(96, seed=3), 43)[:4096].copy(),
        "one_packet_random": rng.integers(0, 256, 4096, dtype=np.uint8),
        "float_256k": datagen.float_bytes(262144),
    }


def _first_diff(a, b):
    if a.size != b.size:
        return "size %d want %d" % (a.size, b.size)
    d = np.nonzero(a != b)[0]
    return "first mismatch at %d (%d differ): got %s want %s" % (
        d[0], d.size, a[d[0]:d[0] + 8].tolist(), b[d[0]:d[0] + 8].tolist()) if d.size else "equal"


@pytest.mark.parametrize("name", list(_inputs().keys()))
def test_device_encode_matches_oracle(glc, cuda, name):
    import torch
    L = glc.lib()
    x = _inputs()[name]
    n = x.size
    stride = L.glcLzssPackStride(n)
    d_in = torch.from_numpy(x.copy()).cuda()
    d_cand = torch.zeros(2 * n, dtype=torch.uint8, device=cuda)
    d_packed = torch.zeros(stride, dtype=torch.uint8, device=cuda)
    d_size = torch.full((1,), -7, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, 1), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, 1, d_cand.data_ptr(), d_packed.data_ptr(),
                                 d_size.data_ptr(), d_work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    cand = d_cand.cpu().numpy()
    want_cand = O.lzss_candidates(x)
    assert np.array_equal(cand, want_cand), name + " candidates " + _first_diff(cand, want_cand)
    want_packed = O.lzss_pack(want_cand, n)
    size = int(d_size.item())
    if want_packed is None or want_packed.size >= n:               # (>= n: see test_packed_form_not_smaller_than_the_buffer)
        assert size == 0, "oracle says store-raw, gpu packed %d bytes" % size
        assert np.array_equal(d_packed.cpu().numpy()[:n], x)          # slot keeps the input
    else:
        got = d_packed.cpu().numpy()[:size]
        assert np.array_equal(got, want_packed), name + " packed " + _first_diff(got, want_packed)
    # decode on the device
    d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
    assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_size.data_ptr(), n, 1, d_out.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), x), name + " round trip"


def test_store_raw_fallback(glc, cuda):
    """10-symbol i.i.d. letters: packed form outgrows the buffer"""
    import torch
    L = glc.lib()
    x = np.random.default_rng(1).integers(97, 107, size=MiB, dtype=np.uint8)
    assert O.lzss_pack(O.lzss_candidates(x), MiB) is None
    out = np.zeros(L.glcLzssPackStride(MiB), dtype=np.uint8)
    n = C.c_int(0)
    assert L.culzss_compress(x.ctypes.data, MiB, out.ctypes.data, C.byref(n)) == 2
    assert n.value == MiB and np.array_equal(out[:MiB], x)


def test_batch_of_buffers(glc, cuda):
    import torch
    L = glc.lib()
    n, nb = 65536, 5
    bufs = [datagen.log_bytes(n, seed=i + 1) for i in range(3)] + [datagen.text_bytes(n, seed=8),
                                                                   np.random.default_rng(2).integers(97, 107, n, dtype=np.uint8)]
    stride = L.glcLzssPackStride(n)
    d_in = torch.from_numpy(np.concatenate(bufs)).cuda()
    d_packed = torch.zeros(stride * nb, dtype=torch.uint8, device=cuda)
    d_size = torch.zeros(nb, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, nb), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, nb, None, d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    d_out = torch.zeros(n * nb, dtype=torch.uint8, device=cuda)
    assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_size.data_ptr(), n, nb, d_out.data_ptr(), None) == 1
    torch.cuda.synchronize()
    sizes = d_size.cpu().numpy()
    packed = d_packed.cpu().numpy()
    for i, b in enumerate(bufs):
        want = O.lzss_pack(O.lzss_candidates(b), n)
        if want is None:
            assert sizes[i] == 0
        else:
            assert sizes[i] == want.size
            assert np.array_equal(packed[i * stride:i * stride + want.size], want), "buffer %d" % i
    assert np.array_equal(d_out.cpu().numpy(), np.concatenate(bufs))


def test_reference_wrapper_abi(glc, cuda):
    """the call sequence of culzss.c:85-86,108-109,170,176 and deculzss.c:98"""
    L = glc.lib()
    x = datagen.log_bytes(MiB, seed=21)
    L.initGPU()
    buf = L.initCPUmem(MiB)
    bufout = L.initCPUmem(2 * MiB)
    in_d = L.initGPUmem(MiB)
    out_d = L.initGPUmem(2 * MiB)
    assert buf and bufout and in_d and out_d
    C.memmove(buf, x.ctypes.data, MiB)
    for slot in (0, 3):
        C.memmove(buf, x.ctypes.data, MiB)
        assert L.compression_kernel_wrapper(buf, MiB, bufout, 0, 0, 128, 0, slot, in_d, out_d) == 1
        assert L.onestream_finish_GPU(slot) == 1
        cand = np.ctypeslib.as_array(C.cast(bufout, C.POINTER(C.c_uint8)), shape=(2 * MiB,)).copy()
        want_cand = O.lzss_candidates(x)
        assert np.array_equal(cand, want_cand), _first_diff(cand, want_cand)
        n = C.c_int(0)
        assert L.aftercompression_wrapper(buf, MiB, bufout, C.byref(n)) == 1
        want = O.lzss_pack(want_cand, MiB)
        got = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(MiB,))[:n.value].copy()
        assert np.array_equal(got, want), _first_diff(got, want)
        # candidates that were not produced by a tracked call are packed on the GPU as well
        other = L.initCPUmem(2 * MiB)
        C.memmove(other, want_cand.ctypes.data, 2 * MiB)
        n2 = C.c_int(0)
        assert L.aftercompression_wrapper(buf, MiB, other, C.byref(n2)) == 1 and n2.value == want.size
        L.deleteCPUmem(other)
        # in-place decode
        m = C.c_int(0)
        assert L.decompression_kernel_wrapper(buf, n.value, C.byref(m), 0, 1, 1) == 1 and m.value == MiB
        back = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(MiB,)).copy()
        assert np.array_equal(back, x)
    L.deleteCPUmem(buf); L.deleteCPUmem(bufout); L.deleteGPUmem(in_d); L.deleteGPUmem(out_d)
    L.deleteGPUStreams()


_synthetic_candidates = datagen.lzss_synthetic_candidates


def test_token_walk_on_synthetic_candidates(glc, cuda):
    """aftercompression_wrapper on candidates that did not come from a tracked call packs them on the GPU
    (gpu_compress.cu:462-566 semantics): packed bytes identical to the oracle's serial walk"""
    L = glc.lib()
    n, cases = _synthetic_candidates()
    L.initGPU()
    buf = L.initCPUmem(n)
    cand = L.initCPUmem(2 * n)
    bad = []
    for name, c in cases.items():
        want = O.lzss_pack(c, n)
        C.memmove(cand, c.ctypes.data, 2 * n)
        m = C.c_int(-1)
        rc = L.aftercompression_wrapper(buf, n, cand, C.byref(m))
        if want is None or want.size >= n:
            assert rc == 0, name + ": oracle says store-raw"
            continue
        assert rc == 1, name
        got = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(n,))[:m.value].copy()
        if not np.array_equal(got, want):
            bad.append(name + " " + _first_diff(got, want))
    L.deleteCPUmem(buf); L.deleteCPUmem(cand)
    L.deleteGPUStreams()
    assert not bad, "; ".join(bad)


def test_culzss_compress_decompress_roundtrip(glc, cuda):
    L = glc.lib()
    x = datagen.text_bytes(MiB, seed=77)
    out = np.zeros(L.glcLzssPackStride(MiB), dtype=np.uint8)
    n = C.c_int(0)
    assert L.culzss_compress(x.ctypes.data, MiB, out.ctypes.data, C.byref(n)) == 1
    want = O.lzss_pack(O.lzss_candidates(x), MiB)
    assert n.value == want.size and np.array_equal(out[:n.value], want)
    back = np.zeros(MiB, dtype=np.uint8)
    m = C.c_int(0)
    assert L.culzss_decompress(out.ctypes.data, n.value, back.ctypes.data, C.byref(m)) == 1
    assert m.value == MiB and np.array_equal(back, x)
    # rejects lengths that are not whole packets
    assert L.culzss_compress(x.ctypes.data, 5000, out.ctypes.data, C.byref(n)) == 0


def test_container_matches_oracle_and_round_trips(glc, cuda, tmp_path):
    """the file format of main.c / culzss.c / deculzss.c, incl. padding of the
    last buffer and a buffer that is stored raw"""
    L = glc.lib()
    rng = np.random.default_rng(5)
    x = np.concatenate([datagen.log_bytes(2 * MiB, seed=31), rng.integers(97, 107, MiB, dtype=np.uint8),
                        datagen.text_bytes(MiB + 12345, seed=32)])
    want = O.lzss_container_compress(x)
    cap = L.culzss_container_bound(x.size)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_ulonglong(0)
    assert L.culzss_container_compress(x.ctypes.data, x.size, out.ctypes.data, cap, C.byref(n)) == 1
    got = out[: n.value]
    assert np.array_equal(got, want), _first_diff(got, want)
    hdr = got[:8].view(np.uint32)
    assert hdr[0] == 5 and hdr[1] == 5 * MiB - x.size
    cum = got[8:8 + 20].view(np.uint32)
    assert cum[2] - cum[1] == MiB                                  # the incompressible buffer is stored raw
    back = np.zeros(5 * MiB, dtype=np.uint8)
    m = C.c_ulonglong(0)
    assert L.culzss_container_decompress(got.ctypes.data, got.size, back.ctypes.data, back.size, C.byref(m)) == 1
    assert m.value == x.size and np.array_equal(back[: x.size], x)
    assert np.array_equal(O.lzss_container_decompress(got), x)    # and the oracle reads our file
    # file API (./main -i / ./main -d 1 -i)
    fin, fc, fo = tmp_path / "in.bin", tmp_path / "c.bin", tmp_path / "out.bin"
    fin.write_bytes(x.tobytes())
    assert L.culzss_compress_file(str(fin).encode(), str(fc).encode()) == 1
    assert fc.read_bytes() == want.tobytes()
    assert L.culzss_decompress_file(str(fc).encode(), str(fo).encode()) == 1
    assert fo.read_bytes() == x.tobytes()
    # shorter than one buffer: refused (main.c:228-232)
    assert L.culzss_container_compress(x.ctypes.data, MiB - 1, out.ctypes.data, cap, C.byref(n)) == 0


def test_container_many_buffers(glc, cuda):
    L = glc.lib()
    x = np.tile(datagen.log_bytes(3 * MiB, seed=77), 14)[: 40 * MiB + 777]     # > 2 groups of 16 buffers
    cap = L.culzss_container_bound(x.size)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_ulonglong(0)
    assert L.culzss_container_compress(x.ctypes.data, x.size, out.ctypes.data, cap, C.byref(n)) == 1
    back = np.zeros(41 * MiB, dtype=np.uint8)
    m = C.c_ulonglong(0)
    assert L.culzss_container_decompress(out.ctypes.data, n.value, back.ctypes.data, back.size, C.byref(m)) == 1
    assert m.value == x.size and np.array_equal(back[: x.size], x)
    # spot-check buffers 0, 17 and the last against the oracle's per-buffer packing
    cum = np.concatenate([[0], out[8:8 + 4 * 41].view(np.uint32)])
    for i in (0, 17, 40):
        blk = np.zeros(MiB, dtype=np.uint8)
        src = x[i * MiB:(i + 1) * MiB]
        blk[: src.size] = src
        want = O.lzss_pack(O.lzss_candidates(blk), MiB)
        got = out[8 + 4 * 41 + cum[i]: 8 + 4 * 41 + cum[i + 1]]
        assert np.array_equal(got, want), "buffer %d" % i


@pytest.mark.parametrize("run", [118685, 119175])
def test_packed_form_not_smaller_than_the_buffer(glc, cuda, run):
    """aftercomp only gives up when the bytes flushed BEFORE the last group outgrow the buffer
    (gpu_compress.cu:492-497), so the packed form incl. trailer can reach BUFSIZE + 535 bytes: the reference
    then writes past its 1 MiB slot, and a payload of exactly BUFSIZE bytes is read back as raw
    (deculzss.c:94-95).  Both inputs (a run of 'A' followed by random bytes: 1 049 103 and exactly 1 048 576
    packed bytes by the reference's rules) must be stored raw here and survive the container round trip."""
    L = glc.lib()
    x = np.concatenate([np.full(run, 65, dtype=np.uint8),
                        np.random.default_rng(20260928).integers(0, 256, MiB, dtype=np.uint8)[: MiB - run]])
    ref_rule = O.lzss_pack(O.lzss_candidates(x), MiB)
    assert ref_rule is not None and ref_rule.size == {118685: 1049103, 119175: MiB}[run]   # the reference's rule lets it pass
    out = np.zeros(L.glcLzssPackStride(MiB), dtype=np.uint8)
    n = C.c_int(0)
    assert L.culzss_compress(x.ctypes.data, MiB, out.ctypes.data, C.byref(n)) == 2
    assert n.value == MiB and np.array_equal(out[:MiB], x)
    y = np.concatenate([datagen.log_bytes(MiB, seed=3), x])
    cap = L.culzss_container_bound(y.size)
    cont = np.zeros(cap, dtype=np.uint8)
    m = C.c_ulonglong(0)
    assert L.culzss_container_compress(y.ctypes.data, y.size, cont.ctypes.data, cap, C.byref(m)) == 1
    assert m.value <= cap
    got = cont[: m.value]
    assert np.array_equal(got, O.lzss_container_compress(y))
    back = np.zeros(2 * MiB, dtype=np.uint8)
    k = C.c_ulonglong(0)
    assert L.culzss_container_decompress(got.ctypes.data, got.size, back.ctypes.data, back.size, C.byref(k)) == 1
    assert k.value == y.size and np.array_equal(back, y)


def test_malformed_streams_are_rejected(glc, cuda):
    """the decoder must not trust the stream: a payload too short for its trailer, a trailer that names another
    buffer length, and packet sizes that run past the body all fail cleanly (no out-of-range device reads)"""
    L = glc.lib()
    x = datagen.log_bytes(2 * MiB, seed=41)
    cap = L.culzss_container_bound(x.size)
    cont = np.zeros(cap, dtype=np.uint8)
    m = C.c_ulonglong(0)
    assert L.culzss_container_compress(x.ctypes.data, x.size, cont.ctypes.data, cap, C.byref(m)) == 1
    good = cont[: m.value].copy()
    back = np.zeros(2 * MiB, dtype=np.uint8)
    k = C.c_ulonglong(0)
    cum = good[8:16].view(np.uint32).copy()
    # (a) first payload claims 100 bytes
    bad = good.copy(); bad[8:12].view(np.uint32)[0] = 100
    assert L.culzss_container_decompress(bad.ctypes.data, bad.size, back.ctypes.data, back.size, C.byref(k)) == 0
    # (b) trailer of payload 0 names a 2 MiB buffer
    bad = good.copy(); bad[16 + cum[0] - 6] = 0; bad[16 + cum[0] - 5] = 0x20
    assert L.culzss_container_decompress(bad.ctypes.data, bad.size, back.ctypes.data, back.size, C.byref(k)) == 0
    # (c) packet size table of payload 0 sums to far more than the body
    bad = good.copy(); tr = 16 + cum[0] - 6 - 512
    bad[tr:tr + 512] = 0xFF
    assert L.culzss_container_decompress(bad.ctypes.data, bad.size, back.ctypes.data, back.size, C.byref(k)) == 0
    # the wrapper ABI on the same corrupted payload
    buf = np.zeros(L.glcLzssPackStride(MiB), dtype=np.uint8)
    buf[: cum[0]] = bad[16:16 + cum[0]]
    dl = C.c_int(0)
    assert L.decompression_kernel_wrapper(buf.ctypes.data, int(cum[0]), C.byref(dl), 0, 1, 1) == 0
    # and the intact container still decodes
    assert L.culzss_container_decompress(good.ctypes.data, good.size, back.ctypes.data, back.size, C.byref(k)) == 1
    assert np.array_equal(back, x)


@pytest.mark.parametrize("shift", [1, 5, 13])
def test_decode_from_unaligned_device_pointers(glc, cuda, shift):
    """the decoder's 16-byte staging loads must not assume the caller's buffers are aligned"""
    import torch
    L = glc.lib()
    n, nb = 4 * 4096, 3
    bufs = [datagen.log_bytes(n, seed=70 + i) for i in range(nb)]
    d_in = torch.from_numpy(np.concatenate(bufs)).to(cuda)
    stride = L.glcLzssPackStride(n)
    d_packed = torch.zeros(stride * nb, dtype=torch.uint8, device=cuda)
    d_size = torch.zeros(nb, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, nb), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, nb, None, d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    moved = torch.zeros(stride * nb + 64, dtype=torch.uint8, device=cuda)
    moved[shift:shift + stride * nb] = d_packed
    d_out = torch.zeros(n * nb + 64, dtype=torch.uint8, device=cuda)
    assert L.glcLzssDecodeDevice(moved.data_ptr() + shift, d_size.data_ptr(), n, nb, d_out.data_ptr() + shift, None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy()[shift:shift + n * nb], np.concatenate(bufs))


def test_randomised_parity_sweep(glc, cuda):
    """30 seeded random buffers (1..8 packets; i.i.d. small alphabets, runs, periodic text with
    mutations): candidates, packed bytes and decode against the lock-step oracle."""
    import torch
    L = glc.lib()
    rng = np.random.default_rng(int(os.environ.get("GLC_FUZZ_SEED", "4096")))      # other seeds / more cases: one-off sweeps
    for case in range(int(os.environ.get("GLC_FUZZ_CASES", "30"))):
        n = 4096 * int(rng.integers(1, 9))
        a = int(rng.choice([2, 4, 26, 256]))
        kind = case % 3
        if kind == 0:
            x = rng.integers(0, a, n, dtype=np.uint16).astype(np.uint8)
        elif kind == 1:
            x = np.resize(np.repeat(rng.integers(0, a, n // 5 + 1, dtype=np.uint16), rng.integers(1, 200, n // 5 + 1)), n).astype(np.uint8)
        else:
            x = np.resize(rng.integers(32, 32 + min(a, 90), int(rng.integers(3, 400)), dtype=np.uint16).astype(np.uint8), n).copy()
            flips = rng.integers(0, n, n // 300 + 1)
            x[flips] = rng.integers(32, 122, flips.size, dtype=np.uint16).astype(np.uint8)
        x = np.ascontiguousarray(x)
        tag = "case %d n=%d alphabet=%d kind=%d" % (case, n, a, kind)
        d_in = torch.from_numpy(x).to(cuda)
        stride = L.glcLzssPackStride(n)
        d_cand = torch.zeros(2 * n, dtype=torch.uint8, device=cuda)
        d_packed = torch.zeros(stride, dtype=torch.uint8, device=cuda)
        d_size = torch.full((1,), -7, dtype=torch.int32, device=cuda)
        d_work = torch.zeros(L.glcLzssWorkBytes(n, 1), dtype=torch.uint8, device=cuda)
        assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, 1, d_cand.data_ptr(), d_packed.data_ptr(),
                                     d_size.data_ptr(), d_work.data_ptr(), None) == 1
        torch.cuda.synchronize()
        want_cand = O.lzss_candidates(x)
        assert np.array_equal(d_cand.cpu().numpy(), want_cand), tag + " candidates"
        want = O.lzss_pack(want_cand, n)
        size = int(d_size.item())
        if want is None:
            assert size == 0, tag
        else:
            assert size == want.size and np.array_equal(d_packed.cpu().numpy()[:size], want), tag + " packed"
        d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
        assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_size.data_ptr(), n, 1, d_out.data_ptr(), None) == 1
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), x), tag + " round trip"


def test_container_with_a_packed_chunk_longer_than_the_buffer(glc, cuda):
    """ADVICE.md (round 2): a file written by the REFERENCE can hold a packed chunk of up to BUFSIZE + 535 bytes (its packer
    gives up only when the bytes flushed before the last group outgrow the buffer), and its decoder takes anything that
    is not exactly BUFSIZE as packed (deculzss.c:92-98).  Such a container -- built here by hand from the packed form
    the reference's rules give for the 118 685-byte-run input, 1 049 103 bytes -- must decode to the input."""
    L = glc.lib()
    x = datagen.lzss_gold_inputs()["run_118685"]
    packed = O.lzss_pack(O.lzss_candidates(x), MiB)
    assert packed is not None and packed.size == 1049103 > MiB
    blob = np.concatenate([np.array([1, 0, packed.size], dtype=np.uint32).view(np.uint8), packed])
    back = np.zeros(MiB, dtype=np.uint8)
    k = C.c_ulonglong(0)
    assert L.culzss_container_decompress(blob.ctypes.data, blob.size, back.ctypes.data, back.size, C.byref(k)) == 1
    assert k.value == MiB and np.array_equal(back, x)
    # one byte more than a slot can hold is refused, not followed
    too_long = np.concatenate([np.array([1, 0, int(L.glcLzssPackStride(MiB)) + 1], dtype=np.uint32).view(np.uint8),
                               np.zeros(int(L.glcLzssPackStride(MiB)) + 1, dtype=np.uint8)])
    assert L.culzss_container_decompress(too_long.ctypes.data, too_long.size, back.ctypes.data, back.size, C.byref(k)) == 0
