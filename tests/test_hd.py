"""Row f3 (CUHD-shaped Huffman-only stream): host table/encoder on CPU, device decoder on GPU.

Parity definition for this row is the reference's own: decoded == original
(cuhd-icpp/src/demo.cc:176-178); the oracle's bit-serial decoder is the independent
checker of the stream the host encoder writes."""
import numpy as np
import pytest

import oracle_lib as O


def binomial_bytes(n, seed):
    # demo.cc:93-105: symbols ~ Binomial(255, 0.5)
    return np.random.default_rng(seed).binomial(255, 0.5, size=n).astype(np.uint8)


def cases():
    rng = np.random.default_rng(7)
    yield "binomial_1m", binomial_bytes(1 << 20, 1)
    yield "uniform", rng.integers(0, 256, 300_000, dtype=np.uint8)
    yield "two_symbols", rng.integers(0, 2, 100_001, dtype=np.uint8) * 200
    yield "one_symbol", np.full(70_000, 42, dtype=np.uint8)
    yield "single_byte", np.array([9], dtype=np.uint8)
    z = np.minimum(rng.zipf(1.3, 400_000) - 1, 255).astype(np.uint8)
    yield "zipf_deep", z
    # fibonacci-like counts force the unrestricted tree far deeper than 11 bits
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    yield "fibonacci", np.repeat(np.arange(30, dtype=np.uint8), fib)[rng.permutation(sum(fib))]
    yield "short_7", rng.integers(0, 5, 7, dtype=np.uint8)


CASES = list(cases())


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_table_is_prefix_free_limited_and_optimal(glc, name, data):
    hist = np.bincount(data, minlength=256).astype(np.uint64)
    lens, codes = glc.hd_build_table(hist)
    used = lens > 0
    assert np.array_equal(used, hist > 0)
    assert lens.max() <= glc.GLC_HD_MAX_LEN
    kraft = np.sum(2.0 ** -lens[used].astype(np.float64))
    assert kraft <= 1.0 + 1e-12
    if used.sum() > 1:
        assert abs(kraft - 1.0) < 1e-12          # complete code
    # canonical: codes ascend with (length, symbol)
    order = sorted(np.nonzero(used)[0], key=lambda s: (lens[s], s))
    vals = [int(codes[s]) << (11 - int(lens[s])) for s in order]
    assert vals == sorted(vals) and len(set(vals)) == len(vals)
    cost = int(np.sum(hist * lens.astype(np.uint64)))
    hcost, depth = O.huffman_cost(hist)
    assert cost >= hcost
    if depth <= glc.GLC_HD_MAX_LEN:
        assert cost == hcost


@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_host_encoder_against_oracle_decoder(glc, name, data):
    hist = np.bincount(data, minlength=256).astype(np.uint64)
    lens, codes = glc.hd_build_table(hist)
    units = glc.hd_encode_host(data, lens, codes)
    assert units[-1] == 0                        # pad unit
    bits = int(np.sum(hist * lens.astype(np.uint64)))
    assert units.size == (bits + 31) // 32 + 1
    assert np.array_equal(O.hd_decode(units, lens, codes, data.size), data)


def test_encoder_rejects_symbol_without_code(glc):
    lens, codes = glc.hd_build_table(np.bincount([1, 2, 2], minlength=256))
    with pytest.raises(glc.HdError):
        glc.hd_encode_host(np.array([3], dtype=np.uint8), lens, codes)
    with pytest.raises(glc.HdError):
        glc.hd_build_table(np.zeros(256))


# ------------------------------------------------------------------ GPU -----
@pytest.mark.gpu
@pytest.mark.parametrize("name,data", CASES, ids=[c[0] for c in CASES])
def test_device_decoder_round_trip(glc, cuda, name, data):
    import torch
    hist = np.bincount(data, minlength=256).astype(np.uint64)
    lens, codes = glc.hd_build_table(hist)
    units = glc.hd_encode_host(data, lens, codes)
    d_units = torch.from_numpy(units.view(np.int32)).to(cuda)
    out = glc.hd_decode_device(d_units, lens, codes, data.size)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, data)
    assert np.array_equal(got, O.hd_decode(units, lens, codes, data.size))


@pytest.mark.gpu
def test_device_decoder_large_multi_chunk(glc, cuda):
    """> 512 workgroup functions (two walk chunks) and a non-multiple-of-span tail."""
    import torch
    data = binomial_bytes(40_000_003, 3)
    hist = np.bincount(data, minlength=256).astype(np.uint64)
    lens, codes = glc.hd_build_table(hist)
    units = glc.hd_encode_host(data, lens, codes)
    assert units.size > 512 * 8192
    d_units = torch.from_numpy(units.view(np.int32)).to(cuda)
    out = glc.hd_decode_device(d_units, lens, codes, data.size)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), torch.from_numpy(data))


# --------------------------------------------------------------------------
# streams written by the REFERENCE's encoder (tests/golden/ref_cuhd_gold.npz, made by make_cuhd_gold.py from
# cuhd-icpp/encoder/src/llhuffman_encoder.cc + src/cuhd_codetable.cc compiled unmodified)
# --------------------------------------------------------------------------
import os  # noqa: E402

import datagen  # noqa: E402

REF = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cuhd_gold.npz"))
REF_CASES = [str(s) for s in REF["cases"]]


def _ref_symbols(name):
    if name + "_symbols" in REF.files:
        return REF[name + "_symbols"]
    return datagen.symbols_from_hist(REF[name + "_hist"])


@pytest.mark.parametrize("name", REF_CASES)
def test_reference_encoded_stream_through_oracle_decoder(name):
    """CPU: the reference's units + its dictionary decode (bit-serially) to the symbols -- pins the stream shape
    this repo assumes (MSB-first 32-bit units, pad unit) and the oracle's decoder to the reference's encoder"""
    sym = _ref_symbols(name)
    got = O.hd_decode(REF[name + "_units"], REF[name + "_lens"], REF[name + "_codes"].astype(np.uint16), sym.size)
    assert np.array_equal(got, sym)
    # the reference's decoder table says the same as its dictionary
    t = REF[name + "_table"].reshape(2048, 2)
    for s in np.nonzero(REF[name + "_lens"])[0]:
        ln, code = int(REF[name + "_lens"][s]), int(REF[name + "_codes"][s])
        lo = code << (11 - ln)
        assert (t[lo:lo + (1 << (11 - ln)), 0] == ln).all() and (t[lo:lo + (1 << (11 - ln)), 1] == s).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_CASES)
def test_device_decodes_reference_encoded_stream(glc, cuda, name):
    """config 5 driven by the reference's own encoder: glcHdDecodeDeviceTable(reference units, reference table)
    == original (the reference's pass criterion, demo.cc:176-178), and the dictionary form agrees"""
    import torch
    L = glc.lib()
    sym = _ref_symbols(name)
    units = REF[name + "_units"]
    d_units = torch.from_numpy(units.view(np.int32).copy()).cuda()
    work = torch.empty(L.glcHdWorkBytes(units.size), dtype=torch.uint8, device=cuda)
    out = torch.zeros(sym.size, dtype=torch.uint8, device=cuda)
    table = np.ascontiguousarray(REF[name + "_table"])
    assert L.glcHdDecodeDeviceTable(d_units.data_ptr(), units.size, table.ctypes.data, out.data_ptr(), sym.size,
                                    work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), sym)
    back = glc.hd_decode_device(d_units, REF[name + "_lens"], REF[name + "_codes"].astype(np.uint16), sym.size)
    torch.cuda.synchronize()
    assert np.array_equal(back.cpu().numpy(), sym)
    # the table where cuhd::CUHDGPUCodetable keeps it: in device memory (glcHdDecodeDeviceTableOnDevice)
    d_table = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
    out.zero_()
    assert L.glcHdDecodeDeviceTableOnDevice(d_units.data_ptr(), units.size, d_table.data_ptr(), out.data_ptr(), sym.size,
                                            work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), sym)
