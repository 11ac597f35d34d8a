"""GPU parity tests for the cudppCompress path: HIP kernels (through the C ABI)
vs the CPU oracle, bit-exact.  Mirrors the reference's own test matrix
(apps/cudpp_testrig/test_compress.cpp:375-377,552-556,687-692; test_sa.cpp:44-46,124-126)
and adds the edge cases it lacks."""
import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu

# test_compress.cpp:375-377 (MTF sizes) / test_sa.cpp:44-46 (SA sizes)
MTF_SIZES = [39, 128, 256, 512, 1000, 1024, 1025, 32768, 45537, 65536, 131072, 262144, 500001, 524288,
             1048577, 1048576, 1048581]
SA_SIZES = [39, 128, 256, 512, 513, 1000, 1024, 1025, 32768, 45537, 65536, 131072, 262144, 500001, 524288,
            1048576]


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return "first mismatch at %d of %d (%d differ): got %s want %s" % (
        d[0], a.size, d.size, a[d[0]:d[0] + 8].tolist(), b[d[0]:d[0] + 8].tolist()) if d.size else "equal"


@pytest.fixture(scope="module")
def ctx(glc, cuda):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.fixture(scope="module")
def rand_1m():
    return O.glibc_rand_bytes(1 << 20, 255)          # test_compress.cpp:552-556


def _planted():
    x = datagen.zipf_bytes(200000, seed=77)
    motif = datagen.zipf_bytes(300, seed=78)
    for p in (1000, 50000, 50400, 120000, 199000):
        x[p:p + 300] = motif
    x[150000:150700] = 7
    return x


def edge_inputs():
    rng = np.random.default_rng(7)
    return {
        "mississippi": np.frombuffer(b"mississippi", dtype=np.uint8),
        "banana": np.frombuffer(b"banana", dtype=np.uint8),
        "single": np.array([42], dtype=np.uint8),
        "two_equal": np.array([7, 7], dtype=np.uint8),
        "all_equal_5000": np.full(5000, 0, dtype=np.uint8),
        "all_ff_4097": np.full(4097, 255, dtype=np.uint8),
        "period2_10000": np.tile(np.array([1, 2], dtype=np.uint8), 5000),
        "period3_zero_tail": np.concatenate([np.tile(np.array([9, 0, 0], dtype=np.uint8), 3000),
                                             np.zeros(17, dtype=np.uint8)]),
        "long_repeat": np.tile(rng.integers(0, 256, 777, dtype=np.uint8), 40),
        "zeros_then_random": np.concatenate([np.zeros(30000, dtype=np.uint8),
                                             rng.integers(0, 256, 30000, dtype=np.uint8)]),
        "two_symbols": rng.integers(0, 2, 50000, dtype=np.uint8),
        "text_64k": datagen.text_bytes(65536),
        # mostly shallow (text-refinement rounds) with a few long planted repeats, so the sorter
        # runs its 3 text rounds and THEN has to switch to prefix doubling for the leftovers
        "zipf_planted_repeats": _planted(),
    }


# --------------------------------------------------------------------------
# suffix array (cudppSuffixArray): test_sa.cpp
# --------------------------------------------------------------------------
def test_suffix_array_reference_sizes(glc, ctx, cuda):
    import torch
    data = O.glibc_rand_bytes(max(SA_SIZES), 128)          # test_sa.cpp:124-126
    with glc.Plan(ctx, glc.CUDPP_SA, max(SA_SIZES)) as plan:
        for n in SA_SIZES:
            x = data[:n]
            d_in = _dev(torch, x)
            d_out = torch.zeros(n + 1, dtype=torch.int32, device=cuda)
            rc = glc.lib().cudppSuffixArray(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n)
            assert rc == glc.CUDPP_SUCCESS
            got = d_out.cpu().numpy().view(np.uint32)
            want = O.suffix_array(x)
            assert got[0] == n
            assert np.array_equal(got[1:], want), "n=%d %s" % (n, _first_diff(got[1:], want))


# --------------------------------------------------------------------------
# BWT (cudppBurrowsWheelerTransform): test_compress.cpp:501-660
# --------------------------------------------------------------------------
def _bwt_gpu(glc, plan, torch, x):
    n = x.size
    d_in = _dev(torch, x)
    d_out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    d_idx = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    rc = glc.lib().cudppBurrowsWheelerTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n)
    assert rc == glc.CUDPP_SUCCESS
    return d_out.cpu().numpy(), int(d_idx.item())


def test_bwt_reference_vector(glc, ctx, cuda, rand_1m):
    import torch
    with glc.Plan(ctx, glc.CUDPP_BWT, 1 << 20) as plan:
        got, idx = _bwt_gpu(glc, plan, torch, rand_1m)
        want, widx = O.bwt(rand_1m)
        assert idx == widx == 296638
        assert np.array_equal(got, want), _first_diff(got, want)
        assert O.crc32(got) == 0xBD22BE99
        # plan reuse (the reference's plans drift after a few calls, sa_app.cu:201-202)
        for _ in range(2):
            got2, idx2 = _bwt_gpu(glc, plan, torch, rand_1m[:500001])
            want2, widx2 = O.bwt(rand_1m[:500001])
            assert idx2 == widx2 and np.array_equal(got2, want2)


@pytest.mark.parametrize("name", list(edge_inputs().keys()))
def test_bwt_edge_inputs(glc, ctx, cuda, name):
    import torch
    x = edge_inputs()[name]
    with glc.Plan(ctx, glc.CUDPP_BWT, max(x.size, 64)) as plan:
        got, idx = _bwt_gpu(glc, plan, torch, x)
        want, widx = O.bwt(x)
        assert idx == widx, "%s: index %d want %d" % (name, idx, widx)
        assert np.array_equal(got, want), "%s %s" % (name, _first_diff(got, want))


def test_bwt_worst_case_all_equal_1m(glc, ctx, cuda):
    """deepest prefix doubling: every suffix shares its whole prefix"""
    import torch
    x = np.full(1 << 20, 0x61, dtype=np.uint8)
    with glc.Plan(ctx, glc.CUDPP_BWT, 1 << 20) as plan:
        got, idx = _bwt_gpu(glc, plan, torch, x)
        assert idx == (1 << 20) - 1
        assert np.array_equal(got, x)


def test_bwt_batch_matches_single(glc, ctx, cuda):
    import torch
    n, nb = 70000, 5
    blocks = [datagen.zipf_bytes(n, seed=100 + i) for i in range(nb - 1)] + [datagen.text_bytes(n)]
    d_in = _dev(torch, np.concatenate(blocks))
    d_out = torch.zeros(n * nb, dtype=torch.uint8, device=cuda)
    d_idx = torch.zeros(nb, dtype=torch.int32, device=cuda)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=nb) as plan:
        rc = glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, nb)
        assert rc == glc.CUDPP_SUCCESS
    got = d_out.cpu().numpy().reshape(nb, n)
    idx = d_idx.cpu().numpy()
    for i, blk in enumerate(blocks):
        want, widx = O.bwt(blk)
        assert idx[i] == widx and np.array_equal(got[i], want), "block %d %s" % (i, _first_diff(got[i], want))


# --------------------------------------------------------------------------
# MTF (cudppMoveToFrontTransform): test_compress.cpp:367-499
# --------------------------------------------------------------------------
def test_mtf_reference_sizes(glc, ctx, cuda):
    import torch
    data = O.glibc_rand_bytes(max(MTF_SIZES), 255)          # test_compress.cpp:439-441
    with glc.Plan(ctx, glc.CUDPP_MTF, max(MTF_SIZES)) as plan:
        for n in MTF_SIZES:
            x = data[:n]
            d_in = _dev(torch, x)
            d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
            rc = glc.lib().cudppMoveToFrontTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n)
            assert rc == glc.CUDPP_SUCCESS
            got = d_out.cpu().numpy()
            want = O.mtf(x)
            assert np.array_equal(got, want), "n=%d %s" % (n, _first_diff(got, want))


@pytest.mark.parametrize("name", list(edge_inputs().keys()))
def test_mtf_edge_inputs(glc, ctx, cuda, name):
    import torch
    x = edge_inputs()[name]
    with glc.Plan(ctx, glc.CUDPP_MTF, max(x.size, 64)) as plan:
        d_in = _dev(torch, x)
        d_out = torch.zeros(x.size, dtype=torch.uint8, device=cuda)
        assert glc.lib().cudppMoveToFrontTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), x.size) == 0
        got = d_out.cpu().numpy()
        want = O.mtf(x)
        assert np.array_equal(got, want), "%s %s" % (name, _first_diff(got, want))


def test_mtf_of_real_bwt_output(glc, ctx, cuda):
    """skewed MTF input (long runs, rare symbols far apart)"""
    import torch
    x, _ = O.bwt(datagen.text_bytes(1 << 20))
    with glc.Plan(ctx, glc.CUDPP_MTF, 1 << 20) as plan:
        d_in = _dev(torch, x)
        d_out = torch.zeros(x.size, dtype=torch.uint8, device=cuda)
        assert glc.lib().cudppMoveToFrontTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), x.size) == 0
        got = d_out.cpu().numpy()
        want = O.mtf(x)
        assert np.array_equal(got, want), _first_diff(got, want)


# --------------------------------------------------------------------------
# cudppCompress: test_compress.cpp:662-899 (the reference only round-trips;
# here every output array is compared with the oracle)
# --------------------------------------------------------------------------
def _compress_gpu(glc, plan, torch, x):
    n = x.size
    nsub = (n + 4095) // 4096
    d_in = _dev(torch, x)
    d_idx = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    d_hist = torch.full((256,), -1, dtype=torch.int32, device="cuda")       # not pre-zeroed on purpose
    d_off = torch.full((nsub,), -1, dtype=torch.int32, device="cuda")
    d_size = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    d_comp = torch.full(((1536 + 1) * nsub,), -1, dtype=torch.int32, device="cuda")
    rc = glc.lib().cudppCompress(plan.handle, d_in.data_ptr(), d_idx.data_ptr(), None, d_hist.data_ptr(),
                                 d_off.data_ptr(), d_size.data_ptr(), d_comp.data_ptr(), n)
    assert rc == glc.CUDPP_SUCCESS
    plan.synchronize()
    size = int(d_size.item())
    return dict(bwt_index=int(d_idx.item()), hist=d_hist.cpu().numpy().view(np.uint32),
                offsets=d_off.cpu().numpy().view(np.uint32), size=size,
                words=d_comp.cpu().numpy().view(np.uint32)[:size])


def _assert_same_stream(got, want, tag=""):
    assert got["bwt_index"] == want["bwt_index"], tag
    assert np.array_equal(got["hist"], want["hist"]), tag + " hist " + _first_diff(got["hist"], want["hist"])
    assert got["size"] == want["size"], "%s size %d want %d" % (tag, got["size"], want["size"])
    assert np.array_equal(got["offsets"], want["offsets"]), tag + " offsets " + _first_diff(got["offsets"], want["offsets"])
    assert np.array_equal(got["words"], want["words"]), tag + " words " + _first_diff(got["words"], want["words"])


def test_compress_reference_vector_kat(glc, ctx, cuda, rand_1m):
    import torch
    x = rand_1m.copy()
    x[-1] = 0                                               # test_compress.cpp:687-692
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, 1 << 20) as plan:
        got = _compress_gpu(glc, plan, torch, x)
    want = O.compress(x)
    _assert_same_stream(got, want, "compressTest vector")
    # BASELINE.md section 4 known answers
    assert got["bwt_index"] == 296638 and got["size"] == 262491
    assert O.crc32(got["words"].view(np.uint8)) == 0xE12686DC
    assert O.crc32(got["offsets"].view(np.uint8)) == 0x62C9B10A
    assert O.crc32(got["hist"].view(np.uint8)) == 0xAAAAA264
    # and the gold decoder reproduces the input
    back = O.decompress(got["bwt_index"], got["hist"], got["offsets"], got["words"], x.size)
    assert np.array_equal(back, x)


@pytest.mark.parametrize("gen", ["zipf", "float", "text", "log"])
def test_compress_1m_synthetic(glc, ctx, cuda, gen):
    import torch
    x = {"zipf": datagen.zipf_bytes, "float": datagen.float_bytes, "text": datagen.text_bytes,
         "log": datagen.log_bytes}[gen](1 << 20)
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, 1 << 20) as plan:
        got = _compress_gpu(glc, plan, torch, x)
    _assert_same_stream(got, O.compress(x), gen)


@pytest.mark.parametrize("n", [4096, 32768, 65536, 98304, 524288])
def test_compress_smaller_multiples(glc, ctx, cuda, n):
    import torch
    x = datagen.text_bytes(n, seed=n)
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, 1 << 20) as plan:
        got = _compress_gpu(glc, plan, torch, x)
    _assert_same_stream(got, O.compress(x), "n=%d" % n)


@pytest.mark.parametrize("n", [1, 39, 4095, 4097, 45537, 500001])
def test_compress_ragged_sizes(glc, ctx, cuda, n):
    """extension: sizes the reference leaves undefined (tail block < 4096 symbols)"""
    import torch
    x = datagen.zipf_bytes(n, seed=n)
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, 1 << 20) as plan:
        got = _compress_gpu(glc, plan, torch, x)
    want = O.compress(x)
    _assert_same_stream(got, want, "n=%d" % n)
    assert np.array_equal(O.decompress(got["bwt_index"], got["hist"], got["offsets"], got["words"], n), x)


def test_compress_batch_matches_single_calls(glc, ctx, cuda):
    import torch
    n, nb = 1 << 18, 6
    blocks = [datagen.zipf_bytes(n, seed=1), datagen.float_bytes(n, seed=2), datagen.text_bytes(n, seed=3),
              datagen.log_bytes(n, seed=4), np.zeros(n, dtype=np.uint8), datagen.zipf_bytes(n, seed=5, s=2.0)]
    d_in = _dev(torch, np.concatenate(blocks))
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
        out = glc.compress_batch(plan, d_in, n, nb)
        plan.synchronize()
    stride, nsub = out["stride"], out["nsub"]
    for i, blk in enumerate(blocks):
        size = int(out["size"][i].item())
        got = dict(bwt_index=int(out["bwt_index"][i].item()),
                   hist=out["hist"][i * 256:(i + 1) * 256].cpu().numpy().view(np.uint32),
                   offsets=out["offsets"][i * nsub:(i + 1) * nsub].cpu().numpy().view(np.uint32), size=size,
                   words=out["words"][i * stride:i * stride + size].cpu().numpy().view(np.uint32))
        _assert_same_stream(got, O.compress(blk), "block %d" % i)


# --------------------------------------------------------------------------
# error behaviour: cudpp.cpp:782-805, cudpp_plan.cpp:29-46,147-190
# --------------------------------------------------------------------------
def test_error_codes(glc, ctx, cuda):
    import ctypes as C
    import torch
    L = glc.lib()
    h = C.c_size_t(0)
    # conflicting options
    rc = L.cudppPlan(ctx.handle, C.byref(h), glc.config(glc.CUDPP_COMPRESS, options=glc.CUDPP_OPTION_FORWARD | glc.CUDPP_OPTION_BACKWARD), 1 << 20, 1, 0)
    assert rc == glc.CUDPP_ERROR_ILLEGAL_CONFIGURATION and h.value == glc.CUDPP_INVALID_HANDLE
    # algorithm that is not on this path
    assert L.cudppPlan(ctx.handle, C.byref(h), glc.config(glc.CUDPP_SCAN), 1024, 1, 0) == glc.CUDPP_ERROR_ILLEGAL_CONFIGURATION
    # too large for COMPRESS (cudpp-inpar/README.md:96)
    assert L.cudppPlan(ctx.handle, C.byref(h), glc.config(glc.CUDPP_COMPRESS), (1 << 20) + 1, 1, 0) == glc.CUDPP_ERROR_ILLEGAL_CONFIGURATION
    d = torch.zeros(4096, dtype=torch.uint8, device=cuda)
    o = torch.zeros(4096 * 8, dtype=torch.int32, device=cuda)
    with glc.Plan(ctx, glc.CUDPP_MTF, 4096) as mtf_plan, glc.Plan(ctx, glc.CUDPP_COMPRESS, 4096, datatype=glc.CUDPP_UINT) as bad_dt:
        # NULL plan -> INVALID_HANDLE ; wrong plan type -> INVALID_PLAN ; wrong datatype -> ILLEGAL_CONFIGURATION
        assert L.cudppCompress(0, d.data_ptr(), o.data_ptr(), None, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr(), 4096) == glc.CUDPP_ERROR_INVALID_HANDLE
        assert L.cudppCompress(mtf_plan.handle, d.data_ptr(), o.data_ptr(), None, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr(), 4096) == glc.CUDPP_ERROR_INVALID_PLAN
        assert L.cudppCompress(bad_dt.handle, d.data_ptr(), o.data_ptr(), None, o.data_ptr(), o.data_ptr(), o.data_ptr(), o.data_ptr(), 4096) == glc.CUDPP_ERROR_ILLEGAL_CONFIGURATION
        assert L.cudppBurrowsWheelerTransform(mtf_plan.handle, d.data_ptr(), d.data_ptr(), o.data_ptr(), 4096) == glc.CUDPP_ERROR_INVALID_PLAN
        assert L.cudppMoveToFrontTransform(mtf_plan.handle, d.data_ptr(), d.data_ptr(), 8192) == glc.CUDPP_ERROR_ILLEGAL_CONFIGURATION
    assert L.cudppDestroyPlan(glc.CUDPP_INVALID_HANDLE) == glc.CUDPP_ERROR_INVALID_HANDLE


def test_pipelined_calls_match_plain_calls(glc, ctx, cuda):
    """glcPlanSetPipelining: six back-to-back batched calls (suffix sort of call i+1 overlapping the
    MTF + Huffman stages of call i on a second stream) give the same streams as plain calls, with the
    outputs read through ordinary stream-ordered copies on the plan's stream."""
    import torch
    n, nb, calls = 1 << 17, 4, 6
    batches = [np.concatenate([datagen.zipf_bytes(n, seed=1000 + 10 * c + b) if (b + c) % 2 else
                               datagen.text_bytes(n, seed=2000 + 10 * c + b) for b in range(nb)]) for c in range(calls)]
    d_in = [torch.from_numpy(x).cuda() for x in batches]

    def run(pipelined):
        outs = []
        with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
            if pipelined:
                plan.set_pipelining(True)
            res = [glc.compress_batch(plan, d_in[c], n, nb) for c in range(calls)]   # no sync in between
            plan.synchronize()
            for r in res:
                sizes = r["size"].cpu().numpy()
                outs.append((r["bwt_index"].cpu().numpy().copy(), sizes.copy(), r["hist"].cpu().numpy().copy(),
                             [r["words"][b * r["stride"]: b * r["stride"] + int(sizes[b])].cpu().numpy().copy()
                              for b in range(nb)]))
        return outs

    plain, piped = run(False), run(True)
    for c in range(calls):
        assert np.array_equal(plain[c][0], piped[c][0]) and np.array_equal(plain[c][1], piped[c][1])
        assert np.array_equal(plain[c][2], piped[c][2])
        for b in range(nb):
            assert np.array_equal(plain[c][3][b], piped[c][3][b])
    want = O.compress(batches[3][2 * n:3 * n])                  # and one block against the oracle
    assert np.array_equal(piped[3][3][2].view(np.uint32), want["words"])


def test_randomised_parity_sweep(glc, ctx, cuda):
    """60 seeded random inputs (sizes 1..70 000, alphabets 1..256, i.i.d. / runs / periodic / tandem
    repeats) through ONE reused plan, every output array compared with the oracle."""
    import torch
    rng = np.random.default_rng(20260928)
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, 70000, rows=1) as plan:
        for case in range(60):
            n = int(rng.integers(1, 70001)) if case % 5 else int(rng.choice([1, 2, 3, 5, 64, 4095, 4096, 4097, 8192, 65536]))
            a = int(rng.choice([1, 2, 3, 16, 255, 256]))
            kind = case % 4
            if kind == 0:
                x = rng.integers(0, a, n, dtype=np.uint16).astype(np.uint8)
            elif kind == 1:                                        # long runs
                x = np.repeat(rng.integers(0, a, n // 7 + 1, dtype=np.uint16), rng.integers(1, 14, n // 7 + 1))[:n].astype(np.uint8)
                x = np.resize(x, n)
            elif kind == 2:                                        # periodic
                per = int(rng.integers(1, 40))
                x = np.resize(rng.integers(0, a, per, dtype=np.uint16).astype(np.uint8), n)
            else:                                                  # tandem repeats with mutations
                unit = rng.integers(0, a, int(rng.integers(2, 300)), dtype=np.uint16).astype(np.uint8)
                x = np.resize(unit, n).copy()
                flips = rng.integers(0, n, max(1, n // 500))
                x[flips] = rng.integers(0, a, flips.size, dtype=np.uint16).astype(np.uint8)
            want = O.compress(x)
            d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
            r = glc.compress_batch(plan, d_in, n, 1)
            plan.synchronize()
            size = int(r["size"].item())
            tag = "case %d n=%d alphabet=%d kind=%d" % (case, n, a, kind)
            assert int(r["bwt_index"].item()) == want["bwt_index"], tag
            assert size == want["size"], tag
            assert np.array_equal(r["hist"].cpu().numpy().view(np.uint32), want["hist"]), tag
            assert np.array_equal(r["words"][:size].cpu().numpy().view(np.uint32), want["words"]), tag
            nsub = (n + 4095) // 4096
            assert np.array_equal(r["offsets"][:nsub].cpu().numpy().view(np.uint32), want["offsets"]), tag
            back = glc.decompress_batch(plan, r, n, 1)
            plan.synchronize()
            assert np.array_equal(back.cpu().numpy(), x), tag + " round trip"
