"""GPU results against REFERENCE-PRODUCED vectors, with nothing of this repo's oracle in between:

  tests/golden/ref_gold_small.npz  suffix arrays of the reference's computeSaGold (sa_gold.cpp, compiled unmodified),
                                   BWT / MTF at the small sizes of its test matrix
  tests/golden/ref_huff_gold.npz   code lengths of the reference's huffman_build_tree_cpu + FindMinimumCountTest on
                                   tie-heavy histograms, and streams its own gold decoder accepts
                                   (tests/golden/make_huff_gold.py; oracle/mk_ref_compress_gold.sh)
Checksums are zlib's CRC-32 (the same reflected 0xEDB88320 the fixtures were made with)."""
import os
import zlib

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = np.load(os.path.join(GOLD, "ref_gold_small.npz"))
HUFF = np.load(os.path.join(GOLD, "ref_huff_gold.npz"))
N = 1 << 20


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def ctx(glc, cuda):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.mark.parametrize("n", [39, 128, 256, 512, 513, 1000, 1024, 1025, 32768, 45537, 65536])
def test_suffix_array_vs_reference_gold(glc, ctx, cuda, n):
    """cudppSuffixArray == computeSaGold on the reference's own test input (test_sa.cpp:124-126)"""
    import torch
    x, want = SMALL["sa_in_%d" % n], SMALL["sa_out_%d" % n]
    with glc.Plan(ctx, glc.CUDPP_SA, n) as plan:
        d_out = torch.zeros(n + 1, dtype=torch.int32, device=cuda)
        rc = glc.lib().cudppSuffixArray(plan.handle, torch.from_numpy(x.copy()).cuda().data_ptr(), d_out.data_ptr(), n)
        assert rc == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
    assert got[0] == n and np.array_equal(got[1:], want)       # result at d_out + 1 (test_sa.cpp:112-113,161)


@pytest.mark.parametrize("n", [39, 128, 1000, 1025, 45537, 65536])
def test_bwt_and_mtf_vs_reference_gold(glc, ctx, cuda, n):
    import torch
    L = glc.lib()
    x = SMALL["bwt_in_%d" % n]
    d_in = torch.from_numpy(x.copy()).cuda()
    with glc.Plan(ctx, glc.CUDPP_BWT, n) as plan:
        d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
        d_idx = torch.zeros(1, dtype=torch.int32, device=cuda)
        assert L.cudppBurrowsWheelerTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n) == 0
        torch.cuda.synchronize()
        assert int(d_idx.item()) == int(SMALL["bwt_idx_%d" % n][0])
        assert np.array_equal(d_out.cpu().numpy(), SMALL["bwt_out_%d" % n])
    with glc.Plan(ctx, glc.CUDPP_MTF, n) as plan:
        d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
        assert L.cudppMoveToFrontTransform(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), SMALL["mtf_of_in_%d" % n])


def _check_stream(out, key, tag):
    size = int(out["size"][0].item())
    assert size == int(HUFF[key + "_size"][0]), tag + ": words"
    words = out["words"][:size].cpu().numpy().view(np.uint32)
    assert np.array_equal(words[:64], HUFF[key + "_head_words"][:64]), tag + ": first words"
    assert _crc(words) == int(HUFF[key + "_crc_words"][0]), tag + ": stream CRC"
    assert _crc(out["offsets"].cpu().numpy().view(np.uint32)) == int(HUFF[key + "_crc_offsets"][0]), tag + ": offsets"
    assert np.array_equal(out["hist"].cpu().numpy().view(np.uint32), HUFF[key + "_hist"]), tag + ": histogram"


@pytest.mark.parametrize("name", [str(s) for s in HUFF["hist_cases"]])
def test_huffman_stage_vs_reference_tree(glc, ctx, cuda, name):
    """tree tie-breaks (count, level, slot), relocation rule, code assignment, packer, offsets: the HIP stream for
    symbols with a chosen histogram == the stream built from the REFERENCE's tree (and read by its gold decoder)"""
    import torch
    sym = datagen.symbols_from_hist(HUFF["h_%s_hist" % name])
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=1) as plan:
        out = glc.huffman_encode_batch(plan, torch.from_numpy(sym).cuda(), N, 1)
        plan.synchronize()
        _check_stream(out, "h_" + name, name)


@pytest.mark.parametrize("name", [str(s) for s in HUFF["e2e_cases"]])
def test_cudppcompress_vs_reference_gold_chain(glc, ctx, cuda, name):
    """input -> computeBwtGold -> computeMtfGold -> huffman_build_tree_cpu (all the reference's lines) -> stream"""
    import torch
    if name == "ref_compressTest":
        x = datagen.glibc_rand_bytes(N, 255); x[-1] = 0
    else:
        x = {"zipf": datagen.zipf_bytes, "float": datagen.float_bytes, "text": datagen.text_bytes}[name](N)
    assert _crc(x) == int(HUFF["e_%s_crc_in" % name][0]), "input generator drifted"
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=1) as plan:
        out = glc.compress_batch(plan, torch.from_numpy(x).cuda(), N, 1)
        plan.synchronize()
        assert int(out["bwt_index"][0].item()) == int(HUFF["e_%s_bwt_index" % name][0])
        _check_stream(out, "e_" + name, name)
    with glc.Plan(ctx, glc.CUDPP_BWT, N) as plan:
        d_out = torch.zeros(N, dtype=torch.uint8, device=cuda)
        d_idx = torch.zeros(1, dtype=torch.int32, device=cuda)
        assert glc.lib().cudppBurrowsWheelerTransform(plan.handle, torch.from_numpy(x).cuda().data_ptr(),
                                                      d_out.data_ptr(), d_idx.data_ptr(), N) == 0
        torch.cuda.synchronize()
        assert _crc(d_out.cpu().numpy()) == int(HUFF["e_%s_crc_bwt" % name][0])
