"""The bucket sorter (bwt_bucket.hip) and its hand-over to the general sorter (bwt_sa.hip): inputs chosen to sit on
every exit of the fast path -- no ties, ties resolved by text comparison at growing common-prefix lengths, common
prefixes past the cap, work lists that fill up, buckets that overflow -- each checked bit-exactly against the oracle,
and the three sorter modes (glcPlanSetSorter) against each other."""
import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
N = 1 << 20


def _bwt(glc, plan, torch, x, rows=1):
    n = x.size // rows
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_out = torch.zeros(x.size, dtype=torch.uint8, device=d_in.device)
    d_idx = torch.zeros(rows, dtype=torch.int32, device=d_in.device)
    assert glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_idx.cpu().numpy()


def _planted(base, seg_len, copies, seed):
    """random bytes with one random segment planted `copies` times: suffixes inside the copies tie for up to seg_len bytes"""
    rng = np.random.default_rng(seed)
    x = base.copy()
    seg = rng.integers(0, 256, seg_len, dtype=np.uint8)
    for p in rng.choice((x.size - seg_len) // seg_len, copies, replace=False):
        x[p * seg_len:(p + 1) * seg_len] = seg
    return x


CASES = {
    # name: (generator, expect_general_sorter)   None = either
    "zipf": (lambda: datagen.zipf_bytes(N), False),
    "float": (lambda: datagen.float_bytes(N), False),
    "uniform": (lambda: np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8), False),
    # 6 symbols of a 1-bit source carry 6 bits of code: everything ties, the block takes the general sorter
    "two_symbols_iid": (lambda: np.random.default_rng(2).integers(0, 2, N, dtype=np.uint8) * 255, True),
    "sixteen_symbols_iid": (lambda: np.random.default_rng(3).integers(0, 16, N, dtype=np.uint8), None),
    "ties_lcp_40": (lambda: _planted(datagen.zipf_bytes(N, seed=5), 40, 64, 5), False),
    "ties_lcp_300": (lambda: _planted(datagen.zipf_bytes(N, seed=6), 300, 16, 6), False),
    "ties_lcp_500": (lambda: _planted(np.random.default_rng(7).integers(0, 256, N, dtype=np.uint8), 500, 8, 7), False),
    "lcp_past_cap_2000": (lambda: _planted(np.random.default_rng(8).integers(0, 256, N, dtype=np.uint8), 2000, 4, 8), True),
    "chunk_repeated_16x": (lambda: np.tile(np.random.default_rng(9).integers(0, 256, N // 16, dtype=np.uint8), 16), True),
    "text": (lambda: datagen.text_bytes(N), True),
    "log": (lambda: datagen.log_bytes(N), True),
    "zeros": (lambda: np.zeros(N, dtype=np.uint8), False),               # one symbol: finished by k_fs_tables, no tier runs
    "zeros_but_one": (lambda: np.concatenate([np.zeros(N - 1, dtype=np.uint8), np.ones(1, dtype=np.uint8)]), True),
    "run_inside_random": (lambda: np.concatenate([np.random.default_rng(10).integers(0, 256, N // 2, dtype=np.uint8),
                                                  np.full(5000, 7, dtype=np.uint8),
                                                  np.random.default_rng(11).integers(0, 256, N // 2 - 5000, dtype=np.uint8)]), None),
    "ends_in_zeros": (lambda: np.concatenate([datagen.zipf_bytes(N - 9, seed=12), np.zeros(9, dtype=np.uint8)]), False),
}


@pytest.fixture(scope="module")
def ctx(glc, cuda):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_bucket_sorter_exits(glc, ctx, cuda, name):
    import torch
    gen, expect_general = CASES[name]
    x = gen()
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=1) as plan:
        got, gidx = _bwt(glc, plan, torch, x)
        flagged = plan.last_flagged_blocks()
        assert int(gidx[0]) == widx and np.array_equal(got, want), name
        if expect_general is not None:
            assert (flagged == 1) == expect_general, "%s: %d block(s) went to the general sorter" % (name, flagged)
        for mode in (1, 2):                                    # general sorter, text refinement / prefix doubling only
            plan.set_sorter(mode)
            g2, i2 = _bwt(glc, plan, torch, x)
            assert int(i2[0]) == widx and np.array_equal(g2, want), "%s mode %d" % (name, mode)
            assert plan.last_flagged_blocks() == 1
        plan.set_sorter(0)


def test_mixed_batch_only_flagged_blocks_fall_back(glc, ctx, cuda):
    """a batch in which some blocks are flagged: the others must keep the bucket sorter's result"""
    import torch
    blocks = [datagen.zipf_bytes(N, seed=1), datagen.text_bytes(N, seed=2), datagen.float_bytes(N, seed=3),
              np.zeros(N, dtype=np.uint8), datagen.zipf_bytes(N, seed=4), datagen.log_bytes(N, seed=5)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=len(blocks)) as plan:
        got, gidx = _bwt(glc, plan, torch, x, rows=len(blocks))
        assert plan.last_flagged_blocks() == 2                 # text and log; the all-zero block is no tier's business
        for i, blk in enumerate(blocks):
            want, widx = O.bwt(blk)
            assert int(gidx[i]) == widx and np.array_equal(got[i * N:(i + 1) * N], want), "block %d" % i


@pytest.mark.parametrize("n", [1, 2, 7, 64, 2047, 2048, 2049, 4096, 100000, 524289, 1048575])
def test_bucket_sorter_block_sizes(glc, ctx, cuda, n):
    """bucket count and slot geometry change with n (1 bucket below 2049 suffixes, 512 at 1 MiB)"""
    import torch
    x = datagen.zipf_bytes(max(n, 8), seed=n)[:n]
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
        got, gidx = _bwt(glc, plan, torch, x)
        assert int(gidx[0]) == widx and np.array_equal(got, want)
        assert plan.last_flagged_blocks() == 0


def test_suffix_array_result_through_the_bucket_sorter(glc, ctx, cuda):
    """cudppSuffixArray asks for the array itself: rows written by k_fs_sort / k_fs_ties"""
    import torch
    x = _planted(datagen.zipf_bytes(N, seed=21), 100, 32, 21)
    want = O.suffix_array(x)
    with glc.Plan(ctx, glc.CUDPP_SA, N) as plan:
        d_out = torch.zeros(N + 1, dtype=torch.int32, device=cuda)
        assert glc.lib().cudppSuffixArray(plan.handle, torch.from_numpy(x).cuda().data_ptr(), d_out.data_ptr(), N) == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
        assert plan.last_flagged_blocks() == 0
    assert got[0] == N and np.array_equal(got[1:], want)
