"""Blocks with repeats deeper than the sample sorter's cap INSIDE otherwise ordinary data (zero pages, a duplicated region,
a long phrase, long runs in log lines): the sample sorter runs once more in its tolerant form (suffixes that agree in more
than SS_TOL_CAP symbols stay as they come), the groups of rows that still tie are found by looking, and prefix doubling
RESUMES from that depth (bwt_sa.hip sa_build_finish, k_grp_*).  Same bytes as the oracle and as the general sorter from
scratch (glcPlanSetSorter 5); glcPlanLastSortResumed says which way a call's blocks went."""
import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
N = 1 << 20


def _bwt(glc, plan, torch, x, rows):
    n = x.size // rows
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_out = torch.zeros(x.size, dtype=torch.uint8, device=d_in.device)
    d_idx = torch.zeros(rows, dtype=torch.int32, device=d_in.device)
    assert glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), d_idx.cpu().numpy()


def _zero_pages(n, seed):
    x = datagen.zipf_bytes(n, seed=seed).copy()
    for off in (n // 25, n // 3 + 123, n - n // 9):
        x[off:off + 4096] = 0
    return x


def _duplicate(n, seed):
    x = datagen.text_bytes(n, seed=seed).copy()
    ln = n // 50
    x[n // 2 + 77:n // 2 + 77 + ln] = x[n // 10:n // 10 + ln]
    return x


def _phrase(n, seed):
    x = datagen.text_bytes(n, seed=seed).copy()
    ph = np.random.default_rng(seed).integers(97, 123, 2000, dtype=np.uint8)
    for off in range(5000, n - 2000, 16384):
        x[off:off + 2000] = ph
    return x


def _log_runs(n, seed):
    x = datagen.log_bytes(n, seed=seed).copy()
    x[n // 5:n // 5 + 1500] = 32
    x[n - n // 3:n - n // 3 + 9000] = 0
    return x


def _tail_run(n, seed):
    """the deep part reaches the end of the block: suffixes shorter than the cap inside a run"""
    x = datagen.text_bytes(n, seed=seed).copy()
    x[n - 3000:] = 65
    return x


GENS = {"zero_pages": _zero_pages, "duplicate": _duplicate, "phrase": _phrase, "log_runs": _log_runs, "tail_run": _tail_run}


@pytest.fixture(scope="module")
def ctx(glc):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.mark.parametrize("name", list(GENS.keys()))
def test_partly_deep_blocks_resume(glc, ctx, cuda, name):
    """four different blocks of one kind in a call: all given up on for depth, all finished by the resumed doubling"""
    import torch
    blocks = [GENS[name](N, 100 + 7 * i) for i in range(4)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=4) as plan:
        for mode, resumed in ((0, 4), (5, 0), (0, 4)):          # resumed, from scratch, and the plan is reusable
            plan.set_sorter(mode)
            got, gidx = _bwt(glc, plan, torch, x, 4)
            assert plan.last_sort_stats() == (4, 4) and plan.last_sort_resumed() == resumed, (mode, plan.last_sort_stats(), plan.last_sort_resumed())
            for i, blk in enumerate(blocks):
                want, widx = O.bwt(blk)
                assert int(gidx[i]) == widx and np.array_equal(got[i * N:(i + 1) * N], want), "%s block %d (sorter %d)" % (name, i, mode)


def test_resume_in_a_mixed_batch(glc, ctx, cuda):
    """every way a block can go, in one call: bucket sorter, sample sorter, resumed doubling, general sorter from scratch
    (periodic data: the tolerant form gives it up at sampling), a block of one symbol"""
    import torch
    blocks = [datagen.zipf_bytes(N, seed=1), _duplicate(N, 2), datagen.text_bytes(N, seed=3), _phrase(N, 4),
              np.tile(np.frombuffer(b"xy", dtype=np.uint8), N // 2), _log_runs(N, 5), np.zeros(N, dtype=np.uint8),
              _tail_run(N, 6), np.tile(np.random.default_rng(7).integers(0, 256, 4096, dtype=np.uint8), N // 4096)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=len(blocks)) as plan:
        for rep in range(2):
            got, gidx = _bwt(glc, plan, torch, x, len(blocks))
            assert plan.last_sort_stats() == (7, 6), plan.last_sort_stats()   # flagged: all but Zipf and the zeros; given up on: the six deep ones
            assert plan.last_sort_resumed() == 4, plan.last_sort_resumed()    # the two periodic blocks take the general sorter from scratch
            for i, blk in enumerate(blocks):
                want, widx = O.bwt(blk)
                assert int(gidx[i]) == widx and np.array_equal(got[i * N:(i + 1) * N], want), "block %d (call %d)" % (i, rep)


@pytest.mark.parametrize("n", [70000, 300001, 999999])
def test_resume_other_block_sizes(glc, ctx, cuda, n):
    import torch
    blocks = [_duplicate(n, 11), _tail_run(n, 12), _log_runs(n, 13), _zero_pages(n, 14)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=4) as plan:
        plan.set_sorter(6)                                     # resume whatever the count
        got, gidx = _bwt(glc, plan, torch, x, 4)
        given_up = plan.last_sort_stats()[1]
        assert plan.last_sort_resumed() == given_up
        for i, blk in enumerate(blocks):
            want, widx = O.bwt(blk)
            assert int(gidx[i]) == widx and np.array_equal(got[i * n:(i + 1) * n], want), "n = %d block %d" % (n, i)


def test_suffix_array_resumed(glc, ctx, cuda):
    """cudppSuffixArray (one block, the suffix array itself is the result) through the resumed path"""
    import torch
    n = 400000
    x = _duplicate(n, 21)
    with glc.Plan(ctx, glc.CUDPP_SA, n) as plan:
        plan.set_sorter(6)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros(n + 1, dtype=torch.int32, device=d_in.device)
        assert glc.lib().cudppSuffixArray(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n) == glc.CUDPP_SUCCESS
        got = d_out.cpu().numpy().view(np.uint32)
        assert got[0] == n and np.array_equal(got[1:], O.suffix_array(x))
        assert plan.last_sort_stats() == (1, 1) and plan.last_sort_resumed() == 1


def test_compress_round_trip_with_resumed_blocks(glc, ctx, cuda):
    """cudppCompress-path batch whose blocks take the resumed path: streams equal the oracle's, decode gives the input back"""
    import torch
    blocks = [_duplicate(N, 31), _phrase(N, 32), _log_runs(N, 33), _tail_run(N, 34), datagen.zipf_bytes(N, seed=35)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=len(blocks)) as plan:
        d_in = torch.from_numpy(x).cuda()
        out = glc.compress_batch(plan, d_in, N, len(blocks))
        plan.synchronize()
        assert plan.last_sort_resumed() == 4
        back = glc.decompress_batch(plan, out, N, len(blocks))
        plan.synchronize()
        assert np.array_equal(back.cpu().numpy(), x)
        for i, blk in enumerate(blocks):                       # and it is the reference's stream: same words as the oracle's
            want = O.compress(blk)
            size = int(out["size"][i].item())
            assert size == want["size"] and int(out["bwt_index"][i].item()) == want["bwt_index"], i
            words = out["words"][i * out["stride"]:i * out["stride"] + size].cpu().numpy().view(np.uint32)
            assert np.array_equal(words, want["words"]), "block %d" % i
