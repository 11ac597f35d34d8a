"""Seeded random mixtures through every sorter tier and the whole pipeline: blocks glued from pieces of different
kinds (i.i.d. bytes, text, log lines, runs, repeats, periodic data, small alphabets), random block sizes, random
batch shapes.  BWT bytes + index against the oracle for every block; compress -> decompress back to the input;
streams identical to the oracle's for a sample."""
import os

import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu

_OFFSET = 100000 * int(os.environ.get("GLC_FUZZ_OFFSET", "0"))      # other seed ranges for one-off sweeps


def _piece(rng, n):
    kind = int(rng.integers(0, 10))
    seed = int(rng.integers(1, 1 << 30))
    if kind == 0:
        return datagen.zipf_bytes(n, seed=seed, s=float(rng.uniform(0.5, 2.0)))
    if kind == 1:
        return datagen.text_bytes(n, seed=seed)
    if kind == 2:
        return datagen.log_bytes(n, seed=seed)
    if kind == 3:
        return datagen.float_bytes(n, seed=seed)
    if kind == 4:
        return np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
    if kind == 5:
        per = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
        return np.tile(per, n // per.size + 1)[:n]
    if kind == 6:
        return rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8) + np.uint8(rng.integers(0, 250))
    if kind == 7:
        chunk = rng.integers(0, 256, max(1, n // int(rng.integers(2, 9))), dtype=np.uint8)
        return np.tile(chunk, n // chunk.size + 1)[:n]
    if kind == 8:
        return rng.integers(0, 256, n, dtype=np.uint8)
    base = datagen.text_bytes(n, seed=seed)
    base[rng.integers(0, n, max(1, n // 50))] = rng.integers(0, 256, max(1, n // 50), dtype=np.uint8)   # text with typos
    return base


def _block(rng, n):
    parts, left = [], n
    while left > 0:
        m = left if rng.random() < 0.35 else int(rng.integers(1, left + 1))
        parts.append(_piece(rng, m))
        left -= m
    return np.concatenate(parts)[:n]


@pytest.fixture(scope="module")
def ctx(glc, cuda):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.mark.parametrize("seed", list(range(24)))
def test_fuzz_bwt_batches(glc, ctx, cuda, seed):
    import torch
    rng = np.random.default_rng(1000 + seed + _OFFSET)
    n = int(rng.choice([1, 3, 64, 2048, 4097, 50000, 262144, 1 << 20, int(rng.integers(2, 1 << 20))]))
    rows = int(rng.integers(1, 6))
    x = np.concatenate([_block(rng, n) for _ in range(rows)])
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
        for mode in (0, 4):
            plan.set_sorter(mode)
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.zeros(x.size, dtype=torch.uint8, device=d_in.device)
            d_idx = torch.zeros(rows, dtype=torch.int32, device=d_in.device)
            assert glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
            torch.cuda.synchronize()
            got, gidx = d_out.cpu().numpy(), d_idx.cpu().numpy()
            for i in range(rows):
                want, widx = O.bwt(x[i * n:(i + 1) * n])
                ok = int(gidx[i]) == widx and np.array_equal(got[i * n:(i + 1) * n], want)
                assert ok, "seed %d n %d block %d mode %d tiers %r" % (seed, n, i, mode, plan.last_sort_stats())


@pytest.mark.parametrize("seed", list(range(6)))
def test_fuzz_compress_round_trip(glc, ctx, cuda, seed):
    import torch
    rng = np.random.default_rng(2000 + seed + _OFFSET)
    n = int(rng.choice([4096, 70000, 1 << 19, 1 << 20]))
    rows = int(rng.integers(1, 5))
    x = np.concatenate([_block(rng, n) for _ in range(rows)])
    d_in = torch.from_numpy(x).cuda()
    # a 4096-symbol block whose codes need more than the 1536 words the format gives it (cudpp_globals.h:66; glued
    # pieces can do that: rare symbols of one piece under the tree of the whole block): the oracle says so and the
    # library must report it (include/cudpp.h: CUDPP_ERROR_UNKNOWN from glcPlanSynchronize), not produce a stream
    overflow = any(O.compress(x[i * n:(i + 1) * n])["rc"] != 0 for i in range(rows))
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
        if overflow:
            glc.compress_batch(plan, d_in, n, rows)
            with pytest.raises(glc.CudppError):
                plan.synchronize()
            ok = torch.from_numpy(datagen.zipf_bytes(n * rows, seed=7)).cuda()     # the plan is usable afterwards
            comp = glc.compress_batch(plan, ok, n, rows)
            plan.synchronize()
            assert torch.equal(glc.decompress_batch(plan, comp, n, rows), ok)
            return
        for rep in range(2):                                   # the second call sees the plan's memory of the first
            comp = glc.compress_batch(plan, d_in, n, rows)
            plan.synchronize()
            back = glc.decompress_batch(plan, comp, n, rows)
            torch.cuda.synchronize()
            assert np.array_equal(back.cpu().numpy(), x), "seed %d n %d call %d" % (seed, n, rep)
        want = O.compress(x[:n])
        size = int(comp["size"][0].item())
        assert size == want["size"] and int(comp["bwt_index"][0].item()) == want["bwt_index"]
        assert np.array_equal(comp["words"][:size].cpu().numpy().view(np.uint32), want["words"])


def test_compress_reports_a_block_that_does_not_fit(glc, ctx, cuda):
    """one of the one-off sweeps' inputs (GLC_FUZZ_OFFSET=4, seed 5): glued pieces whose third 512 KiB block has a
    4096-symbol piece that needs 1544 words under the block's tree -- more than the format's 1536 (cudpp_globals.h:66).
    The oracle reports it, the library must too, and the plan must go on working."""
    import torch
    rng = np.random.default_rng(2000 + 5 + 400000)
    n = int(rng.choice([4096, 70000, 1 << 19, 1 << 20]))
    rows = int(rng.integers(1, 5))
    x = np.concatenate([_block(rng, n) for _ in range(rows)])
    assert [O.compress(x[i * n:(i + 1) * n])["rc"] for i in range(rows)] == [0, 0, 1, 0]
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
        glc.compress_batch(plan, torch.from_numpy(x).cuda(), n, rows)
        with pytest.raises(glc.CudppError):
            plan.synchronize()
        ok = torch.from_numpy(datagen.zipf_bytes(n * rows, seed=7)).cuda()
        comp = glc.compress_batch(plan, ok, n, rows)
        plan.synchronize()
        assert torch.equal(glc.decompress_batch(plan, comp, n, rows), ok)


# --- the sweeps DESIGN.md quotes, reproducible: python -m pytest tests/test_gpu_fuzz.py -m gpu_long -q  (about 10 minutes) ---
@pytest.mark.gpu_long
@pytest.mark.parametrize("offset", list(range(1, 58)))
def test_fuzz_sweep_long(glc, ctx, cuda, offset):
    """57 further seed ranges of the two fuzz tests above (24 BWT batches + 6 round trips each: 1 710 cases)"""
    global _OFFSET
    saved = _OFFSET
    _OFFSET = 100000 * offset
    try:
        for seed in range(24):
            test_fuzz_bwt_batches(glc, ctx, cuda, seed)
        for seed in range(6):
            test_fuzz_compress_round_trip(glc, ctx, cuda, seed)
    finally:
        _OFFSET = saved


@pytest.mark.parametrize("kind", ["two_symbols_sampled", "sample_constant", "high_symbols_missed", "low_symbols_missed"])
def test_symbol_statistics_from_a_sample_of_the_block(glc, ctx, cuda, kind):
    """blocks of 16 slices or more take their {C, p} table from every fourth 32 KB slice (k_fs_hist / k_fs_tables): blocks
    whose sampled slices are NOT like the rest -- symbols the sample never saw above, below and between the ones it saw
    (seed 2 of the round-trip fuzz failed on the first form before every symbol had a floor of one count), a sample of one
    symbol in a block that is not constant."""
    import torch
    n, S = 1 << 20, 32768
    rng = np.random.default_rng(77)
    x = np.empty(n, dtype=np.uint8)
    sampled = np.zeros(n, dtype=bool)
    for lo in range(0, n, 4 * S):
        sampled[lo:lo + S] = True
    if kind == "two_symbols_sampled":
        x[:] = rng.integers(0, 256, n, dtype=np.uint8)
        x[sampled] = rng.choice(np.array([97, 101], dtype=np.uint8), int(sampled.sum()))
    elif kind == "sample_constant":
        x[:] = rng.integers(0, 4, n, dtype=np.uint8) + 60
        x[sampled] = 61
    elif kind == "high_symbols_missed":
        x[:] = rng.integers(200, 256, n, dtype=np.uint8)
        x[sampled] = rng.integers(0, 100, int(sampled.sum()), dtype=np.uint8)
    else:
        x[:] = rng.integers(0, 50, n, dtype=np.uint8)
        x[sampled] = rng.integers(100, 256, int(sampled.sum()), dtype=np.uint8)
    rows = 3
    y = datagen.zipf_bytes(n, seed=5)
    blocks = [x] * (rows - 1) + [y]
    d_in = torch.from_numpy(np.concatenate(blocks)).cuda()
    d_out = torch.zeros_like(d_in)
    d_idx = torch.zeros(rows, dtype=torch.int32, device=d_in.device)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
        assert glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
        plan.synchronize()
    got, gidx = d_out.cpu().numpy().reshape(rows, n), d_idx.cpu().numpy()
    for k, blk in ((0, x), (1, x), (2, y)):
        want, widx = O.bwt(blk)
        assert int(gidx[k]) == widx and np.array_equal(got[k], want), (kind, k)
