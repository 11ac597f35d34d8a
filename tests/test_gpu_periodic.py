"""The periodic tier (csrc/bwt_periodic.hip): blocks that are ONE periodic stretch with a short tail get their suffix array in
closed form from the sorted rotations of the period.  BWT + index against the oracle (the suffix array of a block is unique:
whatever tier finishes a block, the bytes are the reference's, sa_app.cu:125-298 + compress_kernel.cuh:55-74), with the tier
that finished each block asserted; the same blocks with the tier switched off (glcPlanSetSorter 7) give the same bytes."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
MiB = 1 << 20


def _bwt_batch(glc, cuda, blocks, n, mode=0):
    import torch
    L = glc.lib()
    nb = len(blocks)
    d_in = torch.from_numpy(np.concatenate(blocks)).to(cuda)
    d_out = torch.zeros_like(d_in)
    d_idx = torch.full((nb,), -1, dtype=torch.int32, device=cuda)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=nb) as plan:
        plan.set_sorter(mode)
        assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, nb) == 0
        plan.synchronize()
        return d_out.cpu().numpy().reshape(nb, n), d_idx.cpu().numpy(), plan.last_sort_periodic(), plan.last_sort_stats()


def _periodic(n, p, t, rng, alphabet, exit_smaller=None):
    """w^k cut at n - t, then a tail of t bytes whose first byte breaks the period"""
    w = rng.choice(np.asarray(alphabet, dtype=np.uint8), size=p)
    x = np.tile(w, n // p + 2)[:n].copy()
    if t:
        e = n - t
        tail = rng.choice(np.asarray(alphabet, dtype=np.uint8), size=t)
        cont = int(x[e])                                       # what the period would continue with
        others = [a for a in alphabet if a != cont] or [(cont + 1) & 0xFF]
        if exit_smaller is True:
            smaller = [a for a in others if a < cont]
            tail[0] = smaller[0] if smaller else others[0]
        elif exit_smaller is False:
            larger = [a for a in others if a > cont]
            tail[0] = larger[-1] if larger else others[0]
        else:
            tail[0] = others[int(rng.integers(0, len(others)))]
        x[e:] = tail
    return x


def test_the_three_periodic_kinds_of_the_bench(glc, cuda):
    """a 4 KiB page repeated, one byte up to a different last one, a two-byte period: bench.py's deep_repeats kinds"""
    rng = np.random.default_rng(7)
    onebyte = np.full(MiB, 65, dtype=np.uint8)
    onebyte[-1] = 66
    blocks = [np.tile(rng.integers(0, 256, 4096, dtype=np.uint8), MiB // 4096), onebyte,
              np.tile(np.frombuffer(b"ab", dtype=np.uint8), MiB // 2)]
    got, idx, nper, (f1, f2) = _bwt_batch(glc, cuda, blocks, MiB)
    assert nper == 3, (nper, f1, f2)
    for k, x in enumerate(blocks):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx, k
        assert np.array_equal(got[k], want), k
    # the tier switched off: the general sorter's doubling rounds give the same bytes
    got7, idx7, nper7, _ = _bwt_batch(glc, cuda, blocks, MiB, mode=7)
    assert nper7 == 0 and np.array_equal(got7, got) and np.array_equal(idx7, idx)


@pytest.mark.parametrize("seed", range(6))
def test_random_periods_tails_and_alphabets(glc, cuda, seed):
    """periods 1 .. 4096, tails 0 .. p with either exit, alphabets of two to four symbols (zero among them: the padding of the
    text of representatives is zeros; few symbols: rotations that share long prefixes)"""
    rng = np.random.default_rng(1000 + seed)
    n = 1 << 17
    alphabets = ([0, 1], [0, 255], [7, 8, 9], [0, 1, 2, 200], [65, 66], [0, 3])
    blocks, expect = [], 0
    for p in (1, 2, 3, 5, 16, 100, 777, 4096):
        for case in range(2):
            al = alphabets[int(rng.integers(0, len(alphabets)))]
            t = 0 if case == 0 else int(rng.integers(1, [p + 1, 4 * p + 2, 3000][int(rng.integers(0, 3))]))   # shorter / longer than the period
            x = _periodic(n, p, t, rng, al, exit_smaller=[None, True, False][int(rng.integers(0, 3))])
            if len(set(x.tolist())) == 1:
                x[-1] ^= 1                                      # (a block of one symbol is finished before any tier)
            blocks.append(x)
    got, idx, nper, (f1, f2) = _bwt_batch(glc, cuda, blocks, n)
    for k, x in enumerate(blocks):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx, (seed, k)
        assert np.array_equal(got[k], want), (seed, k, int(np.nonzero(got[k] != want)[0][0]))
    assert nper >= len(blocks) // 2, (nper, f1, f2)            # (random words of few symbols may have a smaller period or sort early)


def test_blocks_the_tier_must_leave_alone(glc, cuda):
    """a period longer than the tier takes, a tail longer than it takes, a break in the middle, a word that is itself periodic
    (its smallest period has a tail too long for it): same bytes by the other tiers"""
    rng = np.random.default_rng(5)
    n = 1 << 17
    long_p = np.tile(rng.integers(0, 4, 6000, dtype=np.uint8), n // 6000 + 1)[:n].copy()
    long_tail = _periodic(n, 64, 0, rng, [1, 2, 3])
    long_tail[-5000:] = rng.integers(0, 256, 5000, dtype=np.uint8)
    broken = _periodic(n, 32, 0, rng, [4, 5])
    broken[n // 2] ^= 1
    ones = np.ones(n, dtype=np.uint8)
    ones[-4500:] = rng.integers(0, 2, 4500, dtype=np.uint8)      # period 1 with a tail of 4500: not p = 5 (or any multiple) either
    ones[-4500] = 0
    blocks = [long_p, long_tail, broken, ones]
    got, idx, nper, _ = _bwt_batch(glc, cuda, blocks, n)
    for k, x in enumerate(blocks):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx and np.array_equal(got[k], want), k
    assert nper == 0


def test_a_tail_longer_than_the_period(glc, cuda):
    """zeros up to a 3000-byte trailer, a 7-byte pattern up to a 100-byte trailer: taken (the tail is covered by whole periods)"""
    rng = np.random.default_rng(6)
    n = 1 << 18
    a = np.zeros(n, dtype=np.uint8)
    a[-3000:] = rng.integers(0, 256, 3000, dtype=np.uint8)
    a[-3000] = 9
    b = _periodic(n, 7, 0, rng, [0, 1, 2])
    b[-100:] = rng.integers(0, 3, 100, dtype=np.uint8)
    b[-100] = (int(b[-107]) + 1) % 3                            # breaks the period right where the trailer starts
    blocks = [a, b]
    got, idx, nper, _ = _bwt_batch(glc, cuda, blocks, n)
    for k, x in enumerate(blocks):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx and np.array_equal(got[k], want), k
    assert nper == 2


def test_periodic_blocks_inside_a_mixed_compress_batch(glc, cuda):
    """cudppCompress outputs of a batch with periodic, text and Zipf blocks: every stream equals the oracle's"""
    import torch
    import datagen
    rng = np.random.default_rng(11)
    n = MiB
    blocks = [datagen.zipf_bytes(n, seed=3), np.tile(rng.integers(0, 256, 1024, dtype=np.uint8), n // 1024),
              datagen.text_bytes(n, seed=4), _periodic(n, 3, 2, rng, [0, 1, 2], exit_smaller=True)]
    d_in = torch.from_numpy(np.concatenate(blocks)).to(cuda)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=len(blocks)) as plan:
        out = glc.compress_batch(plan, d_in, n, len(blocks))
        plan.synchronize()
        assert plan.last_sort_periodic() == 2
        for k, x in enumerate(blocks):
            want = O.compress(x)
            size = int(out["size"][k].item())
            assert int(out["bwt_index"][k].item()) == want["bwt_index"] and size == want["size"], k
            assert np.array_equal(out["words"][out["stride"] * k: out["stride"] * k + size].cpu().numpy().view(np.uint32), want["words"]), k
        back = glc.decompress_batch(plan, out, n, len(blocks))
        assert torch.equal(back, d_in)


def test_exit_suffixes_that_agree_with_a_rotation_through_the_tail(glc, cuda):
    """'abaa' * k + 'baab' and 47 more (tests/periodic_model.ADVERSARIAL): an exit suffix with fewer than p periodic symbols left
    that goes on agreeing with ANOTHER class's rotation through the tail -- deeper than the L + 1 symbols round 5's layout gave a far
    suffix.  Every phase of the first period, three block lengths."""
    import periodic_model as M
    for n in (1 << 16, (1 << 16) + 5, 1 << 17):
        blocks = [np.frombuffer(M.adversarial_block(per, cut, tail, n), dtype=np.uint8) for per, cut, tail in M.ADVERSARIAL]
        got, idx, nper, (f1, f2) = _bwt_batch(glc, cuda, blocks, n)
        for k, x in enumerate(blocks):
            want, widx = O.bwt(x)
            assert int(idx[k]) == widx, (n, k)
            assert np.array_equal(got[k], want), (n, k, M.ADVERSARIAL[k], int(np.nonzero(got[k] != want)[0][0]))
        assert nper == len(blocks), (n, nper, f1, f2)


@pytest.mark.parametrize("seed", range(4))
def test_small_alphabet_sweep_short_periods_short_tails(glc, cuda, seed):
    """periods 1..7, tails 0..10, two or three symbols, every phase of the break: 256 blocks of 64 KiB per seed (the shapes the
    CPU model sweeps by the ten thousand, tests/test_cpu_periodic_model.py)"""
    rng = np.random.default_rng(4242 + seed)
    n = 1 << 16
    blocks = []
    while len(blocks) < 256:
        p, t, A = int(rng.integers(1, 8)), int(rng.integers(0, 11)), int(rng.integers(2, 4))
        per = rng.integers(0, A, p, dtype=np.uint8)
        tail = rng.integers(0, A, t, dtype=np.uint8)
        x = np.concatenate([np.tile(per, n // p + 1)[:n - t], tail])
        if len(set(x.tolist())) > 1:
            blocks.append(x)
    got, idx, nper, (f1, f2) = _bwt_batch(glc, cuda, blocks, n)
    for k, x in enumerate(blocks):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx, (seed, k)
        assert np.array_equal(got[k], want), (seed, k, int(np.nonzero(got[k] != want)[0][0]))
    assert nper >= 200, (nper, f1, f2)
