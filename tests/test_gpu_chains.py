"""Chain groups of the doubling rounds (csrc/bwt_sa.hip, k_chain_*): a group whose members are an arithmetic progression
i_0, i_0 + d, ... over d-periodic text is ONE monotone chain -- ascending or descending by position, decided where its last two
members differ -- and gets its final order in one round instead of log2(stretch / depth) doubling rounds.  Blocks that only the
general sorter takes (two or more periodic regions, a periodic stretch inside ordinary data, a run of one byte, either direction
of the exit), BWT + index against the oracle (the suffix array of a block is unique: sa_app.cu:125-298 + compress_kernel.cuh:55-74),
with the periodic tier on and off."""
import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu


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
        return d_out.cpu().numpy().reshape(nb, n), d_idx.cpu().numpy(), plan.last_sort_stats(), plan.last_sort_periodic()


def _regions(n, rng, periods, alphabet):
    """len(periods) periodic regions of (about) equal length, random words over `alphabet`"""
    cuts = [n * k // len(periods) for k in range(len(periods) + 1)]
    x = np.empty(n, dtype=np.uint8)
    for k, p in enumerate(periods):
        w = rng.choice(np.asarray(alphabet, dtype=np.uint8), size=p)
        m = cuts[k + 1] - cuts[k]
        x[cuts[k]:cuts[k + 1]] = np.tile(w, m // p + 1)[:m]
    return x


def _check(glc, cuda, blocks, n, what):
    for mode in (0, 7):                                        # the periodic tier on / off: what it leaves (or everything) is the general sorter's
        got, idx, (f1, f2), nper = _bwt_batch(glc, cuda, blocks, n, mode)
        for k, x in enumerate(blocks):
            want, widx = O.bwt(x)
            assert int(idx[k]) == widx, (what, mode, k)
            assert np.array_equal(got[k], want), (what, mode, k, int(np.nonzero(got[k] != want)[0][0]))


@pytest.mark.parametrize("seed", range(4))
def test_two_and_more_periodic_regions(glc, cuda, seed):
    rng = np.random.default_rng(9000 + seed)
    n = 1 << 17
    blocks = []
    for periods, al in (((3, 5), [0, 1]), ((7, 7), [1, 2, 3]), ((1, 2), [0, 255]), ((40, 300), list(range(256))), ((17, 4, 90), [5, 6, 7, 8]),
                        ((400, 2, 31, 31), list(range(97, 123))), ((5000, 3), list(range(256))), ((2, 2, 2), [0, 1])):
        blocks.append(_regions(n, rng, periods, al))
    _check(glc, cuda, blocks, n, "regions")


def test_periodic_stretches_and_runs_inside_ordinary_data(glc, cuda):
    rng = np.random.default_rng(77)
    n = 1 << 18
    z = datagen.zipf_bytes(n, seed=5).copy()
    t = datagen.text_bytes(n, seed=6).copy()
    a = z.copy(); a[50000:150000] = np.tile(rng.integers(0, 256, 37, dtype=np.uint8), 100000 // 37 + 1)[:100000]
    b = t.copy(); b[100000:180000] = 32                           # a run of blanks inside text (d = 1)
    c = z.copy(); c[n - 70000:] = np.tile(np.frombuffer(b"xyz", dtype=np.uint8), 70000 // 3 + 1)[:70000]   # a stretch up to the end of the block
    d = t.copy(); d[:90000] = 0; d[200000:260000] = np.tile(np.frombuffer(b"ab", dtype=np.uint8), 30000)
    e = z.copy()                                                  # the same pattern twice, far apart: groups hold two chains until the doubling separates them
    w = rng.integers(0, 256, 29, dtype=np.uint8)
    e[20000:60000] = np.tile(w, 40000 // 29 + 1)[:40000]
    e[150000:200000] = np.tile(w, 50000 // 29 + 1)[:50000]
    _check(glc, cuda, [a, b, c, d, e], n, "stretches")


def test_either_direction_of_the_chain(glc, cuda):
    """the symbol behind the stretch smaller / larger than the one the period would continue with; the stretch ending the block"""
    n = 1 << 17
    blocks = []
    for nxt in (0, 255):
        x = np.tile(np.frombuffer(b"mnop", dtype=np.uint8), n // 4).copy()
        x[n // 2:] = nxt
        x[n - 1] = 7
        blocks.append(x)
    y = np.tile(np.frombuffer(b"mnop", dtype=np.uint8), n // 4).copy()
    y[:1000] = np.arange(1000, dtype=np.uint32).astype(np.uint8)
    blocks.append(y)
    _check(glc, cuda, blocks, n, "direction")


def test_full_size_blocks_of_the_bench(glc, cuda):
    """bench.py's two_regions kinds at 1 MiB: two periodic halves, a 256 KiB stretch inside Zipf data"""
    rng = np.random.default_rng(11)
    n = 1 << 20
    h = _regions(n, rng, (211, 97), list(range(256)))
    z = datagen.zipf_bytes(n, seed=9).copy()
    z[300000:300000 + 262144] = np.tile(rng.integers(0, 256, 123, dtype=np.uint8), 262144 // 123 + 1)[:262144]
    got, idx, (f1, f2), nper = _bwt_batch(glc, cuda, [h, z], n, 0)
    for k, x in enumerate((h, z)):
        want, widx = O.bwt(x)
        assert int(idx[k]) == widx and np.array_equal(got[k], want), k


def test_progressions_that_are_not_chains(glc, cuda):
    """groups whose members ARE an arithmetic progression with a stride the chain pass tries (<= 4096) but whose text is not periodic
    over it: a 600-byte phrase every 2048 bytes with different random bytes in between; a periodic stretch with ONE byte changed in
    its middle (every residue class still has a member every d bytes across the defect); two stretches of one pattern a whole number
    of periods apart.  The verification must refuse them (plain doubling orders them) -- the bytes are the oracle's either way."""
    rng = np.random.default_rng(31)
    n = 1 << 17
    a = rng.integers(0, 256, n, dtype=np.uint8)
    phrase = rng.integers(97, 123, 600, dtype=np.uint8)
    for o in range(1000, n - 700, 2048):
        a[o:o + 600] = phrase
    w = rng.integers(0, 4, 23, dtype=np.uint8)
    b = np.tile(w, n // 23 + 1)[:n].copy()
    b[n // 2 + 7] ^= 1
    c = rng.integers(0, 256, n, dtype=np.uint8)
    w2 = rng.integers(0, 256, 50, dtype=np.uint8)
    c[10000:30000] = np.tile(w2, 400)
    c[30000 + 50 * 100:30000 + 50 * 100 + 20000] = np.tile(w2, 400)      # the same phase, 100 periods of other bytes in between
    _check(glc, cuda, [a, b, c], n, "not chains")


@pytest.mark.gpu_long
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GLC_MOSAIC_SEEDS", "6"))))
def test_random_mosaics_of_periodic_pieces(glc, cuda, seed):
    """blocks glued from random pieces -- periodic stretches of random period and alphabet, runs, random bytes, copies of earlier
    pieces -- through the general sorter alone (mode 1), the tiers' own choice (0) and with the periodic tier off (7)"""
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([1 << 16, (1 << 16) + 123, 1 << 17, 200000]))
    blocks = []
    for _ in range(12):
        x = np.empty(n, dtype=np.uint8)
        o = 0
        pieces = []
        while o < n:
            kind = int(rng.integers(0, 5))
            m = int(min(n - o, rng.integers(50, n // 2)))
            if kind == 0:
                p = int(rng.integers(1, 600))
                al = int(rng.choice([2, 3, 4, 26, 256]))
                seg = np.tile(rng.integers(0, al, p, dtype=np.uint8), m // p + 1)[:m]
            elif kind == 1:
                seg = np.full(m, int(rng.integers(0, 256)), dtype=np.uint8)
            elif kind == 2 and pieces:
                src = pieces[int(rng.integers(0, len(pieces)))]
                seg = np.resize(src, m)
            else:
                seg = rng.integers(0, int(rng.choice([2, 4, 256])), m, dtype=np.uint8)
            x[o:o + m] = seg
            pieces.append(seg[:min(m, 5000)])
            o += m
        blocks.append(x)
    for mode in (1, 0, 7):
        got, idx, _, _ = _bwt_batch(glc, cuda, blocks, n, mode)
        for k, x in enumerate(blocks):
            want, widx = O.bwt(x)
            assert int(idx[k]) == widx, (seed, mode, k)
            assert np.array_equal(got[k], want), (seed, mode, k, int(np.nonzero(got[k] != want)[0][0]))
