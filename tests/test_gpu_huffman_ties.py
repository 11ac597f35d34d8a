"""Huffman trees whose merges tie (-m gpu).  The tree builder (csrc/huff_tree.h) keeps the composites in a queue in
creation order and only re-orders inside a run of equal counts at its tail -- the place where FindMinimumCount's
(count, level, slot) order (cudpp-inpar/src/cudpp/cta/compress_cta.cuh:550-571) is not creation order.  Real data
almost never gets there; histograms of small counts get there dozens of times per tree.  The stream of the HIP
Huffman stage must equal the oracle's (whose tree is pinned to the reference's huffman_build_tree_cpu) for them."""
import os

import numpy as np
import pytest

import oracle_lib as O

NMAX = 1 << 16
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_huff_ties_gold.npz")

CHOICES = {
    "zeros_ones": [0, 1],
    "ones_twos": [1, 2],
    "ones_twos_fours": [1, 1, 2, 4],
    "small_mix": [0, 1, 1, 2, 2, 3, 4, 5, 8],
    "powers": [1, 2, 4, 8, 16, 32, 64],
    "all_equal_3": [3],
    "fib": [1, 1, 2, 3, 5, 8, 13, 21],
    "dense_small": [1, 2, 3],
}


def reorderings(hist256):
    """composites that FindMinimumCount's order puts ahead of an older live composite (a model of the reference's rule
    with a heap; independent of both the oracle and the HIP builder)"""
    import heapq
    h = [(int(c), 0, i) for i, c in enumerate([c for c in list(hist256) + [1] if c > 0])]
    heapq.heapify(h)
    live, n = [], 0
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        live = [x for x in live if x != a and x != b]
        new = (a[0] + b[0], max(a[1], b[1]) + 1, a[2])
        if live and live[-1] > new:
            n += 1
            live = sorted(live + [new])
        else:
            live.append(new)
        heapq.heappush(h, new)
    return n


def _cases():
    out = []
    for name, ch in CHOICES.items():
        for seed in range(3):
            rng = np.random.default_rng(1000 * seed + len(name))
            h = rng.choice(ch, size=256).astype(np.int64)
            if h.sum() == 0:
                h[7] = 1
            out.append((name + "_%d" % seed, h))
    h = np.zeros(256, dtype=np.int64); h[255] = 4096                     # one symbol + EOF
    out.append(("one_symbol", h))
    h = np.ones(256, dtype=np.int64)                                      # 257 leaves of count 1
    out.append(("all_ones", h))
    return out


CASES = _cases()


def test_the_cases_reach_the_reordering_path():
    hits = [reorderings(h) for _, h in CASES]
    assert sum(1 for x in hits if x > 0) >= 12 and max(hits) >= 20, hits


def test_oracle_codes_equal_the_reference_tree_on_the_tied_cases():
    """tests/golden/ref_huff_ties_gold.npz (make_huff_ties_gold.py): codes of the reference's huffman_build_tree_cpu"""
    ref = np.load(GOLD)
    assert [str(c) for c in ref["cases"]] == [c[0] for c in CASES]
    for name, hist in CASES:
        assert np.array_equal(ref[name + "_hist"], hist.astype(np.uint32)), name + ": the case generator drifted"
        codes, lens, _ = O.huff_codes(hist)
        assert np.array_equal(lens, ref[name + "_lens"]), name
        present = lens > 0
        assert np.array_equal(codes[present].astype(np.uint64), ref[name + "_codes"][present]), name


@pytest.mark.gpu
@pytest.mark.parametrize("name,hist", CASES, ids=[c[0] for c in CASES])
def test_tied_trees_give_the_oracle_stream(glc, cuda, name, hist):
    import torch
    rng = np.random.default_rng(len(name))
    sym = rng.permutation(np.repeat(np.arange(256, dtype=np.uint8), hist))
    n = sym.size
    assert 0 < n <= NMAX
    want = O.huff_encode(sym)
    assert want["rc"] == 0
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, NMAX, rows=1) as plan:
        out = glc.huffman_encode_batch(plan, torch.from_numpy(sym).to(cuda), n, 1)
        plan.synchronize()
    size = int(out["size"][0].item())
    assert size == want["size"]
    assert np.array_equal(out["hist"].cpu().numpy().view(np.uint32), want["hist"])
    nsub = (n + 4095) // 4096
    assert np.array_equal(out["offsets"].cpu().numpy().view(np.uint32)[:nsub], want["offsets"])
    assert np.array_equal(out["words"][:size].cpu().numpy().view(np.uint32), want["words"])


@pytest.mark.gpu
def test_random_small_count_histograms_in_one_batch(glc, cuda):
    """96 more histograms (random menus of small counts, geometric and near-equal ones), one call: the batched merge loop of
    huff_tree.h -- pairs of the merged queue order formed 30-60 at a time -- against the oracle's tree, block by block"""
    import torch
    rng = np.random.default_rng(20260929)
    rows, syms = 96, []
    for r in range(rows):
        kind = r % 4
        if kind == 0:
            menu = rng.integers(0, 9, size=int(rng.integers(2, 7)))
            h = rng.choice(menu, size=256)
        elif kind == 1:
            h = np.minimum(rng.geometric(0.08, size=256), 200)
        elif kind == 2:
            h = rng.integers(5, 8, size=256)
        else:
            h = (2 ** rng.integers(0, 6, size=256)) * (rng.random(256) < 0.7)
        h = h.astype(np.int64)
        if h.sum() == 0:
            h[r] = 3
        while h.sum() > NMAX:
            h = h // 2
        s = rng.permutation(np.repeat(np.arange(256, dtype=np.uint8), h))
        syms.append(s)
    n = max(s.size for s in syms)
    # one length per call: shorter blocks are padded with their own most frequent symbol (the histogram changes, the menu does not)
    pad = [np.concatenate([s, np.full(n - s.size, np.bincount(s, minlength=256).argmax(), dtype=np.uint8)]) for s in syms]
    x = np.concatenate(pad)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, NMAX, rows=rows) as plan:
        out = glc.huffman_encode_batch(plan, torch.from_numpy(x).to(cuda), n, rows)
        plan.synchronize()
    sizes = out["size"].cpu().numpy()
    words = out["words"].cpu().numpy().view(np.uint32)
    for r in range(rows):
        want = O.huff_encode(pad[r])
        assert want["rc"] == 0
        assert int(sizes[r]) == want["size"], r
        assert np.array_equal(words[r * out["stride"]:r * out["stride"] + want["size"]], want["words"]), r
