"""The second sorter tier (string sample sort, bwt_bucket.hip k_ss_*): the blocks the bucket sorter flags -- text, logs,
heavy repeated phrases, low-entropy sources -- must come out bit-exact, on this tier where its design says so
(glcPlanLastSortStatsEx: how many blocks each tier gave up on), and on the general sorter beyond its depth cap.
Every case is checked against the oracle and against the other sorter modes (glcPlanSetSorter)."""
import os

import numpy as np
import pytest

import datagen
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def _records(n, seed, key_len=24):
    """log-like records that share a long fixed prefix and differ late: long common prefixes inside every bucket"""
    rng = np.random.default_rng(seed)
    out = bytearray()
    i = 0
    while len(out) < n:
        out += b"2026-09-28T12:00:00Z host-17 svc-auth[4242]: request completed id=" + (b"%0*d" % (key_len, int(rng.integers(0, 10 ** 9)))) + b"\n"
        i += 1
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def _phrases(n, seed, nphrases=40, plen=90, nsep=3):
    """a few long phrases repeated thousands of times between random separators: runs of equal keys hundreds long, for a dozen rounds"""
    rng = np.random.default_rng(seed)
    ph = [bytes(rng.integers(97, 123, plen, dtype=np.uint8)) for _ in range(nphrases)]
    out = bytearray()
    while len(out) < n:
        out += ph[int(rng.integers(0, nphrases))] + bytes(rng.integers(48, 58, nsep, dtype=np.uint8))
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


CASES = {
    # name: (generator, blocks the bucket sorter gives up on, blocks the sample sorter gives up on)   None = either
    "text": (lambda: datagen.text_bytes(N), 1, 0),
    "log": (lambda: datagen.log_bytes(N), 1, 0),
    "records_long_prefix": (lambda: _records(N, 21), 1, 0),
    "phrases": (lambda: _phrases(N, 22), 1, 0),
    # four phrases of 200 bytes, ~1270 copies each: every offset inside a phrase is a run of > 1024 suffixes that agree for up
    # to 28 rounds (k_ss_long's four-wave instance and its all-equal rounds), then differ in the six digits behind the phrase
    "few_long_phrases": (lambda: _phrases(N, 26, nphrases=4, plen=200, nsep=6), 1, 0),
    "two_symbols_iid": (lambda: np.random.default_rng(2).integers(0, 2, N, dtype=np.uint8) * 255, 1, 0),
    "dna": (lambda: np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(23).integers(0, 4, N)], 1, 0),
    "text_then_zipf": (lambda: np.concatenate([datagen.text_bytes(N // 2, seed=24), datagen.zipf_bytes(N // 2, seed=25)]), 1, 0),
    # beyond the depth cap of the tier (~500 symbols past the common prefix of a bucket): the general sorter takes over
    # (a block of ONE symbol is no tier's business any more: k_fs_tables writes its rows -- see test_constant_blocks)
    "zeros": (lambda: np.zeros(N, dtype=np.uint8), 0, 0),
    "period_1_then_one_byte": (lambda: np.concatenate([np.zeros(N - 1, dtype=np.uint8), np.ones(1, dtype=np.uint8)]), 1, 1),
    "period_3": (lambda: np.tile(np.frombuffer(b"abc", dtype=np.uint8), N // 3 + 1)[:N], 1, 1),
    "chunk_repeated_16x": (lambda: np.tile(np.random.default_rng(9).integers(0, 256, N // 16, dtype=np.uint8), 16), 1, 1),
    "text_with_a_long_run": (lambda: np.concatenate([datagen.text_bytes(N // 2, seed=26), np.full(6000, 32, dtype=np.uint8),
                                                     datagen.text_bytes(N // 2 - 6000, seed=27)]), 1, 1),
}


@pytest.fixture(scope="module")
def ctx(glc, cuda):
    c = glc.Cudpp()
    yield c
    c.close()


@pytest.mark.parametrize("name", list(CASES.keys()))
def test_sample_sorter_cases(glc, ctx, cuda, name):
    import torch
    gen, flagged1, flagged2 = CASES[name]
    x = gen()
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=1) as plan:
        got, gidx = _bwt(glc, plan, torch, x)
        assert int(gidx[0]) == widx and np.array_equal(got, want), name
        assert plan.last_sort_stats() == (flagged1, flagged2), "%s: tiers gave up on %r" % (name, plan.last_sort_stats())
        plan.set_sorter(3)                                     # bucket sorter, then the general sorter: same bytes
        g3, i3 = _bwt(glc, plan, torch, x)
        assert int(i3[0]) == widx and np.array_equal(g3, want), "%s without the sample tier" % name
        assert plan.last_sort_stats() == (flagged1, flagged1)
        plan.set_sorter(4)                                     # sample sorter first (the caller's hint): same bytes, no attempt
        g4, i4 = _bwt(glc, plan, torch, x)
        assert int(i4[0]) == widx and np.array_equal(g4, want), "%s with the sample sorter first" % name
        assert plan.last_sort_stats() == ((1, flagged2) if name != "zeros" else (0, 0))
        plan.set_sorter(0)
        g0, i0 = _bwt(glc, plan, torch, x)                     # and the plan is reusable afterwards
        assert int(i0[0]) == widx and np.array_equal(g0, want)


@pytest.mark.parametrize("n", [40, 1000, 2049, 4097, 65537, 300001, 1048575])
def test_sample_sorter_block_sizes(glc, ctx, cuda, n):
    """number of samples, splitters and buckets change with n; the last bytes exercise the end-of-block keys"""
    import torch
    x = datagen.text_bytes(n + 8, seed=n)[:n]
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
        got, gidx = _bwt(glc, plan, torch, x)
        assert int(gidx[0]) == widx and np.array_equal(got, want)
        f1, f2 = plan.last_sort_stats()
        assert f2 == 0 and f1 <= 1


def _text_with_word(n, seed, every, word=b"QZXJKVWY", tail=12):
    """text with ONE 8-byte word (and `tail` random letters behind it) every `every` bytes on average: the samples that fall on it
    share one code -- a run of 16384 * (1 / every) equal codes in k_ss_sample"""
    rng = np.random.default_rng(seed)
    x = datagen.text_bytes(n, seed=seed).copy()
    at = np.sort(rng.choice((n - 64) // 32, size=n // every, replace=False)) * 32
    w = np.frombuffer(word, dtype=np.uint8)
    for o in at:
        x[o:o + 8] = w
        x[o + 8:o + 8 + tail] = rng.integers(97, 123, tail, dtype=np.uint8)
    return x


def _text_with_tail_copy(n, seed, length, src=12345):
    """the block's last `length` bytes repeat an earlier stretch: samples in the tail tie with samples in the stretch up to the
    END of the text (k_ss_sample: the pair-by-pair form of a window with a tied member near the end)"""
    x = datagen.text_bytes(n, seed=seed).copy()
    x[n - length:] = x[src:src + length]
    return x


@pytest.mark.parametrize("name,gen", [
    ("word_every_90", lambda: _text_with_word(N, 41, 90)),        # a run of ~180 equal codes: ordered on its own, four to a lane
    ("word_every_200", lambda: _text_with_word(N, 42, 200)),      # ~80: inside a window
    ("word_every_40", lambda: _text_with_word(N, 43, 40, tail=20)),   # ~410: past the cap, the network with text comparisons
    ("tail_copy_40", lambda: _text_with_tail_copy(N, 44, 40)),
    ("tail_copy_300", lambda: _text_with_tail_copy(N, 45, 300)),
    ("tail_copy_3000_log", lambda: np.concatenate([datagen.log_bytes(N - 3000, seed=46), datagen.log_bytes(N, seed=46)[5000:8000]])),
    # a 100-byte phrase 40 times at the very end: samples of equal phase tie up to the end of the text, past the exact form's cap
    # and (sorter mode 6) inside the tolerant form's
    ("tail_phrase_x40", lambda: np.concatenate([datagen.text_bytes(N - 4000, seed=47), np.tile(datagen.text_bytes(100, seed=48), 40)])),
])
def test_sample_step_paths(glc, ctx, cuda, name, gen):
    """k_ss_sample orders its samples in three ways (windows of runs, long runs, the network with text comparisons) and ranks a
    window pair by pair when a tied member is near the end of the text: every one of them must give the same splitters' order"""
    import torch
    x = gen()
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=1) as plan:
        for mode in (0, 4, 6):
            plan.set_sorter(mode)
            got, gidx = _bwt(glc, plan, torch, x)
            assert int(gidx[0]) == widx and np.array_equal(got, want), "%s, sorter mode %d" % (name, mode)


def test_sample_sorter_mixed_batch(glc, ctx, cuda):
    """one batch, all three tiers at work: every block must carry its own tier's result"""
    import torch
    blocks = [datagen.text_bytes(N, seed=31), datagen.zipf_bytes(N, seed=32), np.zeros(N, dtype=np.uint8),
              datagen.log_bytes(N, seed=33), datagen.float_bytes(N, seed=34), _phrases(N, 35),
              np.tile(np.frombuffer(b"xy", dtype=np.uint8), N // 2), datagen.text_bytes(N, seed=36)]
    x = np.concatenate(blocks)
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=len(blocks)) as plan:
        for rep in range(2):                                   # twice: scratch of the first call must not leak into the second
            got, gidx = _bwt(glc, plan, torch, x, rows=len(blocks))
            assert plan.last_sort_stats() == (5, 1)             # (the all-zero block is finished by k_fs_tables, not by a tier)
            for i, blk in enumerate(blocks):
                want, widx = O.bwt(blk)
                assert int(gidx[i]) == widx and np.array_equal(got[i * N:(i + 1) * N], want), "block %d (call %d)" % (i, rep)


def test_sample_sorter_first_on_iid_data(glc, ctx, cuda):
    """the hint on data that did not need it: slower, same bytes"""
    import torch
    x = np.concatenate([datagen.zipf_bytes(N, seed=51), datagen.float_bytes(N, seed=52)])
    with glc.Plan(ctx, glc.CUDPP_BWT, N, rows=2) as plan:
        plan.set_sorter(4)
        got, gidx = _bwt(glc, plan, torch, x, rows=2)
        assert plan.last_sort_stats() == (2, 0)
        for i in range(2):
            want, widx = O.bwt(x[i * N:(i + 1) * N])
            assert int(gidx[i]) == widx and np.array_equal(got[i * N:(i + 1) * N], want)


@pytest.mark.parametrize("mode", [0, 3, 4])
def test_suffix_array_of_text(glc, ctx, cuda, mode):
    """cudppSuffixArray on text: the tiers write the suffix array itself (not only BWT bytes); a deep block in the same
    plan afterwards takes the general sorter's"""
    import torch
    n = 300000
    with glc.Plan(ctx, glc.CUDPP_SA, n) as plan:
        plan.set_sorter(mode)
        for x, general in ((datagen.text_bytes(n, seed=61), 0), (np.tile(np.frombuffer(b"abcd", dtype=np.uint8), n // 4), 1),
                           (datagen.log_bytes(n, seed=62), 0)):
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.zeros(n + 1, dtype=torch.int32, device=d_in.device)
            assert glc.lib().cudppSuffixArray(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n) == glc.CUDPP_SUCCESS
            got = d_out.cpu().numpy().view(np.uint32)
            want = O.suffix_array(x)
            assert got[0] == n and np.array_equal(got[1:], want)
            assert plan.last_sort_stats()[1] == (1 if mode == 3 else general)


def test_compress_text_round_trip(glc, ctx, cuda):
    """whole pipeline on text (the speculative MTF + Huffman pass is redone for the flagged blocks): stream decodes to the input"""
    import torch
    rows = 4
    x = np.concatenate([datagen.text_bytes(N, seed=41), datagen.log_bytes(N, seed=42), datagen.zipf_bytes(N, seed=43), _records(N, 44)])
    d_in = torch.from_numpy(x).cuda()
    with glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=rows) as plan:
        comp = glc.compress_batch(plan, d_in, N, rows)
        plan.synchronize()
        assert plan.last_sort_stats() == (3, 0)
        back = glc.decompress_batch(plan, comp, N, rows)
        torch.cuda.synchronize()
        assert np.array_equal(back.cpu().numpy(), x)
        for i in range(rows):                                  # and it is the reference's stream: same words as the oracle's
            want = O.compress(x[i * N:(i + 1) * N])
            size = int(comp["size"][i].item())
            assert size == want["size"] and int(comp["bwt_index"][i].item()) == want["bwt_index"]
            words = comp["words"][i * comp["stride"]:i * comp["stride"] + size].cpu().numpy().view(np.uint32)
            assert np.array_equal(words, want["words"]), "block %d" % i


def test_pipelined_calls_across_tiers(glc, ctx, cuda):
    """glcPlanSetPipelining with batches that change character from call to call (all text, all Zipf, text with a block
    for the general sorter, ...): the plan's memory of the previous call decides whether the stages behind the sort are
    queued speculatively, and every combination must give the streams of plain calls"""
    import torch
    n, nb = 1 << 17, 3
    kinds = ["text", "text", "zipf", "deep", "text", "zipf", "zipf", "text"]

    def batch(c, kind):
        if kind == "zipf":
            return np.concatenate([datagen.zipf_bytes(n, seed=3000 + 10 * c + b) for b in range(nb)])
        x = [datagen.text_bytes(n, seed=4000 + 10 * c + b) for b in range(nb)]
        if kind == "deep":
            x[1] = np.tile(np.frombuffer(b"ab", dtype=np.uint8), n // 2)
        return np.concatenate(x)
    batches = [batch(c, k) for c, k in enumerate(kinds)]
    d_in = [torch.from_numpy(x).cuda() for x in batches]

    def run(pipelined):
        outs = []
        with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
            plan.set_pipelining(pipelined)
            res = [glc.compress_batch(plan, d_in[c], n, nb) for c in range(len(kinds))]   # no sync in between
            plan.synchronize()
            for r in res:
                sizes = r["size"].cpu().numpy()
                outs.append((r["bwt_index"].cpu().numpy().copy(), sizes.copy(), r["hist"].cpu().numpy().copy(),
                             [r["words"][b * r["stride"]: b * r["stride"] + int(sizes[b])].cpu().numpy().copy()
                              for b in range(nb)]))
        return outs

    plain, piped = run(False), run(True)
    for c in range(len(kinds)):
        assert np.array_equal(plain[c][0], piped[c][0]) and np.array_equal(plain[c][1], piped[c][1]), "call %d" % c
        assert np.array_equal(plain[c][2], piped[c][2]), "call %d" % c
        for b in range(nb):
            assert np.array_equal(plain[c][3][b], piped[c][3][b]), "call %d block %d" % (c, b)
    for c in (1, 3, 5):                                        # and against the oracle
        for b in range(nb):
            want = O.compress(batches[c][b * n:(b + 1) * n])
            assert np.array_equal(plain[c][3][b].view(np.uint32), want["words"]), "call %d block %d vs oracle" % (c, b)


def test_sample_sorter_first_on_short_blocks_ending_in_zero_bytes(glc, ctx, cuda):
    """ADVICE.md (round 2): with the sample tier forced (sorter 4) on blocks shorter than 128 bytes a suffix inside
    the trailing zero bytes can be a splitter; a word and a splitter whose zero-padded first 8 bytes agree and which
    both end inside them must compare by LENGTH (fs_suffix_less called with k = 8, both past the end)."""
    import torch
    rng = np.random.default_rng(4)
    cases = []
    for n in (16, 23, 40, 64, 65, 100, 127):
        for z in (2, 3, 6, 8):
            x = rng.integers(0, 4, n, dtype=np.uint8)
            x[n - z:] = 0
            cases.append(x)
        cases.append(np.zeros(n, dtype=np.uint8))
        y = rng.integers(1, 256, n, dtype=np.uint8)
        y[n - 5:] = 0
        cases.append(y)
    for x in cases:
        n = x.size
        want, widx = O.bwt(x)
        with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
            for mode in (4, 0, 3):
                plan.set_sorter(mode)
                got, gidx = _bwt(glc, plan, torch, x)
                assert int(gidx[0]) == widx and np.array_equal(got, want), "n %d mode %d: %r" % (n, mode, x.tolist())


def test_bucket_past_its_slot_gets_a_second_attempt(glc, cuda):
    """tests/golden/log_block_bucket_overflow.bin.bz2: a log-style block (max LCP 51) whose splitters, as the first sample set
    draws them, leave one bucket with more than the 4032 words a slot holds -- ~1 % of log blocks.  It used to go to the
    general sorter (ten times slower, host-held rounds); now the sample sorter draws other samples once.  BWT == oracle,
    nothing left for the general sorter, and the block is alone in a batch of Zipf blocks that must not be touched."""
    import bz2
    import torch
    import oracle_lib as O
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "log_block_bucket_overflow.bin.bz2")
    blk = np.frombuffer(bz2.decompress(open(path, "rb").read()), dtype=np.uint8)
    n = blk.size
    assert n == 1 << 20
    others = [datagen.zipf_bytes(n, seed=900 + i) for i in range(2)]
    x = np.concatenate([others[0], blk, others[1]])
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=3) as plan:
        d_in = torch.from_numpy(x).to(cuda)
        d_out = torch.zeros_like(d_in)
        d_idx = torch.zeros(3, dtype=torch.int32, device=cuda)
        assert glc.lib().glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, 3) == 0
        plan.synchronize()
        flagged, general = plan.last_sort_stats()
        assert (flagged, general) == (1, 0), (flagged, general)
        assert plan.last_sort_retries() == 1
        got = d_out.cpu().numpy()
        for i, b in enumerate((others[0], blk, others[1])):
            want, idx = O.bwt(b)
            assert np.array_equal(got[i * n:(i + 1) * n], want) and int(d_idx[i].item()) == idx, i


@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, 65536, 1048576])
def test_constant_blocks(glc, ctx, cuda, n):
    """a block of one symbol: BWT = the symbol n times, index n - 1, SA = n-1 .. 0, written by k_fs_tables -- no tier runs
    (it used to be every tier's worst case: 1.3 ms per 1 MiB block on the general sorter).  Alone, under every sorter
    mode that goes through the bucket sorter's front end, and inside a batch whose other blocks must not notice."""
    import torch
    for sym in (0, 65, 255):
        x = np.full(n, sym, dtype=np.uint8)
        want, widx = O.bwt(x)
        assert widx == n - 1 and np.array_equal(want, x)
        with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
            for mode in (0, 3, 4):
                plan.set_sorter(mode)
                got, gidx = _bwt(glc, plan, torch, x)
                assert int(gidx[0]) == widx and np.array_equal(got, want), (n, sym, mode)
                assert plan.last_sort_stats() == (0, 0), (n, sym, mode)
    if n >= 4096:
        blocks = [datagen.zipf_bytes(n, seed=61), np.full(n, 7, dtype=np.uint8), datagen.text_bytes(n, seed=62), np.zeros(n, dtype=np.uint8)]
        x = np.concatenate(blocks)
        with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=4) as plan:
            got, gidx = _bwt(glc, plan, torch, x, rows=4)
            for i, blk in enumerate(blocks):
                want, widx = O.bwt(blk)
                assert int(gidx[i]) == widx and np.array_equal(got[i * n:(i + 1) * n], want), i
        with glc.Plan(ctx, glc.CUDPP_SA, n, rows=1) as plan:                   # cudppSuffixArray: out[0] = n, out[1 ..] = SA
            d_in = torch.from_numpy(np.full(n, 9, dtype=np.uint8)).to(cuda)
            d_out = torch.zeros(n + 1, dtype=torch.int32, device=cuda)
            assert glc.lib().cudppSuffixArray(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n) == 0
            plan.synchronize()
            assert np.array_equal(d_out.cpu().numpy()[1:], np.arange(n - 1, -1, -1, dtype=np.int32))
        with glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=2) as plan:             # through the whole pipeline and back
            two = torch.from_numpy(np.concatenate([np.zeros(n, dtype=np.uint8), datagen.zipf_bytes(n, seed=63)])).to(cuda)
            out = glc.compress_batch(plan, two, n, 2)
            plan.synchronize()
            w0 = O.compress(np.zeros(n, dtype=np.uint8))
            assert int(out["bwt_index"][0].item()) == w0["bwt_index"] and int(out["size"][0].item()) == w0["size"]
            assert np.array_equal(out["words"][: w0["size"]].cpu().numpy().view(np.uint32), w0["words"])
            back = glc.decompress_batch(plan, out, n, 2)
            plan.synchronize()
            assert torch.equal(back, two)



def test_stages_of_finished_blocks_beside_the_second_attempt(glc, cuda):
    """A batch of text blocks in which the sample sorter gives a block a SECOND attempt (a bucket past its slot): the blocks
    its first attempt finished get their MTF + Huffman on a side stream beside that attempt (cudpp_api.cpp: stage_partial), the
    others afterwards.  Every output equals the run with the overlap off (the stage timer keeps everything on one stream), the
    strided and the compact layout, all blocks decode, and the retried blocks equal the oracle."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_for_ss", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench._GLC = glc
    n, nb = 1 << 20, 256                                       # (bench.py's text_like leg: one of these 256 blocks takes the second attempt)
    d_in = bench.text_blocks_on_device(torch, cuda, nb).view(-1)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
        out = glc.compress_batch(plan, d_in, n, nb)
        plan.synchronize()
        retried = plan.last_sort_retries()
        assert plan.last_sort_stats()[0] == nb and retried >= 1, (plan.last_sort_stats(), retried)   # (else this test exercises nothing)
        keys = ("bwt_index", "hist", "offsets", "size", "words")
        first = {k: out[k].clone() for k in keys}
        plan.enable_timing(1)                                    # one stream, no side stages
        glc.compress_batch_into(plan, d_in, n, nb, out)
        plan.synchronize()
        plan.enable_timing(0)
        sizes = out["size"].cpu().numpy()
        for k in ("bwt_index", "hist", "offsets", "size"):
            assert torch.equal(first[k], out[k]), k
        for b in range(nb):
            s0 = b * out["stride"]
            assert torch.equal(first["words"][s0: s0 + int(sizes[b])], out["words"][s0: s0 + int(sizes[b])]), b
        back = glc.decompress_batch(plan, out, n, nb)
        plan.synchronize()
        assert torch.equal(back, d_in)
        # the compact layout takes the same path
        comp = glc.compress_batch_compact(plan, d_in, n, nb)
        plan.synchronize()
        assert plan.last_sort_retries() == retried
        assert torch.equal(comp["size"], out["size"]) and torch.equal(comp["bwt_index"], out["bwt_index"])
        back = glc.decompress_batch_compact(plan, comp, n, nb)
        plan.synchronize()
        assert torch.equal(back, d_in)
        # two blocks against the oracle: one that took the second attempt is among them if the flags say which
        fs, ss = plan.debug_sort_flags(nb) if hasattr(plan, "debug_sort_flags") else (None, None)
        for b in (0, nb - 1):
            x = d_in[b * n:(b + 1) * n].cpu().numpy()
            want = O.compress(x)
            assert int(out["bwt_index"][b].item()) == want["bwt_index"] and int(sizes[b]) == want["size"], b
            assert np.array_equal(out["words"][b * out["stride"]: b * out["stride"] + int(sizes[b])].cpu().numpy().view(np.uint32), want["words"]), b


def test_small_calls_behind_a_text_like_streak_skip_the_bucket_sorter(glc, cuda):
    """The reference's callers hand over ONE block per call (test_compress.cpp:744).  After eight calls in a row whose every block the
    text-likeness probe flagged, a call of up to four blocks goes straight to the sample sorter (glcPlanLastSortSkipped); the first
    block the probe does not flag -- evaluated in skipped calls too -- ends the streak.  Whatever the guess, every call's outputs
    are the oracle's."""
    import torch
    n = 1 << 20
    text = datagen.text_bytes(n, seed=77)
    zipf = datagen.zipf_bytes(n, seed=78)
    want = {"T": O.compress(text), "Z": O.compress(zipf)}
    dev = {"T": torch.from_numpy(text).to(cuda), "Z": torch.from_numpy(zipf).to(cuda)}
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
        seen = []
        for kind in "TTTTTTTTTTZZTZ":
            out = glc.compress_batch(plan, dev[kind], n, 1)
            plan.synchronize()
            w = want[kind]
            size = int(out["size"][0].item())
            assert int(out["bwt_index"][0].item()) == w["bwt_index"] and size == w["size"], (kind, seen)
            assert np.array_equal(out["words"][:size].cpu().numpy().view(np.uint32), w["words"]), (kind, seen)
            assert np.array_equal(out["hist"][:256].cpu().numpy().view(np.uint32), w["hist"]), (kind, seen)
            seen.append((kind, plan.last_sort_skipped()[0], plan.last_sort_stats()[0]))
        skipped = "".join("s" if s else "-" for _, s, _ in seen)
        #            T T T T T T T T T T Z Z T Z      (the streak is eight calls long before a call skips)
        assert skipped == "--------sss---", seen               # a Zipf block behind a streak is the one wrong guess; it ends the streak
        assert [f for _, _, f in seen] == [1] * 10 + [1, 0, 1, 0], seen   # (a skipped call sends its block on whatever it is)
    # a batch of more than four blocks never skips
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=5) as plan:
        d5 = dev["T"].repeat(5)
        for _ in range(4):
            glc.compress_batch(plan, d5, n, 5)
            plan.synchronize()
            assert plan.last_sort_skipped()[0] is False


@pytest.mark.parametrize("n,cut", [(70000, 0), (70003, 5), (262144, 11), (1 << 20, 13)])
def test_first_cut_keys_tie_in_their_high_half(glc, ctx, cuda, n, cut):
    """k_ss_cut's keys are fourteen text bytes {bytes 0..7, bytes 8..13}: records of 8 fixed bytes + 6 bytes of a three-letter alphabet +
    2 fixed bytes make most pivots tie in the high half and differ in the low one (and in its last byte: the fourteenth); the block is
    cut off `cut` bytes into a record, so the keys of its last suffixes reach the end of the text (the 9-bit digits) while the
    others are bytes.  The sample sorter first (mode 4) and by the tiers' own choice."""
    import torch
    rng = np.random.default_rng(n + cut)
    nrec = n // 16 + 2
    rec = np.empty((nrec, 16), dtype=np.uint8)
    rec[:, :8] = np.frombuffer(b"HEADER__", dtype=np.uint8)
    rec[:, 8:14] = rng.integers(0, 3, (nrec, 6), dtype=np.uint8) + 120
    rec[:, 14:] = np.frombuffer(b";\n", dtype=np.uint8)
    x = rec.reshape(-1)[:n].copy()
    want, widx = O.bwt(x)
    with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
        for mode in (4, 0):
            plan.set_sorter(mode)
            got, gidx = _bwt(glc, plan, torch, x)
            assert int(gidx[0]) == widx and np.array_equal(got, want), (mode, int(np.nonzero(got != want)[0][0]) if not np.array_equal(got, want) else -1)
