"""GPU decoder for the cudppCompress stream.  The reference has
no GPU decoder; parity = round trip through the HIP encoder + HIP decoder, and
HIP decoder == oracle decoder (the gold decoder semantics of
test_compress.cpp:192-311) on streams produced by the ORACLE encoder."""
import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(17)
    return {
        "zipf_1m": datagen.zipf_bytes(1 << 20),
        "text_1m": datagen.text_bytes(1 << 20),
        "float_1m": datagen.float_bytes(1 << 20),
        "ref_vector_1m": O.glibc_rand_bytes(1 << 20, 255),
        "zeros_1m": np.zeros(1 << 20, dtype=np.uint8),
        "log_300001": datagen.log_bytes(300001),
        "one": np.array([9], dtype=np.uint8),
        "n_4097": rng.integers(0, 256, 4097, dtype=np.uint8),
        "two_symbols": rng.integers(0, 2, 70000, dtype=np.uint8),
        "ends_without_unique_min": np.frombuffer(b"abracadabra" * 3000, dtype=np.uint8),   # gold LF walk fails here
    }


@pytest.mark.parametrize("name", list(_cases().keys()))
def test_round_trip_and_oracle_stream(glc, cuda, name):
    import torch
    x = _cases()[name]
    n = x.size
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, max(n, 64), rows=1) as plan:
        d_in = torch.from_numpy(x.copy()).cuda()
        comp = glc.compress_batch(plan, d_in, n, 1)
        back = glc.decompress_batch(plan, comp, n, 1)
        plan.synchronize()
        assert np.array_equal(back.cpu().numpy(), x), name + ": HIP encode -> HIP decode"
        # decode a stream made by the oracle encoder
        r = O.compress(x)
        nsub, stride = comp["nsub"], comp["stride"]
        words = np.zeros(stride, dtype=np.uint32); words[: r["size"]] = r["words"]
        oc = dict(bwt_index=torch.tensor([r["bwt_index"]], dtype=torch.int32, device=cuda),
                  hist=torch.from_numpy(r["hist"].view(np.int32).copy()).cuda(),
                  offsets=torch.from_numpy(r["offsets"].view(np.int32).copy()).cuda(),
                  words=torch.from_numpy(words.view(np.int32)).cuda(), nsub=nsub, stride=stride)
        back2 = glc.decompress_batch(plan, oc, n, 1)
        plan.synchronize()
        assert np.array_equal(back2.cpu().numpy(), x), name + ": oracle encode -> HIP decode"


def test_batch_round_trip(glc, cuda):
    import torch
    n, nb = 1 << 19, 8
    blocks = [datagen.zipf_bytes(n, seed=i) for i in range(4)] + [datagen.text_bytes(n, seed=9), datagen.float_bytes(n),
                                                                 np.zeros(n, dtype=np.uint8), datagen.log_bytes(n)]
    x = np.concatenate(blocks)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
        d_in = torch.from_numpy(x).cuda()
        comp = glc.compress_batch(plan, d_in, n, nb)
        back = glc.decompress_batch(plan, comp, n, nb)
        plan.synchronize()
        assert torch.equal(back, d_in)


@pytest.mark.parametrize("plan_n,n,nb", [(45537, 45537, 5), (70001, 4099, 3), (1 << 16, 65535, 4), (8191, 1, 7)])
def test_batch_round_trip_misaligned_rows(glc, cuda, plan_n, n, nb):
    """rows of odd length: every row but the first starts at an unaligned address inside the plan's
    scratch (vector paths must fall back), and n may be smaller than the plan's block size"""
    import torch
    rng = np.random.default_rng(plan_n + n)
    blocks = []
    for i in range(nb):
        kind = i % 3
        if kind == 0:
            blocks.append(datagen.zipf_bytes(n, seed=100 + i))
        elif kind == 1:
            blocks.append(rng.integers(0, 4, n, dtype=np.uint8))
        else:
            blocks.append(np.frombuffer((b"the quick brown fox " * (n // 20 + 1))[:n], dtype=np.uint8))
    x = np.concatenate(blocks)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, plan_n, rows=nb) as plan:
        d_in = torch.from_numpy(x.copy()).cuda()
        comp = glc.compress_batch(plan, d_in, n, nb)
        back = glc.decompress_batch(plan, comp, n, nb)
        plan.synchronize()
        assert torch.equal(back, d_in)
        sizes = comp["size"].cpu().numpy()
        for b in range(nb):                                    # and every block equals the oracle's stream
            want = O.compress(blocks[b])
            got = comp["words"][b * comp["stride"]: b * comp["stride"] + int(sizes[b])].cpu().numpy().view(np.uint32)
            assert int(comp["bwt_index"][b].item()) == want["bwt_index"] and int(sizes[b]) == want["size"]
            assert np.array_equal(got, want["words"])


def test_pipelined_decode_calls(glc, cuda):
    """glcPlanSetPipelining on the decode side: Huffman + inverse MTF of call i+1 on the second stream
    while the inverse BWT of call i runs on the plan's stream; six back-to-back calls, outputs read
    through the plan's stream."""
    import torch
    n, nb, calls = 1 << 17, 4, 6
    batches = [np.concatenate([datagen.zipf_bytes(n, seed=3000 + 10 * c + b) if (b + c) % 2 else
                               datagen.text_bytes(n, seed=4000 + 10 * c + b) for b in range(nb)]) for c in range(calls)]
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
        comps = [glc.compress_batch(plan, torch.from_numpy(x).cuda(), n, nb) for x in batches]
        plan.synchronize()
        plan.set_pipelining(True)
        outs = [glc.decompress_batch(plan, c, n, nb) for c in comps]           # no sync in between
        plan.synchronize()
        for x, o in zip(batches, outs):
            assert np.array_equal(o.cpu().numpy(), x)
        plan.set_pipelining(False)
        back = glc.decompress_batch(plan, comps[2], n, nb)
        plan.synchronize()
        assert np.array_equal(back.cpu().numpy(), batches[2])


def test_corrupt_stream_is_reported_not_followed(glc, cuda):
    """row index, block offsets and block lengths are stream data: out-of-range values must not be
    dereferenced, and glcPlanSynchronize has to report them"""
    import torch
    n = 1 << 16
    x = datagen.zipf_bytes(n, seed=3)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
        comp = glc.compress_batch(plan, torch.from_numpy(x).cuda(), n, 1)
        plan.synchronize()
        for what in ("index", "offset", "length", "hist_wrap", "hist_sum"):
            bad = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in comp.items()}
            if what == "hist_wrap":
                bad["hist"][5] = 1 << 23                           # would wrap the tree builder's count << 9 | slot key
            elif what == "hist_sum":
                bad["hist"][7] += 3                                # counts that do not add up to n
            elif what == "index":
                bad["bwt_index"][0] = n + 12345
            elif what == "offset":
                bad["offsets"][3] = 0x7FFFFFF0
            else:
                bad["words"][int(comp["offsets"][2].item())] = 0x7FFFFFFF
            glc.decompress_batch(plan, bad, n, 1)
            with pytest.raises(glc.CudppError):
                plan.synchronize()
        back = glc.decompress_batch(plan, comp, n, 1)           # the plan is still usable
        plan.synchronize()
        assert np.array_equal(back.cpu().numpy(), x)
