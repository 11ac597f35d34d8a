"""The compact output layout (-m gpu): glcCompressBatchCompact writes every block where glcCompressBatch +
glcCompactStreams would have copied it -- same words, same offsets, same records -- and glcDecompressBatchCompact reads
that layout back.  Batches chain through a device-side start offset."""
import ctypes as C

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu


def _blocks(n, kinds):
    gens = {"zipf": datagen.zipf_bytes, "float": datagen.float_bytes, "text": datagen.text_bytes,
            "zeros": lambda n, seed=0: np.zeros(n, dtype=np.uint8)}
    return np.concatenate([gens[k](n, seed=11 + i) for i, k in enumerate(kinds)])


@pytest.mark.parametrize("n,kinds", [(1 << 18, ["zipf", "float", "zipf", "zipf", "float"]),
                                     (1 << 16, ["zipf", "text", "zipf", "zeros", "float", "text", "zipf"]),   # blocks that leave the bucket sorter
                                     (40961, ["zipf", "float", "zipf"]), (1 << 20, ["zipf", "float"]),
                                     (1, ["zipf"]), (7, ["zeros", "zipf", "float"]), (4097, ["zipf", "zeros"]), (4096, ["float"])])
def test_compact_equals_strided_then_compacted(glc, cuda, n, kinds):
    import torch
    L = glc.lib()
    nblk = len(kinds)
    x = _blocks(n, kinds)
    d_in = torch.from_numpy(x).to(cuda)
    stride = glc.compressed_stride_words(n)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nblk) as plan:
        ref = glc.compress_batch(plan, d_in, n, nblk)
        want = torch.zeros(nblk * stride, dtype=torch.int32, device=cuda)
        woff = torch.zeros(nblk + 1, dtype=torch.int64, device=cuda)
        assert L.glcCompactStreams(plan.handle, ref["words"].data_ptr(), stride, ref["size"].data_ptr(), nblk,
                                   want.data_ptr(), woff.data_ptr()) == 0
        plan.synchronize()
        got = glc.compress_batch_compact(plan, d_in, n, nblk)
        plan.synchronize()
        total = int(woff[nblk].item())
        assert torch.equal(got["block_off"], woff)
        assert torch.equal(got["words"][:total], want[:total])
        for k in ("bwt_index", "hist", "offsets", "size"):
            assert torch.equal(got[k], ref[k]), k
        back = glc.decompress_batch_compact(plan, got, n, nblk)
        plan.synchronize()
        assert torch.equal(back, d_in)
        # pipelined plans take the same path on the side stream
        plan.set_pipelining(True)
        got2 = glc.compress_batch_compact(plan, d_in, n, nblk)
        plan.synchronize()                                     # pipelined: the outputs are complete after this, not before
        back2 = glc.decompress_batch_compact(plan, got2, n, nblk)
        plan.synchronize()
        plan.set_pipelining(False)
        assert torch.equal(got2["words"][:total], want[:total]) and torch.equal(back2, d_in)


def test_batches_chain_through_the_device_start_offset(glc, cuda):
    """three batches into ONE array, each starting where the one before ended, no host read in between; decode of a
    batch in the middle reads its slice of the offsets"""
    import torch
    L = glc.lib()
    n, per, nb = 1 << 17, 3, 3
    total_blocks = per * nb
    x = _blocks(n, ["zipf", "float", "zipf"] * nb)
    d_in = torch.from_numpy(x).to(cuda)
    stride = glc.compressed_stride_words(n)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=total_blocks) as plan:
        ref = glc.compress_batch(plan, d_in, n, total_blocks)
        want = torch.zeros(total_blocks * stride, dtype=torch.int32, device=cuda)
        woff = torch.zeros(total_blocks + 1, dtype=torch.int64, device=cuda)
        assert L.glcCompactStreams(plan.handle, ref["words"].data_ptr(), stride, ref["size"].data_ptr(), total_blocks,
                                   want.data_ptr(), woff.data_ptr()) == 0
        plan.synchronize()
        words = torch.zeros(total_blocks * stride, dtype=torch.int32, device=cuda)
        boff = torch.zeros(total_blocks + 1, dtype=torch.int64, device=cuda)
        meta = None
        for k in range(nb):
            start = None if k == 0 else boff.data_ptr() + 8 * (k * per)       # = the end of the batch before, on the device
            out = glc.compress_batch_compact(plan, d_in[k * per * n:(k + 1) * per * n], n, per, words=words, block_off=boff,
                                             start=start, meta=meta, first=k * per)
            meta = {kk: out[kk] for kk in ("bwt_index", "hist", "offsets", "size", "nsub")} if meta is None else meta
            if k == 0:                                                        # grow the metadata arrays to all batches
                meta = dict(bwt_index=torch.empty(total_blocks, dtype=torch.int32, device=cuda),
                            hist=torch.empty(total_blocks * 256, dtype=torch.int32, device=cuda),
                            offsets=torch.empty(total_blocks * out["nsub"], dtype=torch.int32, device=cuda),
                            size=torch.empty(total_blocks, dtype=torch.int32, device=cuda), nsub=out["nsub"])
                for kk in ("bwt_index", "hist", "offsets", "size"):
                    meta[kk][:out[kk].numel()] = out[kk]
        plan.synchronize()
        total = int(woff[total_blocks].item())
        assert torch.equal(boff, woff) and torch.equal(words[:total], want[:total])
        comp = dict(meta); comp.update(words=words, block_off=boff)
        mid = glc.decompress_batch_compact(plan, comp, n, per, first=per)
        plan.synchronize()
        assert torch.equal(mid, d_in[per * n: 2 * per * n])


def test_compact_array_too_small_is_reported_and_not_overrun(glc, cuda):
    import torch
    n, nblk = 1 << 16, 4
    d_in = torch.from_numpy(_blocks(n, ["float"] * nblk)).to(cuda)            # ~incompressible: > n / 4 words per block
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nblk) as plan:
        cap = 3 * (n // 4)
        words = torch.full((cap + 4096,), -1, dtype=torch.int32, device=cuda)
        out = glc.compress_batch_compact(plan, d_in, n, nblk, words=words[:cap])
        with pytest.raises(glc.CudppError):
            plan.synchronize()
        assert int((words[cap:] != -1).sum().item()) == 0                     # nothing behind the stated capacity


def test_compact_decoder_refuses_offsets_that_leave_the_array(glc, cuda):
    import torch
    n, nblk = 1 << 16, 3
    d_in = torch.from_numpy(_blocks(n, ["zipf"] * nblk)).to(cuda)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nblk) as plan:
        out = glc.compress_batch_compact(plan, d_in, n, nblk)
        plan.synchronize()
        total = int(out["block_off"][nblk].item())
        for bad in ([0, total + 5, total + 9, total + 20], [0, 900, 400, total]):
            comp = dict(out)
            comp["words"] = out["words"][:total]
            comp["block_off"] = torch.tensor(bad, dtype=torch.int64, device=cuda)
            glc.decompress_batch_compact(plan, comp, n, nblk)
            with pytest.raises(glc.CudppError):
                plan.synchronize()
