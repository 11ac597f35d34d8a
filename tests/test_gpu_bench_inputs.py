"""The inputs bench.py times, through the C ABI, against the oracle: until now only bench.py's own
post-check compared them (64 of 4096 blocks, builder-printed).

  * configs[1]: bytes [g MiB, (g + 1) MiB) of the Philox4x32-10 Zipf stream for three g (first, middle, last block of the 4 GiB
    input), generated on the device by glcGenZipfPhilox exactly as bench.zipf_blocks_on_device does, regenerated on the host
    by datagen.zipf_philox_bytes for the oracle;
  * configs[0]-style text and configs[2] log lines: blocks BUILT ON THE DEVICE by bench.text_blocks_on_device /
    bench.log_buffers_on_device (the vectorised generators of the text_like / culzss legs), copied back for the oracle;
  * one rows = 1024 batch (the batch size of the timed region): every block's outputs against its own single-block call,
    sampled blocks against the oracle, all blocks round-tripped through the HIP decoder.
"""
import importlib.util
import os

import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
MiB = 1 << 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench(glc):
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._GLC = glc
    return mod


def _check_block_against_oracle(out, k, x, what):
    """block k of a compress_batch result == orc_compress(x), every output"""
    want = O.compress(x)
    assert want["rc"] == 0
    nsub, stride = out["nsub"], out["stride"]
    assert int(out["bwt_index"][k].item()) == want["bwt_index"], what + ": BWT index"
    assert np.array_equal(out["hist"][256 * k: 256 * k + 256].cpu().numpy().view(np.uint32), want["hist"]), what + ": histogram"
    assert np.array_equal(out["offsets"][nsub * k: nsub * (k + 1)].cpu().numpy().view(np.uint32), want["offsets"]), what + ": offsets"
    size = int(out["size"][k].item())
    assert size == want["size"], what + ": size"
    got = out["words"][stride * k: stride * k + size].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want["words"]), what + ": stream words"


@pytest.mark.parametrize("g", [0, 2047, 4095])
def test_philox_zipf_block_g_equals_oracle(glc, cuda, bench, g):
    import torch
    d = bench.zipf_blocks_on_device(torch, cuda, 1, g, 1)       # block g of the stream, as the timed region's input is made
    x = datagen.zipf_philox_bytes(g * MiB, MiB)
    assert np.array_equal(d.cpu().numpy(), x), "device block %d != host stream" % g
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, MiB, rows=1) as plan:
        out = glc.compress_batch(plan, d, MiB, 1)
        plan.synchronize()
        _check_block_against_oracle(out, 0, x, "Philox-Zipf block %d" % g)
        assert plan.last_sort_stats() == (0, 0)                  # finished by the bucket sorter, as in the timed region
        back = glc.decompress_batch(plan, out, MiB, 1)
        assert np.array_equal(back.cpu().numpy(), x)


def test_device_built_text_block_equals_oracle(glc, cuda, bench):
    import torch
    d = bench.text_blocks_on_device(torch, cuda, 2, seed=0x5EED0001)
    assert d.numel() == 2 * MiB
    x = d.cpu().numpy()
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, MiB, rows=2) as plan:
        out = glc.compress_batch(plan, d, MiB, 2)
        plan.synchronize()
        for k in range(2):
            _check_block_against_oracle(out, k, x[k * MiB:(k + 1) * MiB], "device-built text block %d" % k)
        f1, f2 = plan.last_sort_stats()
        assert f1 == 2 and f2 == 0                               # recognised as text-like, finished by the sample sorter


def test_device_built_log_buffer_culzss_equals_oracle(glc, cuda, bench):
    import torch
    L = glc.lib()
    d = bench.log_buffers_on_device(torch, cuda, 2, seed=0x5EED0003)
    n = MiB
    x = d.cpu().numpy()[n:2 * n].copy()                          # the second buffer (its own seed)
    stride = L.glcLzssPackStride(n)
    d_in = d[n:2 * n].contiguous()
    d_cand = torch.zeros(2 * n, dtype=torch.uint8, device=cuda)
    d_packed = torch.zeros(stride, dtype=torch.uint8, device=cuda)
    d_size = torch.full((1,), -7, dtype=torch.int32, device=cuda)
    d_work = torch.zeros(L.glcLzssWorkBytes(n, 1), dtype=torch.uint8, device=cuda)
    assert L.glcLzssEncodeDevice(d_in.data_ptr(), n, 1, d_cand.data_ptr(), d_packed.data_ptr(), d_size.data_ptr(),
                                 d_work.data_ptr(), None) == 1
    torch.cuda.synchronize()
    want_cand = O.lzss_candidates(x)
    assert np.array_equal(d_cand.cpu().numpy(), want_cand)
    want_packed = O.lzss_pack(want_cand, n)
    size = int(d_size.item())
    assert want_packed is not None and size == want_packed.size
    assert np.array_equal(d_packed.cpu().numpy()[:size], want_packed)
    d_out = torch.zeros(n, dtype=torch.uint8, device=cuda)
    assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_size.data_ptr(), n, 1, d_out.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), x)
    # ... and the same buffer through the BWT pipeline (config 3's data as a cudppCompress input)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, MiB, rows=1) as plan:
        out = glc.compress_batch(plan, d_in, MiB, 1)
        plan.synchronize()
        _check_block_against_oracle(out, 0, x, "device-built log buffer")


def test_rows_1024_batch(glc, cuda, bench):
    """the timed region's batch: 1024 blocks of the Philox-Zipf stream in ONE glcCompressBatch call on a 1024-row plan"""
    import torch
    rows = 1024
    d = bench.zipf_blocks_on_device(torch, cuda, rows, 1024, 1)  # blocks 1024 .. 2047 of the stream
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, MiB, rows=rows) as plan:
        out = glc.compress_batch(plan, d, MiB, rows)
        plan.synchronize()
        assert plan.last_sort_stats() == (0, 0)
        for k in (0, 1, 511, 777, 1023):                         # sampled blocks against the oracle, every output
            _check_block_against_oracle(out, k, datagen.zipf_philox_bytes((1024 + k) * MiB, MiB), "batch block %d" % k)
        # every block: the batch position does not matter -- same outputs as 64-block calls of the same blocks
        with glc.Plan(ctx, glc.CUDPP_COMPRESS, MiB, rows=64) as small:
            for b0 in range(0, rows, 64):
                o2 = glc.compress_batch(small, d[b0 * MiB:(b0 + 64) * MiB], MiB, 64)
                small.synchronize()
                assert torch.equal(o2["bwt_index"], out["bwt_index"][b0:b0 + 64])
                assert torch.equal(o2["size"], out["size"][b0:b0 + 64])
                assert torch.equal(o2["hist"], out["hist"][256 * b0:256 * (b0 + 64)])
                nsub, stride = out["nsub"], out["stride"]
                assert torch.equal(o2["offsets"], out["offsets"][nsub * b0:nsub * (b0 + 64)])
                w1 = out["words"][stride * b0:stride * (b0 + 64)].view(64, stride)
                w2 = o2["words"].view(64, stride)
                col = torch.arange(stride, device=cuda).unsqueeze(0)
                mask = col < o2["size"].unsqueeze(1)
                assert torch.equal(torch.where(mask, w1, 0), torch.where(mask, w2, 0))
        back = glc.decompress_batch(plan, out, MiB, rows)
        assert torch.equal(back, d)
