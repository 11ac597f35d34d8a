"""The N > 1 path end to end on ONE GPU box (-m gpu): two processes share device 0 and talk over gloo (RCCL refuses
two ranks on one device), blocks dealt round-robin, HIP encoder on each rank, gather of records + streams to rank
0, and then what SURVEY.md 8(e) asks of the result:
    * the gathered stream == what ONE process produces for the same global blocks (HIP encoder, no oracle),
    * rank 0 can decode everything it gathered (glcExpandStreams + glcDecompressBatch) back to the input,
    * the scatter mirror hands every rank its own blocks, which it decodes independently.
Data: config 4's float32-as-bytes."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gpu-lossless-compression_amd")
N = 1 << 20


def _mod(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, fname))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, q, nblk_per_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    glc = _mod("glc_binding", "glc_binding.py")
    ex = _mod("glc_dist", "dist_gather.py")
    L = glc.lib()
    nsub, stride = N // 4096, glc.compressed_stride_words(N)
    gen = lambda g: datagen.float_bytes(N, seed=0x5EED0004 + g)
    nblk = nblk_per_rank[rank]
    nglobal = sum(nblk_per_rank)
    ok = True
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=max(nglobal, 1)) as plan:
        mine = np.concatenate([gen(rank + i * world) for i in range(nblk)])
        d_in = torch.from_numpy(mine).to(dev)
        out = glc.compress_batch(plan, d_in, N, nblk)
        compact = torch.empty(nblk * stride, dtype=torch.int32, device=dev)
        off = torch.empty(nblk + 1, dtype=torch.int64, device=dev)
        assert L.glcCompactStreams(plan.handle, out["words"].data_ptr(), stride, out["size"].data_ptr(), nblk,
                                   compact.data_ptr(), off.data_ptr()) == 0
        plan.synchronize()
        rec = ex.pack_records(torch, out, nblk, nsub)
        res = ex.gather_blocks(dist, torch, compact, off, rec, dst=0)
        if rank == 0:
            # (1) one process, all global blocks, same encoder: streams must be identical block by block
            allin = np.concatenate([gen(g) for g in range(nglobal)])
            d_all = torch.from_numpy(allin).to(dev)
            ref = glc.compress_batch(plan, d_all, N, nglobal)
            plan.synchronize()
            sizes = ref["size"].cpu().numpy()
            for g in range(nglobal):
                words, record = ex.block_of(res, g)
                want = ref["words"][g * stride: g * stride + int(sizes[g])]
                ok = ok and int(record[0].item()) == int(sizes[g]) and torch.equal(words, want)
                ok = ok and int(record[1].item()) == int(ref["bwt_index"][g].item())
            # (2) decode everything on the root from what was gathered: global order g = r + i * world
            order = [(g % world, g // world) for g in range(nglobal)]
            recs = torch.stack([res["records"][r][i] for r, i in order])
            f = ex.unpack_records(torch, recs, nsub)
            cat = torch.cat([ex.block_of(res, g)[0] for g in range(nglobal)])
            goff = torch.zeros(nglobal + 1, dtype=torch.int64, device=dev)
            goff[1:] = torch.cumsum(f["size"].to(torch.int64), 0)
            strided = torch.zeros(nglobal * stride, dtype=torch.int32, device=dev)
            assert L.glcExpandStreams(plan.handle, cat.data_ptr(), goff.data_ptr(), nglobal, strided.data_ptr(), stride, None) == 0
            comp = dict(bwt_index=f["bwt_index"], hist=f["hist"], offsets=f["offsets"], words=strided, nsub=nsub, stride=stride)
            back = glc.decompress_batch(plan, comp, N, nglobal)
            plan.synchronize()
            ok = ok and torch.equal(back, d_all)
        # (3) the mirror: own blocks back from the root, decoded independently on every rank
        buf, boff, brec = ex.scatter_blocks(dist, torch, res, src=0, device=dev)
        ok = ok and torch.equal(brec, rec) and torch.equal(boff, off)
        f = ex.unpack_records(torch, brec, nsub)
        strided = torch.zeros(nblk * stride, dtype=torch.int32, device=dev)
        assert L.glcExpandStreams(plan.handle, buf.data_ptr(), boff.data_ptr(), nblk, strided.data_ptr(), stride, None) == 0
        comp = dict(bwt_index=f["bwt_index"], hist=f["hist"], offsets=f["offsets"], words=strided, nsub=nsub, stride=stride)
        back = glc.decompress_batch(plan, comp, N, nblk)
        plan.synchronize()
        ok = ok and torch.equal(back, d_in)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nblk_per_rank", [(3, 3), (3, 2)])
def test_two_ranks_hip_encoder_gather_decode_scatter(cuda, nblk_per_rank):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + sum(nblk_per_rank)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, nblk_per_rank)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}
