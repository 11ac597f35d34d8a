"""world_size-2 gloo tests (CPU) of the multi-GPU exchange: round-robin block assignment, the
gather of records + exact-length streams to a root, decodability of what the root holds, and the scatter mirror.
There is no GPU here: the encoder/decoder are the oracle (host logic under test is dist_gather.py);
tests/test_gpu_dist.py runs the same exchange with the HIP encoder on the GPU box."""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location(
        "glc_dist", os.path.join(ROOT, "gpu-lossless-compression_amd", "dist_gather.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, q, nblk_per_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import datagen
    mod = _load()
    n, nsub = 8192, 2
    nblk = nblk_per_rank[rank]
    gen = lambda g: datagen.float_bytes(n, seed=1000 + g)          # config 4's data: float32 as bytes
    enc = [O.compress(gen(rank + i * world)) for i in range(nblk)]   # global block g = rank + i * world
    total = sum(e["size"] for e in enc)
    compact = torch.zeros(total + 100, dtype=torch.int32)
    off = torch.zeros(nblk + 1, dtype=torch.int64)
    pos = 0
    for i, e in enumerate(enc):
        compact[pos:pos + e["size"]] = torch.from_numpy(e["words"].view(np.int32).copy())
        off[i] = pos
        pos += e["size"]
    off[nblk] = pos
    out = dict(size=torch.tensor([e["size"] for e in enc], dtype=torch.int32),
               bwt_index=torch.tensor([e["bwt_index"] for e in enc], dtype=torch.int32),
               hist=torch.from_numpy(np.concatenate([e["hist"] for e in enc] or [np.zeros(0, np.uint32)]).view(np.int32).copy()),
               offsets=torch.from_numpy(np.concatenate([e["offsets"] for e in enc] or [np.zeros(0, np.uint32)]).view(np.int32).copy()))
    rec = mod.pack_records(torch, out, nblk, nsub) if nblk else torch.zeros((0, 258 + nsub), dtype=torch.int32)
    res = mod.gather_blocks(dist, torch, compact, off, rec, dst=0)
    ok = True
    if rank == 0:
        ok = res is not None and res["nblk"] == list(nblk_per_rank)
        nglobal = sum(nblk_per_rank)
        for r in range(world):
            for i in range(nblk_per_rank[r]):
                g = r + i * world
                words, record = mod.block_of(res, g)
                want = O.compress(gen(g))                            # what a single process produces for block g
                ok = ok and np.array_equal(words.numpy().view(np.uint32), want["words"])
                f = mod.unpack_records(torch, record.view(1, -1), nsub)
                ok = ok and int(f["size"][0]) == want["size"] and int(f["bwt_index"][0]) == want["bwt_index"]
                # the root can DECODE what it holds: words + record are a complete description of the block
                back = O.decompress(int(f["bwt_index"][0]), f["hist"].numpy().view(np.uint32),
                                    f["offsets"].numpy().view(np.uint32), words.numpy().view(np.uint32), n)
                ok = ok and np.array_equal(back, gen(g))
        ok = ok and sum(res["words"]) == sum(int(o[-1]) for o in res["offsets"]) and nglobal > 0
    else:
        ok = res is None
    # the mirror: every rank gets its own blocks back, bit for bit
    buf, boff, brec = mod.scatter_blocks(dist, torch, res, src=0)
    ok = ok and torch.equal(buf, compact[:total]) and torch.equal(boff, off) and torch.equal(brec, rec)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run(nblk_per_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + len(nblk_per_rank) + sum(nblk_per_rank)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, tuple(nblk_per_rank))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}


def test_gather_decode_scatter_world2_gloo():
    _run([3, 3])


def test_gather_ragged_block_counts_world2_gloo():
    _run([3, 2])            # 5 global blocks over 2 ranks: rank 1 holds one block less
