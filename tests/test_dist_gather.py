"""world_size-2 gloo test (CPU) of the multi-GPU result collection and of the
round-robin block assignment used by bench.py."""
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import datagen
    mod = _load()
    # global blocks g = rank, rank+world, ... (bench.py's assignment); the encoder here is the
    # oracle (this is a host-logic test, there is no GPU)
    nblk, n = 3, 8192
    streams, sizes = [], []
    for i in range(nblk):
        g = rank + i * world
        r = O.compress(datagen.zipf_bytes(n, seed=1000 + g))
        streams.append(r["words"].astype(np.int64).astype(np.uint32).view(np.int32))
        sizes.append(r["size"])
    compact = torch.zeros(sum(sizes) + 100, dtype=torch.int32)
    off = torch.zeros(nblk + 1, dtype=torch.int64)
    pos = 0
    for i, s in enumerate(streams):
        compact[pos:pos + s.size] = torch.from_numpy(s.copy())
        off[i] = pos
        pos += s.size
    off[nblk] = pos
    res = mod.gather_streams(dist, torch, compact, off, dst=0)
    if rank == 0:
        ok = res is not None and len(res["buffers"]) == world
        # the gathered stream of rank r must equal what a single process would produce for blocks r, r+world, ...
        for r in range(world):
            exp = np.concatenate([O.compress(datagen.zipf_bytes(n, seed=1000 + r + i * world))["words"] for i in range(nblk)])
            ok = ok and np.array_equal(res["buffers"][r].numpy().view(np.uint32), exp)
        ok = ok and res["total_words"] == sum(res["per_rank_words"])
        q.put(bool(ok))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_streams_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
