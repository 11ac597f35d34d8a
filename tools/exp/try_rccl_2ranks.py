#!/usr/bin/env python3
"""Can two ranks of an RCCL communicator share ONE device on this box?  If so, the C-ABI exchange (glcGatherStreams /
glcScatterStreams, two ranks, real ncclSend / ncclRecv) is exercised end to end; if not, prints RCCL's refusal.
usage: try_rccl_2ranks.py"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "gpu-lossless-compression_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))
N = 1 << 18


def _mod(name, fname):
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, fname))
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m)
    return m


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import numpy as np, torch, torch.distributed as dist, datagen
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    glc = _mod("glc_binding", "glc_binding.py"); ex = _mod("glc_dist", "dist_gather.py")
    L = glc.lib()
    try:
        xch = ex.RcclExchange(glc, torch, dist)
    except Exception as e:
        q.put((rank, "init failed: %r" % (e,))); return
    nblk = 3 - rank
    nsub, stride = N // 4096, glc.compressed_stride_words(N)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, N, rows=8) as plan:
        x = np.concatenate([datagen.float_bytes(N, seed=100 + rank + i * world) for i in range(nblk)])
        d_in = torch.from_numpy(x).to(dev)
        out = glc.compress_batch(plan, d_in, N, nblk)
        compact = torch.empty(nblk * stride, dtype=torch.int32, device=dev); off = torch.empty(nblk + 1, dtype=torch.int64, device=dev)
        assert L.glcCompactStreams(plan.handle, out["words"].data_ptr(), stride, out["size"].data_ptr(), nblk, compact.data_ptr(), off.data_ptr()) == 0
        plan.synchronize()
        rec = xch.pack_records(out, nblk, nsub)
        g = xch.gather(compact, off.data_ptr() + 8 * nblk, rec, dst=0)
        torch.cuda.synchronize()
        ok = True
        counts = list(xch.counts)
        if rank == 0:
            g = xch.finish(g)
            ok = ok and g["nblk"] == [3, 2] and torch.equal(g["buffers"][0], compact[:int(off[nblk].item())]) and torch.equal(g["records"][0], rec)
        buf, boff, brec = xch.scatter(g, counts, rec.shape[1], src=0)
        ok = ok and torch.equal(buf, compact[:int(off[nblk].item())]) and torch.equal(brec, rec) and torch.equal(boff, off)
        # rank 1 sends what it got back to rank 0 through gloo for a cross-check of the received bytes
        if rank == 0:
            other = torch.empty(g["words"][1], dtype=torch.int32)
            dist.recv(other, 1)
            ok = ok and torch.equal(other.to(dev), g["buffers"][1])
        else:
            dist.send(compact[:int(off[nblk].item())].cpu(), 0)
    xch.close()
    q.put((rank, "ok" if ok else "MISMATCH"))
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, 29871, q)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(120)
    while not q.empty(): print(q.get())
    print("exit codes", [p.exitcode for p in ps])
    for p in ps:
        if p.is_alive(): p.kill()
