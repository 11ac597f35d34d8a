#!/usr/bin/env python3
"""bench.py's partly_deep blocks (64: text + duplicated 20 KB, text + a phrase every 16 KiB, log + runs, log + duplicated 20 KB):
time of glcBwtBatch per kind and for all 64.  usage: pd_batch.py [iters]   (under kstats.sh: per-kernel totals)"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
bench._GLC = glc
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["all", "textdup", "phrase", "logruns", "logdup"]
n = 1 << 20
dev = torch.device("cuda:0")
tb = bench.text_blocks_on_device(torch, dev, 32, seed=0x5EED0011).view(32, n).clone()
lb = bench.log_buffers_on_device(torch, dev, 32, seed=0x5EED0013).view(32, n).clone()
tb[:16, 600000:620000] = tb[:16, 100000:120000]
for o in range(5000, n - 2000, 16384):
    tb[16:, o:o + 2000] = tb[16:, :2000]
lb[:16, 200000:201500] = 32
lb[:16, 700000:709000] = 0
lb[16:, 500000:520000] = lb[16:, 40000:60000]
sets = {"all": torch.cat([tb, lb]), "textdup": tb[:16], "phrase": tb[16:], "logruns": lb[:16], "logdup": lb[16:]}
L = glc.lib()
for name in kinds:
    x = sets[name]
    rows = x.shape[0]
    d_in = x.reshape(-1).contiguous()
    d_out = torch.zeros_like(d_in); d_idx = torch.zeros(rows, dtype=torch.int32, device=dev)
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
        ts = []
        for it in range(iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
            plan.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("%-8s %3d blocks  %7.2f ms = %.3f ms per block (%.2f GB/s)  (flagged, given up, resumed) = %r" % (
            name, rows, min(ts), min(ts) / rows, rows * n / min(ts) / 1e6, plan.last_sort_stats() + (plan.last_sort_resumed(),)), flush=True)
