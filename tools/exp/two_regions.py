#!/usr/bin/env python3
"""bench.py's two_regions blocks through glcCompressBatch (for kstats.sh / timeline_cmd.sh).  usage: two_regions.py [both|halves|stretch] [iters]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
bench._GLC = glc
kinds = sys.argv[1] if len(sys.argv) > 1 else "both"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
d = bench.two_region_blocks_on_device(torch, dev, kinds)
n = 1 << 20
rows = d.numel() // n
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    if os.environ.get("GLC_SORTER"): plan.set_sorter(int(os.environ["GLC_SORTER"]))
    out = glc.compress_batch(plan, d, n, rows); plan.synchronize()
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        glc.compress_batch_into(plan, d, n, rows, out); plan.synchronize()
        print("%s: %d blocks %.2f ms" % (kinds, rows, (time.perf_counter() - t0) * 1e3), plan.last_sort_stats(), "periodic", plan.last_sort_periodic(), "resumed", plan.last_sort_resumed())
