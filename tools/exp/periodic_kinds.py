#!/usr/bin/env python3
"""32 blocks of two periodic halves over a chosen alphabet size (small alphabets: the period's 5-grams repeat, so a group of the first
doubling round holds SEVERAL residue classes -- not one chain).  usage: periodic_kinds.py <alphabet size> [iters]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
A = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n, rows = 1 << 20, 32
rng = np.random.default_rng(5)
blocks = []
for k in range(rows):
    p1, p2 = int(rng.integers(3, 400)), int(rng.integers(3, 400))
    a = np.tile(rng.integers(0, A, p1, dtype=np.uint8), n // (2 * p1) + 1)[:n // 2]
    b = np.tile(rng.integers(0, A, p2, dtype=np.uint8), n // (2 * p2) + 1)[:n - n // 2]
    blocks.append(np.concatenate([a, b]))
d = torch.from_numpy(np.concatenate(blocks)).cuda()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    out = glc.compress_batch(plan, d, n, rows); plan.synchronize()
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        glc.compress_batch_into(plan, d, n, rows, out); plan.synchronize()
        print("alphabet %d: %d blocks %.2f ms" % (A, rows, (time.perf_counter() - t0) * 1e3), plan.last_sort_stats())
    back = glc.decompress_batch(plan, out, n, rows); plan.synchronize()
    print("round trip", bool(torch.equal(back, d)))
