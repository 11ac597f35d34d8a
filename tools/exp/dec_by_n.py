#!/usr/bin/env python3
"""decode kernel times as a function of the block size n at a fixed total (256 MiB): does the walk get faster per byte when a
block's LF table is small enough to stay in L2?   usage: dec_by_n.py   (prints the plan's live kernel profile per n)"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
dev = torch.device("cuda:0")
total = 256 << 20
d_all = bench.zipf_blocks_on_device(torch, dev, 256, 0, 1)
for lg in (20, 19, 18, 17, 16):
    n = 1 << lg
    rows = total // n
    if rows > 1024:
        rows = 1024
    x = d_all[: rows * n]
    with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
        comp = glc.compress_batch(plan, x, n, rows)
        back = glc.decompress_batch(plan, comp, n, rows); plan.synchronize()
        plan.enable_timing(3)
        for _ in range(3):
            back = glc.decompress_batch(plan, comp, n, rows)
        plan.synchronize()
        prof = plan.kernel_profiles()
        mib = rows * n / (1 << 20)
        print("n=2^%d rows=%d (%d MiB): " % (lg, rows, mib) + "  ".join("%s %.3f" % (k.split("+")[-1][:14], v["ms"] / v["launches"] * 256 / mib) for k, v in prof.items() if k.startswith(("k_i", "k_d"))) + "   [ms per 256 MiB]  ok=%s" % bool(torch.equal(back, x)))
