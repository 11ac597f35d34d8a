#!/usr/bin/env python3
"""Timing probe (no verification unless --check): glcDecompressBatch on a Zipf batch.
usage: probe_dec.py [rows] [iters] [--check]  -- prints ms per batch; run under rocprofv3 for A/B."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = int(args[0]) if len(args) > 0 else 256
iters = int(args[1]) if len(args) > 1 else 5
dev = torch.device("cuda:0")
n = 1 << 20
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    comp = glc.compress_batch(plan, d_in, n, rows)
    for it in range(iters + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        back = glc.decompress_batch(plan, comp, n, rows)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if it: print("decode batch of %d: %.3f ms  (%.2f GB/s)" % (rows, (t1 - t0) * 1e3, rows * n / (t1 - t0) / 1e9))
    if "--check" in sys.argv:
        print("round trip:", bool(torch.equal(back, d_in)))
