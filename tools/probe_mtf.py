#!/usr/bin/env python3
"""Timing probe (no verification): glcMtfBatch (k_mtf_encode<false>: no histogram) on Zipf blocks.
usage: probe_mtf.py [rows] [iters]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
n = 1 << 20
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
d_out = torch.empty_like(d_in)
L = glc.lib()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_MTF, n, rows=rows) as plan:
    plan.enable_timing(3)
    for it in range(iters + 1):
        assert L.glcMtfBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), n, rows) == 0
    plan.synchronize()
    for k, v in plan.kernel_profiles().items():
        print("%-40s avg %.4f ms over %d launches" % (k, v["ms"] / v["launches"], v["launches"]))
