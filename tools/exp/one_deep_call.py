#!/usr/bin/env python3
"""ONE block of bench.py's two_regions kinds through glcCompressBatch a few times (for timeline_cmd.sh).  usage: one_deep_call.py [halves|stretch]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py")); bench._GLC = glc
kind = sys.argv[1] if len(sys.argv) > 1 else "halves"
n = 1 << 20
d = bench.two_region_blocks_on_device(torch, torch.device("cuda:0"), kind)[:n].clone()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
    out = glc.compress_batch(plan, d, n, 1); plan.synchronize()
    for _ in range(3):
        glc.compress_batch_into(plan, d, n, 1, out); plan.synchronize()
