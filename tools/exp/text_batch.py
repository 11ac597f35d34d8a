#!/usr/bin/env python3
"""A batch of text-like blocks through glcCompressBatch a few times (for rocprofv3 --kernel-trace: tools/exp/kstats.sh).
usage: text_batch.py [text|log|text256|log256|zipf] [rows] [iters]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n, distinct = 1 << 20, 8
dev = torch.device("cuda:0")
if kind == "zipf":
    d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
elif kind in ("text256", "log256"):                          # as bench.py's text_like leg: every block distinct
    bench._GLC = glc
    d_in = (bench.text_blocks_on_device(torch, dev, rows) if kind == "text256" else bench.log_buffers_on_device(torch, dev, rows)).view(-1)
else:
    x = {"text": datagen.text_bytes, "log": datagen.log_bytes}[kind](n * distinct).reshape(distinct, n)
    d_in = torch.from_numpy(np.tile(x, (rows // distinct, 1))).to(dev).contiguous().view(-1)
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    out = glc.compress_batch(plan, d_in, n, rows)
    plan.synchronize()
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        glc.compress_batch_into(plan, d_in, n, rows, out)
        plan.synchronize()
        print("%s batch of %d: %.3f ms" % (kind, rows, (time.perf_counter() - t0) * 1e3), plan.last_sort_stats())
