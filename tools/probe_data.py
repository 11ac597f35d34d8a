#!/usr/bin/env python3
"""Timing probe: glcCompressBatch on 256 blocks of each synthetic data class of tests/datagen.py (8 distinct blocks,
tiled).  Prints ms per batch, GB/s, blocks handed to the general sorter, per-kernel times.
usage: probe_data.py [classes...]   (default: zipf float text log)"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
sorter = int(os.environ.get("GLC_SORTER", "0"))
classes = sys.argv[1:] or ["zipf", "float", "text", "log"]
dev = torch.device("cuda:0")
n, rows, distinct = 1 << 20, int(os.environ.get("GLC_ROWS", "256")), 8
gen = {"zipf": datagen.zipf_bytes, "float": datagen.float_bytes, "text": datagen.text_bytes, "log": datagen.log_bytes}
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    plan.set_sorter(sorter)
    for name in classes:
        x = gen[name](n * distinct).reshape(distinct, n)
        d_in = torch.from_numpy(np.tile(x, (rows // distinct, 1))).to(dev).contiguous()
        d_in = d_in.view(-1)
        out = glc.compress_batch(plan, d_in, n, rows)
        plan.enable_timing(3)
        ts = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            glc.compress_batch_into(plan, d_in, n, rows, out)
            plan.synchronize(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ms = min(ts[1:])
        print("%-6s %8.3f ms per batch = %6.1f GB/s   blocks on the general sorter: %d" % (name, ms, n * rows / ms / 1e6, plan.last_flagged_blocks()) + "  sample tier gave up on: %d" % plan.last_sort_stats()[1])
        prof = plan.kernel_profiles()
        print("       " + ", ".join("%s %.2f" % (k.replace("glc::", ""), v["ms"] / 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:24]))
