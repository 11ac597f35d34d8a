#!/usr/bin/env python3
"""Where a k_fs_sort2 workgroup spends its life: needs a library built with -DGLC_FS2_CLOCKS (GLC_LIB points at it).
usage: GLC_LIB=... fs2_clocks.py [rows]   -- s_memtime ticks (100 MHz) per phase, summed over wave 0 of every workgroup"""
import ctypes as C, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
n = 1 << 20
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
d_out = torch.empty_like(d_in); d_idx = torch.empty(rows, dtype=torch.int32, device=dev)
L = glc.lib()
names = ["prologue", "loop top", "fetch+atomics", "scan", "scatter", "zero+rank", "ties", "take (wait words)", "rows"]
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
    for it in range(3):
        out = (C.c_ulonglong * 16)()
        L.glcFs2Clocks(out, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
        torch.cuda.synchronize(); t1 = time.perf_counter()
        L.glcFs2Clocks(out, 0)
        tot = sum(out[:9])
        nbk = rows * 512
        print("batch %.3f ms; per bucket (wave 0): " % ((t1 - t0) * 1e3) +
              ", ".join("%s %.1f%%" % (names[i], 100.0 * out[i] / tot) for i in range(9)) +
              "; %d workgroups, %.2f us each by s_memrealtime (100 MHz) = %.2f us per bucket; s_memtime ticks per us: %.0f"
              % (out[14], out[15] / max(1, out[14]) / 100.0, out[15] / max(1, out[14]) / 100.0 / 8, tot / max(1, out[15]) * 100.0))
