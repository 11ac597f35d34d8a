#!/usr/bin/env python3
"""Which blocks of a generated batch does the sample sorter give up on?  usage: find_flagged.py [log|text] [rows]
Writes the first such block to gpurun_out/flagged_block.bin."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
kind = sys.argv[1] if len(sys.argv) > 1 else "log"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
n = 1 << 20
d_in = (bench.log_buffers_on_device if kind == "log" else bench.text_blocks_on_device)(torch, dev, rows)
bad = []
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=1) as plan:
    L = glc.lib()
    out = torch.empty(n, dtype=torch.uint8, device=dev); idx = torch.empty(1, dtype=torch.int32, device=dev)
    for b in range(rows):
        assert L.glcBwtBatch(plan.handle, d_in.data_ptr() + b * n, out.data_ptr(), idx.data_ptr(), n, 1) == 0
        torch.cuda.synchronize()
        f = plan.last_sort_stats()
        if f[1]:
            import ctypes as C
            a, c2 = (C.c_uint * 1)(), (C.c_uint * 1)()
            L.glcPlanDebugSortFlags.restype = C.c_int
            L.glcPlanDebugSortFlags(C.c_size_t(plan.handle) if not isinstance(plan.handle, C.c_size_t) else plan.handle, a, c2, C.c_size_t(1))
            fillv = (C.c_uint * 512)()
            L.glcPlanDebugBucketFill.restype = C.c_int
            L.glcPlanDebugBucketFill(C.c_size_t(plan.handle), C.c_size_t(0), fillv)
            fl = sorted(fillv, reverse=True)
            bad.append((b, a[0], c2[0], "fills: max %s sum %d" % (fl[:4], sum(fillv))))
print("blocks the sample sorter gave up on:", bad)
if bad:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    d_in[bad[0][0] * n:(bad[0][0] + 1) * n].cpu().numpy().tofile(os.path.join(ROOT, "gpurun_out", "flagged_block.bin"))
