#!/usr/bin/env python3
"""One cudppCompress call on a rows = 1 plan: wall time per call against the GPU's own span (hipEvents at the first and behind
the last launch: glcPlanEnableTiming(1)) and the time the host spends inside the call.  usage: single_timing.py [zipf|text]"""
import importlib.util, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
bench._GLC = glc
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
d = bench.zipf_blocks_on_device(torch, dev, 1, 0, 1) if kind == "zipf" else bench.text_blocks_on_device(torch, dev, 1)
n = 1 << 20
L = glc.lib()
nsub, stride = n // 4096, glc.compressed_stride_words(n)
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
    o = dict(idx=torch.empty(1, dtype=torch.int32, device=dev), hist=torch.empty(256, dtype=torch.int32, device=dev),
             off=torch.empty(nsub, dtype=torch.int32, device=dev), size=torch.empty(1, dtype=torch.int32, device=dev),
             words=torch.empty(stride, dtype=torch.int32, device=dev))
    def call():
        t0 = time.perf_counter()
        rc = L.cudppCompress(plan.handle, d.data_ptr(), o["idx"].data_ptr(), None, o["hist"].data_ptr(), o["off"].data_ptr(),
                             o["size"].data_ptr(), o["words"].data_ptr(), n)
        t1 = time.perf_counter()
        plan.synchronize()
        t2 = time.perf_counter()
        assert rc == 0
        return (t1 - t0) * 1e3, (t2 - t0) * 1e3
    for mode in (0, 1):
        plan.enable_timing(mode)
        for _ in range(5): call()
        r = [call() + (tuple(plan.last_timing()) if mode else ()) for _ in range(50)]
        med = [statistics.median(x[i] for x in r) for i in range(len(r[0]))]
        print(kind, "timing events %s: host in call %.4f ms, wall per call %.4f ms" % ("on " if mode else "off", med[0], med[1]) +
              (", GPU span sort %.4f + mtf %.4f + huffman %.4f = %.4f ms" % tuple(med[2:6]) if mode else ""))
