#!/usr/bin/env python3
"""Do two of the encoder's stages share the GPU when they are queued on two streams?  glcMtfBatch (k_mtf_encode, no
histogram) on one plan and glcBwtBatch (k_fs_part2 + k_fs_sort_bwt) on another, alone and together.
usage: overlap_probe.py [rows]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
n = 1 << 20
bench._GLC = glc
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
d_bwt = torch.empty_like(d_in); d_idx = torch.empty(rows, dtype=torch.int32, device=dev)
d_mtf = torch.empty_like(d_in)
L = glc.lib()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as pb, glc.Plan(ctx, glc.CUDPP_MTF, n, rows=rows) as pm:
    pb.set_stream(sa.cuda_stream); pm.set_stream(sb.cuda_stream)
    def bwt(): assert L.glcBwtBatch(pb.handle, d_in.data_ptr(), d_bwt.data_ptr(), d_idx.data_ptr(), n, rows) == 0
    def mtf(): assert L.glcMtfBatch(pm.handle, d_in.data_ptr(), d_mtf.data_ptr(), n, rows) == 0
    def timed(f):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    bwt(); mtf(); torch.cuda.synchronize()
    for it in range(3):
        tb, tm = timed(bwt), timed(mtf)
        both = timed(lambda: (mtf(), bwt()))
        print("bwt alone %.3f ms, mtf alone %.3f ms, sum %.3f; mtf queued then bwt on another stream: %.3f ms (%.0f %% of the sum)" % (tb, tm, tb + tm, both, 100 * both / (tb + tm)))
