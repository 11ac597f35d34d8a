#!/usr/bin/env python3
"""The bucketing pass alone (k_fs_part2), per-kernel time from the plan's hipEvent profile.  For builds with -DGLC_EXP_PART=<k>
(0: as it is, 1: no stores, 2: no global atomic, 4: 6-byte words as a dword + a halfword array) and GLC_FS_STOP_AFTER_PART=1
(nothing behind the pass runs: its output is not a sort).  usage: part_probe.py [rows] [iters]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
n = 1 << 20
bench._GLC = glc
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
d_out = torch.empty_like(d_in); d_idx = torch.empty(rows, dtype=torch.int32, device=dev)
L = glc.lib()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
    def run():
        assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
    run(); plan.synchronize()
    plan.enable_timing(3)
    for _ in range(iters): run()
    plan.synchronize()
    for k, v in sorted(plan.kernel_profiles().items()):
        if v["launches"]: print("   %-36s %.3f ms per launch of %d blocks" % (k, v["ms"] / v["launches"], rows))
