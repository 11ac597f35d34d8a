import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
bench._GLC = glc
dev = torch.device("cuda:0")
d = bench.text_blocks_on_device(torch, dev, 1).view(-1)
n = 1 << 20
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
    out = glc.compress_batch(plan, d, n, 1); plan.synchronize()
    for _ in range(5):
        glc.compress_batch_into(plan, d, n, 1, out); plan.synchronize()
