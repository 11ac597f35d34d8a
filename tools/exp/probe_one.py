import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
name = sys.argv[1]
gen = {"zipf": datagen.zipf_bytes, "float": datagen.float_bytes, "text": datagen.text_bytes, "log": datagen.log_bytes}
n = 1 << 20
x = gen[name](n)
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=1) as plan:
    d_in = torch.from_numpy(x).cuda()
    out = glc.compress_batch(plan, d_in, n, 1)
    plan.synchronize(); torch.cuda.synchronize()
    print(plan.last_sort_stats())
